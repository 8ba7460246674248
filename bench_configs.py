"""Network definitions and synthetic data of the benchmark / parity configurations (plain torch.nn only, so the
same definitions feed the HIP engine, the CPU oracle and the golden-vector generator).

Architectures: the DCGAN generator/critic pair of the PyTorch DCGAN tutorial (credited by the reference's
README.md:101) at 64x64 and 128x128; the MLP pair of examples/example_input_formats.py:141-165; the conditional
convolutional pair of notebooks/03_mnist-conditional.ipynb; for the image-to-image configuration a small
encoder-decoder / patch critic pair and the pix2pix U-Net (skip concatenations) / 70x70 PatchGAN pair (the reference
ships neither; its own image-to-image example is a plain encoder-decoder, examples/example_image_to_image.py:35-99).
"""
import numpy as np
import torch
from torch import nn


class Reshape(nn.Module):
    """[B, F] -> [B, *shape] (same role as vegans.utils.layers.LayerReshape, without importing the engine)."""

    def __init__(self, shape):
        super().__init__()
        self.shape = tuple(shape)

    def forward(self, x):
        return torch.reshape(x, (-1, *self.shape))


def dcgan_generator(nz=100, ngf=64, nc=3, size=64):
    """ConvT(nz->8ngf,4,1,0) BN ReLU, then (ConvT(.,4,2,1) BN ReLU) doubling the resolution, ConvT(ngf->nc) Tanh."""
    stages = {16: 2, 32: 4, 64: 8, 128: 16}[size]
    layers = [nn.ConvTranspose2d(nz, ngf * stages, 4, 1, 0, bias=False), nn.BatchNorm2d(ngf * stages), nn.ReLU(True)]
    ch = ngf * stages
    while ch > ngf:
        layers += [nn.ConvTranspose2d(ch, ch // 2, 4, 2, 1, bias=False), nn.BatchNorm2d(ch // 2), nn.ReLU(True)]
        ch //= 2
    layers += [nn.ConvTranspose2d(ngf, nc, 4, 2, 1, bias=False), nn.Tanh()]
    return nn.Sequential(*layers)


def dcgan_critic(ndf=64, nc=3, size=64, batch_norm=True, sigmoid=False):
    """Conv(nc->ndf,4,2,1) LReLU, then (Conv(.,4,2,1) [BN] LReLU) halving the resolution down to 4x4,
    Conv(->1,4,1,0), Flatten, Identity (critic) or Sigmoid (discriminator)."""
    layers = [nn.Conv2d(nc, ndf, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True)]
    ch, res = ndf, size // 2
    while res > 4:
        layers.append(nn.Conv2d(ch, ch * 2, 4, 2, 1, bias=False))
        if batch_norm:
            layers.append(nn.BatchNorm2d(ch * 2))
        layers.append(nn.LeakyReLU(0.2, inplace=True))
        ch, res = ch * 2, res // 2
    layers += [nn.Conv2d(ch, 1, 4, 1, 0, bias=False), nn.Flatten(), nn.Sigmoid() if sigmoid else nn.Identity()]
    return nn.Sequential(*layers)


def dcgan_init(module, seed):
    """Weights N(0, 0.02), BatchNorm gamma N(1, 0.02), beta 0 — from a private generator so construction order
    elsewhere does not matter."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.02)
                if m.bias is not None:
                    m.bias.zero_()
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            with torch.no_grad():
                m.weight.copy_(1.0 + torch.randn(m.weight.shape, generator=g) * 0.02)
                m.bias.zero_()
    return module


def seeded_init(module, seed, scale=0.1):
    """Generic deterministic init used by the small parity networks."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * scale)
            elif "weight" in name:
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return module


def mlp_generator(z_dim=64, im_dim=(1, 28, 28), width=128):
    out = int(np.prod(im_dim))
    return nn.Sequential(
        nn.Linear(z_dim, width), nn.LeakyReLU(0.2),
        nn.Linear(width, 2 * width), nn.LeakyReLU(0.2), nn.BatchNorm1d(2 * width),
        nn.Linear(2 * width, 4 * width), nn.LeakyReLU(0.2), nn.BatchNorm1d(4 * width),
        nn.Linear(4 * width, 8 * width), nn.LeakyReLU(0.2), nn.BatchNorm1d(8 * width),
        nn.Linear(8 * width, out), Reshape(im_dim), nn.Sigmoid())


def mlp_discriminator(im_dim=(1, 28, 28), width=128, sigmoid=True):
    inp = int(np.prod(im_dim))
    return nn.Sequential(
        nn.Flatten(), nn.Linear(inp, 4 * width), nn.LeakyReLU(0.2),
        nn.Linear(4 * width, 2 * width), nn.LeakyReLU(0.2),
        nn.Linear(2 * width, 1), nn.Sigmoid() if sigmoid else nn.Identity())


def mlp_encoder(im_dim=(1, 28, 28), z_dim=64, width=128):
    inp = int(np.prod(im_dim))
    return nn.Sequential(
        nn.Flatten(), nn.Linear(inp, 2 * width), nn.LeakyReLU(0.2), nn.Linear(2 * width, z_dim), nn.Identity())


def cond_conv_generator(in_channels=11, ngf=20, out_channels=1):
    """Conditional generator on a [in_channels, 4, 4] input -> [out_channels, 32, 32] (three stride-2 ConvT)."""
    return nn.Sequential(
        nn.ConvTranspose2d(in_channels, ngf * 4, 4, 2, 1), nn.BatchNorm2d(ngf * 4), nn.LeakyReLU(0.2),
        nn.ConvTranspose2d(ngf * 4, ngf * 2, 4, 2, 1), nn.BatchNorm2d(ngf * 2), nn.LeakyReLU(0.2),
        nn.ConvTranspose2d(ngf * 2, ngf, 4, 2, 1), nn.BatchNorm2d(ngf), nn.LeakyReLU(0.2),
        nn.Conv2d(ngf, out_channels, 3, 1, 1), nn.Sigmoid())


def cond_conv_critic(in_channels=11, ncf=8, sigmoid=False):
    """Conditional critic on [in_channels, 32, 32]: three stride-2 convs, then Flatten -> Linear -> ReLU -> Linear."""
    return nn.Sequential(
        nn.Conv2d(in_channels, ncf, 4, 2, 1), nn.LeakyReLU(0.2),
        nn.Conv2d(ncf, ncf * 2, 4, 2, 1), nn.BatchNorm2d(ncf * 2), nn.LeakyReLU(0.2),
        nn.Conv2d(ncf * 2, ncf * 4, 4, 2, 1), nn.BatchNorm2d(ncf * 4), nn.LeakyReLU(0.2),
        nn.Flatten(), nn.Linear(ncf * 4 * 16, 32), nn.ReLU(), nn.Linear(32, 1),
        nn.Sigmoid() if sigmoid else nn.Identity())


def img2img_generator(in_channels, out_channels=3, width=16):
    """Small encoder-decoder (stride-2 convs down, stride-2 transposed convs up) ending in Sigmoid."""
    return nn.Sequential(
        nn.Conv2d(in_channels, width, 5, 2, 2), nn.LeakyReLU(0.2),
        nn.Conv2d(width, width * 2, 5, 2, 2), nn.LeakyReLU(0.2), nn.BatchNorm2d(width * 2),
        nn.Conv2d(width * 2, width * 2, 3, 1, 1), nn.LeakyReLU(0.2), nn.BatchNorm2d(width * 2),
        nn.ConvTranspose2d(width * 2, width, 4, 2, 1), nn.LeakyReLU(0.2), nn.BatchNorm2d(width),
        nn.ConvTranspose2d(width, out_channels, 4, 2, 1), nn.Sigmoid())


def patch_discriminator(in_channels, width=16):
    """PatchGAN-style discriminator: per-patch probabilities [B, 1, h, w] ending in Sigmoid."""
    return nn.Sequential(
        nn.Conv2d(in_channels, width, 4, 2, 1), nn.LeakyReLU(0.2),
        nn.Conv2d(width, width * 2, 4, 2, 1), nn.BatchNorm2d(width * 2), nn.LeakyReLU(0.2),
        nn.Conv2d(width * 2, 1, 3, 1, 1), nn.Sigmoid())


class UNet(nn.Module):
    """pix2pix U-Net: `depth` stride-2 4x4 convolutions down to a bottleneck, as many transposed convolutions up, each
    up block's output concatenated with the mirrored down block's (skip connections); BatchNorm everywhere except the
    outermost and innermost blocks; no dropout (its random stream could not be shared between implementations).
    The blocks are direct attributes, first convolution first, which is what VeGANs' input check walks."""

    def __init__(self, in_channels, out_channels=3, width=64, depth=8):
        super().__init__()
        self.depth = depth
        chans = [min(width * 2 ** i, width * 8) for i in range(depth)]
        c = in_channels
        for i, co in enumerate(chans):
            layers = [] if i == 0 else [nn.LeakyReLU(0.2)]
            layers.append(nn.Conv2d(c, co, 4, 2, 1, bias=(i == 0 or i == depth - 1)))
            if 0 < i < depth - 1:
                layers.append(nn.BatchNorm2d(co))
            setattr(self, "down%d" % i, nn.Sequential(*layers))
            c = co
        for i in reversed(range(depth)):
            cin = chans[i] if i == depth - 1 else chans[i] * 2
            cout = chans[i - 1] if i > 0 else out_channels
            layers = [nn.ReLU(), nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=(i == 0))]
            if i > 0:
                layers.append(nn.BatchNorm2d(cout))
            setattr(self, "up%d" % i, nn.Sequential(*layers))
        self.output = nn.Sigmoid()

    def forward(self, x):
        skips = []
        for i in range(self.depth):
            x = getattr(self, "down%d" % i)(x)
            skips.append(x)
        skips.pop()
        for i in reversed(range(self.depth)):
            x = getattr(self, "up%d" % i)(x)
            if skips:
                x = torch.cat([x, skips.pop()], 1)
        return self.output(x)


def patchgan_discriminator(in_channels, width=64):
    """The 70x70 PatchGAN of pix2pix: C64-C128-C256 at stride 2, C512 and the 1-channel head at stride 1, Sigmoid."""
    return nn.Sequential(
        nn.Conv2d(in_channels, width, 4, 2, 1), nn.LeakyReLU(0.2),
        nn.Conv2d(width, width * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(width * 2), nn.LeakyReLU(0.2),
        nn.Conv2d(width * 2, width * 4, 4, 2, 1, bias=False), nn.BatchNorm2d(width * 4), nn.LeakyReLU(0.2),
        nn.Conv2d(width * 4, width * 8, 4, 1, 1, bias=False), nn.BatchNorm2d(width * 8), nn.LeakyReLU(0.2),
        nn.Conv2d(width * 8, 1, 4, 1, 1), nn.Sigmoid())


def synthetic_images(n, shape, low=-1.0, high=1.0, seed=0):
    return np.random.default_rng(seed).uniform(low, high, (n, *shape)).astype(np.float32)


def one_hot_labels(n, classes=10, seed=0, dtype=np.float32):
    idx = np.random.default_rng(seed).integers(0, classes, n)
    return np.eye(classes, dtype=dtype)[idx]


# forward GEMM work of the DCGAN pair per image (multiply-accumulates), used by the roofline arithmetic
def dcgan_forward_macs(size=64, nz=100, ngf=64, ndf=64, nc=3):
    g = nz * (ngf * (size // 8)) * 16            # first ConvT: 1x1 -> 4x4
    ch, res = ngf * (size // 8), 4
    while ch > ngf:
        res *= 2
        g += res * res * (ch // 2) * ch * 4      # stride-2 k4 ConvT: 4 taps hit each output pixel
        ch //= 2
    res *= 2
    g += res * res * nc * ngf * 4
    d = (size // 2) ** 2 * ndf * nc * 16
    ch, res = ndf, size // 2
    while res > 4:
        res //= 2
        d += res * res * (ch * 2) * ch * 16
        ch *= 2
    d += ch * 16
    return g, d
