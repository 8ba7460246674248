"""TEST INFRASTRUCTURE — the reduced-precision modes of the HIP engine restated on the CPU, on top of the fp32 oracle
(oracle/restated.py).  Only tests/ may import this module.

The reference (tneuer/GAN-pytorch) has no reduced-precision path: its arithmetic is whatever torch does in float32
(vegans/utils/networks.py:45-47 hands the user's module to torch as it is).  The engine's two bf16 modes are extensions
the task statement asks for, and what pins them is this restatement of WHERE they round, applied to the oracle's own
float32 networks: everything else (losses, penalty, optimizers, BatchNorm buffers) stays the oracle's code.  Parity
status: the fp32 oracle underneath is pinned against the reference; the rounding rules below are pinned against
nothing but the engine's documentation (there is no reference for them) — "parity unpinned" for the bf16 modes.

"bf16" (a bf16 data path): every feature map — an activation whose channel / feature count is a multiple of 8 — is
    stored as bfloat16, round to nearest even, and so is its gradient and, in the double backward of the gradient
    penalty, the gradient of that gradient.  Images, scores, latent vectors, parameters and parameter gradients stay
    float32.  A BatchNorm and the ReLU / LeakyReLU behind it are one kernel (nothing is stored in between), and so are
    the critic's image stem and its activation.  Weights enter the matrix cores as a bf16 copy in the layers the bf16
    contraction kernels serve (`rounds_weights`), unrounded elsewhere.  Sums (batch statistics, contractions) are f32.
"bf16_compute": float32 tensors throughout; in the layers the bf16 contraction kernels serve (the same `rounds_weights`
    rule: the image stem, the critic head, the generator tail and the latent layer keep their f32 kernels) the two
    operands of every contraction (forward, input gradient, weight gradient, and the same three inside the double
    backward) are rounded to bf16 on their way into the matrix cores.

`emulate(net, mode)` returns an nn.Sequential that shares Parameters and BatchNorm modules with `net` (gradients land in
the oracle's own parameters, running statistics in its own buffers).
"""
import torch
from torch import nn
import torch.nn.functional as F


def bf16(t):
    """Round to the nearest bfloat16 (ties to even), kept as float32."""
    return t.to(torch.bfloat16).to(t.dtype)


class Stored(torch.autograd.Function):
    """A tensor that lives in memory as bf16: its value is rounded, and so is every gradient that arrives for it (a
    gradient of a feature map is a feature map), to any order."""

    @staticmethod
    def forward(ctx, x):
        return bf16(x)

    @staticmethod
    def backward(ctx, g):
        return Stored.apply(g)


class Operand(torch.autograd.Function):
    """An operand rounded on its way into the matrix cores: the value is rounded, the gradient passes (through
    `Cotangent`, so that a second-order cotangent arriving for it is rounded on ITS way into the contraction that
    consumes it)."""

    @staticmethod
    def forward(ctx, x):
        return bf16(x)

    @staticmethod
    def backward(ctx, g):
        return Cotangent.apply(g)


class Cotangent(torch.autograd.Function):
    """Identity on the value; the gradient arriving for it is an operand of the backward contractions and is rounded."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return Operand.apply(g)


def rounds_weights(conv):
    """Whether the bf16 data path feeds this layer's weights to the bf16 matrix cores (as a bf16 copy): both channel
    counts multiples of 8 and above 32, and not the 1x1-input transposed convolution that runs as a plain GEMM. The
    image stem, the 1-channel critic head, the 3-channel generator tail and the latent layer run float32 kernels on
    float32 weights. (Layers with 8 ... 32 channels on one side are served per contraction; the test configurations
    have none, and tests/test_bf16_emulation_gpu.py checks this rule against the library's own for every layer.)"""
    ci, co = conv.in_channels, conv.out_channels
    return ci % 8 == 0 and co % 8 == 0 and min(ci, co) > 32


class _Conv(nn.Module):
    def __init__(self, conv, mode, latent):
        super().__init__()
        self.conv, self.mode = conv, mode
        self.served = (not isinstance(conv, nn.Linear)) and rounds_weights(conv) and not latent

    def forward(self, x):
        c = self.conv
        w = c.weight
        both = self.mode == "bf16_compute" and self.served
        if both:
            x, w = Operand.apply(x), Operand.apply(w)
        elif self.served:
            w = Operand.apply(w)
        if isinstance(c, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, w, c.bias, c.stride, c.padding, c.output_padding)
        elif isinstance(c, nn.Conv2d):
            y = F.conv2d(x, w, c.bias, c.stride, c.padding)
        else:
            y = F.linear(x, w, c.bias)
        return Cotangent.apply(y) if both else y


class _Store(nn.Module):
    def forward(self, x):
        return Stored.apply(x)


def _features(t_channels):
    return t_channels % 8 == 0


def emulate(net, mode):
    """`net` (an nn.Sequential of Conv2d / ConvTranspose2d / Linear / BatchNorm / activation / shape layers) with the
    roundings of engine mode `mode` ("bf16" or "bf16_compute") inserted."""
    assert mode in ("bf16", "bf16_compute") and isinstance(net, nn.Sequential)
    mods = list(net)
    out = []
    channels = None
    i = 0
    seen_contraction = False
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        act_next = isinstance(nxt, (nn.ReLU, nn.LeakyReLU))
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            is_conv = not isinstance(m, nn.Linear)
            ci = m.in_channels if is_conv else m.in_features
            channels = m.out_channels if is_conv else m.out_features
            # the generator's first layer sees a 1x1 latent "image": a plain GEMM on float32 weights
            latent = isinstance(m, nn.ConvTranspose2d) and not seen_contraction and not _features(ci)
            if not is_conv:
                raise NotImplementedError("Linear stacks are not restated here (convolutional networks only)")
            out.append(_Conv(m, mode, latent))
            seen_contraction = True
            stem_fused = is_conv and isinstance(m, nn.Conv2d) and ci <= 4 and act_next
            if mode == "bf16" and _features(channels) and not stem_fused:
                out.append(_Store())
            if stem_fused:
                out.append(nxt)
                if mode == "bf16" and _features(channels):
                    out.append(_Store())
                i += 1
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            out.append(m)
            if act_next:
                out.append(nxt)
                i += 1
            if mode == "bf16" and channels is not None and _features(channels):
                out.append(_Store())
        else:
            out.append(m)
            if mode == "bf16" and isinstance(m, (nn.ReLU, nn.LeakyReLU, nn.Tanh, nn.Sigmoid)) and channels is not None \
                    and _features(channels):
                out.append(_Store())
        i += 1
    return nn.Sequential(*out)
