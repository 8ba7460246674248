"""Parity cases shared by the golden-vector generator and the tests: network builders, constructor arguments and
synthetic data of each fixture (data and plain torch.nn definitions only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench_configs as cfg  # noqa: E402

FIXED = 4  # fixed_noise_size in every case


def build_nets(case):
    """Fresh, deterministically initialised networks of a case (stock torch modules)."""
    k = case["nets"]
    if k == "dcgan16":
        g = cfg.seeded_init(cfg.dcgan_generator(nz=16, ngf=8, nc=3, size=16), 1, 0.08)
        d = cfg.seeded_init(cfg.dcgan_critic(ndf=8, nc=3, size=16, sigmoid=case.get("sigmoid", False)), 2, 0.08)
        return g, d, None
    if k == "dcgan64":
        g = cfg.seeded_init(cfg.dcgan_generator(nz=32, ngf=8, nc=3, size=64), 1, 0.05)
        d = cfg.seeded_init(cfg.dcgan_critic(ndf=8, nc=3, size=64), 2, 0.05)
        return g, d, None
    if k == "mlp":
        g = cfg.seeded_init(cfg.mlp_generator(z_dim=16, im_dim=(1, 28, 28), width=8), 1)
        d = cfg.seeded_init(cfg.mlp_discriminator(im_dim=(1, 28, 28), width=8), 2)
        e = cfg.seeded_init(cfg.mlp_encoder(im_dim=(1, 28, 28), z_dim=16, width=8), 3)
        return g, d, e
    if k == "cond_conv":
        g = cfg.seeded_init(cfg.cond_conv_generator(in_channels=11, ngf=4), 1)
        d = cfg.seeded_init(cfg.cond_conv_critic(in_channels=11, ncf=4, sigmoid=case.get("sigmoid", False)), 2)
        return g, d, None
    if k == "img2img":
        g = cfg.seeded_init(cfg.img2img_generator(in_channels=2 + 3, out_channels=3, width=4), 1)
        d = cfg.seeded_init(cfg.patch_discriminator(in_channels=6, width=4), 2)
        return g, d, None
    if k == "unet":
        g = cfg.seeded_init(cfg.UNet(in_channels=2 + 3, out_channels=3, width=4, depth=5), 1)
        d = cfg.seeded_init(cfg.patchgan_discriminator(in_channels=6, width=4), 2)
        return g, d, None
    if k == "cond_vec":
        from torch import nn
        g = cfg.seeded_init(nn.Sequential(nn.Linear(18, 32), nn.LeakyReLU(0.2), nn.BatchNorm1d(32), nn.Linear(32, 20),
                                          nn.Tanh()), 1)
        d = cfg.seeded_init(nn.Sequential(nn.Linear(30, 32), nn.LeakyReLU(0.2), nn.Linear(32, 1), nn.Sigmoid()), 2)
        e = cfg.seeded_init(nn.Sequential(nn.Linear(20, 32), nn.ReLU(), nn.Linear(32, 8), nn.Identity()), 3)
        return g, d, e
    raise KeyError(k)


CASES = {
    "wgangp_dcgan16": dict(cls="WassersteinGANGP", algo="wgangp", nets="dcgan16", x_dim=(3, 16, 16), z_dim=(16, 1, 1),
                           batch=8, n=24, optim_kwargs={"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 5e-4}}),
    "wgangp_dcgan64": dict(cls="WassersteinGANGP", algo="wgangp", nets="dcgan64", x_dim=(3, 64, 64), z_dim=(32, 1, 1),
                           batch=8, n=16, optim_kwargs={"Generator": {"lr": 1e-4}, "Adversariat": {"lr": 5e-5}}),
    "wgan_clip_dcgan16": dict(cls="WassersteinGAN", algo="wgan", nets="dcgan16", x_dim=(3, 16, 16), z_dim=(16, 1, 1),
                              batch=8, n=20, steps={"Adversariat": 3},
                              optim_kwargs={"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 1e-3}}),
    "lsgan_dcgan16": dict(cls="LSGAN", algo="lsgan", nets="dcgan16", sigmoid=True, x_dim=(3, 16, 16), z_dim=(16, 1, 1),
                          batch=8, n=16),
    "vanilla_mlp": dict(cls="VanillaGAN", algo="vanilla", nets="mlp", x_dim=(1, 28, 28), z_dim=16, batch=16, n=48,
                        data_range=(0.0, 1.0)),
    "lrgan_mlp": dict(cls="LRGAN", algo="lrgan", nets="mlp", x_dim=(1, 28, 28), z_dim=16, batch=16, n=32,
                      data_range=(0.0, 1.0)),
    "cwgan_clip_onehot": dict(cls="ConditionalWassersteinGAN", algo="wgan", nets="cond_conv", x_dim=(1, 32, 32),
                              z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=16, steps={"Adversariat": 5, "Generator": 1},
                              optim_kwargs={"Generator": {"lr": 5e-4}, "Adversariat": {"lr": 1e-4}},
                              data_range=(0.0, 1.0)),
    "cwgangp_onehot": dict(cls="ConditionalWassersteinGANGP", algo="wgangp", nets="cond_conv", x_dim=(1, 32, 32),
                           z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=16, data_range=(0.0, 1.0),
                           optim_kwargs={"Generator": {"lr": 5e-4}, "Adversariat": {"lr": 1e-4}}),
    "cvanilla_onehot": dict(cls="ConditionalVanillaGAN", algo="vanilla", nets="cond_conv", sigmoid=True,
                            x_dim=(1, 32, 32), z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=16, data_range=(0.0, 1.0)),
    "clsgan_onehot": dict(cls="ConditionalLSGAN", algo="lsgan", nets="cond_conv", sigmoid=True, x_dim=(1, 32, 32),
                          z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=24, data_range=(0.0, 1.0),
                          optim_kwargs={"Generator": {"lr": 5e-4}, "Adversariat": {"lr": 2e-4}}),
    "pix2pix_small": dict(cls="ConditionalPix2Pix", algo="pix2pix", nets="img2img", x_dim=(3, 16, 16), z_dim=2,
                          y_dim=(3, 16, 16), batch=4, n=8, data_range=(0.0, 1.0)),
    "pix2pix_unet": dict(cls="ConditionalPix2Pix", algo="pix2pix", nets="unet", x_dim=(3, 32, 32), z_dim=2,
                         y_dim=(3, 32, 32), batch=4, n=8, data_range=(0.0, 1.0)),
    "clrgan_vec": dict(cls="ConditionalLRGAN", algo="lrgan", nets="cond_vec", x_dim=(20,), z_dim=(8,), y_dim=(10,), batch=8,
                       n=16),
}


def build_data(case):
    lo, hi = case.get("data_range", (-1.0, 1.0))
    xs = (case["x_dim"],) if isinstance(case["x_dim"], int) else tuple(case["x_dim"])
    X = cfg.synthetic_images(case["n"], xs, lo, hi, seed=11)
    y = None
    if "y_dim" in case:
        yd = case["y_dim"]
        if isinstance(yd, int) or len(yd) == 1:
            classes = yd if isinstance(yd, int) else yd[0]
            y = cfg.one_hot_labels(case["n"], classes, seed=12)
        else:
            y = cfg.synthetic_images(case["n"], tuple(yd), 0.0, 1.0, seed=12)
    return X, y


