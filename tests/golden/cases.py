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
    if k == "dcgan128":
        # the DCGAN-128 topology (one stage more than DCGAN-64: generator from 16*ngf channels, discriminator up to
        # 16*ndf) at a width the CPU reference finishes in seconds
        g = cfg.seeded_init(cfg.dcgan_generator(nz=16, ngf=4, nc=3, size=128), 1, 0.05)
        d = cfg.seeded_init(cfg.dcgan_critic(ndf=4, nc=3, size=128, sigmoid=True), 2, 0.05)
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
    if k == "vec_mlp":
        from torch import nn
        g = cfg.seeded_init(nn.Sequential(nn.Linear(8, 32), nn.LeakyReLU(0.2), nn.Linear(32, 20), nn.Tanh()), 1)
        d = cfg.seeded_init(nn.Sequential(nn.Linear(20, 32), nn.LeakyReLU(0.2), nn.Linear(32, 16), nn.LeakyReLU(0.2),
                                          nn.Linear(16, 1), nn.Identity()), 2)
        return g, d, None
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
    # Adam's first step moves every weight by lr = 1e-3 in the direction of sign(g), and at width 4 single weights
    # matter: the reference's own losses after the first update move by 2e-3 when its gradients are perturbed by 1e-6
    # of their scale (tools/oracle_sensitivity.py). The free-running comparison therefore allows 5e-3 on the losses
    # after the first optimizer step and 2e-2 on the final parameters and samples; the teacher-forced comparison of the
    # same case holds 1e-4.
    "lsgan_dcgan128": dict(cls="LSGAN", algo="lsgan", nets="dcgan128", x_dim=(3, 128, 128), z_dim=(16, 1, 1), batch=4,
                           n=8, loss_rel=5e-3, param_rel=2e-2),
    # logging passes on (AbstractGenerativeModel.py:329-351): extra train-mode forwards, alpha and latent draws
    "wgangp_logged": dict(cls="WassersteinGANGP", algo="wgangp", nets="dcgan16", x_dim=(3, 16, 16), z_dim=(16, 1, 1),
                          batch=8, n=32, optim_kwargs={"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 5e-4}},
                          fit={"print_every": 2, "save_losses_every": 1}),
    "cwgangp_logged": dict(cls="ConditionalWassersteinGANGP", algo="wgangp", nets="cond_conv", x_dim=(1, 32, 32),
                           z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=24, n_test=16, data_range=(0.0, 1.0),
                           optim_kwargs={"Generator": {"lr": 5e-4}, "Adversariat": {"lr": 1e-4}},
                           fit={"print_every": 1, "save_losses_every": 2}),
    # float64 images and float64 one-hot labels, as numpy produces them by default (the reference interpolates the
    # penalty's inputs in float64 and casts afterwards, WassersteinGANGP.py:69)
    "cwgangp_f64": dict(cls="ConditionalWassersteinGANGP", algo="wgangp", nets="cond_conv", x_dim=(1, 32, 32),
                        z_dim=(1, 4, 4), y_dim=(10,), batch=8, n=16, data_range=(0.0, 1.0), dtype="float64",
                        optim_kwargs={"Generator": {"lr": 5e-4}, "Adversariat": {"lr": 1e-4}}),
    # vector data: the reference's [B,1,1,1] alpha broadcasts against [B,D] samples to all B x B pairs
    # (WassersteinGANGP.py:68-69)
    "wgangp_vec": dict(cls="WassersteinGANGP", algo="wgangp", nets="vec_mlp", x_dim=(20,), z_dim=(8,), batch=6, n=18,
                       optim_kwargs={"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 5e-4}}),
    "clrgan_vec": dict(cls="ConditionalLRGAN", algo="lrgan", nets="cond_vec", x_dim=(20,), z_dim=(8,), y_dim=(10,), batch=8,
                       n=16),
}


def build_data(case, test=False):
    """(X, y) of a case: the training set, or with test=True the test set of the cases that log test losses."""
    lo, hi = case.get("data_range", (-1.0, 1.0))
    xs = (case["x_dim"],) if isinstance(case["x_dim"], int) else tuple(case["x_dim"])
    n = case["n_test"] if test else case["n"]
    shift = 100 if test else 0
    dtype = np.dtype(case.get("dtype", "float32"))
    if dtype == np.float64:
        # full double precision values (not float32 values widened), as an un-cast numpy array holds them
        X = np.random.default_rng(11 + shift).uniform(lo, hi, (n, *xs))
    else:
        X = cfg.synthetic_images(n, xs, lo, hi, seed=11 + shift)
    y = None
    if "y_dim" in case:
        yd = case["y_dim"]
        if isinstance(yd, int) or len(yd) == 1:
            classes = yd if isinstance(yd, int) else yd[0]
            y = cfg.one_hot_labels(n, classes, seed=12 + shift, dtype=dtype)
        else:
            y = cfg.synthetic_images(n, tuple(yd), 0.0, 1.0, seed=12 + shift)
    return X, y


