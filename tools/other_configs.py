"""Throughput of the other benchmark configurations' per-GPU shapes (not the headline metric): LSGAN on DCGAN-128 at
batch 256 (the per-GPU share of the 8-GPU configuration) and ConditionalPix2Pix (U-Net-256 + PatchGAN) at 3x256x256, batch 64.
    python tools/other_configs.py [steps]
"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_configs as cfg  # noqa: E402
import vegans.GAN as gan  # noqa: E402


def timed(model, args, steps):
    for _ in range(4):
        model.train_on_batch(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_on_batch(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.manual_seed(0)
    np.random.seed(0)
    # secondary variants of the headline configuration: BatchNorm-free critic (WGAN-GP paper practice), Adam
    for label, bn, okw in (("BatchNorm-free critic, RMSprop", False, None),
                           ("BatchNorm critic, Adam(1e-4, betas=(0.5, 0.9))", True,
                            {"Generator": {"lr": 1e-4, "betas": (0.5, 0.9)}, "Adversariat": {"lr": 1e-4, "betas": (0.5, 0.9)}})):
        with contextlib.redirect_stdout(io.StringIO()):
            G = cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 64), 1)
            D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64, batch_norm=bn), 2)
            kw = dict(optim_kwargs=okw or {"Generator": {"lr": 1e-4}, "Adversariat": {"lr": 5e-5}})
            if okw is not None:
                kw["optim"] = {"Generator": torch.optim.Adam, "Adversariat": torch.optim.Adam}
            m = gan.WassersteinGANGP(G, D, x_dim=(3, 64, 64), z_dim=(100, 1, 1), folder=None, fixed_noise_size=4, **kw)
        m._create_steps(None)
        m.train()
        X = m._to_device(cfg.synthetic_images(256, (3, 64, 64), seed=1))
        Z = torch.randn(256, 100, 1, 1, device="cuda")
        dt = timed(m, (X, Z), steps)
        print("WGAN-GP DCGAN-64, batch 256, %s, fp32: %.2f ms/iteration, %.0f images/s" % (label, 1e3 * dt, 256 / dt))
        del m, X, Z
        torch.cuda.empty_cache()

    with contextlib.redirect_stdout(io.StringIO()):
        G = cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 128), 1)
        D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 128, batch_norm=True, sigmoid=True), 2)
        m = gan.LSGAN(G, D, x_dim=(3, 128, 128), z_dim=(100, 1, 1), folder=None, fixed_noise_size=4)
    m._create_steps(None)
    m.train()
    X = m._to_device(cfg.synthetic_images(256, (3, 128, 128), seed=1))
    Z = torch.randn(256, 100, 1, 1, device="cuda")
    from vegans.engine import ops
    for prec in ("fp32", "fp32_split", "bf16", "bf16_compute"):
        previous = ops.set_matmul_precision(prec)
        try:
            dt = timed(m, (X, Z), steps)
        finally:
            ops.set_matmul_precision(previous)
        print("LSGAN DCGAN-128, batch 256, Adam, %s: %.2f ms/iteration, %.0f images/s" % (prec, 1e3 * dt, 256 / dt))
    del m, X, Z
    torch.cuda.empty_cache()

    with contextlib.redirect_stdout(io.StringIO()):
        G = cfg.seeded_init(cfg.UNet(in_channels=4 + 3, out_channels=3, width=64, depth=8), 1, 0.02)
        D = cfg.seeded_init(cfg.patchgan_discriminator(in_channels=6, width=64), 2, 0.02)
        m = gan.ConditionalPix2Pix(G, D, x_dim=(3, 256, 256), z_dim=4, y_dim=(3, 256, 256), folder=None,
                                   fixed_noise_size=2)
    m._create_steps(None)
    m.train()
    X = m._to_device(cfg.synthetic_images(64, (3, 256, 256), 0.0, 1.0, seed=1))
    y = m._to_device(cfg.synthetic_images(64, (3, 256, 256), 0.0, 1.0, seed=2))
    Z = torch.randn(64, 4, device="cuda")
    for prec in ("fp32", "fp32_split", "bf16", "bf16_compute"):
        previous = ops.set_matmul_precision(prec)
        try:
            dt = timed(m, (X, y, Z), steps)
        finally:
            ops.set_matmul_precision(previous)
        print("ConditionalPix2Pix (U-Net-256 with skip connections, 8 down / 8 up, width 64; 70x70 PatchGAN), 3x256x256, "
              "batch 64, Adam, %s: %.2f ms/iteration, %.0f images/s" % (prec, 1e3 * dt, 64 / dt))

if __name__ == "__main__":
    main()
