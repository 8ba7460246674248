"""Forward precision of the full-size U-Net generator (train mode, batch 2): engine vs fp32 oracle vs float64, at the
initial weights and after one Adam step of the oracle."""
import copy
import os
import sys
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_configs as cfg  # noqa: E402
from oracle.restated import OracleGAN  # noqa: E402
import test_models_gpu as T  # noqa: E402


def main():
    case = dict(cls="ConditionalPix2Pix", algo="pix2pix", x_dim=(3, 256, 256), z_dim=2, y_dim=(3, 256, 256),
                fixed_noise_size=2)

    def nets():
        return (cfg.seeded_init(cfg.UNet(in_channels=2 + 3, out_channels=3, width=64, depth=8), 1, 0.05),
                cfg.seeded_init(cfg.patchgan_discriminator(in_channels=6, width=64), 2, 0.05), None)

    torch.manual_seed(3)
    np.random.seed(3)
    Go, Do, _ = nets()
    oracle = OracleGAN("pix2pix", Go, Do, case["x_dim"], case["z_dim"], y_dim=case["y_dim"], fixed_noise_size=2)
    torch.manual_seed(3)
    np.random.seed(3)
    model = T._build_model(case, nets())
    model.use_step_graph = False
    model._create_steps(None)
    model.train()
    oracle._set_mode(True)
    B = 2
    g = torch.Generator().manual_seed(5)
    low = torch.rand(2 * B, 3, 8, 8, generator=g)
    big = torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
    big = (big + 0.1 * torch.rand(2 * B, 3, 256, 256, generator=g)).clamp(0, 1)
    X, y = big[:B].contiguous(), big[B:].contiguous()
    Z = torch.randn(B, 2, generator=torch.Generator().manual_seed(12))
    yd, Zd = model._to_device(y.numpy()), Z.to("cuda")

    def report(tag):
        T._sync_from_oracle(oracle, model)
        with torch.no_grad():
            f32 = oracle.generate(Z.clone(), y).double()
            twin = {k: copy.deepcopy(v).double() for k, v in oracle.nets.items()}
            with mock.patch.object(torch.Tensor, "float", lambda self: self.double()):
                from oracle.restated import concatenate
                f64 = twin["Generator"].train()(concatenate(Z.double(), y.double()))
            fe = model.generate(y=yd, z=Zd).detach().cpu().double()
        n = float(f64.norm())
        print("%s: fake images, l2 error over l2 norm: fp32 oracle vs f64 %.2e, engine vs f64 %.2e, engine vs fp32 %.2e; "
              "largest element error: %.2e / %.2e / %.2e"
              % (tag, float((f32 - f64).norm()) / n, float((fe - f64).norm()) / n, float((fe - f32).norm()) / n,
                 float((f32 - f64).abs().max()), float((fe - f64).abs().max()), float((fe - f32).abs().max())))

    report("initial weights")
    oracle.sub_step("Generator", X, Z.clone(), y, None)
    report("after one generator step")
    oracle.sub_step("Adversariat", X, Z.clone(), y, None)
    oracle.sub_step("Generator", X, Z.clone(), y, None)
    report("after two generator steps")


if __name__ == "__main__":
    main()
