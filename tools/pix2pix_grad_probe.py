"""Full-size pix2pix (U-Net-256 + PatchGAN, 3x256x256, batch 2): gradients of one generator sub-step from the fp32
oracle, from the same oracle code in float64, and from the engine, layer by layer. Tells rounding of fp32 arithmetic
on an ill-conditioned input (BatchNorm over 8 ... 32 nearly equal samples in the bottleneck blocks) apart from an
error of the engine: both fp32 columns should sit at a similar distance from the float64 column."""
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
    data = sys.argv[1] if len(sys.argv) > 1 else "noise"
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
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if data == "noise":
        X = torch.from_numpy(cfg.synthetic_images(B, (3, 256, 256), 0.0, 1.0, seed=70))
        y = torch.from_numpy(cfg.synthetic_images(B, (3, 256, 256), 0.0, 1.0, seed=71))
    else:   # smooth images that differ from each other at every scale
        g = torch.Generator().manual_seed(5)
        low = torch.rand(2 * B, 3, 8, 8, generator=g)
        big = torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
        big = (big + 0.1 * torch.rand(2 * B, 3, 256, 256, generator=g)).clamp(0, 1)
        X, y = big[:B].contiguous(), big[B:].contiguous()
    Z = torch.randn(B, 2, generator=torch.Generator().manual_seed(12))
    who = sys.argv[3] if len(sys.argv) > 3 else "Generator"
    advance = sys.argv[4] if len(sys.argv) > 4 else "none"      # "oracle": one generator step on the oracle first;
    batch = (model._to_device(X.numpy()), model._to_device(y.numpy()), Z.to("cuda"))     # "both": on the engine too
    if advance in ("oracle", "both"):
        T._sync_from_oracle(oracle, model)
        oracle.sub_step("Generator", X, Z.clone(), y, None)
        if advance == "both":
            T._engine_sub_step(model, "Generator", *batch)
    T._sync_from_oracle(oracle, model)
    params = list(oracle.nets[who].parameters())
    names = [k for k, _ in oracle.nets[who].named_parameters()]
    loss32 = oracle.losses(who, X, Z.clone(), y, None)[who]
    g32 = torch.autograd.grad(loss32, params)
    twin = object.__new__(type(oracle))
    twin.__dict__.update(oracle.__dict__)
    twin.nets = {k: copy.deepcopy(v).double() for k, v in oracle.nets.items()}
    twin._set_mode(True)
    with mock.patch.object(torch.Tensor, "float", lambda self: self.double()):
        loss64 = twin.losses(who, X.double(), Z.double(), y.double(), None)[who]
    g64 = torch.autograd.grad(loss64, list(twin.nets[who].parameters()))
    got = T._engine_sub_step(model, who, *batch)
    ge = [p.grad.detach().cpu().double() for p in model.neural_nets[who].network.parameters()]
    print("data = %s; loss: oracle fp32 %.8f  float64 %.8f  engine %.8f" % (data, float(loss32), float(loss64), float(got[who])))
    print("largest element error over the tensor's largest magnitude | l2 norm of the error over l2 norm of the tensor")
    print("%-22s %12s %12s %12s | %12s %12s %12s" % ("parameter", "fp32 vs f64", "engine vs f64", "engine vs fp32",
                                                      "fp32 vs f64", "engine vs f64", "engine vs fp32"))
    for k, a, b, c in zip(names, g32, g64, ge):
        a = a.double()
        s = max(float(b.abs().max()), 1e-30)
        n = max(float(b.norm()), 1e-30)
        fit = float((c * b).sum() / (b * b).sum())      # best-fit scale of the engine's gradient against float64
        print("%-22s %12.2e %12.2e %12.2e | %12.2e %12.2e %12.2e | scale %.6f, residual %.2e"
              % (k, float((a - b).abs().max()) / s, float((c - b).abs().max()) / s, float((c - a).abs().max()) / s,
                 float((a - b).norm()) / n, float((c - b).norm()) / n, float((c - a).norm()) / n, fit,
                 float((c - fit * b).norm()) / n))


if __name__ == "__main__":
    main()
