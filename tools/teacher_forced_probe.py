"""Where do the engine and the fp32 CPU oracle disagree, and who is closer to the truth?

Full-width DCGAN-64 WGAN-GP at batch 64, teacher-forced before every sub-step (as the parity test does). Per
iteration it prints, for the generator loss and the gradient penalty, the error of the engine and of the fp32 oracle
against the same sub-step evaluated by the oracle in FLOAT64 on the same weights, and for the generator's first layer
the gradient error of both fp32 implementations against float64 together with the share of elements whose update
ends up further than 1e-4 of the tensor's scale from the fp32 oracle's.
    python tools/teacher_forced_probe.py [iterations]
"""
import copy
import os
import sys
from unittest import mock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_configs as cfg  # noqa: E402
from oracle.restated import OracleGAN  # noqa: E402
from test_models_gpu import _build_model, _engine_sub_step, _sync_from_oracle  # noqa: E402


def f64_copy(oracle):
    G = copy.deepcopy(oracle.nets["Generator"]).double()
    D = copy.deepcopy(oracle.nets["Adversariat"]).double()
    o = object.__new__(OracleGAN)
    o.__dict__.update(oracle.__dict__)
    o.nets = {"Generator": G, "Adversariat": D}
    return o


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = 64
    torch.manual_seed(3)
    np.random.seed(3)
    nets = lambda: (cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 64), 1), cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64), 2))
    okw = {"Generator": {"lr": 1e-4}, "Adversariat": {"lr": 5e-5}}
    Go, Do = nets()
    oracle = OracleGAN("wgangp", Go, Do, (3, 64, 64), (100, 1, 1), optim_kwargs=copy.deepcopy(okw), fixed_noise_size=4)
    case = dict(algo="wgangp", cls="WassersteinGANGP", x_dim=(3, 64, 64), z_dim=(100, 1, 1), optim_kwargs=okw,
                fixed_noise_size=4)
    model = _build_model(case, nets() + (None,))
    model.use_step_graph = False
    model._create_steps(None)
    model.train()
    oracle._set_mode(True)
    pool = [cfg.synthetic_images(B, (3, 64, 64), seed=40 + i) for i in range(10)]
    zgen = torch.Generator().manual_seed(10)
    grads_m = {}
    plain_step = model._step

    def step(who=None):
        grads_m[who] = [p.grad.detach().cpu().double() for p in model.neural_nets[who].network.parameters()]
        plain_step(who=who)

    model._step = step
    print("it | G loss: eng-f64 f32-f64 | penalty: value eng-f64 f32-f64 (relative) | G.0.weight grad: max|g| med|g| "
          "eng-f64 f32-f64 (of max) | share of G.0.weight beyond 1e-4 after the step")
    for it in range(iters):
        X = torch.from_numpy(pool[it % len(pool)])
        Z = torch.randn(B, 100, 1, 1, generator=zgen)
        A = np.random.default_rng(1000 + it).random((B, 1, 1, 1))
        model._draw_alpha = lambda n, A=A: torch.from_numpy(A.astype(np.float32)).to("cuda")
        Xd, Zd = model._to_device(X.numpy()), Z.to("cuda")
        row = {}
        for who in ("Generator", "Adversariat"):
            _sync_from_oracle(oracle, model)
            o64 = f64_copy(oracle)
            o64._set_mode(True)
            # the restatement casts with .float() where the reference does; the float64 evaluation keeps double
            with mock.patch.object(torch.Tensor, "float", lambda self: self.double()):
                l64 = o64.losses(who, X.double(), Z.double(), None, A if who == "Adversariat" else None)
            g64 = torch.autograd.grad(l64[who], list(o64.nets[who].parameters()))
            want = oracle.sub_step(who, X, Z.clone(), None, A if who == "Adversariat" else None)
            g32 = [p.grad.detach().double() for p in oracle.nets[who].parameters()]
            got = _engine_sub_step(model, who, Xd, Zd)
            if who == "Generator":
                t = float(l64["Generator"])
                row["g"] = (abs(float(got["Generator"]) - t) / abs(t), abs(want["Generator"] - t) / abs(t))
                ref = g64[0]
                mx = float(ref.abs().max())
                row["grad"] = (mx, float(ref.abs().median()), float((grads_m[who][0] - ref).abs().max()) / mx,
                               float((g32[0] - ref).abs().max()) / mx)
                wo = oracle.nets[who][0].weight.detach().double()
                wm = model.neural_nets[who].network[0].weight.detach().cpu().double()
                row["share"] = float(((wo - wm).abs() > 1e-4 * max(float(wo.abs().max()), 1e-2)).double().mean())
            else:
                t = float(l64["Adversariat_grad"])
                row["gp"] = (t, abs(float(got["Adversariat_grad"]) - t) / abs(t), abs(want["Adversariat_grad"] - t) / abs(t))
        print("%3d | %.1e %.1e | %.5f %.1e %.1e | %.2e %.2e %.1e %.1e | %.1e" % (
            (it,) + row["g"] + row["gp"] + row["grad"] + (row["share"],)), flush=True)


if __name__ == "__main__":
    main()
