"""Teacher-forced gradient comparison for one golden case: same initial weights on the CPU oracle and on the engine,
the first sub-step of every network, every parameter gradient element by element.
    python tools/grad_check_case.py <case name> [...]
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from oracle.restated import OracleGAN  # noqa: E402
from test_models_gpu import _build_model  # noqa: E402


def main(name):
    fx = gu.load(name)
    case = fx["meta"]
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    G, D, E = gu.build_nets(case)
    oracle = OracleGAN(case["algo"], G, D, case["x_dim"], case["z_dim"], y_dim=case.get("y_dim"), encoder=E,
                       optim_kwargs=copy.deepcopy(case.get("optim_kwargs")), fixed_noise_size=case["fixed_noise_size"])
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    model = _build_model(case, gu.build_nets(case))
    B = case["batch"]
    X, y = fx["X"][:B], (fx["y"][:B] if fx["y"] is not None else None)
    Z = torch.from_numpy(fx["Z"][0])
    A = fx["alpha"][0] if fx["alpha"] else None
    oracle._set_mode(True)
    model.train()
    if A is not None:
        model._draw_alpha = lambda n: torch.from_numpy(np.asarray(A, dtype=np.float32)).to("cuda")
    Xm, Zm = model._to_device(X), Z.to("cuda")
    ym = model._to_device(y) if y is not None else None
    for who in oracle.nets:
        lo = oracle.losses(who, torch.from_numpy(X), Z.clone(), torch.from_numpy(y) if y is not None else None, A)
        oracle.optimizers[who].zero_grad()
        lo[who].backward()
        model._losses = {}
        with model._training_only(who):
            if ym is None:
                model.calculate_losses(X_batch=Xm, Z_batch=Zm, who=who)
            else:
                model.calculate_losses(X_batch=Xm, Z_batch=Zm, y_batch=ym, who=who)
            model._zero_grad(who=who)
            model._backward(who=who)
        print("%s %s loss engine %.8g oracle %.8g" % (name, who, float(model._losses[who]), float(lo[who])))
        for (k, po), (_, pm) in zip(oracle.nets[who].named_parameters(), model.neural_nets[who].network.named_parameters()):
            if po.grad is None or pm.grad is None:
                print("   %-28s grad missing: oracle %s engine %s" % (k, po.grad is None, pm.grad is None))
                continue
            a, b = pm.grad.detach().cpu().double(), po.grad.double()
            scale = float(b.abs().max())
            print("   %-28s max|grad| %.3e  max err %.3e  (%.1e of scale)  shape %s" % (
                k, scale, float((a - b).abs().max()), float((a - b).abs().max()) / max(scale, 1e-30), tuple(b.shape)))


if __name__ == "__main__":
    for n in sys.argv[1:]:
        main(n)
