"""Prints, for every golden case, the largest loss / parameter / sample deviation of the HIP engine from the
reference fixture (run on the GPU box: python tests/parity_report.py). Diagnostic companion of test_models_gpu.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402  (sets sys.path)
import golden_util as gu  # noqa: E402
from test_models_gpu import _build_model, _fit, _inject  # noqa: E402


def main(names):
    for name in names or gu.names():
        fx = gu.load(name)
        case = fx["meta"]
        nets = gu.build_nets(case)
        torch.manual_seed(case["seed"])
        np.random.seed(case["seed"])
        model = _build_model(case, nets)
        model.fixed_noise = torch.from_numpy(fx["fixed_noise"]).to("cuda")
        record = []
        _inject(model, [torch.from_numpy(z) for z in fx["Z"]], fx["alpha"], record)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            _fit(model, case, fx["X"], fx["y"])
        rows = gu.loss_rows(fx)
        worst = (0.0, "")
        for i, ((who, got), (_, want)) in enumerate(zip(record, rows)):
            for k, v in want.items():
                if k == "RealFakeRatio" or not np.isfinite(v):
                    continue
                e = abs(float(got[k]) - v) / max(abs(v), 1e-1)
                if e > worst[0]:
                    worst = (e, "step %d %s" % (i, k))
        print("%-20s losses: worst rel %.2e (%s)" % (name, worst[0], worst[1]))
        for net in fx["final"]:
            state = model.neural_nets[net].network.state_dict()
            for k, v in fx["final"][net].items():
                if k.endswith("num_batches_tracked"):
                    continue
                got = state[k].detach().cpu().numpy().astype(np.float64)
                d = np.abs(got - v)
                scale = max(np.abs(v).max(), 1e-3)
                upd = np.abs(v - fx["init"][net][k]).max()
                if d.max() / scale > 2e-5:
                    print("    %-12s %-28s max|d| %.2e  rel %.2e  n(>1e-5*scale) %d/%d  max|update| %.2e" % (
                        net, k, d.max(), d.max() / scale, int((d > 1e-5 * scale).sum()), d.size, upd))


if __name__ == "__main__":
    main(sys.argv[1:])
