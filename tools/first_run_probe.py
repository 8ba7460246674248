"""Does the first training run of a process differ from later runs with the same seeds? Records the gradients every
optimizer step receives in the first iteration(s) of several identical runs and reports, per parameter, how far the
runs are apart.
    python tools/first_run_probe.py [eager|graph] [iterations]
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402


def run(batches, rows):
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    X = cfg.synthetic_images(rows * batches, (3, 64, 64), seed=1)
    seen = []
    plain = model._step

    def step(who=None, clamp=None):
        net = model.neural_nets[who]
        seen.append((who, [(n, p.grad.detach().clone(), p.detach().clone()) for n, p in net.named_parameters()]))
        plain(who=who, clamp=clamp) if clamp is not None else plain(who=who)

    model._step = step
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X, epochs=1, batch_size=rows, print_every=None, save_losses_every=None, enable_tensorboard=False)
    torch.cuda.synchronize()
    return seen


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    os.environ["VEGANS_HIP_GRAPH"] = "1" if mode == "graph" else "0"
    runs = [run(batches, 128) for _ in range(3)]
    for other in (1, 2):
        print("run 0 against run %d" % other)
        for k, ((who, a), (_, b)) in enumerate(zip(runs[0], runs[other])):
            for (name, ga, pa), (_, gb, pb) in zip(a, b):
                dg = (ga - gb).abs().max().item()
                dp = (pa - pb).abs().max().item()
                scale = gb.abs().max().item()
                if dg != 0.0 or dp != 0.0:
                    print("   step %d %-12s %-28s grad differs by %.3e (largest |grad| %.3e), weights before by %.3e" %
                          (k, who, name, dg, scale, dp))
    print("done")


if __name__ == "__main__":
    main()
