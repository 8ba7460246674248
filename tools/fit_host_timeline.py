"""Where the host spends its time inside `fit()`, and whether it stays ahead of the GPU.
    python tools/fit_host_timeline.py [batches] [f64]
Wraps the pieces of one iteration (upload of the next batch, latent draw, graph replay) with wall-clock timers and,
right before every graph launch, asks whether the previous iteration has already finished on the device (then the GPU
sat idle waiting for the host).
"""
import contextlib
import io
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402
from vegans.engine import graph  # noqa: E402

spent = defaultdict(list)


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        spent[name].append(time.perf_counter() - t0)
        return out
    return wrapper


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    X = cfg.synthetic_images(256 * batches, (3, 64, 64), seed=1)
    if "f64" in sys.argv[2:]:
        X = X.astype(np.float64)      # numpy's default dtype, what users of the reference typically pass
    print("input dtype", X.dtype)
    kw = dict(batch_size=256, print_every=None, save_losses_every=None, enable_tensorboard=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X[:256 * 6], epochs=1, **kw)
    torch.cuda.synchronize()

    model._to_device = timed("upload (pin + async copy + layout)", model._to_device)
    model._draw_latent = timed("latent draw", model._draw_latent)
    idle = []
    last = [None]
    plain_replay = graph._StepGraph.replay

    def replay(self, mdl, inputs):
        t0 = time.perf_counter()
        if last[0] is not None:
            idle.append(last[0].query())
        plain_replay(self, mdl, inputs)
        ev = torch.cuda.Event()
        ev.record()
        last[0] = ev
        spent["replay (static copies + host script + launch)"].append(time.perf_counter() - t0)

    graph._StepGraph.replay = replay
    marks = []
    plain_train = model.train_on_batch

    def train(Xb, Zb):
        marks.append(time.perf_counter())
        plain_train(Xb, Zb)

    model.train_on_batch = train
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        model.fit(X, epochs=1, **kw)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("fit(): %.2f ms per batch; the host finished issuing after %.2f ms per batch" %
          (1e3 * dt / batches, 1e3 * t_host / batches))
    for name, v in spent.items():
        v = np.array(v) * 1e3
        print("  %-48s n=%3d  mean %.3f ms  median %.3f  max %.3f" % (name, len(v), v.mean(), np.median(v), v.max()))
    gaps = np.diff(np.array(marks)) * 1e3
    print("  host time between iteration starts: mean %.3f ms  median %.3f  max %.3f" %
          (gaps.mean(), np.median(gaps), gaps.max()))
    print("  previous iteration already finished when the next graph was launched: %d of %d" %
          (sum(idle), len(idle)))


if __name__ == "__main__":
    main()
