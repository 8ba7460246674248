"""How long the host needs to enqueue one benchmark iteration versus how long the GPU needs to run it, plus a
cProfile of the enqueue path.
    python tools/host_overhead.py [steps]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402
from vegans.engine import parallel  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    model._create_steps(None)
    model.train()
    parallel.context(model)
    X = model._to_device(cfg.synthetic_images(256, (3, 64, 64), seed=1))
    Z = torch.randn(256, 100, 1, 1, device="cuda")
    for _ in range(5):
        model.train_on_batch(X, Z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_on_batch(X, Z)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.2f ms/step, enqueue+drain %.2f ms/step" % (1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps))
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(steps):
        model.train_on_batch(X, Z)
    prof.disable()
    torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
