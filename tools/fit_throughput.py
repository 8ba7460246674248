"""Images/s of the public `fit()` call (host batches in numpy, logging off) next to the resident-batch figure of
bench.py: what the input pipeline costs.
    python tools/fit_throughput.py [batches]
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

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    X = cfg.synthetic_images(256 * batches, (3, 64, 64), seed=1)
    kw = dict(batch_size=256, print_every=None, save_losses_every=None, enable_tensorboard=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X[:256 * 6], epochs=1, **kw)          # warm-up: eager calls + graph capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(X, epochs=1, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("fit(): %.1f images/s, %.2f ms per batch of 256 (%d batches, float32 numpy input)" %
          (256 * batches / dt, 1e3 * dt / batches, batches))


if __name__ == "__main__":
    main()
