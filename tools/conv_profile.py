"""Per-layer timing of the convolution kernels inside one benchmark iteration (HIP events on the launch stream).
    python tools/conv_profile.py [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402
from vegans.engine import ops, parallel  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    model._create_steps(None)
    model.train()
    parallel.context(model)
    X = model._to_device(cfg.synthetic_images(256, (3, 64, 64), seed=1))
    Z = torch.randn(256, 100, 1, 1, device="cuda")
    for _ in range(3):
        model.train_on_batch(X, Z)
    timer = ops.ConvTimer()
    ops.conv_timer = timer
    torch.cuda.synchronize()
    for _ in range(steps):
        model.train_on_batch(X, Z)
    torch.cuda.synchronize()
    ops.conv_timer = None
    agg = {}
    for _, g, start, stop, mode in timer.records:
        key = (mode, g.N, g.H, g.W, g.Ci, g.Co, g.kh, g.sh, g.ph)
        n, ms = agg.get(key, (0, 0.0))
        agg[key] = (n + 1, ms + start.elapsed_time(stop))
    total = sum(v[1] for v in agg.values())
    print("%-6s %-34s %6s %9s %8s %7s" % ("mode", "geometry N,H,W,Ci,Co,k,s,p", "n/it", "us/launch", "TFLOP/s", "share"))
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        mode, N, H, W, Ci, Co, k, s, p = key
        Ho = (H + 2 * p - k) // s + 1
        flops = 2.0 * N * Ho * Ho * Co * k * k * Ci
        print("%-6s %-34s %6.1f %9.1f %8.1f %6.1f%%" % (("fwd", "dgrad", "wgrad")[mode], str(key[1:]), n / steps,
                                                       1e3 * ms / n, flops / (ms / n * 1e-3) / 1e12, 100 * ms / total))
    print("conv total per iteration: %.2f ms" % (total / steps))


if __name__ == "__main__":
    main()
