"""Runs single convolution contractions in a loop (for rocprofv3 counter passes and A/B timing of kernel variants).
    python tools/conv_micro.py MODE N H W Ci Co k s p [reps]      MODE = fwd | dgrad | wgrad
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

from vegans.engine import ops  # noqa: E402


def main():
    mode = sys.argv[1]
    N, H, W, Ci, Co, k, s, p = (int(v) for v in sys.argv[2:10])
    reps = int(sys.argv[10]) if len(sys.argv) > 10 else 20
    g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    x = torch.randn(N, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, Co, g.Ho, g.Wo, device="cuda").contiguous(memory_format=torch.channels_last)
    fn = {"fwd": lambda: ops.raw_conv_fwd(g, x, w, None),
          "dgrad": lambda: ops.raw_conv_dgrad(g, gy, w, None),
          "wgrad": lambda: ops.raw_conv_wgrad(g, gy, x, (Co, Ci, k, k))}[mode]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize()
    us = 1e3 * start.elapsed_time(stop) / reps
    flops = 2.0 * N * g.Ho * g.Wo * Co * k * k * Ci
    print("%s %s: %.1f us  %.1f TFLOP/s" % (mode, sys.argv[2:10], us, flops / us / 1e6))


if __name__ == "__main__":
    main()
