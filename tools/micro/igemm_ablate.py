"""Times the three DCGAN-64 conv geometries in the three contraction modes against a library named by VEGANS_ABLATE_LIB
(a build of csrc/ with parts of the igemm main loop switched off), to see where the time of the loop goes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

from vegans import _hip  # noqa: E402

if os.environ.get("VEGANS_ABLATE_LIB"):
    _hip.LIB_PATH = os.environ["VEGANS_ABLATE_LIB"]
from vegans.engine import ops  # noqa: E402

SHAPES = [(256, 32, 32, 64, 128), (256, 16, 16, 128, 256), (256, 8, 8, 256, 512)]


def main():
    reps = 100
    out = []
    for (N, H, W, Ci, Co) in SHAPES:
        k, s, p = 4, 2, 1
        g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
        scale = float(os.environ.get("VEGANS_ABLATE_DATA_SCALE", "1"))     # 0: all-zero operands (data-dependent power)
        x = (torch.randn(N, Ci, H, W, device="cuda") * scale).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05 * scale).contiguous(memory_format=torch.channels_last)
        gy = (torch.randn(N, Co, g.Ho, g.Wo, device="cuda") * scale).contiguous(memory_format=torch.channels_last)
        fns = {"fwd": lambda: ops.raw_conv_fwd(g, x, w, None),
               "dgrad": lambda: ops.raw_conv_dgrad(g, gy, w, None),
               "wgrad": lambda: ops.raw_conv_wgrad(g, gy, x, (Co, Ci, k, k))}
        for mode, fn in fns.items():
            for _ in range(300):          # some 40 ms of the same kernel: after an idle stretch the chip runs it ~10 % slower at first
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
            out.append("%s Ci=%d: %.1f us %.0f TF" % (mode, Ci, us, flops / us / 1e6))
    print(os.environ.get("VEGANS_ABLATE_LIB", "product"), " | ".join(out))


if __name__ == "__main__":
    main()
