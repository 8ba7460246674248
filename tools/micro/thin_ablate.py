"""Times the three image-side contractions of DCGAN-64 at batch 256 (stem forward 3 -> 64, its weight gradient, and the
scatter-form input gradient 64 -> 3) against a library named by VEGANS_ABLATE_LIB (a build of csrc/ with parts of a
kernel switched off), f32 and bf16 feature maps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

from vegans import _hip  # noqa: E402

if os.environ.get("VEGANS_ABLATE_LIB"):
    _hip.LIB_PATH = os.environ["VEGANS_ABLATE_LIB"]
from vegans.engine import ops  # noqa: E402


def timed(fn, reps=200, warm=100):
    """Average duration of the contraction kernel itself (events attached to its dispatch), in microseconds."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ops.conv_timer = ops.ConvTimer(capacity=reps)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    rows = ops.conv_timer.summary()
    ops.conv_timer = None
    return sum(1e3 * ms for _, ms, _ in rows.values()) / sum(n for n, _, _ in rows.values())


def main():
    N, H, W, Ci, Co, k, s, p = 256, 64, 64, 3, 64, 4, 2, 1
    g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    out = []
    for mode in ("fp32", "bf16"):
        previous = ops.set_matmul_precision(mode)
        feat = torch.bfloat16 if mode == "bf16" else torch.float32
        x = torch.randn(N, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(N, Co, g.Ho, g.Wo, device="cuda").to(feat).contiguous(memory_format=torch.channels_last)
        which = os.environ.get("THIN_WHICH", "fwd,dgrad,wgrad").split(",")
        fns = {"fwd": lambda: ops.raw_conv_fwd(g, x, w, None, out_dtype=feat),
               "dgrad": lambda: ops.raw_conv_dgrad(g, gy, w, None, out_dtype=torch.float32),
               "wgrad": lambda: ops.raw_conv_wgrad(g, gy, x, (Co, Ci, k, k))}
        for name in which:
            out.append("%s %s %.1f us" % (mode, name, timed(fns[name])))
        ops.set_matmul_precision(previous)
    print(os.path.basename(os.environ.get("VEGANS_ABLATE_LIB", "product")), " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
