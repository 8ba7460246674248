"""Accuracy and speed of the three multiplier modes of the convolution contractions on the DCGAN-64 layers.

For every layer and contraction (fwd / dgrad / wgrad) the result of each mode ("fp32" = f32 MFMA, "fp32_split" =
three-term bf16 split with six products, "bf16") and of torch's own f32 convolution is compared with a float64
evaluation of the same contraction:  err = max |y - y64| / max |y64|  and  rms = ||y - y64|| / ||y64||.
    python tools/precision_study.py [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vegans.engine import ops  # noqa: E402

# (H, Ci, Co) of the k=4, s=2, p=1 convolutions of the DCGAN-64 critic at batch 256; the generator's transposed
# convolutions are the same contractions with the roles of fwd and dgrad exchanged.
LAYERS = [(32, 64, 128), (16, 128, 256), (8, 256, 512)]
N = 256


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        out = fn()
    stop.record()
    torch.cuda.synchronize()
    return out, 1e3 * start.elapsed_time(stop) / reps


def errors(y, y64):
    d = (y.double() - y64)
    return (d.abs().max() / y64.abs().max()).item(), (d.norm() / y64.norm()).item()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    print("%-6s %-16s %-11s %9s %8s %10s %10s" % ("mode", "H,Ci,Co", "precision", "us", "TFLOP/s", "max err", "rms err"))
    for H, Ci, Co in LAYERS:
        g = ops.Geom(N, H, H, Ci, Co, 4, 4, 2, 2, 1, 1)
        cl = torch.channels_last
        x = torch.randn(N, Ci, H, H, device="cuda").contiguous(memory_format=cl)
        w = (torch.randn(Co, Ci, 4, 4, device="cuda") * 0.05).contiguous(memory_format=cl)
        gy = torch.randn(N, Co, g.Ho, g.Wo, device="cuda").contiguous(memory_format=cl)
        x64, w64, gy64 = x.double().requires_grad_(), w.double().requires_grad_(), gy.double()
        y64 = F.conv2d(x64, w64, None, 2, 1)
        gx64, gw64 = torch.autograd.grad(y64, (x64, w64), gy64)
        y64 = y64.detach()
        flops = 2.0 * N * g.Ho * g.Wo * Co * 16 * Ci
        xt, wt = x.clone().requires_grad_(), w.clone().requires_grad_()
        cases = {
            "fwd": (lambda: ops.raw_conv_fwd(g, x, w, None), y64, lambda: F.conv2d(x, w, None, 2, 1)),
            "dgrad": (lambda: ops.raw_conv_dgrad(g, gy, w, None), gx64,
                      lambda: torch.autograd.grad(F.conv2d(xt, w, None, 2, 1), xt, gy)[0]),
            "wgrad": (lambda: ops.raw_conv_wgrad(g, gy, x, (Co, Ci, 4, 4)), gw64,
                      lambda: torch.autograd.grad(F.conv2d(x, wt, None, 2, 1), wt, gy)[0]),
        }
        for mode, (fn, ref, torch_fn) in cases.items():
            for prec in ("fp32", "fp32_split", "bf16_compute"):
                previous = ops.set_matmul_precision(prec)
                try:
                    out, us = timed(fn, reps)
                finally:
                    ops.set_matmul_precision(previous)
                e_max, e_rms = errors(out, ref)
                print("%-6s %-16s %-11s %9.1f %8.1f %10.2e %10.2e" % (mode, (H, Ci, Co), prec, us, flops / us / 1e6,
                                                                     e_max, e_rms))
            out = torch_fn()
            e_max, e_rms = errors(out, ref)
            print("%-6s %-16s %-11s %9s %8s %10.2e %10.2e" % (mode, (H, Ci, Co), "torch f32", "-", "-", e_max, e_rms))


if __name__ == "__main__":
    main()
