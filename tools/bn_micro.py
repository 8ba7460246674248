"""Times the BatchNorm reduction / apply kernels on the DCGAN-64 activation shapes (batch 256) and prints achieved GB/s
against their algorithmic bytes.
    python tools/bn_micro.py
"""
import ctypes
import os
import sys
from ctypes import c_float, c_size_t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

from vegans import _hip  # noqa: E402
from vegans._hip import _ptr, _stream, call  # noqa: E402

SHAPES = [(256 * 32 * 32, 64), (256 * 16 * 16, 128), (256 * 8 * 8, 256), (256 * 4 * 4, 512)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    dev = torch.device("cuda")
    for m, c in SHAPES:
        x = torch.randn(m, c, device=dev)
        gy = torch.randn(m, c, device=dev)
        y = torch.randn(m, c, device=dev)
        mean = torch.zeros(c, device=dev)
        invstd = torch.ones(c, device=dev)
        gamma = torch.ones(c, device=dev)
        sums = torch.empty(2, c, device=dev)
        gx = torch.empty_like(x)
        nbytes = _hip.load().vg_bn_workspace_bytes(m, c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        wsp = ctypes.c_void_p(ws.data_ptr())
        t_stats = timed(lambda: call("vg_bn_stats", _ptr(x), m, c, c_float(1e-5), c_float(0.1), _ptr(mean), _ptr(invstd),
                                     None, None, None, wsp, c_size_t(nbytes), _stream()))
        beta = torch.zeros(c, device=dev)
        # slope mask read off the activation's output (three activation-sized reads) ...
        t_red = timed(lambda: call("vg_bn_bwd_reduce", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums), wsp,
                                   c_size_t(nbytes), _ptr(y), _hip.ACT_LRELU, c_float(0.2), None, None, _stream()))
        t_app = timed(lambda: call("vg_bn_bwd_apply", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums),
                                   _ptr(gx), m, c, _ptr(y), _hip.ACT_LRELU, c_float(0.2), None, _stream()))
        # ... and rebuilt from x (two reads): what the training step runs
        t_red2 = timed(lambda: call("vg_bn_bwd_reduce", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums), wsp,
                                    c_size_t(nbytes), None, _hip.ACT_LRELU, c_float(0.2), _ptr(gamma), _ptr(beta), _stream()))
        t_app2 = timed(lambda: call("vg_bn_bwd_apply", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums),
                                    _ptr(gx), m, c, None, _hip.ACT_LRELU, c_float(0.2), _ptr(beta), _stream()))
        mb = m * c * 4 / 1e6
        print("M=%7d C=%3d (%5.1f MB/tensor)  stats %6.1f us %5.0f GB/s | bwd reduce %6.1f us %5.0f GB/s, mask from x %6.1f us "
              "%5.0f GB/s | bwd apply %6.1f us %5.0f GB/s, mask from x %6.1f us %5.0f GB/s"
              % (m, c, mb, t_stats, mb / t_stats * 1e3, t_red, 3 * mb / t_red * 1e3, t_red2, 2 * mb / t_red2 * 1e3,
                 t_app, 4 * mb / t_app * 1e3, t_app2, 3 * mb / t_app2 * 1e3))


def main_h():
    """The same on bf16 activations (the bf16 data path): statistics, backward reduce and apply with the mask from x."""
    dev = torch.device("cuda")
    for m, c in SHAPES[1:]:
        x = torch.randn(m, c, device=dev).bfloat16()
        gy = torch.randn(m, c, device=dev).bfloat16()
        mean = torch.zeros(c, device=dev)
        invstd = torch.ones(c, device=dev)
        gamma = torch.ones(c, device=dev)
        beta = torch.zeros(c, device=dev)
        sums = torch.empty(2, c, device=dev)
        gx = torch.empty_like(x)
        y = torch.empty_like(x)
        nbytes = _hip.load().vg_bn_workspace_bytes(m, c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        wsp = ctypes.c_void_p(ws.data_ptr())
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        t_stats = timed(lambda: call("vg_bn_stats_h", P(x), m, c, c_float(1e-5), c_float(0.1), _ptr(mean), _ptr(invstd),
                                     None, None, None, wsp, c_size_t(nbytes), _stream()), 200)
        t_fwd = timed(lambda: call("vg_bn_apply_h", P(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), P(y), m, c,
                                   _hip.ACT_LRELU, c_float(0.2), _stream()), 200)
        t_red = timed(lambda: call("vg_bn_bwd_reduce_h", P(gy), P(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums), wsp,
                                   c_size_t(nbytes), None, _hip.ACT_LRELU, c_float(0.2), _ptr(gamma), _ptr(beta), _stream()), 200)
        t_app = timed(lambda: call("vg_bn_bwd_apply_h", P(gy), P(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums),
                                   P(gx), m, c, None, _hip.ACT_LRELU, c_float(0.2), _ptr(beta), _stream()), 200)
        mb = m * c * 2 / 1e6
        print("bf16 M=%7d C=%3d (%5.1f MB/tensor)  stats+final %5.1f us %5.0f GB/s | apply %5.1f us %5.0f GB/s | bwd reduce+final "
              "%5.1f us %5.0f GB/s | bwd apply %5.1f us %5.0f GB/s"
              % (m, c, mb, t_stats, mb / t_stats * 1e3, t_fwd, 2 * mb / t_fwd * 1e3, t_red, 2 * mb / t_red * 1e3,
                 t_app, 3 * mb / t_app * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bf16":
        main_h()
    else:
        main()
