"""Times the bf16 contractions (vg_conv2d_{fwd,dgrad,wgrad_partial}_h: pgemm_kernel on bf16 tensors) on the three fat
DCGAN-64 geometries against a library named by VEGANS_ABLATE_LIB, a build of csrc/ with -DPG_ABLATE=<bits> that switches
parts of the kernel off (1 global loads, 2 LDS stores, 4 fragment reads, 8 MFMAs, 16 result stores), to see where the
time goes.  build: make -C gan-pytorch_amd/csrc (product), then per variant
  hipcc <flags of csrc/Makefile> -DPG_ABLATE=<bits> for the three conv_*.hip units, linked with the product's other objects."""
import ctypes
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
    # PG_ONLY="fwd:128" restricts the run to one contraction (mode:Ci), PG_REPS sets the repetitions (counter passes)
    reps = int(os.environ.get("PG_REPS", "200"))
    only = os.environ.get("PG_ONLY", "")
    shapes = SHAPES
    if os.environ.get("PG_SHAPES"):          # "N,H,W,Ci,Co;N,H,W,Ci,Co": other 4x4 stride-2 layers
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["PG_SHAPES"].split(";")]
    out = []
    lib = _hip.load()
    st = _hip._stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    CL = torch.channels_last
    for (N, H, W, Ci, Co) in shapes:
        k, s, p = 4, 2, 1
        g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
        G = ctypes.byref(g.c)
        x = torch.randn(N, Ci, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
        w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).bfloat16().contiguous(memory_format=CL)
        gy = torch.randn(N, Co, g.Ho, g.Wo, device="cuda").bfloat16().contiguous(memory_format=CL)
        y = torch.empty_like(gy)
        gx = torch.empty_like(x)
        gw = torch.empty((Co, Ci, k, k), device="cuda").contiguous(memory_format=CL)
        nbytes = lib.vg_conv2d_wgrad_h_workspace_bytes(G)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        fns = {"fwd": lambda: _hip.call("vg_conv2d_fwd_h", G, P(x), P(w), None, P(y), st),
               "dgrad": lambda: _hip.call("vg_conv2d_dgrad_h", G, P(gy), P(w), None, P(gx), st),
               "wgrad": lambda: _hip.call("vg_conv2d_wgrad_partial_h", G, P(gy), P(x), P(gw), P(ws), ctypes.c_size_t(nbytes), st)}
        for mode, fn in fns.items():
            if only and only != "%s:%d" % (mode, Ci):
                continue
            for _ in range(300 if not only else 3):
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
    print(os.path.basename(os.environ.get("VEGANS_ABLATE_LIB", "product")), " | ".join(out))


if __name__ == "__main__":
    main()
