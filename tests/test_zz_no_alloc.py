"""vg_conv2d_wgrad_noalloc: the weight gradient with its split-K slabs in the library's built-in device array, for
callers that cannot allocate. Called through the C ABI and compared bit for bit with vg_conv2d_wgrad (caller-provided
workspace) on two calls in a row, the second on a smaller input; a geometry beyond the built-in 32 MiB is refused."""
import ctypes
from ctypes import byref, c_size_t, c_void_p

import pytest
import torch

pytestmark = pytest.mark.gpu


def _geom(N, H, W, Ci, Co, k, s, p):
    from vegans import _hip
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    return _hip.ConvGeom(N, H, W, Ci, Co, k, k, Ho, Wo, s, s, p, p), Ho, Wo


def _both(lib, stream, case, seed):
    N, H, W, Ci, Co, k, s, p = case
    g, Ho, Wo = _geom(*case)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, Ci, generator=gen).cuda()           # NHWC
    gy = torch.randn(N, Ho, Wo, Co, generator=gen).cuda()
    need = lib.vg_conv2d_wgrad_workspace_bytes(byref(g))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    ref = torch.full((Co, k, k, Ci), float("nan"), device="cuda")
    rc = lib.vg_conv2d_wgrad(byref(g), c_void_p(gy.data_ptr()), c_void_p(x.data_ptr()), c_void_p(ref.data_ptr()),
                             c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream)
    assert rc == 0, lib.vg_last_error()
    out = torch.full((Co, k, k, Ci), float("nan"), device="cuda")
    before = torch.cuda.memory_allocated()
    rc = lib.vg_conv2d_wgrad_noalloc(byref(g), c_void_p(gy.data_ptr()), c_void_p(x.data_ptr()), c_void_p(out.data_ptr()),
                                     stream)
    assert rc == 0, lib.vg_last_error()
    assert torch.cuda.memory_allocated() == before
    torch.cuda.synchronize()
    return need, out, ref


def test_wgrad_noalloc_matches_wgrad_with_workspace(gpu):
    from vegans import _hip
    lib = _hip.load()
    stream = _hip._stream()
    # a DCGAN-64 critic layer at batch 64 (several splits, megabytes of slabs), then a smaller one that leaves most of
    # the previous call's slabs in the array, then an image stem (its own kernel fills the slabs), then one split only
    cases = [(64, 32, 32, 64, 128, 4, 2, 1), (4, 16, 16, 16, 24, 3, 1, 1), (8, 32, 32, 3, 64, 4, 2, 1),
             (2, 4, 4, 8, 8, 3, 1, 1)]
    needs = []
    for i, case in enumerate(cases):
        need, out, ref = _both(lib, stream, case, seed=10 + i)
        needs.append(need)
        assert torch.isfinite(ref).all()
        assert torch.equal(out, ref), "case %d %s differs from vg_conv2d_wgrad" % (i, case)
    assert needs[0] > needs[1] > 0, needs       # the second call really was the smaller one, and both used slabs
    # the same call twice in a row: the array's leftover contents do not matter
    _, again, ref = _both(lib, stream, cases[1], seed=11)
    assert torch.equal(again, ref)


def test_wgrad_noalloc_refuses_what_does_not_fit(gpu):
    from vegans import _hip
    lib = _hip.load()
    g, Ho, Wo = _geom(64, 64, 64, 8, 384, 4, 2, 1)            # 171 splits of a 384 x 128 tile: just beyond 32 MiB
    assert lib.vg_conv2d_wgrad_workspace_bytes(byref(g)) > (32 << 20)
    dummy = torch.zeros(16, device="cuda")
    rc = lib.vg_conv2d_wgrad_noalloc(byref(g), c_void_p(dummy.data_ptr()), c_void_p(dummy.data_ptr()),
                                     c_void_p(dummy.data_ptr()), _hip._stream())
    assert rc == -3 and b"built-in scratch holds 33554432" in lib.vg_last_error(), (rc, lib.vg_last_error())
    rc = lib.vg_conv2d_wgrad_noalloc(byref(g), None, None, None, None)
    assert rc == -1
