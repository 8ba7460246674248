"""The bf16 data path ("bf16" precision mode): feature maps and their gradients are bf16 tensors in HBM, the `_h`
entry points of the C ABI read and write them directly.

What is compared with what:
* the `_h` kernels against torch f32 on the CPU fed with the SAME bf16 values: a result stored as bf16 may differ
  from the f32 result by the rounding of the store (2^-9 relative, half a unit in the last place of bf16) plus
  summation-order noise; sums and weight gradients stay f32 and must meet the f32 tolerances;
* elementwise `_h` kernels against their f32 twins: bit-identical to rounding the f32 kernel's result (kernels that
  also take column sums: within the rounding of the store);
* whole operators through autograd (first and second order) against torch f32: bf16 tolerance 2e-2 of the scale;
* whole fits against the reference's recorded fp32 run: tests/test_models_gpu.py (2e-2 on the losses).
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_STORE = 2.0 ** -8      # one rounding to bf16 (2^-9 relative) with room for a value that straddles a binade


def _scale_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


def _rel_elem_err(a, b, floor):
    """Largest |a-b| / max(|b|, floor * scale): a per-element relative error with an absolute floor."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-6)
    return float(((a - b).abs() / b.abs().clamp_min(floor * scale)).max())


@pytest.fixture
def bf16_mode():
    from vegans.engine import ops
    previous = ops.set_matmul_precision("bf16")
    yield ops
    ops.set_matmul_precision(previous)


def test_casts_round_to_nearest_even_and_widen_exactly(gpu):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(5)
    for n in (8, 1000, 4099, 1 << 18):
        x = torch.randn(n, generator=g) * 3.0
        x[::7] = 0.0
        got = ops._cast_raw(x.to(gpu), torch.bfloat16)
        assert got.dtype == torch.bfloat16
        assert torch.equal(got.cpu(), x.bfloat16())
        back = ops._cast_raw(got, torch.float32)
        assert torch.equal(back.cpu(), x.bfloat16().float())
    tie = torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -1.0 - 2.0 ** -8])      # halfway cases: to even
    assert torch.equal(ops._cast_raw(tie.to(gpu), torch.bfloat16).cpu(), tie.bfloat16())


H_CONV_CASES = [
    # N, Ci, H, W, Co, k, s, p
    (2, 64, 16, 16, 128, 4, 2, 1),     # fat DCGAN layer: 128-column tiles
    (3, 16, 9, 7, 64, 3, 1, 1),        # ragged row tiles, 3x3 stride 1, 64 columns
    (5, 40, 6, 6, 72, 3, 2, 0),        # channel counts that are multiples of 8 only, unequal parity classes
    (130, 64, 4, 4, 40, 1, 1, 0),      # 1x1, rows not a tile multiple, ragged column tile
    (2, 128, 8, 8, 256, 4, 2, 1),
    # k-steps inside one tap (channel counts in multiples of 64) and weight-gradient k-steps of whole gradient rows:
    (2, 64, 9, 7, 128, 3, 1, 1),       # nine taps with padding on every side, ragged rows; stride-1 input gradient
    (3, 64, 9, 11, 64, 5, 2, 2),       # 25 taps; parity classes with 3x3, 3x2, 2x3 and 2x2 taps
    (4, 8, 24, 32, 64, 3, 2, 1),       # 12 x 16 gradient: four rows per k-step, three k-steps per image
    (48, 64, 4, 4, 64, 3, 2, 1),       # 2 x 2 gradient: sixteen images per k-step
    (2, 16, 64, 64, 64, 3, 1, 1),      # 64-wide gradient: one row per k-step
    (1, 128, 32, 32, 64, 4, 2, 1),
    (4, 16, 13, 13, 64, 5, 3, 1),      # stride 3: nine parity classes (no host table), 4 x 4 gradient, four images per k-step
]


@pytest.mark.parametrize("case", H_CONV_CASES)
def test_h_contractions_against_f32_on_the_same_bf16_values(gpu, case):
    """vg_conv2d_fwd_h / _dgrad_h / _wgrad_partial_h called through the C ABI on bf16 tensors, against torch's f32
    convolution of the same values on the CPU."""
    from vegans import _hip
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    lib = _hip.load()
    gen = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(N, Ci, H, W, generator=gen)).bfloat16()
    w = (torch.randn(Co, Ci, k, k, generator=gen) * 0.1).bfloat16()
    b = torch.randn(Co, generator=gen)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=gen).bfloat16()
    yr.backward(gy.float())
    geo = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    modes = [m for m in (0, 1, 2) if lib.vg_conv2d_h_applies(m, ctypes.byref(geo.c))]
    assert modes, "no contraction of this geometry is served on bf16 tensors"
    CL = torch.channels_last
    xd = x.to(gpu).contiguous(memory_format=CL)
    wd = w.to(gpu).contiguous(memory_format=CL)
    gd = gy.to(gpu).contiguous(memory_format=CL)
    bd = b.to(gpu)
    st = _hip._stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    if 0 in modes:
        y = torch.empty((N, Co, geo.Ho, geo.Wo), dtype=torch.bfloat16, device=gpu, memory_format=CL)
        _hip.call("vg_conv2d_fwd_h", ctypes.byref(geo.c), P(xd), P(wd), P(bd), P(y), st)
        assert _rel_elem_err(y.float(), yr, 2e-3) <= BF16_STORE, "forward"
    if 1 in modes:
        gx = torch.empty((N, Ci, H, W), dtype=torch.bfloat16, device=gpu, memory_format=CL)
        _hip.call("vg_conv2d_dgrad_h", ctypes.byref(geo.c), P(gd), P(wd), None, P(gx), st)
        assert _rel_elem_err(gx.float(), xr.grad, 2e-3) <= BF16_STORE, "input gradient"
    if 2 in modes:
        gw = torch.empty((Co, Ci, k, k), dtype=torch.float32, device=gpu, memory_format=CL)
        nbytes = lib.vg_conv2d_wgrad_h_workspace_bytes(ctypes.byref(geo.c))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gpu)
        _hip.call("vg_conv2d_wgrad_partial_h", ctypes.byref(geo.c), P(gd), P(xd), P(gw), P(ws), ctypes.c_size_t(nbytes), st)
        _hip.call("vg_conv2d_wgrad_reduce_h", ctypes.byref(geo.c), P(gw), P(ws), st)
        assert _scale_err(gw, wr.grad) <= 2e-5, "weight gradient (f32 result of exact bf16 products)"


@pytest.mark.parametrize("case,pairs", [((2, 64, 16, 16, 128, 4, 2, 1), 4), ((3, 16, 9, 7, 64, 3, 1, 1), 3),
                                        ((4, 8, 24, 32, 64, 3, 2, 1), 2), ((5, 40, 6, 6, 72, 3, 2, 0), 1)])
def test_h_weight_gradient_over_several_pairs_in_one_launch(gpu, case, pairs):
    """vg_conv2d_wgrad_partial_h_multi + vg_conv2d_wgrad_reduce_h_multi: the weight gradient of one layer summed over
    several (gy, x) pairs, against the sum of torch's f32 weight gradients of the same bf16 values on the CPU."""
    from vegans import _hip
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    lib = _hip.load()
    gen = torch.Generator().manual_seed(sum(case) + pairs)
    geo = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    assert lib.vg_conv2d_h_applies(2, ctypes.byref(geo.c))
    CL = torch.channels_last
    want = torch.zeros(Co, Ci, k, k)
    xs, gs = [], []
    for _ in range(pairs):
        x = torch.randn(N, Ci, H, W, generator=gen).bfloat16()
        gy = torch.randn(N, Co, geo.Ho, geo.Wo, generator=gen).bfloat16()
        want += torch.nn.grad.conv2d_weight(x.float(), (Co, Ci, k, k), gy.float(), stride=s, padding=p)
        xs.append(x.to(gpu).contiguous(memory_format=CL))
        gs.append(gy.to(gpu).contiguous(memory_format=CL))
    gw = torch.empty((Co, Ci, k, k), dtype=torch.float32, device=gpu, memory_format=CL)
    nbytes = lib.vg_conv2d_wgrad_h_multi_workspace_bytes(ctypes.byref(geo.c), pairs)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gpu)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    gptr = (ctypes.c_void_p * pairs)(*[t.data_ptr() for t in gs])
    xptr = (ctypes.c_void_p * pairs)(*[t.data_ptr() for t in xs])
    st = _hip._stream()
    _hip.call("vg_conv2d_wgrad_partial_h_multi", ctypes.byref(geo.c), pairs, gptr, xptr, P(gw), P(ws), ctypes.c_size_t(nbytes), st)
    _hip.call("vg_conv2d_wgrad_reduce_h_multi", ctypes.byref(geo.c), pairs, P(gw), P(ws), st)
    assert _scale_err(gw, want) <= 2e-5
    assert lib.vg_conv2d_wgrad_partial_h_multi(ctypes.byref(geo.c), 5, gptr, xptr, P(gw), P(ws), ctypes.c_size_t(nbytes), st) == -1


def test_h_entry_points_refuse_shapes_they_do_not_serve(gpu):
    from vegans import _hip
    from vegans.engine import ops
    lib = _hip.load()
    for geo in (ops.Geom(4, 64, 64, 3, 64, 4, 4, 2, 2, 1, 1),        # image stem
                ops.Geom(4, 4, 4, 512, 1, 4, 4, 1, 1, 0, 0),         # critic head
                ops.Geom(4, 8, 8, 12, 64, 3, 3, 1, 1, 1, 1)):        # 12 channels: not whole 16-byte vectors
        assert not lib.vg_conv2d_h_applies(0, ctypes.byref(geo.c))
        dummy = ctypes.c_void_p(256)
        assert lib.vg_conv2d_fwd_h(ctypes.byref(geo.c), dummy, dummy, None, dummy, None) == -1
        assert b"not served" in lib.vg_last_error()


def test_h_forward_refuses_a_result_of_two_gib(gpu):
    """The bf16 kernels store through range-checked 31-bit byte offsets: a 2 GiB result is refused before any launch
    (the operands here are small; the result tensor is really allocated, so a launch that slipped through would stay
    inside it)."""
    from vegans import _hip
    from vegans.engine import ops
    lib = _hip.load()
    geo = ops.Geom(4, 256, 256, 8, 4096, 1, 1, 1, 1, 0, 0)           # 2^18 pixels x 4096 channels x 2 bytes = 2 GiB
    assert lib.vg_conv2d_h_applies(0, ctypes.byref(geo.c))
    x = torch.zeros(4 * 256 * 256 * 8, dtype=torch.bfloat16, device=gpu)
    w = torch.zeros(4096 * 8, dtype=torch.bfloat16, device=gpu)
    y = torch.empty(1 << 30, dtype=torch.bfloat16, device=gpu)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.vg_conv2d_fwd_h(ctypes.byref(geo.c), P(x), P(w), None, P(y), None) == -1
    assert b"2 GiB" in lib.vg_last_error()


@pytest.mark.parametrize("shape,act", [((6, 64, 9, 7), 2), ((33, 128), 1), ((4, 24, 5, 5), 0), ((3, 7, 4, 4), 2), ((6, 28, 9, 7), 2)])
def test_norm_and_activation_h_kernels_equal_their_f32_twins(gpu, shape, act):
    """BatchNorm (statistics, apply, backward, double backward) and activation kernels on bf16 storage: all arithmetic is
    the f32 kernels' own, so the results are those of the f32 kernels on the widened values, rounded once at the
    store; column sums and statistics stay f32 and agree to summation noise."""
    from vegans import _hip
    from vegans.engine import ops
    gen = torch.Generator().manual_seed(11 + len(shape) + act)
    mk = lambda: (torch.randn(shape, generator=gen) * 1.5 + 0.3).bfloat16().to(gpu)
    if len(shape) == 4:
        mk4 = lambda: mk().contiguous(memory_format=torch.channels_last)
    else:
        mk4 = mk
    x, gy, u = mk4(), mk4(), mk4()
    m, c = ops._rows_cols(x)
    gamma = (torch.rand(c, generator=gen) + 0.5).to(gpu)
    beta = (torch.randn(c, generator=gen) * 0.2).to(gpu)
    v = torch.randn(c, generator=gen).to(gpu)
    wv = torch.randn(c, generator=gen).to(gpu)
    slope = 0.2
    st = _hip._stream()
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    nbytes = _hip.load().vg_bn_workspace_bytes(m, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gpu)

    def run(h):
        sfx = "_h" if h else ""
        X, G, U = (x, gy, u) if h else (x.float(), gy.float(), u.float())
        new = lambda: torch.empty_like(X)
        mean, invstd = torch.empty(c, device=gpu), torch.empty(c, device=gpu)
        rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
        _hip.call("vg_bn_stats" + sfx, P(X), m, c, ctypes.c_float(1e-5), ctypes.c_float(0.1), P(mean), P(invstd), P(rm), P(rv),
                  None, P(ws), ctypes.c_size_t(nbytes), st)
        y = new()
        _hip.call("vg_bn_apply" + sfx, P(X), P(mean), P(invstd), P(gamma), P(beta), P(y), m, c, act, ctypes.c_float(slope), st)
        sums = torch.empty((2, c), device=gpu)
        _hip.call("vg_bn_bwd_reduce" + sfx, P(G), P(X), P(mean), P(invstd), m, c, P(sums), P(ws), ctypes.c_size_t(nbytes), None,
                  act, ctypes.c_float(slope), P(gamma), P(beta), st)
        gx = new()
        _hip.call("vg_bn_bwd_apply" + sfx, P(G), P(X), P(mean), P(invstd), P(gamma), P(sums), P(gx), m, c, None, act,
                  ctypes.c_float(slope), P(beta), st)
        sums5 = torch.empty((5, c), device=gpu)
        _hip.call("vg_bn_dbwd_reduce" + sfx, P(G), P(X), P(U), P(mean), P(invstd), m, c, P(sums5), P(ws), ctypes.c_size_t(nbytes),
                  None, act, ctypes.c_float(slope), P(gamma), P(beta), st)
        dgy, dx, dgamma = new(), new(), torch.empty(c, device=gpu)
        _hip.call("vg_bn_dbwd_apply" + sfx, P(G), P(X), P(U), P(v), P(wv), P(mean), P(invstd), P(gamma), P(sums5), P(dgy), P(dx),
                  P(dgamma), m, c, None, act, ctypes.c_float(slope), P(beta), st)
        a, ab = new(), new()
        _hip.call("vg_act_fwd" + sfx, P(X), P(a), ctypes.c_size_t(X.numel()), 3, ctypes.c_float(0.0), st)       # tanh
        _hip.call("vg_act_bwd" + sfx, P(G), P(a), P(ab), ctypes.c_size_t(X.numel()), 3, ctypes.c_float(0.0), st)
        colsum = torch.empty(c, device=gpu)
        _hip.call("vg_colsum" + sfx, P(X), P(colsum), m, c, P(ws), ctypes.c_size_t(nbytes), st)
        added = new()
        _hip.call("vg_add" + sfx, P(X), P(G), P(added), ctypes.c_size_t(X.numel()), st)
        return dict(mean=mean, invstd=invstd, rm=rm, rv=rv, y=y, sums=sums, gx=gx, sums5=sums5, dgy=dgy, dx=dx,
                    dgamma=dgamma, a=a, colsum=colsum, added=added), ab

    h, h_ab = run(True)
    f, _ = run(False)
    for k in ("mean", "invstd", "rm", "rv", "sums", "sums5", "dgamma", "colsum"):        # f32 results
        assert h[k].dtype == torch.float32
        assert _scale_err(h[k], f[k]) <= 1e-6, k
    for k in ("a", "added"):                                      # stored as bf16, no statistics involved: exact
        assert h[k].dtype == torch.bfloat16
        assert torch.equal(h[k], f[k].bfloat16()), "%s is not the f32 kernel's result rounded to bf16" % k
    for k in ("y", "gx", "dgy", "dx"):          # functions of the column sums, which agree to summation noise only
        assert h[k].dtype == torch.bfloat16
        assert _rel_elem_err(h[k].float(), f[k], 1e-2) <= BF16_STORE, k
    # the tanh backward reads its own stored output (already rounded): compare with the f32 kernel on that value
    ref = torch.empty_like(f["a"])
    gy32, a32 = gy.float(), h["a"].float()      # (kept alive: the call only takes their addresses)
    _hip.call("vg_act_bwd", P(gy32), P(a32), P(ref), ctypes.c_size_t(ref.numel()), 3, ctypes.c_float(0.0), st)
    assert torch.equal(h_ab, ref.bfloat16())


OP_CASES = [
    ("conv", 4, 64, 16, 16, 128, 4, 2, 1),
    ("conv", 3, 16, 9, 7, 64, 3, 1, 1),
    ("conv", 4, 3, 32, 32, 64, 4, 2, 1),       # image stem: f32 image in, bf16 feature map out (converted result)
    ("conv", 5, 512, 4, 4, 1, 4, 1, 0),        # critic head: bf16 feature map in, f32 score out
    ("convT", 4, 100, 1, 1, 64, 4, 1, 0),      # latent GEMM: f32 latent in, bf16 out
    ("convT", 2, 64, 4, 4, 32, 4, 2, 1),
    ("convT", 2, 64, 16, 16, 3, 4, 2, 1),      # generator tail: bf16 in, f32 image out
]


@pytest.mark.parametrize("case", OP_CASES)
def test_operators_store_feature_maps_as_bf16_and_return_leaf_gradients_as_f32(gpu, bf16_mode, case):
    ops = bf16_mode
    kind, N, Ci, H, W, Co, k, s, p = case
    gen = torch.Generator().manual_seed(sum(case[1:]))
    x = torch.randn(N, Ci, H, W, generator=gen)
    if kind == "conv":
        w = torch.randn(Co, Ci, k, k, generator=gen) * 0.1
        ref = lambda a, b: F.conv2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv2d(a, b, None, (s, s), (p, p))
    else:
        w = torch.randn(Ci, Co, k, k, generator=gen) * 0.1
        ref = lambda a, b: F.conv_transpose2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv_transpose2d(a, b, None, (s, s), (p, p))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = ref(xr, wr)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd, wd = x.to(gpu).requires_grad_(True), w.to(gpu).requires_grad_(True)
    y = dut(xd, wd)
    assert y.dtype == (torch.bfloat16 if Co % 8 == 0 else torch.float32)
    gyd = gy.to(gpu)
    y.backward(gyd.bfloat16() if y.dtype == torch.bfloat16 else gyd)
    assert xd.grad.dtype == torch.float32 and wd.grad.dtype == torch.float32
    assert _scale_err(y.float(), yr) <= 2e-2
    assert _scale_err(xd.grad, xr.grad) <= 2e-2
    assert _scale_err(wd.grad, wr.grad) <= 2e-2


def test_gradient_penalty_through_a_bf16_critic(gpu, bf16_mode):
    """Second order through the bf16 data path: conv -> LeakyReLU -> conv -> BatchNorm + LeakyReLU -> conv -> head, the
    penalty of WassersteinGANGP.py:67-82 on f32 interpolates, and its gradient with respect to the weights, against
    torch f32 on the CPU."""
    from torch import nn
    from vegans.engine import modules
    ops = bf16_mode
    torch.manual_seed(0)

    def critic():
        return nn.Sequential(nn.Conv2d(3, 64, 4, 2, 1, bias=False), nn.LeakyReLU(0.2),
                             nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), nn.LeakyReLU(0.2),
                             nn.Conv2d(128, 1, 4, 1, 0, bias=False))
    ref = critic()
    dut = critic()
    dut.load_state_dict(ref.state_dict())
    dut = modules.prepare_parameters(modules.bind(dut.to(gpu)))
    x = torch.randn(16, 3, 16, 16)

    def penalty(net, inp):
        inp = inp.clone().requires_grad_(True)
        out = net(inp)
        if out.dtype != torch.float32:
            out = ops.as_f32(out)
        if inp.is_cuda:
            seed = torch.ones_like(out)
            (g,) = torch.autograd.grad(out, inp, grad_outputs=seed, create_graph=True)
            pen = ops.gradient_penalty(g)
        else:
            (g,) = torch.autograd.grad(out, inp, grad_outputs=torch.ones_like(out), create_graph=True)
            pen = ((g.reshape(len(g), -1).norm(2, dim=1) - 1) ** 2).mean()
        return pen

    pr = penalty(ref, x)
    pr.backward()
    pd = penalty(dut, x.to(gpu))
    pd.backward()
    assert abs(float(pd.detach()) - float(pr.detach())) <= 3e-2 * max(abs(float(pr.detach())), 1e-2), (float(pd.detach()), float(pr.detach()))
    for (name, a), b in zip(dut.named_parameters(), ref.parameters()):
        if a.grad is None:      # the norm's shift only enters through the piecewise-constant slope mask
            assert b.grad is None or float(b.grad.abs().max()) == 0.0, name
            continue
        assert a.grad.dtype == torch.float32, name
        # second-order quantities through three bf16 layers: compared in the l2 norm (a single element of a weight
        # gradient, a sum of many rounded terms of both signs, can be off by more than the tensor as a whole)
        d = (a.grad.detach().cpu().double() - b.grad.double()).norm() / b.grad.double().norm()
        assert float(d) <= 5e-2, (name, float(d))


STEM_CASES = [
    # N, Ci, H, W, Co, k, s, p
    (8, 3, 64, 64, 64, 4, 2, 1),      # DCGAN image stem / generator tail
    (3, 1, 12, 12, 64, 5, 2, 2),      # one channel, 5x5 stride 2, ragged tiles
    (6, 3, 11, 9, 128, 3, 1, 1),      # 3x3 stride 1, two channel tiles
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_h_kernels_equal_the_f32_kernels_on_the_same_values(gpu, case):
    """The image-side layers of a bf16 network (f32 image, bf16 feature map) run the f32 stem / scatter kernels with the
    feature-map side loaded or stored as bf16: bit-identical to the f32 entry points on the widened values (forward:
    rounded once at the store)."""
    from vegans import _hip
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    lib = _hip.load()
    CL = torch.channels_last
    gen = torch.Generator().manual_seed(sum(case))
    geo = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    x = torch.randn(N, Ci, H, W, generator=gen).to(gpu).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, generator=gen) * 0.1).to(gpu).contiguous(memory_format=CL)
    b = torch.randn(Co, generator=gen).to(gpu)
    gy_h = torch.randn(N, Co, geo.Ho, geo.Wo, generator=gen).bfloat16().to(gpu).contiguous(memory_format=CL)
    gy_f = gy_h.float()
    st = _hip._stream()
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    G = ctypes.byref(geo.c)
    for mode in (0, 1):
        assert lib.vg_conv2d_stem_h_applies(mode, G), mode
    # (a weight gradient small enough for a single split-K slab stays on the generic kernel)
    assert lib.vg_conv2d_stem_h_applies(2, G) == (1 if lib.vg_conv2d_wgrad_workspace_bytes(G) > 0 else 0)
    assert case != STEM_CASES[0] or lib.vg_conv2d_stem_h_applies(2, G)
    # forward, with and without the fused activation
    for act, slope in ((0, 0.0), (2, 0.2)):
        y_h = torch.empty((N, Co, geo.Ho, geo.Wo), dtype=torch.bfloat16, device=gpu, memory_format=CL)
        y_f = torch.empty((N, Co, geo.Ho, geo.Wo), dtype=torch.float32, device=gpu, memory_format=CL)
        _hip.call("vg_conv2d_stem_fwd_h", G, P(x), P(w), P(b), P(y_h), act, ctypes.c_float(slope), st)
        if act:
            _hip.call("vg_conv2d_fwd_act", G, P(x), P(w), P(b), P(y_f), act, ctypes.c_float(slope), st)
        else:
            _hip.call("vg_conv2d_fwd", G, P(x), P(w), P(b), P(y_f), st)
        assert torch.equal(y_h, y_f.bfloat16()), "stem forward, act %d" % act
    # input gradient
    gx_h = torch.empty((N, Ci, H, W), dtype=torch.float32, device=gpu, memory_format=CL)
    gx_f = torch.empty_like(gx_h)
    _hip.call("vg_conv2d_stem_dgrad_h", G, P(gy_h), P(w), None, P(gx_h), st)
    _hip.call("vg_conv2d_dgrad", G, P(gy_f), P(w), None, P(gx_f), st)
    assert torch.equal(gx_h, gx_f), "stem input gradient"
    # weight gradient
    if not lib.vg_conv2d_stem_h_applies(2, G):
        return
    nbytes = lib.vg_conv2d_wgrad_workspace_bytes(G)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gpu)
    gw_h = torch.empty((Co, Ci, k, k), dtype=torch.float32, device=gpu, memory_format=CL)
    gw_f = torch.empty_like(gw_h)
    _hip.call("vg_conv2d_stem_wgrad_partial_h", G, P(gy_h), P(x), P(ws), ctypes.c_size_t(nbytes), st)
    _hip.call("vg_conv2d_wgrad_reduce", G, P(gw_h), P(ws), st)
    _hip.call("vg_conv2d_wgrad", G, P(gy_f), P(x), P(gw_f), P(ws), ctypes.c_size_t(nbytes), st)
    assert torch.equal(gw_h, gw_f), "stem weight gradient"


@pytest.mark.parametrize("case", [(4, 64, 16, 16, 128, 4, 2, 1), (3, 64, 9, 7, 16, 3, 1, 1), (5, 40, 6, 6, 72, 3, 2, 0)])
@pytest.mark.parametrize("act,slope", [(2, 0.2), (1, 0.0)])
def test_h_input_gradient_with_slope_mask(gpu, case, act, slope):
    """vg_conv2d_dgrad_h_actmask against vg_conv2d_dgrad_h followed by vg_act_bwd_h: identical where the slope is 1 or 0,
    and within one rounding elsewhere (the fused form multiplies the f32 result and rounds once, the two-pass form
    rounds the gradient, multiplies, and rounds again)."""
    from vegans import _hip
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    CL = torch.channels_last
    gen = torch.Generator().manual_seed(sum(case) + act)
    geo = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    G = ctypes.byref(geo.c)
    assert _hip.load().vg_conv2d_h_applies(1, G)
    gy = torch.randn(N, Co, geo.Ho, geo.Wo, generator=gen).bfloat16().to(gpu).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, generator=gen) * 0.1).bfloat16().to(gpu).contiguous(memory_format=CL)
    ref = torch.randn(N, Ci, H, W, generator=gen).bfloat16().to(gpu).contiguous(memory_format=CL)
    ref[0, :, 0, 0] = 0.0                                   # zeros take the lower branch
    st = _hip._stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fused = torch.empty_like(ref)
    plain = torch.empty_like(ref)
    two = torch.empty_like(ref)
    _hip.call("vg_conv2d_dgrad_h_actmask", G, P(gy), P(w), P(fused), P(ref), act, ctypes.c_float(slope), st)
    _hip.call("vg_conv2d_dgrad_h", G, P(gy), P(w), None, P(plain), st)
    _hip.call("vg_act_bwd_h", P(plain), P(ref), P(two), ctypes.c_size_t(ref.numel()), act, ctypes.c_float(slope), st)
    up = ref > 0
    assert torch.equal(fused[up], plain[up])
    if act == 1:
        assert float(fused[~up].abs().max()) == 0.0
    else:
        # one rounding against two: up to a unit and a half in the last place of bf16
        assert _rel_elem_err(fused.float(), two.float(), 1e-3) <= 2.05 * BF16_STORE
