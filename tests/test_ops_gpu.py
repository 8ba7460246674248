"""Operator-level parity of the HIP kernels (called through the C ABI via the engine's autograd Functions)
against plain torch fp32 on the CPU.  Tolerances: the kernels are exact-f32 FMA chains in a different
summation order than torch's CPU kernels, so outputs agree to ~1e-6 relative to the operand scale; the tests
use 2e-5 of the reference tensor's largest magnitude for first-order quantities (forward, input and weight
gradients) and 1e-4 for second-order ones (penalty gradients)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


CONV_CASES = [
    # N, Ci, H, W, Co, k, s, p
    (4, 3, 16, 16, 8, 4, 2, 1),      # image stem, Ci=3 (scalar loader)
    (2, 64, 16, 16, 128, 4, 2, 1),   # fat DCGAN layer
    (3, 16, 9, 7, 24, 3, 1, 1),      # odd sizes, 3x3 stride 1
    (2, 32, 8, 8, 1, 4, 1, 0),       # Co=1
    (5, 512, 4, 4, 1, 4, 1, 0),      # DCGAN critic head (GEMV)
    (2, 11, 32, 32, 8, 5, 2, 2),     # Ci=11 conditional input, 5x5
    (2, 8, 12, 12, 20, 3, 2, 0),     # stride 2, no padding, odd phases
    (130, 4, 4, 4, 36, 1, 1, 0),     # 1x1 conv, M not a tile multiple
    (2, 3, 9, 7, 8, 3, 1, 1),        # few input channels: scatter-form dgrad, 3x3 stride 1, odd sizes
    (3, 1, 12, 12, 12, 5, 2, 2),     # Ci=1, 5x5 stride 2
    (1, 2, 140, 150, 4, 4, 2, 1),    # Ci=2, gradient image larger than one tile in both dimensions
    (2, 4, 37, 21, 8, 4, 2, 1),      # Ci=4, several row tiles, odd sizes
    (8, 3, 64, 64, 16, 4, 2, 1),     # image stem at full resolution: per-pixel stem kernel forward
    (4, 1, 40, 40, 8, 3, 1, 1),      # stem kernel, 3x3 stride 1, Ci=1
    (8, 3, 64, 64, 64, 4, 2, 1),     # DCGAN image stem: MFMA stem kernel, one block tile per 4 x 32 output pixels
    (3, 1, 12, 12, 32, 5, 2, 2),     # stem kernel, Ci=1, 5x5 stride 2, ragged tile (6 x 6 outputs), one channel tile
    (2, 3, 9, 7, 96, 3, 1, 1),       # stem kernel, 3x3 stride 1 (odd K = 27), 96 channels: second grid row half filled
    (5, 1, 21, 70, 64, 4, 2, 1),     # stem kernel, Ci=1, several column tiles with a ragged last one
    (6, 3, 11, 9, 128, 3, 1, 1),     # stem weight-gradient kernel: odd K = 27, two channel tiles, ragged pixel tiles
    (3, 1, 30, 30, 64, 5, 2, 2),     # stem weight-gradient kernel: Ci=1, 5x5 stride 2, more slabs than tiles
    # persistent scatter-form input gradient (4x4 stride 2, 3 or 4 image channels, 32 or 64 feature channels)
    (3, 3, 20, 28, 64, 4, 2, 1),     # one tile spans the gradient image's height
    (2, 4, 37, 21, 32, 4, 2, 1),     # Ci=4, 32 channels, two row tiles, odd sizes
    (5, 3, 70, 40, 32, 4, 2, 0),     # no padding, five row tiles per image, more tiles than one pass of blocks may hold
    (2, 3, 66, 64, 64, 4, 2, 2),     # padding 2: 33 gradient columns in 64-wide tiles of 4 rows
    # channel counts in multiples of 32 (the block-uniform k-walks of the contraction kernels) away from the DCGAN shapes
    (2, 256, 15, 15, 512, 4, 1, 1),  # PatchGAN layer: stride 1, one parity class of 16 taps
    (2, 128, 13, 11, 64, 3, 1, 1),   # 3x3 stride 1, odd sizes
    (3, 64, 12, 12, 96, 5, 2, 2),    # 5x5 stride 2: classes with 9, 6, 6 and 4 taps
    (4, 512, 2, 2, 512, 4, 2, 1),    # U-Net bottleneck: one output pixel per image
    (2, 32, 40, 16, 64, 4, 2, 1),    # wgrad: k-steps of two whole output rows (Wo = 8: four rows of a 20-row image per k-step)
    (40, 64, 8, 8, 128, 4, 2, 1),    # wgrad: a k-step covers two whole images (Ho = Wo = 4)
    (2, 6, 20, 20, 64, 4, 2, 1),     # PatchGAN stem on a 6-channel pair: input gradient with 4-byte weight loads, 64 channels
    (2, 64, 10, 10, 6, 3, 1, 1),     # 6 output channels: weight gradient and forward with 4-byte loads beside 64 input channels
    # the benchmark's own launches: DCGAN-64 critic layers at batch 256 (the split-K plan, grid sizes and slab set of
    # that batch; the other cases run at a few images)
    (256, 3, 64, 64, 64, 4, 2, 1),   # image stem: 262144 output pixels
    (256, 64, 32, 32, 128, 4, 2, 1),  # fat layer 1: M = 65536, K = 1024
    (256, 128, 16, 16, 256, 4, 2, 1),  # fat layer 2: M = 16384, K = 2048
    (256, 256, 8, 8, 512, 4, 2, 1),   # fat layer 3: M = 4096, K = 4096
    (256, 512, 4, 4, 1, 4, 1, 0),     # critic head
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(gpu, case):
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = x.to(gpu).requires_grad_(True)
    wd = w.to(gpu).requires_grad_(True)
    bd = b.to(gpu).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, (s, s), (p, p))
    _close(y, yr, what="conv fwd")
    y.backward(gy.to(gpu))
    _close(xd.grad, xr.grad, what="conv dgrad")
    _close(wd.grad, wr.grad, what="conv wgrad")
    _close(bd.grad, br.grad, what="conv bias grad")


BF16_CASES = [
    # kind, N, Cin, H, W, Cout, k, s, p
    ("conv", 2, 64, 16, 16, 128, 4, 2, 1),     # 128x128 tiles, all three modes
    ("conv", 3, 16, 9, 7, 24, 3, 1, 1),        # ragged tiles, 3x3 stride 1
    ("conv", 130, 4, 4, 4, 36, 1, 1, 0),       # 1x1, M not a tile multiple, tiny K
    ("conv", 2, 8, 12, 12, 20, 3, 2, 0),       # stride 2 without padding: parity classes of different size
    ("convT", 4, 100, 1, 1, 64, 4, 1, 0),      # latent vector -> 4x4: plain-GEMM path
    ("convT", 2, 64, 4, 4, 32, 4, 2, 1),       # DCGAN upsample: dgrad kernel as forward
    ("convT", 2, 20, 4, 4, 12, 5, 3, 1),       # stride 3, k = 5
]


@pytest.mark.parametrize("case", BF16_CASES)
def test_bf16_mode_is_exact_on_bf16_representable_operands(gpu, case):
    """(The "bf16_compute" mode: f32 tensors, bf16 matrix operands; the bf16 data path has tests/test_bf16_storage_gpu.py.)
    bf16 mode rounds the operands to bf16 inside the kernel and accumulates their exact products in f32. Fed with
    operands that already are bf16 values, every contraction (forward, input gradient, weight gradient) must therefore
    agree with torch's f32 result to summation-order noise, like the f32 kernels do."""
    from vegans.engine import ops
    kind, N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case[1:]))
    rnd = lambda t: t.bfloat16().float()
    x = rnd(torch.randn(N, Ci, H, W, generator=g))
    if kind == "conv":
        w = rnd(torch.randn(Co, Ci, k, k, generator=g) * 0.1)
        ref = lambda a, b: F.conv2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv2d(a, b, None, (s, s), (p, p))
    else:
        w = rnd(torch.randn(Ci, Co, k, k, generator=g) * 0.1)
        ref = lambda a, b: F.conv_transpose2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv_transpose2d(a, b, None, (s, s), (p, p))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = ref(xr, wr)
    gy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    previous = ops.set_matmul_precision("bf16_compute")
    try:
        xd, wd = x.to(gpu).requires_grad_(True), w.to(gpu).requires_grad_(True)
        y = dut(xd, wd)
        y.backward(gy.to(gpu))
    finally:
        ops.set_matmul_precision(previous)
    _close(y, yr, what="bf16 fwd")
    _close(xd.grad, xr.grad, what="bf16 input gradient")
    _close(wd.grad, wr.grad, what="bf16 weight gradient")


def test_bf16_mode_rounds_operands_to_nearest_even(gpu):
    """On general f32 operands the bf16 mode must equal an f32 convolution of the bf16-rounded operands (so its error
    against true f32 is the operand rounding, about 2^-9 relative per product, and nothing else)."""
    from vegans.engine import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 32, 12, 12, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    want = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, stride=1, padding=1)
    exact = F.conv2d(x, w, None, stride=1, padding=1)
    previous = ops.set_matmul_precision("bf16_compute")
    try:
        got = ops.conv2d(x.to(gpu), w.to(gpu), None, (1, 1), (1, 1))
    finally:
        ops.set_matmul_precision(previous)
    _close(got, want, what="bf16 conv vs f32 conv of rounded operands")
    err = float((got.cpu() - exact).abs().max() / exact.abs().max())
    assert 1e-5 < err < 2e-2, err       # really ran in bf16, and stays within the mode's stated tolerance


SPLIT_CASES = [
    # kind, N, Cin, H, W, Cout, k, s, p      (channel counts that are multiples of 8 take the plane GEMM)
    ("conv", 2, 64, 16, 16, 128, 4, 2, 1),     # 128x128 tiles, all three modes
    ("conv", 3, 16, 9, 7, 40, 3, 1, 1),        # ragged rows and columns, K = 144 (partial last k-step)
    ("conv", 130, 8, 4, 4, 72, 1, 1, 0),       # 1x1, K = 8: a single partial k-step; input gradient has 8 columns (f32 path)
    ("conv", 2, 48, 12, 12, 56, 3, 2, 0),      # stride 2 without padding: parity classes of different size
    ("conv", 1, 72, 20, 20, 136, 5, 1, 2),     # 5x5, two column tiles with a ragged second one
    ("conv", 16, 128, 8, 8, 256, 4, 2, 1),     # enough rows for several split-K chunks in the weight gradient
    ("convT", 2, 64, 4, 4, 40, 4, 2, 1),       # DCGAN upsample: dgrad contraction as forward
    ("convT", 2, 24, 4, 4, 40, 5, 3, 1),       # stride 3, k = 5
    ("conv", 4, 12, 8, 8, 20, 3, 1, 1),        # channels not multiples of 8: served by the f32 kernels
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_mode_matches_f32_accuracy(gpu, case):
    """The split mode (three bf16 terms per operand, six products, f32 accumulation) claims f32-grade results: measured
    against a float64 evaluation, its error must stay within 1e-5 of the output scale and within twice the error of the
    f32 MFMA kernels plus rounding noise, on every contraction."""
    from vegans.engine import ops
    kind, N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case[1:]))
    x = torch.randn(N, Ci, H, W, generator=g)
    if kind == "conv":
        w = torch.randn(Co, Ci, k, k, generator=g) * 0.1
        ref = lambda a, b: F.conv2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv2d(a, b, None, (s, s), (p, p))
    else:
        w = torch.randn(Ci, Co, k, k, generator=g) * 0.1
        ref = lambda a, b: F.conv_transpose2d(a, b, None, stride=s, padding=p)
        dut = lambda a, b: ops.conv_transpose2d(a, b, None, (s, s), (p, p))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = ref(xr, wr)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    want = (yr.detach(), xr.grad, wr.grad)

    def run(precision):
        previous = ops.set_matmul_precision(precision)
        try:
            xd, wd = x.to(gpu).requires_grad_(True), w.to(gpu).requires_grad_(True)
            y = dut(xd, wd)
            y.backward(gy.to(gpu))
        finally:
            ops.set_matmul_precision(previous)
        return [float((t.detach().cpu().double() - r).abs().max() / r.abs().max()) for t, r in zip((y, xd.grad, wd.grad), want)]

    base, split = run("fp32"), run("fp32_split")
    for name, e0, e1 in zip(("forward", "input gradient", "weight gradient"), base, split):
        assert e1 <= 1e-5, "%s: split-mode error %.3e" % (name, e1)
        assert e1 <= 2.0 * e0 + 2e-7, "%s: split-mode error %.3e vs %.3e in f32" % (name, e1, e0)


def test_split_mode_uses_the_plane_kernels(gpu):
    """The eligible shapes really go through the split (a workspace for the bf16 terms is requested), others do not."""
    import ctypes
    from vegans import _hip
    from vegans.engine import ops
    lib = _hip.load()
    g = ops.Geom(2, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1)
    for mode in (0, 1):
        assert lib.vg_conv2d_prec_workspace_bytes(mode, ctypes.byref(g.c), 2) > 0
        assert lib.vg_conv2d_prec_workspace_bytes(mode, ctypes.byref(g.c), 0) == 0
    slabs = lib.vg_conv2d_wgrad_workspace_bytes(ctypes.byref(g.c))
    assert lib.vg_conv2d_prec_workspace_bytes(2, ctypes.byref(g.c), 2) > (slabs + 255) // 256 * 256
    stem = ops.Geom(2, 16, 16, 3, 64, 4, 4, 2, 2, 1, 1)
    assert lib.vg_conv2d_prec_workspace_bytes(0, ctypes.byref(stem.c), 2) == 0


CONVT_CASES = [
    # N, Cin, Hi, Wi, Cout, k, s, p
    (4, 100, 1, 1, 64, 4, 1, 0),    # latent vector -> 4x4 (single GEMM path)
    (2, 64, 4, 4, 32, 4, 2, 1),     # DCGAN upsample
    (2, 16, 8, 8, 3, 4, 2, 1),      # image head, Cout=3
    (3, 8, 5, 7, 12, 3, 1, 1),      # stride 1
    (2, 20, 4, 4, 10, 5, 3, 1),     # stride 3, k=5: parity classes with different tap counts
    (2, 512, 1, 1, 512, 4, 2, 1),   # U-Net innermost up-convolution: 1x1 -> 2x2, every class a single tap
    (2, 256, 6, 5, 128, 4, 2, 1),   # channel counts in multiples of 32, odd sizes
    (2, 64, 7, 7, 32, 3, 1, 1),     # stride 1, 3x3, one class of nine taps
    # the benchmark's own launches: DCGAN-64 generator layers at batch 256
    (256, 100, 1, 1, 512, 4, 1, 0),   # latent GEMM
    (256, 512, 4, 4, 256, 4, 2, 1),   # 4 classes x (4096 x 256 x 2048)
    (256, 256, 8, 8, 128, 4, 2, 1),   # 4 classes x (16384 x 128 x 1024)
    (256, 128, 16, 16, 64, 4, 2, 1),  # 4 classes x (65536 x 64 x 512): the short-k, 64-column kernel
    (256, 64, 32, 32, 3, 4, 2, 1),    # image tail: scatter form
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d_fwd_bwd(gpu, case):
    from vegans.engine import ops
    N, Cin, Hi, Wi, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, Hi, Wi, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = x.to(gpu).requires_grad_(True)
    wd = w.to(gpu).requires_grad_(True)
    bd = b.to(gpu).requires_grad_(True)
    y = ops.conv_transpose2d(xd, wd, bd, (s, s), (p, p))
    _close(y, yr, what="convT fwd")
    y.backward(gy.to(gpu))
    _close(xd.grad, xr.grad, what="convT input grad")
    _close(wd.grad, wr.grad, what="convT weight grad")
    _close(bd.grad, br.grad, what="convT bias grad")


@pytest.mark.parametrize("shape", [(32, 64, 128), (7, 784, 512), (256, 100, 1), (33, 5, 3)])
def test_linear(gpu, shape):
    from vegans.engine import ops
    B, I, O = shape
    g = torch.Generator().manual_seed(B + I + O)
    x = torch.randn(B, I, generator=g)
    w = torch.randn(O, I, generator=g) * 0.1
    b = torch.randn(O, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd, wd, bd = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd)
    _close(y, yr, what="linear fwd")
    y.backward(gy.to(gpu))
    _close(xd.grad, xr.grad, what="linear dgrad")
    _close(wd.grad, wr.grad, what="linear wgrad")
    _close(bd.grad, br.grad, what="linear bias grad")


# (6, 28, 9, 7): seven 4-channel groups per row, a grid stride that is no multiple of seven (the elementwise kernels then
# step their channel group instead of staying on it)
@pytest.mark.parametrize("shape", [(8, 16, 6, 6), (4, 3, 5, 5), (64, 32), (16, 130, 4, 4), (6, 28, 9, 7)])
def test_batch_norm_train_and_running_stats(gpu, shape):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(len(shape) + shape[1])
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    yr = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    gy = torch.randn(shape, generator=g)
    yr.backward(gy)
    xd, gd, bd = (t.to(gpu).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
    y = ops.batch_norm(xd, gd, bd, rmd, rvd, True, 0.1, 1e-5)
    _close(y, yr, what="bn fwd")
    _close(rmd, rm, what="running mean")
    _close(rvd, rv, what="running var")
    y.backward(gy.to(gpu))
    _close(xd.grad, xr.grad, what="bn dx")
    _close(gd.grad, gr.grad, what="bn dgamma")
    _close(bd.grad, br.grad, what="bn dbeta")


@pytest.mark.parametrize("act", ["relu", "lrelu"])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_with_fused_activation(gpu, act, training):
    """BatchNorm + ReLU/LeakyReLU as one operator, and the HipSequential container that pairs them up."""
    from torch import nn
    from vegans.engine import modules
    g = torch.Generator().manual_seed(17)
    x = torch.randn(6, 12, 5, 5, generator=g) * 1.5 + 0.3
    gy = torch.randn(6, 12, 5, 5, generator=g)

    def make():
        torch.manual_seed(5)
        bn = nn.BatchNorm2d(12)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
            bn.running_mean.normal_()
            bn.running_var.uniform_(0.5, 1.5)
        return nn.Sequential(bn, nn.ReLU() if act == "relu" else nn.LeakyReLU(0.2), nn.Identity())

    ref, dut = make().train(training), modules.bind(make()).to(gpu).train(training)
    assert type(dut) is modules.HipSequential
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy)
    xd = x.to(gpu).requires_grad_(True)
    y = dut(xd)
    assert "BatchNorm" in type(y.grad_fn).__name__, type(y.grad_fn).__name__   # one operator, no separate activation
    y.backward(gy.to(gpu))
    _close(y, yr, what="fused bn+act fwd")
    _close(xd.grad, xr.grad, what="fused bn+act dx")
    _close(dut[0].weight.grad, ref[0].weight.grad, what="fused bn+act dgamma")
    _close(dut[0].bias.grad, ref[0].bias.grad, what="fused bn+act dbeta")
    _close(dut[0].running_mean, ref[0].running_mean, what="running mean")
    _close(dut[0].running_var, ref[0].running_var, what="running var")
    # a hook on the activation must still see a separate call
    dut[1].register_forward_hook(lambda m, i, o: None)
    y2 = dut(x.to(gpu))
    _close(y2, ref(x), what="unfused because of the hook")
    assert modules._fused_act(dut[0], dut[1]) is None


@pytest.mark.parametrize("shape,k,s,p", [((3, 8, 13, 11), 5, 2, 2), ((2, 3, 8, 8), 2, 2, 0), ((2, 16, 9, 9), 3, 1, 1)])
def test_max_pool_first_and_second_order(gpu, shape, k, s, p):
    """nn.MaxPool2d as the reference's image-to-image critic uses it (kernel 5, stride 2, padding 2), overlapping
    windows included; the double-backward term (a penalty on the input gradient) goes through the winner pick."""
    from torch import nn
    from vegans.engine import modules
    g = torch.Generator().manual_seed(31)
    x = torch.randn(shape, generator=g)
    gy_shape = F.max_pool2d(x, k, s, p).shape
    gy = torch.randn(gy_shape, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    (gxr,) = torch.autograd.grad(yr, xr, gy, create_graph=True)
    pool = modules.bind(nn.Sequential(nn.MaxPool2d(k, s, p)))
    xd = x.to(gpu).requires_grad_(True)
    y = pool(xd)
    gyd = gy.to(gpu).requires_grad_(True)
    (gxd,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
    assert torch.equal(y.detach().cpu(), yr.detach())            # selection, not arithmetic: exact
    _close(gxd, gxr, what="max pool dx")
    # second order: d/d(gy) of sum(w * gx) picks w at the winners
    w = torch.randn(shape, generator=g)
    gyr = gy.clone().requires_grad_(True)
    (gxr2,) = torch.autograd.grad(F.max_pool2d(xr, k, s, p), xr, gyr, create_graph=True)
    (gxr2 * w).sum().backward()
    (d_gy,) = torch.autograd.grad(gxd, gyd, w.to(gpu))
    _close(d_gy, gyr.grad, what="max pool second order")


def test_batch_norm_eval(gpu):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 8, 5, 5, generator=g)
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    rm, rv = torch.randn(8, generator=g), torch.rand(8, generator=g) + 0.5
    yr = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    y = ops.batch_norm(x.to(gpu), gamma.to(gpu), beta.to(gpu), rm.to(gpu), rv.to(gpu), False, 0.1, 1e-5)
    _close(y, yr, what="bn eval")


@pytest.mark.parametrize("act", ["relu", "lrelu", "tanh", "sigmoid"])
def test_activations_first_and_second_order(gpu, act):
    from vegans import _hip
    from vegans.engine import ops
    code = {"relu": _hip.ACT_RELU, "lrelu": _hip.ACT_LRELU, "tanh": _hip.ACT_TANH, "sigmoid": _hip.ACT_SIGMOID}[act]
    ref = {"relu": torch.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 10, 3, 3, generator=g)
    gy = torch.randn(6, 10, 3, 3, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (gxr,) = torch.autograd.grad(yr, xr, gy, create_graph=True)
    xd = x.to(gpu).requires_grad_(True)
    y = ops.activation(xd, code, 0.2)
    (gxd,) = torch.autograd.grad(y, xd, gy.to(gpu), create_graph=True)
    _close(y, yr, what="act fwd")
    _close(gxd, gxr, what="act bwd")
    if act in ("tanh", "sigmoid"):
        (gxr ** 2).sum().backward()
        ops.loss_mse(gxd, 0.0).backward()   # mean(gx^2): same second-order path up to the 1/n factor
        _close(xd.grad * gxd.numel(), xr.grad, tol=1e-4, what="act second order")


def test_losses(gpu):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(5)
    p = torch.rand(37, 1, generator=g) * 0.98 + 0.01
    for target in (0.0, 1.0):
        pr = p.clone().requires_grad_(True)
        lr = F.binary_cross_entropy(pr, torch.full_like(pr, target))
        lr.backward()
        pd = p.to(gpu).requires_grad_(True)
        ld = ops.loss_bce(pd, target)
        ld.backward()
        _close(ld, lr, what="bce")
        _close(pd.grad, pr.grad, what="bce grad")
        pr = p.clone().requires_grad_(True)
        lr = F.mse_loss(pr, torch.full_like(pr, target))
        lr.backward()
        pd = p.to(gpu).requires_grad_(True)
        ld = ops.loss_mse(pd, target)
        ld.backward()
        _close(ld, lr, what="mse")
        _close(pd.grad, pr.grad, what="mse grad")
    # saturated probabilities hit the -100 clamp exactly like torch
    sat = torch.tensor([[0.0], [1.0], [0.5]])
    _close(ops.loss_bce(sat.to(gpu), 1.0), F.binary_cross_entropy(sat, torch.ones_like(sat)), what="bce clamp")
    for sign in (-1.0, 1.0):
        pr = p.clone().requires_grad_(True)
        lr = sign * pr.mean()
        lr.backward()
        pd = p.to(gpu).requires_grad_(True)
        ld = ops.loss_mean(pd, sign)
        ld.backward()
        _close(ld, lr, what="mean loss")
        _close(pd.grad, pr.grad, what="mean loss grad")
    a, b = torch.randn(5, 3, 8, 8, generator=g), torch.randn(5, 3, 8, 8, generator=g)
    ar = a.clone().requires_grad_(True)
    lr = F.l1_loss(ar, b)
    lr.backward()
    ad = a.to(gpu).requires_grad_(True)
    ld = ops.loss_l1(ad, b.to(gpu))
    ld.backward()
    _close(ld, lr, what="l1")
    _close(ad.grad, ar.grad, what="l1 grad")
    big = torch.randn(300000, generator=g)        # multi-block reduction path
    _close(ops.loss_mean(big.to(gpu), 1.0), big.double().mean().float(), tol=1e-5, what="big mean")


def test_concat_and_layout_bit_exact(gpu):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 3, 8, 8, generator=g)
    onehot = F.one_hot(torch.randint(0, 10, (6,), generator=g), 10).float()
    out = ops.concatenate(x.to(gpu), onehot.to(gpu))
    ref = torch.cat((x, onehot.reshape(6, 10, 1, 1).tile(1, 1, 8, 8)), dim=1)
    assert torch.equal(ops.to_nchw(out).cpu(), ref)
    z = torch.randn(6, 4, generator=g)
    out = ops.concatenate(z.to(gpu), x.to(gpu))
    ref = torch.cat((z.reshape(6, 4, 1, 1).tile(1, 1, 8, 8), x), dim=1)
    assert torch.equal(ops.to_nchw(out).cpu(), ref)
    out = ops.concatenate(z.to(gpu), onehot.to(gpu))
    assert torch.equal(out.cpu(), torch.cat((z, onehot), dim=1))
    y = torch.randn(6, 5, 8, 8, generator=g)
    out = ops.concatenate(x.to(gpu), y.to(gpu))
    assert torch.equal(ops.to_nchw(out).cpu(), torch.cat((x, y), dim=1))
    # gradient of the image part is the channel slice
    xd = x.to(gpu).requires_grad_(True)
    out = ops.concatenate(xd, onehot.to(gpu))
    gout = torch.randn(6, 13, 8, 8, generator=g)
    out.backward(gout.to(gpu))
    assert torch.equal(ops.to_nchw(xd.grad).cpu(), gout[:, :3])
    # layout round trip
    t = torch.randn(5, 7, 9, 3, generator=g).to(gpu)
    assert torch.equal(ops.to_nchw(ops.to_nhwc(t)).cpu(), t.cpu())
    assert torch.equal(ops.to_nhwc(t).cpu(), t.cpu())  # logical values unchanged by a layout change


def test_float64_batches_are_rounded_on_the_device_like_float(gpu):
    """A float64 batch (numpy's default dtype) is uploaded as it is and rounded to float32 on the device, images in the
    pass that re-lays them to NHWC: the result must be, bit for bit, what `.float()` gives (AbstractGAN1v1.py:77), for
    images with ragged tile sizes, for vectors, and for values that sit exactly between two float32 neighbours (ties
    to even)."""
    from vegans.engine import ops
    g = torch.Generator().manual_seed(3)
    for shape in ((5, 3, 9, 7), (2, 11, 32, 32), (3, 1, 33, 65), (7, 20), (1, 1)):
        x = torch.randn(shape, generator=g, dtype=torch.float64) * 3
        flat = x.reshape(-1)
        flat[0] = 1.0 + 2.0 ** -24            # a tie: rounds to 1.0 (even)
        if flat.numel() > 1:
            flat[1] = 1.0 + 3 * 2.0 ** -24    # a tie: rounds up to 1 + 2^-22 (even)
        got = ops.from_float64(x.to(gpu))
        assert got.dtype == torch.float32 and got.shape == x.shape
        if x.dim() == 4:
            assert got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got.cpu(), x.float()), shape


@pytest.mark.parametrize("fused", [False, True])
def test_gradient_penalty_with_bn_critic_double_backward(gpu, fused):
    """Second-order path: conv -> BN -> leaky relu -> conv -> BN -> leaky relu -> conv, penalty on d out / d in;
    once with separate norm and activation operators, once with the activation fused into the norm's kernels."""
    from vegans import _hip
    from vegans.engine import ops
    g = torch.Generator().manual_seed(21)
    B = 6
    x = torch.randn(B, 3, 16, 16, generator=g)
    w1 = torch.randn(8, 3, 4, 4, generator=g) * 0.2
    w2 = torch.randn(16, 8, 4, 4, generator=g) * 0.1
    w3 = torch.randn(1, 16, 4, 4, generator=g) * 0.1
    g1, b1 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    g2, b2 = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1

    def ref_critic(inp, p):
        w1, g1, b1, w2, g2, b2, w3 = p
        h = F.conv2d(inp, w1, None, 2, 1)
        h = F.leaky_relu(F.batch_norm(h, None, None, g1, b1, True, 0.1, 1e-5), 0.2)
        h = F.conv2d(h, w2, None, 2, 1)
        h = F.leaky_relu(F.batch_norm(h, None, None, g2, b2, True, 0.1, 1e-5), 0.2)
        return F.conv2d(h, w3, None, 1, 0).flatten(1)

    def norm_act(h, gamma, beta):
        if fused:
            return ops.batch_norm(h, gamma, beta, None, None, True, 0.1, 1e-5, _hip.ACT_LRELU, 0.2)
        return ops.activation(ops.batch_norm(h, gamma, beta, None, None, True, 0.1, 1e-5), _hip.ACT_LRELU, 0.2)

    def hip_critic(inp, p):
        w1, g1, b1, w2, g2, b2, w3 = p
        h = norm_act(ops.conv2d(inp, w1, None, (2, 2), (1, 1)), g1, b1)
        h = norm_act(ops.conv2d(h, w2, None, (2, 2), (1, 1)), g2, b2)
        return ops.flatten(ops.conv2d(h, w3, None, (1, 1), (0, 0)))

    params = [w1, g1, b1, w2, g2, b2, w3]
    pr = [t.clone().requires_grad_(True) for t in params]
    xr = x.clone().requires_grad_(True)
    out = ref_critic(xr, pr)
    (gr,) = torch.autograd.grad(out, xr, torch.ones_like(out), create_graph=True)
    pen_r = ((gr.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    pen_r.backward()

    pd = [t.to(gpu).requires_grad_(True) for t in params]
    xd = x.to(gpu).requires_grad_(True)
    out = hip_critic(xd, pd)
    ones = ops.fill_(torch.empty_like(out), 1.0)
    (gd,) = torch.autograd.grad(out, xd, ones, create_graph=True)
    pen_d = ops.gradient_penalty(gd)
    pen_d.backward()
    _close(gd, gr, what="d critic / d input")
    _close(pen_d, pen_r, what="penalty")
    for name, a, b in zip(["w1", "g1", "b1", "w2", "g2", "b2", "w3"], pd, pr):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) < 1e-6
            continue
        if a.grad is None:
            # a shift in front of a leaky relu reaches the penalty through the piecewise-constant slope mask only:
            # torch reports an all-zero gradient, the engine none at all
            assert float(b.grad.abs().max()) == 0.0, name
            continue
        _close(a.grad, b.grad, tol=1e-4, what="penalty grad " + name)


@pytest.mark.parametrize("act,slope", [("lrelu", 0.2), ("relu", 0.0)])
def test_stem_convolution_with_fused_activation(gpu, act, slope):
    """Conv2d(3 -> 64) -> (Leaky)ReLU as the first layer of a BatchNorm-free critic stem: the activation runs in the
    convolution's epilogue. Output, first-order gradients and the gradient-penalty (second-order) path against torch;
    and the fused operator must equal the two separate operators bit for bit."""
    from vegans import _hip
    from vegans.engine import ops
    g = torch.Generator().manual_seed(31)
    B = 5
    x = torch.randn(B, 3, 18, 14, generator=g)
    w1 = torch.randn(64, 3, 4, 4, generator=g) * 0.2
    b1 = torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(1, 64, 3, 3, generator=g) * 0.1
    code = _hip.ACT_LRELU if act == "lrelu" else _hip.ACT_RELU
    f_act = (lambda t: F.leaky_relu(t, slope)) if act == "lrelu" else F.relu

    def run(critic, xs, ps, penalty, ones):
        out = critic(xs, ps)
        (gx,) = torch.autograd.grad(out, xs, ones(out), create_graph=True)
        loss = penalty(gx) + out.mean()
        loss.backward()
        return out.detach(), gx.detach(), [p.grad for p in ps]

    ref = lambda inp, p: F.conv2d(f_act(F.conv2d(inp, p[0], p[1], 2, 1)), p[2], None, 1, 0).flatten(1)
    pr = [t.clone().requires_grad_(True) for t in (w1, b1, w2)]
    out_r, gx_r, gr = run(ref, x.clone().requires_grad_(True), pr,
                          lambda t: ((t.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean(), torch.ones_like)

    def hip(fused):
        def critic(inp, p):
            if fused:
                h = ops.conv2d(inp, p[0], p[1], (2, 2), (1, 1), code, slope)
            else:
                h = ops.activation(ops.conv2d(inp, p[0], p[1], (2, 2), (1, 1)), code, slope)
            return ops.flatten(ops.conv2d(h, p[2], None, (1, 1), (0, 0)))
        pd = [t.to(gpu).requires_grad_(True) for t in (w1, b1, w2)]
        return run(critic, x.to(gpu).requires_grad_(True), pd, ops.gradient_penalty,
                   lambda o: ops.fill_(torch.empty_like(o), 1.0))

    lib = _hip.load()
    import ctypes
    geom = ops.Geom(B, 18, 14, 3, 64, 4, 4, 2, 2, 1, 1)
    assert lib.vg_conv2d_fwd_act_applies(ctypes.byref(geom.c), code) == 1
    out_f, gx_f, gf = hip(True)
    out_s, gx_s, gs = hip(False)
    _close(out_f, out_r, what="critic output")
    _close(gx_f, gx_r, what="d critic / d input")
    for name, a, b in zip(("w1", "b1", "w2"), gf, gr):
        _close(a, b, tol=1e-4, what="gradient " + name)
    assert torch.equal(out_f, out_s) and torch.equal(gx_f, gx_s)
    for a, b in zip(gf, gs):
        assert torch.equal(a, b)


def test_gp_interpolate(gpu):
    from vegans.engine import ops
    g = torch.Generator().manual_seed(2)
    real, fake = torch.randn(5, 3, 4, 4, generator=g), torch.randn(5, 3, 4, 4, generator=g)
    alpha = torch.rand(5, 1, 1, 1, generator=g)
    out = ops.gp_interpolate(alpha.to(gpu), real.to(gpu), fake.to(gpu))
    assert out.requires_grad
    _close(out, alpha * real + (1 - alpha) * fake, tol=1e-6, what="interp")


def test_optimizers_match_torch(gpu):
    from vegans.engine import optim as hip_optim
    g = torch.Generator().manual_seed(4)
    shapes = [(16, 8, 4, 4), (33,), (7, 5)]
    cases = [
        (torch.optim.RMSprop, dict(lr=1e-2)),
        (torch.optim.RMSprop, dict(lr=5e-3, momentum=0.9, centered=True, weight_decay=1e-3)),
        (torch.optim.Adam, dict(lr=1e-3)),
        (torch.optim.Adam, dict(lr=1e-3, betas=(0.5, 0.9), amsgrad=True, weight_decay=1e-2)),
        (torch.optim.AdamW, dict(lr=1e-3)),
        (torch.optim.SGD, dict(lr=1e-2, momentum=0.9, nesterov=True)),
    ]
    for cls, kw in cases:
        init = [torch.randn(s, generator=g) for s in shapes]
        pr = [t.clone().requires_grad_(True) for t in init]
        pd = [t.clone().to(gpu).requires_grad_(True) for t in init]
        optr, optd = cls(pr, **kw), cls(pd, **kw)
        for _ in range(4):
            grads = [torch.randn(s, generator=g) for s in shapes]
            for a, b, gg in zip(pr, pd, grads):
                a.grad = gg.clone()
                b.grad = gg.clone().to(gpu)
            optr.step()
            hip_optim.fused_step(optd)
        for a, b in zip(pr, pd):
            _close(b, a, tol=2e-6, what="%s %s" % (cls.__name__, kw))
        # state stays loadable by stock torch
        sd = optd.state_dict()
        assert len(sd["state"]) == len(shapes)


def test_clamp_fused_and_bit_exact(gpu):
    from vegans.engine import optim as hip_optim
    g = torch.Generator().manual_seed(8)
    init = torch.randn(1000, generator=g) * 0.02
    grad = torch.randn(1000, generator=g)
    pr = init.clone().requires_grad_(True)
    pd = init.clone().to(gpu).requires_grad_(True)
    pr.grad, pd.grad = grad.clone(), grad.clone().to(gpu)
    optr, optd = torch.optim.RMSprop([pr], lr=1e-3), torch.optim.RMSprop([pd], lr=1e-3)
    optr.step()
    pr.data.clamp_(-0.01, 0.01)
    hip_optim.fused_step(optd, clamp=(-0.01, 0.01))
    out = pd.detach().cpu()
    assert float(out.abs().max()) == float(np.float32(0.01))
    clipped = pr.detach().abs() >= np.float32(0.01)
    assert torch.equal(out[clipped], pr.detach()[clipped])      # clamped entries are bit-identical
    _close(out, pr, tol=2e-6, what="clamped rmsprop")


def test_grad_sum_multi_adds_in_the_given_order(gpu):
    """vg_grad_sum_multi: out = scale * (((s0 + s1) + s2) + ...), bit for bit the left-to-right f32 sum, for many
    tensors in one launch (odd sizes take the scalar path), in place into the first source, and chained beyond eight
    sources."""
    from vegans.engine import grads
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 4, 4, 16), (7,), (33, 5), (128,), (3, 3, 3, 3)]
    counts = [4, 1, 3, 11, 2]
    items, want = [], []
    for shape, k in zip(shapes, counts):
        srcs = [(torch.randn(shape, generator=g) * 10.0 ** (i % 5 - 2)) for i in range(k)]
        acc = srcs[0].clone()
        for t in srcs[1:]:
            acc = acc + t
        want.append(acc * np.float32(0.25))
        p = torch.empty(shape, device=gpu)
        items.append((p, [t.to(gpu) for t in srcs], None))
    got = grads.sum_into(items, 0.25)
    for (p, srcs, _), out, ref in zip(items, got, want):
        assert out.data_ptr() == srcs[0].data_ptr()          # in place into the first contribution
        assert torch.equal(out.cpu(), ref)
    # separate outputs (data-parallel bucket slots), scale 1: a single source is copied
    outs = [torch.full(s, 9.0, device=gpu) for s in shapes[:2]]
    srcs = [[torch.randn(s, generator=g).to(gpu) for _ in range(k)] for s, k in zip(shapes[:2], (3, 1))]
    res = grads.sum_into([(o, ss, o) for o, ss in zip(outs, srcs)], 1.0)
    assert torch.equal(res[0].cpu(), ((srcs[0][0].cpu() + srcs[0][1].cpu()) + srcs[0][2].cpu()))
    assert torch.equal(res[1].cpu(), srcs[1][0].cpu())


def test_losses_on_strided_predictions(gpu):
    """A prediction that is a channel slice of a wider activation (not one dense block) must be compacted before the
    loss kernels walk it as flat memory."""
    from vegans.engine import ops
    g = torch.Generator().manual_seed(5)
    wide = torch.rand(6, 5, 3, 3, generator=g) * 0.8 + 0.1
    wide_d = ops.to_nhwc(wide.to(gpu))
    for name, fn, ref in (("mean", lambda p: ops.loss_mean(p, -1.0), lambda p: -p.mean()),
                          ("bce", lambda p: ops.loss_bce(p, 1), lambda p: F.binary_cross_entropy(p, torch.ones_like(p))),
                          ("mse", lambda p: ops.loss_mse(p, 0), lambda p: (p ** 2).mean())):
        base = wide_d.clone().requires_grad_(True)
        pred = base[:, 1:3]                                   # NHWC channel slice: strided
        assert not pred.is_contiguous() and not pred.is_contiguous(memory_format=torch.channels_last)
        loss = fn(pred)
        loss.backward()
        rb = wide.clone().requires_grad_(True)
        rl = ref(rb[:, 1:3])
        rl.backward()
        _close(loss, rl, 2e-5, name)
        _close(base.grad, rb.grad, 2e-5, name + " grad")
    a = wide_d.clone().requires_grad_(True)
    l1 = ops.loss_l1(a[:, 0:2], wide_d[:, 3:5])
    l1.backward()
    ra = wide.clone().requires_grad_(True)
    rl = (ra[:, 0:2] - wide[:, 3:5]).abs().mean()
    rl.backward()
    _close(l1, rl, 2e-5, "l1")
    _close(a.grad, ra.grad, 2e-5, "l1 grad")


def test_clamp_reaches_parameters_without_gradient(gpu):
    """The reference clamps every critic parameter after the critic's step (WassersteinGAN.py:78-79), also those that
    received no gradient in this step."""
    from vegans.engine import optim as hip_optim
    used = torch.nn.Parameter(torch.tensor([0.5, -0.5, 0.001], device=gpu))
    unused = torch.nn.Parameter(torch.tensor([0.3, -0.2, 0.005, float("nan")], device=gpu))
    opt = torch.optim.RMSprop([used, unused], lr=1e-3)
    used.grad = torch.ones(3, device=gpu)
    hip_optim.fused_step(opt, clamp=(-0.01, 0.01))
    got = unused.detach().cpu()
    assert got[:3].tolist() == [float(np.float32(0.01)), float(np.float32(-0.01)), float(np.float32(0.005))]
    assert torch.isnan(got[3])
    assert float(used.detach().abs().max()) == float(np.float32(0.01))


@pytest.mark.gpu
def test_kernel_timing_events_hold_the_kernels_own_duration(gpu):
    """vg_time_next_conv hands a pair of events to the next contraction launch: the pair is consumed by exactly one kernel
    (MFMA tile, stem and thin kernels alike), results are unchanged, and the interval is positive and not longer than
    the same launch bracketed by markers on the stream."""
    from vegans import _hip
    from vegans.engine import ops
    lib = _hip.load()
    torch.manual_seed(5)
    for shape in ((32, 16, 16, 64, 128, 4, 2, 1), (16, 32, 32, 3, 64, 4, 2, 1), (16, 4, 4, 512, 1, 4, 1, 0)):
        N, H, W, Ci, Co, k, s, p = shape
        g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
        x = torch.randn(N, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        want = ops.raw_conv_fwd(g, x, w, None)
        timer = ops.ConvTimer(capacity=8)
        ops.conv_timer = timer
        try:
            for _ in range(3):
                got = ops.raw_conv_fwd(g, x, w, None)
        finally:
            ops.conv_timer = None
        outer0, outer1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outer0.record()
        ops.raw_conv_fwd(g, x, w, None)
        outer1.record()
        torch.cuda.synchronize()
        assert lib.vg_time_pending() == 0 and timer.untimed == 0
        assert torch.equal(got, want)
        (name, (n, ms, flops)), = timer.summary().items()
        assert n == 3 and flops == 3 * 2.0 * N * g.Ho * g.Wo * Co * k * k * Ci
        assert 0.0 < ms / 3 <= outer0.elapsed_time(outer1) * 1.5 + 0.05, (name, ms / 3, outer0.elapsed_time(outer1))


@pytest.mark.gpu
@pytest.mark.parametrize("case,pairs", [((2, 64, 16, 16, 128, 4, 2, 1), 4), ((3, 16, 9, 7, 24, 3, 1, 1), 3),
                                        ((5, 6, 20, 20, 64, 4, 2, 1), 2), ((2, 96, 6, 6, 40, 5, 2, 2), 1)])
def test_weight_gradient_over_several_pairs_in_one_launch(gpu, case, pairs):
    """vg_conv2d_wgrad_partial_multi + vg_conv2d_wgrad_reduce_multi: the weight gradient of one layer summed over several
    (gy, x) pairs, against the sum of torch's weight gradients on the CPU."""
    import ctypes
    from vegans import _hip
    from vegans.engine import ops
    N, Ci, H, W, Co, k, s, p = case
    lib = _hip.load()
    gen = torch.Generator().manual_seed(sum(case) + pairs)
    geo = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    assert lib.vg_conv2d_wgrad_multi_applies(ctypes.byref(geo.c))
    CL = torch.channels_last
    want = torch.zeros(Co, Ci, k, k)
    xs, gs = [], []
    for _ in range(pairs):
        x = torch.randn(N, Ci, H, W, generator=gen)
        gy = torch.randn(N, Co, geo.Ho, geo.Wo, generator=gen)
        want += torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), gy, stride=s, padding=p)
        xs.append(x.to(gpu).contiguous(memory_format=CL))
        gs.append(gy.to(gpu).contiguous(memory_format=CL))
    gw = torch.empty((Co, Ci, k, k), dtype=torch.float32, device=gpu, memory_format=CL)
    nbytes = lib.vg_conv2d_wgrad_multi_workspace_bytes(ctypes.byref(geo.c), pairs)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gpu)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    gptr = (ctypes.c_void_p * pairs)(*[t.data_ptr() for t in gs])
    xptr = (ctypes.c_void_p * pairs)(*[t.data_ptr() for t in xs])
    st = _hip._stream()
    _hip.call("vg_conv2d_wgrad_partial_multi", ctypes.byref(geo.c), pairs, gptr, xptr, P(gw), P(ws), ctypes.c_size_t(nbytes), st)
    _hip.call("vg_conv2d_wgrad_reduce_multi", ctypes.byref(geo.c), pairs, P(gw), P(ws), st)
    _close(gw, want, what="weight gradient over %d pairs" % pairs)
    stem = ops.Geom(4, 64, 64, 3, 64, 4, 4, 2, 2, 1, 1)
    assert not lib.vg_conv2d_wgrad_multi_applies(ctypes.byref(stem.c))
    assert lib.vg_conv2d_wgrad_partial_multi(ctypes.byref(geo.c), 5, gptr, xptr, P(gw), P(ws), ctypes.c_size_t(nbytes), st) == -1


def test_eight_wave_kernels_are_not_given_factors_beyond_24_bits(gpu):
    """The eight-wave f32 contraction kernels form their address products with 24-bit multiplies; a problem with 2^23 or
    more pixels (or any other factor that large) has to run on the four-wave variants, which multiply in 32 bits. The
    library names the variant a geometry gets, and a forward convolution over 8.4 million input pixels agrees with torch."""
    from vegans.engine import ops
    small = ops.Geom(256, 32, 32, 64, 128, 4, 4, 2, 2, 1, 1)
    assert ops.conv_variant(0, small).endswith("0, 8, 0>"), ops.conv_variant(0, small)
    N, Ci, H, W, Co, k, s, p = 1, 8, 2900, 2900, 64, 4, 2, 1            # 8.41e6 input pixels >= 2^23
    g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    assert N * H * W >= 1 << 23
    assert ops.conv_variant(0, g) == "igemm_kernel<0, 128, 64, 2, false, 4, 4, 0, 4, 0>", ops.conv_variant(0, g)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, k, k, generator=gen) * 0.1
    want = F.conv2d(x, w, None, stride=s, padding=p)
    got = ops.conv2d(x.to(gpu), w.to(gpu), None, (s, s), (p, p))
    _close(got, want, what="forward convolution over 2^23 pixels (four-wave variant)")


@pytest.mark.gpu
@pytest.mark.parametrize("act,slope", [("relu", 0.0), ("lrelu", 0.2)])
def test_bn_backward_mask_rebuilt_from_input_equals_mask_from_output(gpu, act, slope):
    """The norm's backward kernels rebuild the fused activation's slope mask from (x, mean, invstd, gamma, beta); the
    result must be bit-identical to reading the mask off the forward pass's output, also for pre-activations that are
    tiny, exactly zero, or whose sign hangs on the last rounding of the affine map."""
    import ctypes
    from ctypes import c_float, c_size_t
    from vegans import _hip
    from vegans.engine import ops
    from vegans.engine.ops import _ptr, _stream, call
    code = {"relu": _hip.ACT_RELU, "lrelu": _hip.ACT_LRELU}[act]
    torch.manual_seed(11)
    m, c = 4096, 64
    x = torch.randn(m, c, device="cuda")
    gamma = torch.randn(c, device="cuda")
    beta = torch.randn(c, device="cuda") * 0.1
    mean = torch.empty(c, device="cuda")
    invstd = torch.empty(c, device="cuda")
    nbytes = _hip.load().vg_bn_workspace_bytes(m, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    wsp = ctypes.c_void_p(ws.data_ptr())
    call("vg_bn_stats", _ptr(x), m, c, c_float(1e-5), c_float(0.1), _ptr(mean), _ptr(invstd), None, None, None, wsp,
         c_size_t(nbytes), _stream())
    # rows whose pre-activation is zero or within an ulp of it: x = mean - beta / (invstd * gamma), and neighbours
    root = mean - beta / (invstd * gamma)
    x[0] = root
    x[1] = torch.nextafter(root, root + 1)
    x[2] = torch.nextafter(root, root - 1)
    x[3] = mean
    y = torch.empty_like(x)
    call("vg_bn_apply", _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y), m, c, code, c_float(slope),
         _stream())
    gy = torch.randn(m, c, device="cuda")
    u = torch.randn(m, c, device="cuda")
    out = {}
    for how, (yo, ga, be) in {"output": (y, None, None), "input": (None, gamma, beta)}.items():
        sums = torch.empty(2, c, device="cuda")
        gx = torch.empty_like(x)
        call("vg_bn_bwd_reduce", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums), wsp, c_size_t(nbytes),
             _ptr(yo), code, c_float(slope), _ptr(ga), _ptr(be), _stream())
        call("vg_bn_bwd_apply", _ptr(gy), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums), _ptr(gx), m, c,
             _ptr(yo), code, c_float(slope), _ptr(be), _stream())
        sums5 = torch.empty(5, c, device="cuda")
        d_gy, d_x, d_gamma = torch.empty_like(x), torch.empty_like(x), torch.empty(c, device="cuda")
        call("vg_bn_dbwd_reduce", _ptr(gy), _ptr(x), _ptr(u), _ptr(mean), _ptr(invstd), m, c, _ptr(sums5), wsp,
             c_size_t(nbytes), _ptr(yo), code, c_float(slope), _ptr(ga), _ptr(be), _stream())
        call("vg_bn_dbwd_apply", _ptr(gy), _ptr(x), _ptr(u), None, None, _ptr(mean), _ptr(invstd), _ptr(gamma),
             _ptr(sums5), _ptr(d_gy), _ptr(d_x), _ptr(d_gamma), m, c, _ptr(yo), code, c_float(slope), _ptr(be), _stream())
        torch.cuda.synchronize()
        out[how] = (sums, gx, sums5, d_gy, d_x, d_gamma)
    names = ("sums", "gx", "sums5", "d_gy", "d_x", "d_gamma")
    for name, a, b in zip(names, out["output"], out["input"]):
        assert torch.equal(a, b), (name, int((a != b).sum()), float((a - b).abs().max()))
    # the constructed rows do sit on the kink: tiny pre-activations of both signs
    assert float(y[:3].abs().max()) < 1e-4 and bool((y[:3] > 0).any()) and bool((y[:3] <= 0).any())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,shape,served_by_epilogue", [
    ("conv", (64, 16, 16, 64, 128, 4, 2, 1), True),       # 64-row tiles, 128 columns
    ("conv", (5, 17, 13, 32, 96, 3, 1, 1), True),         # ragged last row tile, 96 of 128 columns
    ("conv", (16, 8, 8, 128, 256, 4, 2, 1), True),        # two column tiles
    ("conv", (160, 32, 32, 32, 128, 4, 2, 1), True),      # 128-row tiles (320 of them)
    ("conv", (4, 9, 9, 8, 16, 3, 1, 1), False),           # a shape the statistics kernels do not serve: separate pass
    ("convT", (32, 8, 8, 256, 128, 4, 2, 1), True),       # transposed: four parity classes, 64-row tiles
    ("convT", (16, 25, 24, 64, 64, 4, 2, 1), True),       # transposed, 128 x 64 tiles, ragged last tile of every class
    ("convT", (3, 5, 7, 64, 64, 4, 2, 1), False),         # transposed, too small for the eight-wave tiles
    ("convT", (40, 16, 16, 128, 128, 3, 1, 1), True),     # transposed, stride 1: a single class
    # batch 256, the DCGAN-64 layers in front of a BatchNorm: up to 2048 per-tile records merged per layer
    ("conv", (256, 32, 32, 64, 128, 4, 2, 1), True),
    ("conv", (256, 16, 16, 128, 256, 4, 2, 1), True),
    ("conv", (256, 8, 8, 256, 512, 4, 2, 1), True),
    ("convT", (256, 4, 4, 512, 256, 4, 2, 1), True),
    ("convT", (256, 8, 8, 256, 128, 4, 2, 1), True),
    ("convT", (256, 16, 16, 128, 64, 4, 2, 1), True),
])
def test_batch_statistics_from_the_convolution_epilogue(gpu, kind, shape, served_by_epilogue):
    """A (transposed) convolution followed by a training-mode BatchNorm inside a Sequential: the norm takes its batch
    statistics from per-tile sums the convolution's epilogue leaves. Mean, inverse std, running statistics and the
    output must agree with the separate statistics pass to rounding (5e-6 of scale), and both with torch."""
    from torch import nn
    from vegans.engine import modules, ops
    N, H, W, Ci, Co, k, s, p = shape
    torch.manual_seed(7)
    conv = nn.Conv2d(Ci, Co, k, s, p, bias=True) if kind == "conv" else nn.ConvTranspose2d(Ci, Co, k, s, p, bias=True)
    ref = nn.Sequential(conv, nn.BatchNorm2d(Co), nn.LeakyReLU(0.2))
    with torch.no_grad():
        conv.bias.add_(3.0)                      # a mean well away from zero: the shifted sums must cope
    x = torch.randn(N, Ci, H, W)
    want = ref(x)
    outs = {}
    for fused in (True, False):
        import copy
        net = copy.deepcopy(ref)
        net[1].reset_running_stats()
        net = net.to("cuda")
        modules.bind(net)
        modules.prepare_parameters(net)
        net.train()
        seen = []
        orig = ops.norm_stats.claim
        ops.norm_stats.claim = (lambda *a, orig=orig, seen=seen: seen.append(orig(*a)) or seen[-1]) if fused else (lambda *a: None)
        try:
            y = net(x.to("cuda"))
        finally:
            ops.norm_stats.claim = orig
        outs[fused] = (y.detach().cpu(), net[1].running_mean.cpu(), net[1].running_var.cpu(), int(net[1].num_batches_tracked))
        if fused:
            served = seen and seen[0] is not None
    assert bool(served) == served_by_epilogue, (served, shape)
    for a, b in zip(outs[True][:3], outs[False][:3]):
        _close(a, b, 5e-6, "fused statistics vs separate pass")
    assert outs[True][3] == outs[False][3] == 1
    _close(outs[True][0], want, 2e-5, "conv + batch norm output vs torch")
    _close(outs[True][1], ref[1].running_mean, 2e-5, "running mean vs torch")
    _close(outs[True][2], ref[1].running_var, 2e-5, "running var vs torch")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,shape,act,served_by_epilogue", [
    # (N, H, W of the norm's activation, its channels C, the next layer's channels, k, s, p)
    ("conv", (64, 32, 32, 64, 128, 4, 2, 1), "lrelu", True),      # input gradient of a strided convolution: 4 classes
    ("conv", (5, 17, 13, 96, 32, 3, 1, 1), "relu", True),         # ragged last row tile, 96 of 128 columns
    ("conv", (16, 16, 16, 256, 64, 4, 2, 1), "lrelu", True),      # two column tiles
    ("conv", (24, 15, 14, 128, 64, 4, 2, 1), "none", True),       # no activation between norm and convolution
    ("conv", (4, 9, 9, 16, 8, 3, 1, 1), "relu", False),           # too small for the eight-wave tiles: separate pass
    ("convT", (64, 16, 16, 128, 64, 4, 2, 1), "relu", True),      # input gradient of a transposed convolution
    ("convT", (7, 11, 9, 128, 32, 4, 2, 1), "relu", True),        # ragged
    ("convT", (32, 8, 8, 256, 128, 3, 1, 1), "lrelu", True),      # stride 1
    # batch 256, the DCGAN-64 norms and the layer behind each
    ("conv", (256, 16, 16, 128, 256, 4, 2, 1), "lrelu", True),
    ("conv", (256, 8, 8, 256, 512, 4, 2, 1), "lrelu", True),
    ("convT", (256, 4, 4, 512, 256, 4, 2, 1), "relu", True),
    ("convT", (256, 8, 8, 256, 128, 4, 2, 1), "relu", True),
    ("convT", (256, 16, 16, 128, 64, 4, 2, 1), "relu", True),
])
def test_norm_backward_sums_from_the_convolution_epilogue(gpu, kind, shape, act, served_by_epilogue):
    """BatchNorm (+ activation) in front of a (transposed) convolution: in the backward pass the kernel that produces
    the gradient arriving at the norm also leaves the norm's two column sums. Gradients of the norm's input, gamma and
    beta must agree with the separate reduction pass to rounding (5e-6 of scale), and both with torch (float64)."""
    import copy
    from torch import nn
    from vegans.engine import modules, ops
    N, H, W, C, Co, k, s, p = shape
    torch.manual_seed(11)
    tail = nn.Conv2d(C, Co, k, s, p) if kind == "conv" else nn.ConvTranspose2d(C, Co, k, s, p)
    layers = [nn.BatchNorm2d(C)] + ({"relu": [nn.ReLU()], "lrelu": [nn.LeakyReLU(0.2)], "none": []}[act]) + [tail]
    ref = nn.Sequential(*layers)
    with torch.no_grad():
        ref[0].weight.uniform_(0.5, 1.5)
        ref[0].bias.uniform_(-0.5, 0.5)
    x = torch.randn(N, C, H, W) * 1.5 + 0.3
    probe = torch.randn_like(ref(x))

    def run(net, xin, weight):
        xin = xin.clone().requires_grad_(True)
        (net(xin) * weight).sum().backward()
        return [t.detach().double().cpu() for t in (xin.grad, net[0].weight.grad, net[0].bias.grad)]

    want = run(copy.deepcopy(ref).double(), x.double(), probe.double())
    outs = {}
    for fused in (True, False):
        net = copy.deepcopy(ref).to("cuda")
        modules.bind(net)
        modules.prepare_parameters(net)
        net.train()
        seen = []
        orig_claim, orig_source = ops.norm_bwd.claim, ops.norm_bwd.source_of
        ops.norm_bwd.claim = lambda *a: seen.append(orig_claim(*a)) or seen[-1]
        if not fused:
            ops.norm_bwd.source_of = lambda t: None
        try:
            outs[fused] = run(net, x.to("cuda"), probe.to("cuda"))
        finally:
            ops.norm_bwd.claim, ops.norm_bwd.source_of = orig_claim, orig_source
        assert len(seen) == 1
        if fused:
            assert (seen[0] is not None) == served_by_epilogue, shape
        else:
            assert seen[0] is None
    for name, a, b, w in zip(("gx", "ggamma", "gbeta"), outs[True], outs[False], want):
        _close(a, b, 5e-6, "%s: sums from the epilogue vs separate pass" % name)
        _close(a, w, 3e-5, "%s vs torch float64" % name)


@pytest.mark.gpu
def test_norm_backward_sums_are_not_taken_when_the_gradient_was_summed(gpu):
    """The norm's output feeds TWO consumers: the gradient at the norm is a sum autograd forms, not the tensor the
    convolution kernel wrote, so the per-tile sums that kernel left must be ignored."""
    from torch import nn
    from vegans.engine import modules, ops
    torch.manual_seed(3)
    norm, conv = nn.BatchNorm2d(64), nn.Conv2d(64, 64, 3, 1, 1)
    block = nn.Sequential(norm, nn.ReLU())
    x = torch.randn(16, 64, 16, 16)

    probe = torch.randn(16, 64, 16, 16)

    def run(block, conv, xin):
        xin = xin.clone().requires_grad_(True)
        h = block(xin)
        ((conv(h) * probe.to(xin)).sum() + (h * h).sum()).backward()
        return [t.detach().double().cpu() for t in (xin.grad, block[0].weight.grad, block[0].bias.grad)]

    import copy
    want = run(copy.deepcopy(block).double(), copy.deepcopy(conv).double(), x.double())
    gb, gc = copy.deepcopy(block).to("cuda"), nn.Sequential(copy.deepcopy(conv)).to("cuda")
    for net in (gb, gc):
        modules.bind(net)
        modules.prepare_parameters(net)
        net.train()
    seen = []
    orig = ops.norm_bwd.claim
    ops.norm_bwd.claim = lambda *a: seen.append(orig(*a)) or seen[-1]
    try:
        got = run(gb, gc, x.to("cuda"))
    finally:
        ops.norm_bwd.claim = orig
    assert seen == [None]
    for name, a, w in zip(("gx", "ggamma", "gbeta"), got, want):
        _close(a, w, 3e-5, "%s vs torch float64" % name)


@pytest.mark.gpu
@pytest.mark.parametrize("layers,in_shape", [
    # image stem with LeakyReLU in its epilogue, in front of a strided convolution served by the eight-wave tiles
    (lambda nn: [nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1)], (32, 3, 32, 32)),
    # 128 channels in between: the 128 x 128 tile, which reads the activation in its epilogue
    (lambda nn: [nn.Conv2d(3, 128, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(128, 128, 4, 2, 1)], (64, 3, 32, 32)),
    # small second convolution: no kernel with the mask epilogue, the separate mask kernel runs inside its backward
    (lambda nn: [nn.Conv2d(3, 16, 4, 2, 1), nn.ReLU(), nn.Conv2d(16, 8, 3, 1, 1)], (4, 3, 16, 16)),
    # activation as an operator of its own (no fused convolution kernel for this geometry)
    (lambda nn: [nn.Conv2d(64, 64, 3, 1, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1)], (16, 64, 16, 16)),
    # two pairs in a row, ragged tiles, a norm behind the last convolution
    (lambda nn: [nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 3, 1, 1), nn.ReLU(),
                 nn.Conv2d(128, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), nn.LeakyReLU(0.2)], (9, 3, 26, 22)),
    # batch 256: the DCGAN-64 critic's stem, its LeakyReLU and the first fat layer
    (lambda nn: [nn.Conv2d(3, 64, 4, 2, 1, bias=False), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1, bias=False)],
     (256, 3, 64, 64)),
])
def test_activation_gradient_carried_by_the_next_convolutions_input_gradient(gpu, layers, in_shape):
    """conv -> ReLU / LeakyReLU -> conv inside a Sequential: the second convolution's backward returns its input
    gradient already multiplied by the activation's slope mask and the activation's own backward pass is skipped.
    First- and second-order gradients (gradient-penalty shaped) must equal, bit for bit, those of the separate mask
    pass, and agree with torch in float64."""
    import copy
    from torch import nn
    from vegans.engine import modules, ops
    torch.manual_seed(5)
    ref = nn.Sequential(*layers(nn))
    x = torch.randn(*in_shape)
    probe = torch.randn_like(ref(x))
    probe2 = torch.rand(*in_shape) + 0.5

    def run(net, xin, probe, probe2):
        xin = xin.clone().requires_grad_(True)
        out = net(xin)
        (gx,) = torch.autograd.grad((out * probe).sum(), xin, create_graph=True)
        ((gx * gx * probe2).sum() + (out * probe).sum()).backward()
        return [gx.detach().double().cpu(), xin.grad.double().cpu()] + [q.grad.double().cpu() for q in net.parameters()]

    want = run(copy.deepcopy(ref).double(), x.double(), probe.double(), probe2.double())
    outs = {}
    for carried in (True, False):
        net = copy.deepcopy(ref).to("cuda")
        modules.bind(net)
        modules.prepare_parameters(net)
        net.train()
        accepted = []
        orig_take, orig_announce = ops.act_grad.take_input, ops.act_grad.announce
        ops.act_grad.take_input = lambda t: accepted.append(orig_take(t)) or accepted[-1]
        if not carried:
            ops.act_grad.announce = lambda: None
        try:
            outs[carried] = run(net, x.to("cuda"), probe.to("cuda"), probe2.to("cuda"))
        finally:
            ops.act_grad.take_input, ops.act_grad.announce = orig_take, orig_announce
        pairs = sum(1 for a in accepted if a is not None)
        assert pairs == (sum(isinstance(m, (nn.ReLU, nn.LeakyReLU)) and i + 1 < len(ref) and isinstance(ref[i + 1], nn.Conv2d)
                             for i, m in enumerate(ref)) if carried else 0)
    for k, (a, b, w) in enumerate(zip(outs[True], outs[False], want)):
        assert torch.equal(a, b), ("tensor %d differs between the two backward forms" % k, float((a - b).abs().max()))
        _close(a, w, 5e-5, "tensor %d vs torch float64" % k)
