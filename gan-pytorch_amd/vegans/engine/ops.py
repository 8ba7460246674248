"""Differentiable operators of the training step, each a thin `torch.autograd.Function` over the HIP kernels.

torch's autograd engine only sequences these nodes (it is plumbing); every tensor value is produced by a
kernel of libvegans_hip.so.  Backward passes are themselves written with the Functions below, so
`create_graph=True` (the gradient penalty of WassersteinGANGP.py:67-82 in the reference) differentiates through
them a second time:

    conv_fwd(x, w)  --backward-->  conv_dgrad(gy, w), conv_wgrad(gy, x)
    conv_dgrad(gy, w) --backward--> conv_fwd(ggx, w), conv_wgrad(gy, ggx)
    conv_wgrad(gy, x) --backward--> conv_fwd(x, ggw),  conv_dgrad(gy, ggw)

Gradients of parameters are not left to autograd's accumulation: every operator that produces one hands it to the
gradient ledger (engine/grads.py) under a label derived from program order, and the ledger adds the contributions of a
parameter in label order.  An operator called by user code draws a fresh ticket as its label; an operator created
inside a backward pass takes its parent's label extended by a fixed child index.

Memory layout: 4-d activations are NCHW-shaped tensors with channels_last strides (NHWC in memory); 4-d weights
are kept channels_last as well ([Co][kh][kw][Ci] in memory).  Tensors that arrive in another layout are
re-laid-out by the layout kernels, never by torch.
"""
import collections
import ctypes
import os
from ctypes import c_float, c_int, c_size_t

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _hip
from .._hip import ConvGeom, _ptr, _stream, call, workspace
from . import grads, streams

CL = torch.channels_last
F32 = torch.float32
BF16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------------------------
# storage types: activations are f32, or - in the bf16 mode - bf16 wherever the channel count is a multiple of 8
# ----------------------------------------------------------------------------------------------------------------
def _p(t):
    """Device pointer of an activation of either storage type (None -> null). The kernel entry point is chosen with
    `_A` from one of the call's activations; every call site that hands a kernel more than one activation first
    passes them through `_same_storage` (a bf16 buffer read as f32 would be read past its end)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype not in (F32, BF16):
        raise _hip.HipError("activation on %s with dtype %s: expected float32 or bfloat16 on the GPU" % (t.device, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def _A(name, like):
    """Entry point `name` for activations stored like `like`: the `_h` twin for bf16."""
    return name + "_h" if like.dtype == BF16 else name


def _same_storage(*ts):
    kinds = {t.dtype for t in ts if t is not None}
    if len(kinds) > 1:
        raise _hip.HipError("activations of one kernel call differ in storage type: %s" % sorted(str(k) for k in kinds))


def _cast_raw(t, dtype):
    """`t` converted to `dtype` (f32 <-> bf16, round to nearest even) with its memory order kept; no autograd."""
    if t is None or t.dtype == dtype:
        return t
    if not _hip._dense(t):
        t = _plain_contiguous(t)
    out = torch.empty_like(t, dtype=dtype)
    if out.stride() != t.stride():
        raise _hip.HipError("cannot convert a tensor with strides %s in place order" % (tuple(t.stride()),))
    if dtype == BF16:
        call("vg_cast_f32_to_bf16", _ptr(t), _p(out), c_size_t(t.numel()), _stream())
    else:
        call("vg_cast_bf16_to_f32", _p(t), _ptr(out), c_size_t(t.numel()), _stream())
    return out


def _f32(t):
    return _cast_raw(t, F32)


# ----------------------------------------------------------------------------------------------------------------
# layout helpers (not differentiable by themselves: a layout change is the identity on the logical tensor)
# ----------------------------------------------------------------------------------------------------------------
def _require_gpu(t):
    if not t.is_cuda:
        raise _hip.HipError("input tensor is on %s: the engine only runs on the GPU" % t.device)
    if t.dtype != F32 and t.dtype != BF16:
        raise _hip.HipError("input tensor is %s: the engine computes in float32 (activations may be stored as bfloat16)" % t.dtype)


def to_nhwc(t):
    """Return `t` (N,C,H,W) with NHWC memory; no copy when it already is."""
    _require_gpu(t)
    if t.dim() != 4:
        return t if t.is_contiguous() else _plain_contiguous(t)
    if t.is_contiguous(memory_format=CL):
        return t
    if not t.is_contiguous():
        t = _plain_contiguous(t)
        if t.is_contiguous(memory_format=CL):
            return t
    if t.dtype == BF16:       # rare (a bf16 feature vector viewed as an image by the user's module): through f32
        return _cast_raw(to_nhwc(_f32(t)), BF16)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device, memory_format=CL)
    call("vg_nchw_to_nhwc", _ptr(t), _ptr(out), n, c, h * w, _stream())
    return out


def from_float64(t):
    """A contiguous float64 device tensor as the engine wants its inputs: float32 (round to nearest even, `.float()`),
    images (4-d, NCHW in memory) re-laid to NHWC in the same pass."""
    if not t.is_cuda:
        raise _hip.HipError("input tensor is on %s: the engine only runs on the GPU" % t.device)
    if t.dtype != torch.float64 or not t.is_contiguous():
        raise _hip.HipError("from_float64 expects a contiguous float64 tensor")
    if t.dim() == 4:
        n, c, h, w = t.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=t.device, memory_format=CL)
        if t.numel():
            call("vg_nchw_to_nhwc_f64", ctypes.c_void_p(t.data_ptr()), _ptr(out), n, c, h * w, _stream())
        return out
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    call("vg_cast_f64_to_f32", ctypes.c_void_p(t.data_ptr()), _ptr(out), c_size_t(t.numel()), _stream())
    return out


def to_nchw(t):
    """Return `t` (N,C,H,W) with plain contiguous NCHW memory."""
    _require_gpu(t)
    if t.dim() != 4 or t.is_contiguous():
        return t if t.is_contiguous() else _plain_contiguous(t)
    if not t.is_contiguous(memory_format=CL):
        return to_nchw(_plain_contiguous(t))
    if t.dtype == BF16:
        return _cast_raw(to_nchw(_f32(t)), BF16)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device)
    call("vg_nhwc_to_nchw", _ptr(t), _ptr(out), n, c, h * w, _stream())
    return out


def _plain_contiguous(t):
    # A channel slice of an NHWC activation (or a column slice of a [rows][C] one) is what autograd hands back for each
    # input of a user module's torch.cat along dim 1 (U-Net skip connections): compact it. Other exotic strides
    # (expands, spatial slices) do not occur on the training path; refuse instead of silently running a stock copy
    # kernel.
    shape, stride = tuple(t.shape), tuple(t.stride())
    # a bf16 slice moves as 4-byte words (pairs of channels): the copy kernel only moves bits
    half = 2 if t.dtype == BF16 else 1
    if t.dim() == 4:
        n, c, h, w = shape
        ctot = stride[3]
        if stride[1] == 1 and ctot > c and stride[2] == w * ctot and stride[0] == h * w * ctot:
            if half == 2 and (c % 2 or ctot % 2 or t.data_ptr() % 4):
                raise _hip.HipError("bf16 channel slice of %d of %d channels is not made of whole 4-byte words" % (c, ctot))
            out = torch.empty(shape, dtype=t.dtype, device=t.device, memory_format=CL)
            call("vg_slice_channels", _p(t), ctot // half, 0, c // half, _p(out), c_size_t(n * h * w), _stream())
            return out
    if t.dim() == 2 and stride[1] == 1 and stride[0] > shape[1]:
        if half == 2 and (shape[1] % 2 or stride[0] % 2 or t.data_ptr() % 4):
            raise _hip.HipError("bf16 column slice is not made of whole 4-byte words")
        out = torch.empty(shape, dtype=t.dtype, device=t.device)
        call("vg_slice_channels", _p(t), stride[0] // half, 0, shape[1] // half, _p(out), c_size_t(shape[0]), _stream())
        return out
    raise _hip.HipError("tensor with strides %s (shape %s) is neither NCHW- nor NHWC-contiguous" %
                        (tuple(t.stride()), tuple(t.shape)))


def _new_act(n, c, h, w, like, flat, dtype=F32):
    if flat:
        return torch.empty((n, c), dtype=dtype, device=like.device)
    return torch.empty((n, c, h, w), dtype=dtype, device=like.device, memory_format=CL)


def _weight_mem(w):
    """Weight memory must be [Co][kh][kw][Ci]: channels_last for 4-d, contiguous for 2-d."""
    if w.dim() == 4:
        return to_nhwc(w)
    return w if w.is_contiguous() else _plain_contiguous(w)


class Geom:
    """Convolution geometry + the ctypes struct handed to the library. `flat` marks Linear layers (2-d tensors)."""
    __slots__ = ("c", "flat", "N", "H", "W", "Ci", "Co", "kh", "kw", "Ho", "Wo", "sh", "sw", "ph", "pw")

    def __init__(self, N, H, W, Ci, Co, kh, kw, sh, sw, ph, pw, flat=False):
        Ho = (H + 2 * ph - kh) // sh + 1
        Wo = (W + 2 * pw - kw) // sw + 1
        if Ho <= 0 or Wo <= 0:
            raise ValueError("convolution output would be empty: H=%d W=%d k=(%d,%d) s=(%d,%d) p=(%d,%d)" %
                             (H, W, kh, kw, sh, sw, ph, pw))
        self.N, self.H, self.W, self.Ci, self.Co = N, H, W, Ci, Co
        self.kh, self.kw, self.Ho, self.Wo = kh, kw, Ho, Wo
        self.sh, self.sw, self.ph, self.pw = sh, sw, ph, pw
        self.flat = flat
        self.c = ConvGeom(N, H, W, Ci, Co, kh, kw, Ho, Wo, sh, sw, ph, pw)

    def x_shape(self):
        return (self.N, self.Ci) if self.flat else (self.N, self.Ci, self.H, self.W)

    def y_shape(self):
        return (self.N, self.Co) if self.flat else (self.N, self.Co, self.Ho, self.Wo)

    def w_numel(self):
        return self.Co * self.kh * self.kw * self.Ci


def _check_shape(t, shape, what):
    if tuple(t.shape) != tuple(shape):
        raise _hip.HipError("%s has shape %s, geometry expects %s" % (what, tuple(t.shape), tuple(shape)))


# ----------------------------------------------------------------------------------------------------------------
# raw kernels (no autograd)
# ----------------------------------------------------------------------------------------------------------------
_variant_names = {}


def conv_variant(mode, g):
    """Name of the kernel the library runs for convolution `mode` (0 fwd, 1 dgrad, 2 wgrad) on geometry g."""
    c = g.c
    key = (mode, c.N, c.H, c.W, c.Ci, c.Co, c.kh, c.kw, c.Ho, c.Wo, c.sh, c.sw, c.ph, c.pw)
    name = _variant_names.get(key)
    if name is None:
        buf = ctypes.create_string_buffer(128)
        call("vg_conv2d_variant", mode, ctypes.byref(c), buf, c_size_t(128))
        name = _variant_names[key] = buf.value.decode()
    return name


class ConvTimer:
    """Optional timing of the convolution kernels (installed by the benchmark). Every launch gets a pair of events that
    the library hands to the kernel dispatch itself (vg_time_next_conv): they hold the begin and end of that kernel's
    execution, not the moments at which markers queued around it were reached, which would add the dispatch latency
    of the kernel to every interval. Durations are read after the timed region. With `only` set, just the launches of
    that kernel variant are timed. The events are made (and recorded once, which is what creates them) ahead of the
    timed passes, `capacity` pairs at a time, so the passes themselves put nothing extra on the stream."""

    def __init__(self, only=None, capacity=1024):
        self.only = only
        self.records = []   # (variant, geometry, start_event, stop_event, mode)
        self.untimed = 0    # launches whose kernel does not take the events (none today)
        self.pool = []
        self.capacity = capacity
        self._grow()

    def _grow(self):
        for _ in range(self.capacity):
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            pair[1].record()
            self.pool.append(pair)

    def launch(self, mode, g, fn):
        name = conv_variant(mode, g)
        if self.only is not None and name != self.only:
            fn()
            return
        if not self.pool:
            self._grow()
        start, stop = self.pool.pop()
        call("vg_time_next_conv", ctypes.c_void_p(start.cuda_event), ctypes.c_void_p(stop.cuda_event))
        try:
            fn()
        finally:
            if _hip.load().vg_time_pending():
                call("vg_time_next_conv", None, None)
                self.untimed += 1
                return
        self.records.append((name, g, start, stop, mode))

    def summary(self):
        """{kernel variant: (launches, total_ms, total_algorithmic_flops)}; call after torch.cuda.synchronize()."""
        out = {}
        for key, g, start, stop, _ in self.records:
            flops = 2.0 * g.N * g.Ho * g.Wo * g.Co * g.kh * g.kw * g.Ci
            n, ms, fl = out.get(key, (0, 0.0, 0.0))
            out[key] = (n + 1, ms + start.elapsed_time(stop), fl + flops)
        return out


conv_timer = None   # a ConvTimer while a benchmark wants per-kernel durations, else None


class inputs_only:
    """While active, backward passes compute gradients with respect to activations only (no weight / bias
    gradients).  Used around `torch.autograd.grad(outputs, inputs=<activation>)` calls such as the gradient penalty's
    first-order pass: custom autograd Functions cannot see which of their gradients that call will keep, and the
    unused weight gradients there cost as much as a critic forward."""
    active = False

    def __enter__(self):
        self.prev = inputs_only.active
        inputs_only.active = True

    def __exit__(self, *exc):
        inputs_only.active = self.prev


PRECISIONS = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "bf16_compute": 1, "fp32_split": 2,
              "bf16x3": 2}
_CANONICAL = {"f32": "fp32", "float32": "fp32", "bfloat16": "bf16", "bf16x3": "fp32_split"}
_mode = os.environ.get("VEGANS_HIP_PRECISION", "fp32").lower()
_mode = _CANONICAL.get(_mode, _mode)
_precision = PRECISIONS[_mode]


def set_matmul_precision(name):
    """Precision mode of the engine. "fp32" (default; f32 MFMA products, f32 tensors), "fp32_split" (f32-grade products
    computed on the bf16 matrix cores: operands split exactly into three bf16 terms, six cross products accumulated in
    f32; agrees with "fp32" to f32 rounding noise), "bf16" (a bf16 DATA PATH: feature maps and their gradients - every
    activation whose channel count is a multiple of 8 - are stored as bf16 and feed the bf16 matrix cores as they lie in
    memory; parameters, their gradients, optimizer state, batch statistics, losses, images and scores stay f32, and all
    arithmetic between a load and a store is f32) or "bf16_compute" (f32 tensors throughout; only the operands of the
    contractions are rounded to bf16 on their way into the matrix cores). Returns the previous name."""
    global _precision, _mode
    if name not in PRECISIONS:
        raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
    previous = _mode
    _mode = _CANONICAL.get(name, name)
    _precision = PRECISIONS[name]
    return previous


def matmul_precision():
    return _mode


def storage_h():
    """Whether feature maps are stored as bf16 (the "bf16" mode)."""
    return _mode == "bf16"


def _feature_dtype(channels):
    """Storage type of an activation the engine produces with `channels` channels / features: bf16 for feature maps
    in the bf16 mode; images, scores and other narrow tensors (channel counts that are not multiples of 8) stay f32."""
    return BF16 if (_mode == "bf16" and channels % 8 == 0) else F32


class CastFn(Function):
    """Storage-type conversion inside a differentiable graph (its backward converts back): used where a gradient has
    to be returned in the storage type of the tensor it belongs to."""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.src = t.dtype
        return _cast_raw(t, dtype)

    @staticmethod
    def backward(ctx, g):
        return (g if g.dtype == ctx.src else CastFn.apply(g, ctx.src)), None


def _stored_like(g, ref):
    """Gradient `g` in the storage type of `ref`, the tensor it is the gradient of."""
    if g is None or g.dtype == ref.dtype:
        return g
    return CastFn.apply(g, ref.dtype)


def as_f32(t):
    """`t` as an f32 tensor (differentiable): what leaves a network for the losses, the penalty and the caller."""
    return CastFn.apply(t, F32) if t.dtype == BF16 else t


def _launch_conv(mode, g, fn):
    if conv_timer is None:
        fn()
    else:
        conv_timer.launch(mode, g, fn)


def _conv_workspace(mode, g, device):
    """(bytes, pointer) of the scratch a contraction needs under the current precision: split-K slabs of the weight
    gradient and, in the split mode, the bf16 terms of both operands."""
    nbytes = _hip.load().vg_conv2d_prec_workspace_bytes(mode, ctypes.byref(g.c), _precision)
    if not nbytes:
        return 0, None
    return nbytes, ctypes.c_void_p(workspace(nbytes, device).data_ptr())


class _PlaneCache:
    """Split-mode and bf16-mode operands (their bf16 terms) kept for the rest of a training iteration, so a tensor that
    feeds several contractions - an activation its forward convolution and its weight gradient, a back-propagated
    gradient the input and the weight gradient, a weight every pass through its layer - is split once. An entry
    keeps its source tensor alive (so the address cannot be handed to another tensor) and is valid for one tensor
    version; everything is dropped when parameters change (optimizer step, clip) and at the iteration's boundaries.
    The cache is only switched on inside the training iteration: a captured HIP graph then contains exactly the
    splits an eager iteration runs, and never refers to planes made outside itself."""

    def __init__(self):
        self.active = False
        self.entries = {}

    def begin(self):
        self.entries.clear()
        self.active = True

    def end(self):
        self.entries.clear()
        self.active = False

    def clear(self):
        self.entries.clear()

    def drop(self, tensors):
        """Forget what was made of `tensors` (parameters an optimizer is about to rewrite in place, behind torch's version
        counters); the other network's weights keep their copies for the rest of the iteration."""
        for t in tensors:
            ptr = t.data_ptr()
            self.entries.pop(ptr, None)
            self.entries.pop((ptr, "h"), None)

    def pointer(self, mode, g, t):
        """Device pointer to the planes of `t` (already in kernel memory order), or None when the library should split
        the operand itself."""
        if not self.active or _precision == 0 or not _hip.load().vg_conv2d_split_applies(mode, ctypes.byref(g.c),
                                                                                         _precision):
            return None
        key = t.data_ptr()
        hit = self.entries.get(key)
        if hit is not None and hit[0]._version == hit[1] and hit[0].numel() == t.numel():
            return ctypes.c_void_p(hit[2].data_ptr())
        n = t.numel()
        planes = torch.empty(_hip.load().vg_split_planes_bytes(n, _precision), dtype=torch.uint8, device=t.device)
        call("vg_split_planes_f32", _ptr(t), ctypes.c_longlong(n), _precision, ctypes.c_void_p(planes.data_ptr()),
             _stream())
        self.entries[key] = (t, t._version, planes)
        return ctypes.c_void_p(planes.data_ptr())

    def bf16_copy(self, t):
        """bf16 copy of f32 tensor `t` in its memory order (a weight on its way into the bf16 contractions); kept for
        the rest of the iteration like the planes above."""
        key = (t.data_ptr(), "h")
        cur = torch.cuda.current_stream(t.device)
        if self.active:
            hit = self.entries.get(key)
            if hit is not None and hit[0]._version == hit[1] and hit[0].numel() == t.numel():
                # made by a pass that runs on another stream (engine/streams.py): order this stream behind the conversion
                if hit[3] != cur:
                    cur.wait_event(hit[4])
                return hit[2]
        copy = torch.empty(t.numel(), dtype=BF16, device=t.device)
        call("vg_cast_f32_to_bf16", _ptr(t), _p(copy), c_size_t(t.numel()), _stream())
        if self.active:
            done = torch.cuda.Event()
            done.record(cur)
            self.entries[key] = (t, t._version, copy, cur, done)
        return copy


plane_cache = _PlaneCache()


class _NormStats:
    """Hand-over of batch statistics from a convolution to the BatchNorm directly behind it. The container that runs
    the two layers (modules.HipSequential) asks for it with `request()` right before it calls the convolution; the
    convolution's epilogue then leaves per-tile column statistics beside its output (f32 MFMA tiles only; other
    kernels and precisions ignore the request), and the norm, finding statistics that belong to its input tensor, merges
    them instead of reading the whole activation again. One-shot in both directions."""

    def __init__(self):
        self.wanted = False
        self.ready = None       # (data_ptr of the output, mode, geometry, statistics buffer)

    def request(self):
        self.wanted = True
        self.ready = None

    def cancel(self):
        self.wanted = False

    def take_request(self):
        wanted, self.wanted = self.wanted, False
        return wanted and _precision == 0

    def offer(self, y, mode, g, stats):
        self.ready = (y.data_ptr(), mode, g, stats)

    def claim(self, x, m, c):
        """The statistics left for activation `x` viewed as [m][c], or None."""
        ready, self.ready = self.ready, None
        if ready is None or ready[0] != x.data_ptr():
            return None
        mode, g = ready[1], ready[2]
        rows, cols = (g.N * g.Ho * g.Wo, g.Co) if mode == 0 else (g.N * g.H * g.W, g.Ci)
        return ready if (rows, cols) == (m, c) else None


norm_stats = _NormStats()


NormSource = collections.namedtuple("NormSource", "x mean invstd gamma beta act slope")


class _NormBackwardSums:
    """Hand-over, in the backward pass, of the two column sums a training-mode BatchNorm needs (sum g m, sum g m xhat)
    from the convolution kernel that PRODUCES the gradient g arriving at the norm's output. In the forward pass the norm
    notes its output and what describes it (`produced`); the convolution that takes exactly that tensor as its input
    keeps the description (`source_of`); when that convolution's backward computes its input gradient it asks the
    contraction kernel for the sums (`request` ... `take_request`), which leaves them beside the gradient (`offer`); the
    norm's backward operator takes them if, and only if, the gradient it is handed IS that tensor, unmodified
    (`claim`: same memory, same version counter, same norm). A gradient that autograd summed from several consumers
    is another tensor (or a later version) and is reduced the ordinary way.

    Offers wait under the address of their gradient tensor (which the entry keeps alive, so the address names it), not
    in one slot: between a convolution's backward and the backward of the norm in front of it autograd may run nodes of
    other chains (the critic's fake, real and penalty passes are three chains, interleaved by node priority), and with
    one slot whether their offers and claims displaced each other, and so whether a norm took the sums from the
    epilogue or ran its own reduction pass, would depend on autograd's schedule instead of on the program: two runs from
    the same seeds would differ in the last bits. Unclaimed offers are dropped at the ends of a training iteration
    (`sweep`) and beyond MAX_PENDING."""

    MAX_PENDING = 16

    def __init__(self):
        self.last_output = None     # (output tensor of the latest training-mode norm, its NormSource)
        self.wanted = None
        self.ready = {}             # address of a gradient tensor -> (tensor, its version, mode, geometry, per-tile sums, NormSource)

    def produced(self, y, src):
        self.last_output = (y, src)

    def source_of(self, t):
        last = self.last_output
        if last is None or last[0].data_ptr() != t.data_ptr() or last[0].shape != t.shape:
            return None
        return last[1]

    def request(self, src):
        self.wanted = src

    def cancel(self):
        self.wanted = None

    def take_request(self):
        src, self.wanted = self.wanted, None
        return src if _precision == 0 else None

    def offer(self, gx, mode, g, sums, src):
        self.ready.pop(gx.data_ptr(), None)
        while len(self.ready) >= self.MAX_PENDING:
            self.ready.pop(next(iter(self.ready)))
        self.ready[gx.data_ptr()] = (gx, gx._version, mode, g, sums, src)

    def claim(self, gy, mean):
        """The per-tile sums left beside gradient `gy` for the norm whose batch mean is the tensor `mean` (allocated by
        that norm's forward call, so it names the call), or None."""
        ready = self.ready.pop(gy.data_ptr(), None)
        if ready is None:
            return None
        t, version, mode, g, sums, src = ready
        if gy.data_ptr() != t.data_ptr() or gy._version != version or gy.shape != t.shape:
            return None
        if src.mean.data_ptr() != mean.data_ptr():
            return None
        return mode, g, sums

    def sweep(self):
        """Drop offers nobody claimed (the boundaries of a training iteration)."""
        self.ready.clear()

    def clear(self):
        self.last_output = self.wanted = None
        self.ready.clear()


norm_bwd = _NormBackwardSums()


class _ActivationGradient:
    """Agreement between two neighbouring layers of a container, conv_a (+ ReLU / LeakyReLU in its epilogue) -> conv_b,
    whose intermediate activation no one else sees: conv_b's backward returns its input gradient ALREADY multiplied by
    the activation's slope mask (in the epilogue of its input-gradient kernel where there is such a kernel, with the
    separate mask kernel otherwise), and conv_a's backward then skips its own mask pass. The container announces the
    pairing before it runs conv_a (`announce`); conv_a's operator, if it is the fused kind, takes a token
    (`take_token`); the container then hands conv_b the activation (`private_input`), and conv_b's operator marks the
    token when it has accepted (`take_input`). conv_a's backward skips the mask only when the token is marked, so a
    consumer that never accepted leaves everything as it was."""

    def __init__(self):
        self.announced = False
        self.token = None
        self.pending = None      # (activation tensor, act, slope, token)

    def announce(self):
        self.announced, self.token = True, None

    def take_token(self):
        if not self.announced or not (_precision == 0 or _mode == "bf16") or os.environ.get("VEGANS_HIP_ACT_MASK") == "0":
            return None
        self.announced = False
        self.token = {"accepted": False}
        return self.token

    def withdraw(self):
        """End of conv_a's call: the token its operator took, if any."""
        self.announced = False
        token, self.token = self.token, None
        return token

    def private_input(self, y, act, slope, token):
        self.pending = (y, act, slope, token)

    def take_input(self, x):
        pending, self.pending = self.pending, None
        if pending is None or pending[0].data_ptr() != x.data_ptr() or pending[0].shape != x.shape:
            return None
        pending[3]["accepted"] = True
        return pending[1], pending[2]

    def cancel(self):
        self.pending = None


act_grad = _ActivationGradient()


def _conv_with_stats(mode, g, args, out, src=None):
    """Run the convolution through an entry point whose epilogue also leaves per-tile column sums: the batch statistics
    of the output (src None), or the backward sums of the norm `src` whose output gradient `out` is. True when the
    convolution ran."""
    lib = _hip.load()
    nbytes = lib.vg_conv2d_bnstats_workspace_bytes(mode, ctypes.byref(g.c))
    if not nbytes:
        return False
    if src is not None and (_rows_cols(src.x) != _rows_cols(out) or not src.x.is_contiguous(
            memory_format=CL if src.x.dim() == 4 else torch.contiguous_format)):
        return False
    stats = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
    written = ctypes.c_int(0)
    tail = (ctypes.c_void_p(stats.data_ptr()), c_size_t(nbytes), ctypes.byref(written), _stream())
    if src is None:
        name = "vg_conv2d_fwd_bnstats" if mode == 0 else "vg_conv2d_dgrad_bnstats"
        _launch_conv(mode, g, lambda: call(name, ctypes.byref(g.c), *args, _ptr(out), *tail))
        if written.value:
            norm_stats.offer(out, mode, g, stats)
    else:
        name = "vg_conv2d_fwd_bnbwd" if mode == 0 else "vg_conv2d_dgrad_bnbwd"
        _launch_conv(mode, g, lambda: call(name, ctypes.byref(g.c), *args, _ptr(out), _ptr(src.x), _ptr(src.mean),
                                           _ptr(src.invstd), _ptr(src.gamma), _ptr(src.beta), src.act,
                                           c_float(src.slope), *tail))
        if written.value:
            norm_bwd.offer(out, mode, g, stats, src)
    return True


def _stem_h(mode, g):
    """Whether contraction `mode` of an image-side layer (f32 image, bf16 feature map) has a kernel of its own."""
    return _mode == "bf16" and bool(_hip.load().vg_conv2d_stem_h_applies(mode, ctypes.byref(g.c)))


def _h_contraction(mode, g, a, b, out_dtype):
    """Whether contraction `mode` runs on bf16 tensors: bf16 storage mode, a result stored as bf16 (weight gradients are
    f32 always) and a geometry the bf16 kernels serve. Operands that arrive as f32 are converted first."""
    if _mode != "bf16" or (mode != 2 and out_dtype != BF16):
        return False
    return bool(_hip.load().vg_conv2d_h_applies(mode, ctypes.byref(g.c)))


def raw_conv_fwd(g, x, w, bias, out_dtype=None):
    x, w = to_nhwc(x), _weight_mem(w)
    _check_shape(x, g.x_shape(), "conv input")
    if w.numel() != g.w_numel():
        raise _hip.HipError("weight has %d elements, geometry expects %d" % (w.numel(), g.w_numel()))
    if out_dtype is None:
        out_dtype = _feature_dtype(g.Co)
    if _h_contraction(0, g, x, w, out_dtype):
        xh = _cast_raw(x, BF16)
        wh = plane_cache.bf16_copy(w) if w.dtype == F32 else w
        y = _new_act(g.N, g.Co, g.Ho, g.Wo, x, g.flat, BF16)
        _launch_conv(0, g, lambda: call("vg_conv2d_fwd_h", ctypes.byref(g.c), _p(xh), _p(wh), _ptr(bias), _p(y), _stream()))
        return y
    if out_dtype == BF16 and x.dtype == F32 and _stem_h(0, g):      # image stem: f32 image in, bf16 feature map out
        y = _new_act(g.N, g.Co, g.Ho, g.Wo, x, g.flat, BF16)
        _launch_conv(0, g, lambda: call("vg_conv2d_stem_fwd_h", ctypes.byref(g.c), _ptr(x), _ptr(w), _ptr(bias), _p(y),
                                        _hip.ACT_NONE, c_float(0.0), _stream()))
        return y
    x, w = _f32(x), _f32(w)
    y = _new_act(g.N, g.Co, g.Ho, g.Wo, x, g.flat)
    if out_dtype != F32:      # a shape the bf16 kernels do not serve: f32 kernel, result converted
        return _cast_raw(_raw_conv_fwd_f32(g, x, w, bias, y), out_dtype)
    return _raw_conv_fwd_f32(g, x, w, bias, y)


def _raw_conv_fwd_f32(g, x, w, bias, y):
    src = norm_bwd.take_request()
    if norm_stats.take_request():
        src = None
    elif src is None:
        src = False
    if src is not False and _conv_with_stats(0, g, (_ptr(x), _ptr(w), _ptr(bias)), y, src):
        return y
    nbytes, wsp = _conv_workspace(0, g, x.device)
    xp, wp = plane_cache.pointer(0, g, x), plane_cache.pointer(0, g, w)
    _launch_conv(0, g, lambda: call("vg_conv2d_fwd_prec", ctypes.byref(g.c), _ptr(x), _ptr(w), _ptr(bias), _ptr(y),
                                    _precision, wsp, c_size_t(nbytes), xp, wp, _stream()))
    return y


def raw_conv_dgrad(g, gy, w, bias, out_dtype=None):
    gy, w = to_nhwc(gy), _weight_mem(w)
    _check_shape(gy, g.y_shape(), "conv output-side tensor")
    if w.numel() != g.w_numel():
        raise _hip.HipError("weight has %d elements, geometry expects %d" % (w.numel(), g.w_numel()))
    if out_dtype is None:
        out_dtype = _feature_dtype(g.Ci)
    if _h_contraction(1, g, gy, w, out_dtype):
        gh = _cast_raw(gy, BF16)
        wh = plane_cache.bf16_copy(w) if w.dtype == F32 else w
        gx = _new_act(g.N, g.Ci, g.H, g.W, gy, g.flat, BF16)
        _launch_conv(1, g, lambda: call("vg_conv2d_dgrad_h", ctypes.byref(g.c), _p(gh), _p(wh), _ptr(bias), _p(gx), _stream()))
        return gx
    if out_dtype == F32 and gy.dtype == BF16 and _stem_h(1, g):     # bf16 feature-map gradient in, f32 image gradient out
        gx = _new_act(g.N, g.Ci, g.H, g.W, gy, g.flat)
        _launch_conv(1, g, lambda: call("vg_conv2d_stem_dgrad_h", ctypes.byref(g.c), _p(gy), _ptr(w), _ptr(bias), _ptr(gx),
                                        _stream()))
        return gx
    gy, w = _f32(gy), _f32(w)
    gx = _new_act(g.N, g.Ci, g.H, g.W, gy, g.flat)
    if out_dtype != F32:
        return _cast_raw(_raw_conv_dgrad_f32(g, gy, w, bias, gx), out_dtype)
    return _raw_conv_dgrad_f32(g, gy, w, bias, gx)


def _raw_conv_dgrad_f32(g, gy, w, bias, gx):
    src = norm_bwd.take_request()
    if norm_stats.take_request():
        src = None
    elif src is None:
        src = False
    if src is not False and _conv_with_stats(1, g, (_ptr(gy), _ptr(w), _ptr(bias)), gx, src):
        return gx
    nbytes, wsp = _conv_workspace(1, g, gy.device)
    gp, wp = plane_cache.pointer(1, g, gy), plane_cache.pointer(1, g, w)
    _launch_conv(1, g, lambda: call("vg_conv2d_dgrad_prec", ctypes.byref(g.c), _ptr(gy), _ptr(w), _ptr(bias), _ptr(gx),
                                    _precision, wsp, c_size_t(nbytes), gp, wp, _stream()))
    return gx


def raw_conv_wgrad(g, gy, x, w_shape):
    gy, x = to_nhwc(gy), to_nhwc(x)
    _check_shape(gy, g.y_shape(), "conv output-side tensor")
    _check_shape(x, g.x_shape(), "conv input")
    if len(w_shape) == 4:
        gw = torch.empty(w_shape, dtype=torch.float32, device=x.device, memory_format=CL)
    else:
        gw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
    if gw.numel() != g.w_numel():
        raise _hip.HipError("weight gradient shape %s does not match the geometry" % (tuple(w_shape),))
    if _h_contraction(2, g, gy, x, F32) and (gy.dtype == BF16 or x.dtype == BF16):
        gh, xh = _cast_raw(gy, BF16), _cast_raw(x, BF16)
        nbytes = _hip.load().vg_conv2d_wgrad_h_workspace_bytes(ctypes.byref(g.c))
        wsp = ctypes.c_void_p(workspace(nbytes, x.device).data_ptr()) if nbytes else None
        _launch_conv(2, g, lambda: call("vg_conv2d_wgrad_partial_h", ctypes.byref(g.c), _p(gh), _p(xh), _ptr(gw), wsp,
                                        c_size_t(nbytes), _stream()))
        if nbytes:
            call("vg_conv2d_wgrad_reduce_h", ctypes.byref(g.c), _ptr(gw), wsp, _stream())
        return gw
    if gy.dtype == BF16 and x.dtype == F32 and _stem_h(2, g):
        nbytes = _hip.load().vg_conv2d_wgrad_workspace_bytes(ctypes.byref(g.c))
        wsp = ctypes.c_void_p(workspace(nbytes, x.device).data_ptr())
        _launch_conv(2, g, lambda: call("vg_conv2d_stem_wgrad_partial_h", ctypes.byref(g.c), _p(gy), _ptr(x), wsp,
                                        c_size_t(nbytes), _stream()))
        call("vg_conv2d_wgrad_reduce", ctypes.byref(g.c), _ptr(gw), wsp, _stream())
        return gw
    gy, x = _f32(gy), _f32(x)
    nbytes, wsp = _conv_workspace(2, g, x.device)
    gp, xp = plane_cache.pointer(2, g, gy), plane_cache.pointer(2, g, x)
    _launch_conv(2, g, lambda: call("vg_conv2d_wgrad_partial_prec", ctypes.byref(g.c), _ptr(gy), _ptr(x), _ptr(gw), wsp,
                                    c_size_t(nbytes), _precision, gp, xp, _stream()))
    if _hip.load().vg_conv2d_wgrad_workspace_bytes(ctypes.byref(g.c)):
        call("vg_conv2d_wgrad_reduce", ctypes.byref(g.c), _ptr(gw), wsp, _stream())
    return gw


class PendingWgrad:
    """A weight-gradient contribution that has not been computed: the (gy, x) pair of one pass through a layer, left with
    the gradient ledger, which runs the pairs of a layer in ONE contraction when it sums that layer's contributions
    (`wgrad_of_pairs`). On bf16 tensors a weight-gradient launch of these layers is bound by its fixed costs and its
    slab traffic, not by its flops: four passes in one launch take 2.6 times one (measured, batch 256)."""
    __slots__ = ("g", "gy", "x", "w_shape", "key", "half")
    pending = True
    is_cuda = True

    def __init__(self, g, gy, x, w_shape, half):
        self.g, self.gy, self.x, self.w_shape, self.half = g, gy, x, w_shape, half      # half: bf16 tensors
        self.key = (g.N, g.H, g.W, g.Ci, g.Co, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.flat, tuple(w_shape), half)

    def detach(self):
        return self


MAX_WGRAD_PAIRS = 4


def wgrad_of_pairs(items):
    """Sum over `items` (PendingWgrad of one geometry, at most MAX_WGRAD_PAIRS) of conv_wgrad(gy, x), as one contraction
    over all pairs followed by one slab sum."""
    g, w_shape, n = items[0].g, items[0].w_shape, len(items)
    dev = items[0].x.device
    if len(w_shape) == 4:
        gw = torch.empty(w_shape, dtype=torch.float32, device=dev, memory_format=CL)
    else:
        gw = torch.empty(w_shape, dtype=torch.float32, device=dev)
    if gw.numel() != g.w_numel():
        raise _hip.HipError("weight gradient shape %s does not match the geometry" % (tuple(w_shape),))
    names = (("vg_conv2d_wgrad_h_multi_workspace_bytes", "vg_conv2d_wgrad_partial_h_multi", "vg_conv2d_wgrad_reduce_h_multi")
             if items[0].half else
             ("vg_conv2d_wgrad_multi_workspace_bytes", "vg_conv2d_wgrad_partial_multi", "vg_conv2d_wgrad_reduce_multi"))
    nbytes = getattr(_hip.load(), names[0])(ctypes.byref(g.c), n)
    wsp = ctypes.c_void_p(workspace(nbytes, dev).data_ptr()) if nbytes else None
    gys = (ctypes.c_void_p * n)(*[t.gy.data_ptr() for t in items])
    xs = (ctypes.c_void_p * n)(*[t.x.data_ptr() for t in items])
    # (for the benchmark's per-launch timer the launch counts as one over n times the rows)
    timed_as = g if n == 1 else Geom(g.N * n, g.H, g.W, g.Ci, g.Co, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.flat)
    _launch_conv(2, timed_as, lambda: call(names[1], ctypes.byref(g.c), n, gys, xs, _ptr(gw), wsp, c_size_t(nbytes), _stream()))
    if nbytes:
        call(names[2], ctypes.byref(g.c), n, _ptr(gw), wsp, _stream())
    return gw


def _weight_gradient(w, lab, gy, x, g):
    """The contribution conv_wgrad(gy, x) to d loss / d w, routed to the gradient ledger (None) or returned for autograd.
    In the fp32 mode and the bf16 data path, inside a backward pass that builds no graph, with a ledger collecting for
    `w`, the contraction is left to the ledger (PendingWgrad); VEGANS_HIP_MERGED_WGRAD=0 switches that off."""
    if (grads.accepts(w) and not torch.is_grad_enabled() and gy.is_cuda and _mode in ("bf16", "fp32")
            and os.environ.get("VEGANS_HIP_MERGED_WGRAD", "1") != "0"):
        half = _mode == "bf16" and gy.dtype == BF16 and x.dtype == BF16 and bool(
            _hip.load().vg_conv2d_h_applies(2, ctypes.byref(g.c)))
        full = _mode == "fp32" and gy.dtype == F32 and x.dtype == F32 and bool(
            _hip.load().vg_conv2d_wgrad_multi_applies(ctypes.byref(g.c)))
        if half or full:
            gy_m, x_m = to_nhwc(gy), to_nhwc(x)
            _check_shape(gy_m, g.y_shape(), "conv output-side tensor")
            _check_shape(x_m, g.x_shape(), "conv input")
            return grads.route(w, lab, PendingWgrad(g, gy_m, x_m, tuple(w.shape), half))
    return grads.route(w, lab, ConvWgradFn.apply(gy, x, g, tuple(w.shape), lab + (1,)))


def _rows_cols(t):
    """View an activation as [M][C] (C = channel / feature dim, innermost in memory)."""
    if t.dim() == 4:
        n, c, h, w = t.shape
        return n * h * w, c
    if t.dim() == 2:
        return t.shape[0], t.shape[1]
    raise _hip.HipError("activation must be 2-d or 4-d, got %d-d" % t.dim())


def raw_colsum(x):
    x = to_nhwc(x)
    m, c = _rows_cols(x)
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = _hip.load().vg_colsum_workspace_bytes(m, c)
    ws = workspace(nbytes, x.device)
    call(_A("vg_colsum", x), _p(x), _ptr(out), m, c, ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), _stream())
    return out


# ----------------------------------------------------------------------------------------------------------------
# convolution family
# ----------------------------------------------------------------------------------------------------------------
def _label(label):
    return grads.ticket() if label is None else label


def _wants_input_grad(ctx, x):
    """Whether the backward of an operator must produce the gradient of its activation input `x`. A leaf marked
    `_vg_no_grad_sink` (the gradient penalty's interpolates) is differentiated by `torch.autograd.grad` while the loss is
    built, but its gradient in the training step's own backward pass would be computed and thrown away."""
    if not ctx.needs_input_grad[0]:
        return False
    return not (grads.collecting() and getattr(x, "_vg_no_grad_sink", False))


def _input_gradient(ctx, gy, w, x, g, lab):
    """Input gradient of a convolution operator whose forward noted what its input is (ctx.norm_src, ctx.input_act)."""
    if ctx.input_act is not None:
        return ConvDgradMaskFn.apply(gy, w, x, g, ctx.input_act[0], ctx.input_act[1], lab + (0,))
    norm_bwd.request(ctx.norm_src)
    gx = ConvDgradFn.apply(gy, w, None, g, lab + (0,))
    norm_bwd.cancel()
    return gx


class ConvFwdFn(Function):
    @staticmethod
    def forward(ctx, x, w, bias, g, label=None):
        # the tensors are saved as given (not their re-laid-out copies) so second-order graphs stay connected
        ctx.g = g
        ctx.label = _label(label)
        ctx.save_for_backward(x, w, bias)
        ctx.norm_src = norm_bwd.source_of(x)     # x is the output of a training-mode norm: see _NormBackwardSums
        ctx.input_act = act_grad.take_input(x)   # x is a private ReLU / LeakyReLU output: see _ActivationGradient
        return raw_conv_fwd(g, x, w, bias)

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        g, lab = ctx.g, ctx.label
        params = not inputs_only.active
        gx = None
        if _wants_input_grad(ctx, x):
            gx = _stored_like(_input_gradient(ctx, gy, w, x, g, lab), x)
        gw = _weight_gradient(w, lab, gy, x, g) if (params and ctx.needs_input_grad[1]) else None
        gb = ColSumFn.apply(gy) if (params and bias is not None and ctx.needs_input_grad[2]) else None
        return gx, gw, grads.route(bias, lab, gb), None, None


class ConvActFn(Function):
    """y = act(conv(x, w) + bias) with ReLU / LeakyReLU applied in the convolution's epilogue (the pre-activation is
    never written). The slope mask of the backward pass is read off the sign of y, which for a non-negative slope is
    the sign of the pre-activation; from there on the backward is that of the plain convolution."""

    @staticmethod
    def forward(ctx, x, w, bias, g, act, slope):
        xm, wm = to_nhwc(x), _weight_mem(w)
        _check_shape(xm, g.x_shape(), "conv input")
        if wm.numel() != g.w_numel():
            raise _hip.HipError("weight has %d elements, geometry expects %d" % (wm.numel(), g.w_numel()))
        xm = _f32(xm)
        if _feature_dtype(g.Co) == BF16 and _stem_h(0, g):
            y = _new_act(g.N, g.Co, g.Ho, g.Wo, xm, g.flat, BF16)
            _launch_conv(0, g, lambda: call("vg_conv2d_stem_fwd_h", ctypes.byref(g.c), _ptr(xm), _ptr(wm), _ptr(bias),
                                            _p(y), act, c_float(slope), _stream()))
        else:
            y = _new_act(g.N, g.Co, g.Ho, g.Wo, xm, g.flat)
            _launch_conv(0, g, lambda: call("vg_conv2d_fwd_act", ctypes.byref(g.c), _ptr(xm), _ptr(wm), _ptr(bias),
                                            _ptr(y), act, c_float(slope), _stream()))
            y = _cast_raw(y, _feature_dtype(g.Co))
        ctx.g, ctx.act, ctx.slope = g, act, slope
        ctx.label = grads.ticket()
        ctx.norm_src = norm_bwd.source_of(x)
        ctx.input_act = act_grad.take_input(x)
        ctx.consumer = act_grad.take_token()     # the layer behind may return its input gradient already masked
        ctx.save_for_backward(x, w, bias, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias, y = ctx.saved_tensors
        g, lab = ctx.g, ctx.label
        masked = ctx.consumer is not None and ctx.consumer["accepted"]
        gpre = gy if masked else ActBwdFn.apply(gy, y, ctx.act, ctx.slope)
        params = not inputs_only.active
        gx = _stored_like(_input_gradient(ctx, gpre, w, x, g, lab), x) if _wants_input_grad(ctx, x) else None
        gw = _weight_gradient(w, lab, gpre, x, g) if (params and ctx.needs_input_grad[1]) else None
        gb = ColSumFn.apply(gpre) if (params and bias is not None and ctx.needs_input_grad[2]) else None
        return gx, gw, grads.route(bias, lab, gb), None, None, None


class ConvDgradFn(Function):
    """gx = conv_transpose(gy, w) (+bias over the Ci channels): also the forward of ConvTranspose2d."""

    @staticmethod
    def forward(ctx, gy, w, bias, g, label=None):
        ctx.g = g
        ctx.label = _label(label)
        ctx.save_for_backward(gy, w, bias)
        ctx.norm_src = norm_bwd.source_of(gy)    # as in ConvFwdFn: the transposed convolution behind a norm
        return raw_conv_dgrad(g, gy, w, bias)

    @staticmethod
    def backward(ctx, ggx):
        gy, w, bias = ctx.saved_tensors
        g, lab = ctx.g, ctx.label
        params = not inputs_only.active
        d_gy = None
        if _wants_input_grad(ctx, gy):
            norm_bwd.request(ctx.norm_src)
            d_gy = _stored_like(ConvFwdFn.apply(ggx, w, None, g, lab + (0,)), gy)
            norm_bwd.cancel()
        d_w = _weight_gradient(w, lab, gy, ggx, g) if (params and ctx.needs_input_grad[1]) else None
        d_b = ColSumFn.apply(ggx) if (params and bias is not None and ctx.needs_input_grad[2]) else None
        return d_gy, d_w, grads.route(bias, lab, d_b), None, None


class ConvDgradMaskFn(Function):
    """gx = act'(ref) * conv_transpose(gy, w): the input gradient of a convolution whose input `ref` is the output of a
    ReLU / LeakyReLU, carried on to the front of that activation. The mask is applied in the contraction kernel's
    epilogue where the geometry has such a kernel, by the separate mask kernel otherwise."""

    @staticmethod
    def forward(ctx, gy, w, ref, g, act, slope, label=None):
        ctx.g, ctx.act, ctx.slope = g, act, slope
        ctx.label = _label(label)
        ctx.save_for_backward(gy, w, ref)
        gy_m, w_m, ref_m = to_nhwc(gy), _weight_mem(w), to_nhwc(ref)
        _check_shape(gy_m, g.y_shape(), "conv output-side tensor")
        _check_shape(ref_m, g.x_shape(), "activation output")
        if w_m.numel() != g.w_numel():
            raise _hip.HipError("weight has %d elements, geometry expects %d" % (w_m.numel(), g.w_numel()))
        if _mode == "bf16" and ref_m.dtype == BF16 and _h_contraction(1, g, gy_m, w_m, BF16):
            gh = _cast_raw(gy_m, BF16)
            wh = plane_cache.bf16_copy(w_m) if w_m.dtype == F32 else w_m
            gx = _new_act(g.N, g.Ci, g.H, g.W, gy_m, g.flat, BF16)
            _launch_conv(1, g, lambda: call("vg_conv2d_dgrad_h_actmask", ctypes.byref(g.c), _p(gh), _p(wh), _p(gx), _p(ref_m),
                                            act, c_float(slope), _stream()))
            return gx
        if _precision != 0:
            gx = raw_conv_dgrad(g, gy, w, None, out_dtype=ref_m.dtype)
            masked = ctypes.c_int(0)
        else:
            gx = _new_act(g.N, g.Ci, g.H, g.W, gy_m, g.flat)
            masked = ctypes.c_int(0)
            _launch_conv(1, g, lambda: call("vg_conv2d_dgrad_actmask", ctypes.byref(g.c), _ptr(gy_m), _ptr(w_m), _ptr(gx),
                                            _ptr(ref_m), act, c_float(slope), ctypes.byref(masked), _stream()))
        if not masked.value:
            plain, gx = gx, torch.empty_like(gx)
            _same_storage(plain, ref_m)
            call(_A("vg_act_bwd", gx), _p(plain), _p(ref_m), _p(gx), c_size_t(gx.numel()), act, c_float(slope), _stream())
        return gx

    @staticmethod
    def backward(ctx, ggx):
        gy, w, ref = ctx.saved_tensors
        g, lab = ctx.g, ctx.label
        params = not inputs_only.active
        m = ActBwdFn.apply(ggx, ref, ctx.act, ctx.slope)
        d_gy = ConvFwdFn.apply(m, w, None, g, lab + (0,)) if _wants_input_grad(ctx, gy) else None
        d_w = _weight_gradient(w, lab, gy, m, g) if (params and ctx.needs_input_grad[1]) else None
        return d_gy, d_w, None, None, None, None, None


class ConvWgradFn(Function):
    @staticmethod
    def forward(ctx, gy, x, g, w_shape, label=None):
        ctx.g = g
        ctx.label = _label(label)
        ctx.save_for_backward(gy, x)
        return raw_conv_wgrad(g, gy, x, w_shape)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        g, lab = ctx.g, ctx.label
        d_gy = _stored_like(ConvFwdFn.apply(x, ggw, None, g, lab + (0,)), gy) if ctx.needs_input_grad[0] else None
        d_x = _stored_like(ConvDgradFn.apply(gy, ggw, None, g, lab + (1,)), x) if ctx.needs_input_grad[1] else None
        return d_gy, d_x, None, None, None


class ColSumFn(Function):
    """out[c] = sum over rows of x[m][c] (bias gradient)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        ctx.dtype = x.dtype
        return raw_colsum(x)

    @staticmethod
    def backward(ctx, gout):
        shape = ctx.shape
        like = gout
        if len(shape) == 4:
            out = torch.empty(shape, dtype=torch.float32, device=like.device, memory_format=CL)
            m, c = shape[0] * shape[2] * shape[3], shape[1]
        else:
            out = torch.empty(shape, dtype=torch.float32, device=like.device)
            m, c = shape
        call("vg_bcast_rows", _ptr(gout.contiguous()), _ptr(out), m, c, _stream())
        return _cast_raw(out, ctx.dtype)


def conv2d(x, w, bias, stride, padding, act=_hip.ACT_NONE, slope=0.0):
    """Convolution, optionally followed by ReLU / LeakyReLU (`act`, `slope`): fused into the convolution's epilogue
    where the library has such a kernel for the geometry, otherwise applied as a separate operator."""
    n, ci, h, wd = x.shape
    co, ci_w, kh, kw = w.shape
    if ci_w != ci:
        raise _hip.HipError("Conv2d: input has %d channels, weight expects %d" % (ci, ci_w))
    g = Geom(n, h, wd, ci, co, kh, kw, stride[0], stride[1], padding[0], padding[1])
    if act == _hip.ACT_NONE:
        return ConvFwdFn.apply(x, w, bias, g)
    if slope >= 0.0 and _hip.load().vg_conv2d_fwd_act_applies(ctypes.byref(g.c), act):
        return ConvActFn.apply(x, w, bias, g, act, float(slope))
    return activation(ConvFwdFn.apply(x, w, bias, g), act, slope)


def conv_transpose2d(x, w, bias, stride, padding, output_padding=(0, 0)):
    """ConvTranspose2d forward = input-gradient of the convolution whose weight is `w` ([Cin, Cout, kh, kw])."""
    n, cin, hi, wi = x.shape
    cin_w, cout, kh, kw = w.shape
    if cin_w != cin:
        raise _hip.HipError("ConvTranspose2d: input has %d channels, weight expects %d" % (cin, cin_w))
    h = (hi - 1) * stride[0] - 2 * padding[0] + kh + output_padding[0]
    wd = (wi - 1) * stride[1] - 2 * padding[1] + kw + output_padding[1]
    g = Geom(n, h, wd, cout, cin, kh, kw, stride[0], stride[1], padding[0], padding[1])
    if (g.Ho, g.Wo) != (hi, wi):
        raise _hip.HipError("ConvTranspose2d geometry is not invertible for this output_padding")
    return ConvDgradFn.apply(x, w, bias, g)


def linear(x, w, bias):
    if x.dim() != 2:
        raise _hip.HipError("Linear expects a 2-d input [batch, features], got shape %s" % (tuple(x.shape),))
    n, ci = x.shape
    co, ci_w = w.shape
    if ci_w != ci:
        raise _hip.HipError("Linear: input has %d features, weight expects %d" % (ci, ci_w))
    g = Geom(n, 1, 1, ci, co, 1, 1, 1, 1, 0, 0, flat=True)
    return ConvFwdFn.apply(x, w, bias, g)


# ----------------------------------------------------------------------------------------------------------------
# batch norm
# ----------------------------------------------------------------------------------------------------------------
def _bn_ws(m, c, device):
    nbytes = _hip.load().vg_bn_workspace_bytes(m, c)
    return workspace(nbytes, device), nbytes


def _act_args(act, slope):
    """(act_out, act, slope) kernel arguments of a norm's backward kernels. The slope mask of a fused activation is
    rebuilt from the norm's input and affine parameters (act_out stays null), so the backward pass does not read the
    activation's output."""
    if act == _hip.ACT_NONE:
        return None, _hip.ACT_NONE, c_float(0.0)
    return None, act, c_float(slope)


class AddFn(Function):
    """a + b for two activations of one shape (a sum the engine wants to own rather than leave to autograd's input
    buffers)."""

    @staticmethod
    def forward(ctx, a, b):
        a = _dense_like(a)
        b = _match_layout(_dense_like(b), a)
        if a.shape != b.shape:
            raise _hip.HipError("sum operands differ in shape: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        b = _cast_raw(b, a.dtype)
        out = torch.empty_like(a)
        _same_storage(a, b, out)
        call(_A("vg_add", a), _p(a), _p(b), _p(out), c_size_t(a.numel()), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class BatchNormTrainFn(Function):
    """Training-mode batch norm, optionally with a ReLU / LeakyReLU applied in the same pass (`act`). No
    pre-activation tensor exists; the backward kernels rebuild the activation's slope mask from x, the batch statistics
    and (gamma, beta) with the forward kernel's own instruction sequence, so they do not read the output either.

    Returns (y, tap). `tap` is an alias of the input that exists for the second derivative only: the first-order
    backward below is itself an operator of x (through the batch statistics), and when it is differentiated again
    (gradient penalty) its gradient with respect to x has to be added to the gradient that arrives through y. The
    backward operator therefore reads x through `tap`, both gradients arrive HERE, and the sum is formed by the engine
    (one kernel, fixed operand order) instead of by an accumulation inside autograd's scheduler."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act, slope, batches_tracked):
        x_in = x
        x = to_nhwc(x)
        m, c = _rows_cols(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        ws, nbytes = _bn_ws(m, c, x.device)
        # the running statistics are updated once per pass, in program order, also when the passes of a sub-step run
        # on different streams
        counter = ctypes.c_void_p(batches_tracked.data_ptr()) if batches_tracked is not None else None
        left = norm_stats.claim(x, m, c)
        with streams.ordered_write(running_mean):
            if left is not None:      # the convolution in front left per-tile statistics: merge them
                call("vg_bn_stats_from_conv", left[1], ctypes.byref(left[2].c), ctypes.c_void_p(left[3].data_ptr()),
                     c_float(eps), c_float(momentum), _ptr(mean), _ptr(invstd), _ptr(running_mean), _ptr(running_var),
                     counter, _stream())
            else:
                call(_A("vg_bn_stats", x), _p(x), m, c, c_float(eps), c_float(momentum), _ptr(mean), _ptr(invstd),
                     _ptr(running_mean), _ptr(running_var), counter,
                     ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), _stream())
        y = torch.empty_like(x)
        _same_storage(x, y)
        call(_A("vg_bn_apply", x), _p(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _p(y), m, c,
             act, c_float(slope), _stream())
        ctx.act, ctx.slope = act, slope
        ctx.label = grads.ticket()
        norm_bwd.produced(y, NormSource(x, mean, invstd, gamma, beta, act, slope))
        tap = x_in.view_as(x_in)
        ctx.save_for_backward(tap, gamma, beta, mean, invstd)
        ctx.set_materialize_grads(False)
        return y, tap

    @staticmethod
    def backward(ctx, gy, gtap):
        tap, gamma, beta, mean, invstd = ctx.saved_tensors
        lab = ctx.label
        gx = ggamma = gbeta = None
        if gy is not None:
            # beta only enters the piecewise-constant slope mask: detached, so no derivative is asked of it there
            mask_beta = beta.detach() if beta is not None and ctx.act != _hip.ACT_NONE else None
            gx, ggamma, gbeta = BatchNormBwdFn.apply(gy, tap, gamma, mean, invstd, mask_beta, ctx.act, ctx.slope,
                                                     lab + (0,))
        if gtap is not None:
            gx = gtap if gx is None else AddFn.apply(gx, gtap)
        return (gx, grads.route(gamma, lab, ggamma) if gamma is not None else None,
                grads.route(beta, lab, gbeta) if beta is not None else None,
                None, None, None, None, None, None, None)


class BatchNormBwdFn(Function):
    """(gx, ggamma, gbeta) of training-mode batch norm as a function of (gy, x, gamma); mean/invstd are the batch
    statistics of x (their dependence on x is part of the second derivative below). With a fused activation
    (`act`), `mask_beta` is the norm's beta: together with x, the statistics and gamma it gives the sign of the
    pre-activation, i.e. the piecewise-constant slope mask, which carries no gradient."""

    @staticmethod
    def forward(ctx, gy, x, gamma, mean, invstd, mask_beta, act, slope, label=None):
        ctx.save_for_backward(gy, x, gamma, mean, invstd, mask_beta)
        ctx.act, ctx.slope = act, slope
        ctx.label = _label(label)
        left = norm_bwd.claim(gy, mean)
        gy, x = to_nhwc(gy), to_nhwc(x)
        gy = _cast_raw(gy, x.dtype)
        m, c = _rows_cols(x)
        mask = _act_args(act, slope)
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        _same_storage(gy, x)
        if left is not None:          # the kernel that produced gy left the two sums per row tile: add the tiles
            call("vg_bn_bwd_sums_from_conv", left[0], ctypes.byref(left[1].c), ctypes.c_void_p(left[2].data_ptr()),
                 _ptr(sums), _stream())
        else:
            ws, nbytes = _bn_ws(m, c, x.device)
            call(_A("vg_bn_bwd_reduce", x), _p(gy), _p(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums),
                 ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), *mask, _ptr(gamma), _ptr(mask_beta), _stream())
        gx = torch.empty_like(x)
        _same_storage(gy, x, gx)
        call(_A("vg_bn_bwd_apply", x), _p(gy), _p(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums), _p(gx), m, c,
             *mask, _ptr(mask_beta), _stream())
        ctx.set_materialize_grads(False)
        return gx, sums[1], sums[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, u, v, w):
        gy, x, gamma, mean, invstd, mask_beta = ctx.saved_tensors
        gy_dtype = gy.dtype
        gy, x = to_nhwc(gy), to_nhwc(x)
        gy = _cast_raw(gy, x.dtype)
        m, c = _rows_cols(x)
        mask = _act_args(ctx.act, ctx.slope)
        if u is None:
            u = torch.empty_like(x)
            call(_A("vg_fill", x), _p(u), c_float(0.0), c_size_t(u.numel()), _stream())
        u = _cast_raw(to_nhwc(u), x.dtype)
        v = v.contiguous() if v is not None else None
        w = w.contiguous() if w is not None else None
        sums5 = torch.empty((5, c), dtype=torch.float32, device=x.device)
        _same_storage(gy, x, u)
        ws, nbytes = _bn_ws(m, c, x.device)
        call(_A("vg_bn_dbwd_reduce", x), _p(gy), _p(x), _p(u), _ptr(mean), _ptr(invstd), m, c, _ptr(sums5),
             ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), *mask, _ptr(gamma), _ptr(mask_beta), _stream())
        d_gy = torch.empty_like(x)
        d_x = torch.empty_like(x)
        d_gamma = torch.empty(c, dtype=torch.float32, device=x.device) if gamma is not None else None
        call(_A("vg_bn_dbwd_apply", x), _p(gy), _p(x), _p(u), _ptr(v), _ptr(w), _ptr(mean), _ptr(invstd), _ptr(gamma),
             _ptr(sums5), _p(d_gy), _p(d_x), _ptr(d_gamma), m, c, *mask, _ptr(mask_beta), _stream())
        if gamma is not None:
            d_gamma = grads.route(gamma, ctx.label, d_gamma)
        return _cast_raw(d_gy, gy_dtype), d_x, d_gamma, None, None, None, None, None, None


class BatchNormEvalFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, act, slope):
        x = to_nhwc(x)
        m, c = _rows_cols(x)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        call("vg_bn_invstd_from_var", _ptr(running_var), _ptr(invstd), c, c_float(eps), _stream())
        y = torch.empty_like(x)
        call(_A("vg_bn_apply", x), _p(x), _ptr(running_mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _p(y), m, c,
             act, c_float(slope), _stream())
        ctx.act, ctx.slope = act, slope
        ctx.label = grads.ticket()
        ctx.save_for_backward(x, gamma, beta, running_mean, invstd, y if act != _hip.ACT_NONE else None)
        ctx.has_affine = (gamma is not None, beta is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd, y = ctx.saved_tensors
        gy = _cast_raw(to_nhwc(gy), x.dtype)
        _same_storage(gy, x, y)
        m, c = _rows_cols(x)
        if y is not None:  # undo the fused activation first: g = gy * act'(y)
            g = torch.empty_like(x)
            call(_A("vg_act_bwd", x), _p(gy), _p(y), _p(g), c_size_t(x.numel()), ctx.act, c_float(ctx.slope), _stream())
            gy = g
        gx = torch.empty_like(x)
        call(_A("vg_bn_bwd_eval", x), _p(gy), _ptr(invstd), _ptr(gamma), _p(gx), m, c, _stream())
        ggamma = gbeta = None
        if ctx.has_affine[0] or ctx.has_affine[1]:
            sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
            ws, nbytes = _bn_ws(m, c, x.device)
            call(_A("vg_bn_bwd_reduce", x), _p(gy), _p(x), _ptr(mean), _ptr(invstd), m, c, _ptr(sums),
                 ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), None, _hip.ACT_NONE, c_float(0.0), None, None,
                 _stream())
            ggamma = grads.route(gamma, ctx.label, sums[1]) if ctx.has_affine[0] else None
            gbeta = grads.route(beta, ctx.label, sums[0]) if ctx.has_affine[1] else None
        return gx, ggamma, gbeta, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum, eps, act=_hip.ACT_NONE, slope=0.0,
               batches_tracked=None):
    """Batch norm with an optional fused ReLU / LeakyReLU (act, slope) behind it. `batches_tracked` (training only):
    the module's int64 device counter, incremented by the statistics kernel."""
    if act not in (_hip.ACT_NONE, _hip.ACT_RELU, _hip.ACT_LRELU):
        raise ValueError("only relu / leaky relu can be fused behind a batch norm")
    if training:
        if batches_tracked is not None and (batches_tracked.dtype != torch.int64 or not batches_tracked.is_cuda):
            raise _hip.HipError("num_batches_tracked must be an int64 tensor on the GPU")
        return BatchNormTrainFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, act, float(slope),
                                      batches_tracked)[0]
    return BatchNormEvalFn.apply(x, gamma, beta, running_mean, running_var, eps, act, float(slope))


# ----------------------------------------------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------------------------------------------
_SAVES_OUTPUT = (_hip.ACT_TANH, _hip.ACT_SIGMOID)


def _dense_like(t):
    if t.dim() == 4:
        return to_nhwc(t)
    return t if t.is_contiguous() else _plain_contiguous(t)


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x_in = x
        x = _dense_like(x)
        y = torch.empty_like(x)
        call(_A("vg_act_fwd", x), _p(x), _p(y), c_size_t(x.numel()), act, c_float(slope), _stream())
        ctx.act, ctx.slope = act, slope
        # ReLU / LeakyReLU in front of a convolution that may return its input gradient already masked
        ctx.consumer = act_grad.take_token() if act in (_hip.ACT_RELU, _hip.ACT_LRELU) and slope >= 0 else None
        ctx.save_for_backward(y if act in _SAVES_OUTPUT else x_in)
        return y

    @staticmethod
    def backward(ctx, gy):
        (ref,) = ctx.saved_tensors
        if ctx.consumer is not None and ctx.consumer["accepted"]:
            return gy, None, None
        return ActBwdFn.apply(gy, ref, ctx.act, ctx.slope), None, None


class ActBwdFn(Function):
    """gx = gy * act'(ref); ref is the forward input (relu, leaky relu) or output (tanh, sigmoid)."""

    @staticmethod
    def forward(ctx, gy, ref, act, slope):
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(gy, ref)
        ref = _dense_like(ref)
        gy = _cast_raw(_match_layout(_dense_like(gy), ref), ref.dtype)
        gx = torch.empty_like(ref)
        _same_storage(gy, ref, gx)
        call(_A("vg_act_bwd", ref), _p(gy), _p(ref), _p(gx), c_size_t(ref.numel()), act, c_float(slope), _stream())
        return gx

    @staticmethod
    def backward(ctx, gg):
        gy, ref = ctx.saved_tensors
        d_gy = ActBwdFn.apply(gg, ref, ctx.act, ctx.slope) if ctx.needs_input_grad[0] else None
        d_ref = None
        if ctx.act in _SAVES_OUTPUT and ctx.needs_input_grad[1]:
            if torch.is_grad_enabled():
                raise NotImplementedError("third-order derivatives of tanh/sigmoid are not implemented")
            ref_ = _dense_like(ref)
            gg_ = _cast_raw(_match_layout(_dense_like(gg), ref_), ref_.dtype)
            gy_ = _cast_raw(_match_layout(_dense_like(gy), ref_), ref_.dtype)
            d_ref = torch.empty_like(ref_)
            _same_storage(gg_, gy_, ref_, d_ref)
            call(_A("vg_act_dbwd_ref", ref_), _p(gg_), _p(gy_), _p(ref_), _p(d_ref), c_size_t(ref_.numel()), ctx.act,
                 _stream())
        return d_gy, d_ref, None, None


def _match_layout(t, ref):
    """Elementwise kernels index flat memory, so both operands must share one memory order."""
    if t.stride() == ref.stride():
        return t
    if ref.dim() == 4:
        return to_nhwc(t) if ref.is_contiguous(memory_format=CL) else to_nchw(t)
    return t


def activation(x, act, slope=0.0):
    return ActFn.apply(x, act, float(slope))


# ----------------------------------------------------------------------------------------------------------------
# max pooling
# ----------------------------------------------------------------------------------------------------------------
def _iptr(t):
    return ctypes.c_void_p(t.data_ptr())


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        ctx.dtype = x.dtype
        x = _f32(to_nhwc(x))
        n, c, h, w = x.shape
        g = Geom(n, h, w, c, c, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1])
        y = _new_act(n, c, g.Ho, g.Wo, x, False)
        idx = torch.empty((n, c, g.Ho, g.Wo), dtype=torch.int32, device=x.device, memory_format=CL)
        call("vg_maxpool2d_fwd", ctypes.byref(g.c), _ptr(x), _ptr(y), _iptr(idx), _stream())
        ctx.g = g
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        gx = MaxPoolBwdFn.apply(gy, idx, ctx.g)
        return (gx if gx.dtype == ctx.dtype else CastFn.apply(gx, ctx.dtype)), None, None, None


class MaxPoolBwdFn(Function):
    """gx of max pooling as a (linear) function of gy for fixed winners; its own derivative picks the winners."""

    @staticmethod
    def forward(ctx, gy, idx, g):
        gy = _f32(to_nhwc(gy))
        _check_shape(gy, (g.N, g.Ci, g.Ho, g.Wo), "pooling output-side tensor")
        gx = _new_act(g.N, g.Ci, g.H, g.W, gy, False)
        call("vg_maxpool2d_bwd", ctypes.byref(g.c), _ptr(gy), _iptr(idx), _ptr(gx), _stream())
        ctx.g = g
        ctx.save_for_backward(idx)
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx):
        (idx,) = ctx.saved_tensors
        g = ctx.g
        ggx = _f32(to_nhwc(ggx))
        out = _new_act(g.N, g.Ci, g.Ho, g.Wo, ggx, False)
        call("vg_maxpool2d_pick", ctypes.byref(g.c), _ptr(ggx), _iptr(idx), _ptr(out), _stream())
        return out, None, None


def max_pool2d(x, kernel, stride, padding):
    return MaxPoolFn.apply(x, tuple(kernel), tuple(stride), tuple(padding))


# ----------------------------------------------------------------------------------------------------------------
# losses (scalars stay on the device)
# ----------------------------------------------------------------------------------------------------------------
def _reduce_ws(n, device):
    nbytes = _hip.load().vg_reduce_workspace_bytes(c_size_t(n))
    return workspace(nbytes, device), nbytes


def _flat_dense(t):
    """`t` as one gap-free block the loss kernels can walk as flat memory (a channel / column slice is compacted)."""
    _require_gpu(t)
    if not (t.is_contiguous() or t.is_contiguous(memory_format=CL)):
        t = _plain_contiguous(t)
    return _f32(t)      # (network outputs arrive as f32; a bf16 tensor handed in directly is widened)


class MeanLossFn(Function):
    """sign * mean(pred): the Wasserstein loss with the sign fixed by the call site (ones -> -1, zeros -> +1)."""

    @staticmethod
    def forward(ctx, pred, sign):
        ctx.stored = pred.dtype
        pred = _flat_dense(pred)
        n = pred.numel()
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        ws, nbytes = _reduce_ws(n, pred.device)
        call("vg_loss_mean_fwd", _ptr(pred), c_size_t(n), c_float(sign), _ptr(out), ctypes.c_void_p(ws.data_ptr()),
             c_size_t(nbytes), _stream())
        ctx.sign = sign
        ctx.like = pred
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        pred = ctx.like
        d = torch.empty_like(pred)
        call("vg_loss_mean_bwd", _ptr(gout), c_size_t(pred.numel()), c_float(ctx.sign), _ptr(d), _stream())
        return _cast_raw(d, ctx.stored), None


class PointLossFn(Function):
    """BCE ('bce') or least squares ('mse') against a constant target or a target tensor."""

    @staticmethod
    def forward(ctx, pred, target, tconst, kind):
        ctx.stored = pred.dtype
        pred = _flat_dense(pred)
        if target is not None:
            target = _match_layout(_flat_dense(target), pred)
        n = pred.numel()
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        ws, nbytes = _reduce_ws(n, pred.device)
        call("vg_loss_%s_fwd" % kind, _ptr(pred), _ptr(target), c_float(tconst), c_size_t(n), _ptr(out),
             ctypes.c_void_p(ws.data_ptr()), c_size_t(nbytes), _stream())
        ctx.kind, ctx.tconst = kind, tconst
        ctx.save_for_backward(pred, target)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        d = torch.empty_like(pred)
        call("vg_loss_%s_bwd" % ctx.kind, _ptr(gout), _ptr(pred), _ptr(target), c_float(ctx.tconst),
             c_size_t(pred.numel()), _ptr(d), _stream())
        return _cast_raw(d, ctx.stored), None, None, None


class L1LossFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.stored = (a.dtype, b.dtype)
        a = _flat_dense(a)
        b = _match_layout(_flat_dense(b), a)
        if a.shape != b.shape:
            raise _hip.HipError("L1 loss operands differ in shape: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        n = a.numel()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws, nbytes = _reduce_ws(n, a.device)
        call("vg_loss_l1_fwd", _ptr(a), _ptr(b), c_size_t(n), _ptr(out), ctypes.c_void_p(ws.data_ptr()),
             c_size_t(nbytes), _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is None and db is None:
            return None, None
        call("vg_loss_l1_bwd", _ptr(gout), _ptr(a), _ptr(b), c_size_t(a.numel()), _ptr(da), _ptr(db), _stream())
        return _cast_raw(da, ctx.stored[0]), _cast_raw(db, ctx.stored[1])


class CombineFn(Function):
    """out = c0*a + c1*b + c2*c on device scalars (b, c optional); second output b/a is logged only."""

    @staticmethod
    def forward(ctx, a, b, c, c0, c1, c2, want_ratio):
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ratio = torch.empty((), dtype=torch.float32, device=a.device) if want_ratio else None
        call("vg_scalar_combine", _ptr(a), c_float(c0), _ptr(b), c_float(c1), _ptr(c), c_float(c2), _ptr(out),
             _ptr(ratio), _stream())
        ctx.coefs = (c0, c1, c2)
        ctx.present = (True, b is not None, c is not None)
        ctx.set_materialize_grads(False)
        if want_ratio:
            ctx.mark_non_differentiable(ratio)
            return out, ratio
        return out, None

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _gratio):
        if gout is None:
            return (None,) * 7
        grads = []
        for coef, present, need in zip(ctx.coefs, ctx.present, ctx.needs_input_grad[:3]):
            if present and need:
                g = torch.empty_like(gout)
                call("vg_scale", _ptr(gout), c_float(coef), _ptr(g), c_size_t(1), _stream())
                grads.append(g)
            else:
                grads.append(None)
        return grads[0], grads[1], grads[2], None, None, None, None


def loss_mean(pred, sign):
    return MeanLossFn.apply(pred, float(sign))


def loss_bce(pred, target):
    """`target` may be a python number (constant target) or a tensor."""
    if isinstance(target, torch.Tensor):
        return PointLossFn.apply(pred, target, 0.0, "bce")
    return PointLossFn.apply(pred, None, float(target), "bce")


def loss_mse(pred, target):
    if isinstance(target, torch.Tensor):
        return PointLossFn.apply(pred, target, 0.0, "mse")
    return PointLossFn.apply(pred, None, float(target), "mse")


def loss_l1(a, b):
    return L1LossFn.apply(a, b)


def combine(a, c0, b=None, c1=0.0, c=None, c2=0.0, want_ratio=False):
    return CombineFn.apply(a, b, c, float(c0), float(c1), float(c2), want_ratio)


# ----------------------------------------------------------------------------------------------------------------
# WGAN-GP
# ----------------------------------------------------------------------------------------------------------------
def gp_interpolate(alpha, real, fake):
    """alpha[b]*real[b] + (1-alpha[b])*fake[b]; returns a fresh leaf that requires grad (no gradient flows to
    real/fake: the reference detaches fake and real is data, WassersteinGANGP.py:69,89)."""
    real = _dense_like(real.detach())
    fake = _match_layout(_dense_like(fake.detach()), real)
    if real.shape != fake.shape:
        raise _hip.HipError("interpolation operands differ in shape: %s vs %s" % (tuple(real.shape), tuple(fake.shape)))
    b = real.shape[0]
    alpha = alpha.reshape(-1)
    if alpha.numel() != b or not alpha.is_contiguous():
        raise _hip.HipError("alpha must hold one value per sample")
    _require_gpu(alpha)
    out = torch.empty_like(real)
    call("vg_gp_interp", _ptr(alpha), _ptr(real), _ptr(fake), _ptr(out), b, c_size_t(real.numel() // b), _stream())
    return _penalty_leaf(out)


def _penalty_leaf(t):
    # differentiated by torch.autograd.grad while the loss is built; the training step's own backward pass has no use
    # for d loss / d interpolates (see _wants_input_grad)
    t.requires_grad_(True)
    t._vg_no_grad_sink = True
    return t


def gp_interpolate_pairs(alpha, real, fake):
    """The reference's interpolation for VECTOR data (2-d `real` [B, D]): its alpha has shape [B, 1, 1, 1]
    (WassersteinGANGP.py:68), which broadcasts against [B, D] to [B, 1, B, D]: every alpha is paired with every sample,
    out[i][j] = alpha[i]*real[j] + (1-alpha[i])*fake[j]. Returned as the [A*B, D] leaf the critic is applied to; the
    penalty then takes one norm per alpha over all B*D gradient elements of its row block (`gradient_penalty(rows=A)`).
    A, the number of alphas, is B on one GPU; a data-parallel rank holds the alphas of its rows only and pairs them
    with the samples of the WHOLE batch (nn.DataParallel scatters the [B, 1, B, D] critic input along its first axis)."""
    real = _dense_like(real.detach())
    fake = _match_layout(_dense_like(fake.detach()), real)
    if real.dim() != 2 or real.shape != fake.shape:
        raise _hip.HipError("pairwise interpolation expects two [B, D] tensors, got %s and %s" %
                            (tuple(real.shape), tuple(fake.shape)))
    b, d = real.shape
    alpha = alpha.reshape(-1)
    a = alpha.numel()
    if a < 1 or not alpha.is_contiguous():
        raise _hip.HipError("alpha must hold at least one contiguous value")
    _require_gpu(alpha)
    tiled_real = torch.empty((a, b * d), dtype=torch.float32, device=real.device)
    tiled_fake = torch.empty((a, b * d), dtype=torch.float32, device=real.device)
    call("vg_bcast_rows", _ptr(real), _ptr(tiled_real), a, b * d, _stream())
    call("vg_bcast_rows", _ptr(fake), _ptr(tiled_fake), a, b * d, _stream())
    out = torch.empty((a * b, d), dtype=torch.float32, device=real.device)
    call("vg_gp_interp", _ptr(alpha), _ptr(tiled_real), _ptr(tiled_fake), _ptr(out), a, c_size_t(b * d), _stream())
    return _penalty_leaf(out)


class GradPenaltyFn(Function):
    """mean_b (||g_b||_2 - 1)^2 over per-sample gradients g (any layout: the norm runs over all elements of a sample).
    `rows` overrides the number of samples b (the memory is then read as b equal consecutive blocks)."""

    @staticmethod
    def forward(ctx, g, rows=None):
        g = _dense_like(g)
        b = g.shape[0] if rows is None else rows
        if g.numel() % b:
            raise _hip.HipError("gradient of %d elements does not divide into %d samples" % (g.numel(), b))
        ctx.rows = b
        d = g.numel() // b
        norms = torch.empty(b, dtype=torch.float32, device=g.device)
        call("vg_gp_rownorm", _ptr(g), _ptr(norms), b, c_size_t(d), _stream())
        out = torch.empty((), dtype=torch.float32, device=g.device)
        call("vg_gp_penalty_fwd", _ptr(norms), b, _ptr(out), _stream())
        ctx.save_for_backward(g, norms)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g, norms = ctx.saved_tensors
        b = ctx.rows
        dg = torch.empty_like(g)
        call("vg_gp_penalty_bwd", _ptr(gout), _ptr(g), _ptr(norms), _ptr(dg), b, c_size_t(g.numel() // b), _stream())
        return dg, None


def gradient_penalty(gradients, rows=None):
    return GradPenaltyFn.apply(gradients, rows)


_unit = {}


def unit_scalar(device):
    """A 0-d device tensor holding 1.0: the seed gradient of `loss.backward` (created once per device; handed to
    backward explicitly so that no stock fill kernel runs per backward pass)."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _unit.get(key)
    if t is None:
        t = torch.empty((), dtype=torch.float32, device=device)
        call("vg_fill", _ptr(t), c_float(1.0), c_size_t(1), _stream())
        _unit[key] = t
    return t


def fill_(t, value):
    """In-place constant fill (no autograd)."""
    _require_gpu(t)
    call("vg_fill", _ptr(t), c_float(value), c_size_t(t.numel()), _stream())
    return t


# ----------------------------------------------------------------------------------------------------------------
# conditional inputs
# ----------------------------------------------------------------------------------------------------------------
class ConcatFn(Function):
    """Channel concatenation with the vector->image tiling of the reference's utils.concatenate (utils.py:52-90)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        _require_gpu(b)
        if a.shape[0] != b.shape[0]:
            raise AssertionError("Tensors to concatenate must have same dim 0. Tensor1: {}. Tensor2: {}.".format(
                a.shape[0], b.shape[0]))
        n = a.shape[0]
        if a.dim() == 2 and b.dim() == 2:
            a_, b_ = a.contiguous(), b.contiguous()
            out = torch.empty((n, a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
            call("vg_concat_channels", _ptr(a_), a.shape[1], 0, _ptr(b_), b.shape[1], 0, _ptr(out), n, 1, _stream())
            ctx.meta = (a.shape[1], b.shape[1], 0, 0, tuple(a.shape), tuple(b.shape))
            return out
        if a.dim() == 4 and b.dim() == 4:
            if a.shape[2:] != b.shape[2:]:
                raise _hip.HipError("images to concatenate differ in size: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
            a_, b_ = to_nhwc(a), to_nhwc(b)
            h, w = a.shape[2:]
            abc, bbc = 0, 0
        elif a.dim() == 4 and b.dim() == 2:
            a_, b_ = to_nhwc(a), b.contiguous()
            h, w = a.shape[2:]
            abc, bbc = 0, 1
        elif a.dim() == 2 and b.dim() == 4:
            a_, b_ = a.contiguous(), to_nhwc(b)
            h, w = b.shape[2:]
            abc, bbc = 1, 0
        else:
            raise NotImplementedError("tensor1 and tensor2 must have 2 or 4 dimensions. Given: {} and {}.".format(
                a.shape, b.shape))
        c1, c2 = a.shape[1], b.shape[1]
        out = torch.empty((n, c1 + c2, h, w), dtype=torch.float32, device=a.device, memory_format=CL)
        call("vg_concat_channels", _ptr(a_), c1, abc, _ptr(b_), c2, bbc, _ptr(out), n, h * w, _stream())
        ctx.meta = (c1, c2, abc, bbc, tuple(a.shape), tuple(b.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        c1, c2, abc, bbc, ashape, bshape = ctx.meta
        gout = _dense_like(gout)
        pixels = gout.numel() // (c1 + c2)
        grads = []
        for need, c0, cp, bc, shape in ((ctx.needs_input_grad[0], 0, c1, abc, ashape),
                                        (ctx.needs_input_grad[1], c1, c2, bbc, bshape)):
            if not need or bc:
                # a tiled vector (label / latent code) gets no gradient on the training path
                grads.append(None)
                continue
            if len(shape) == 4:
                g = torch.empty(shape, dtype=torch.float32, device=gout.device, memory_format=CL)
            else:
                g = torch.empty(shape, dtype=torch.float32, device=gout.device)
            call("vg_slice_channels", _ptr(gout), c1 + c2, c0, cp, _ptr(g), c_size_t(pixels), _stream())
            grads.append(g)
        return grads[0], grads[1]


def concatenate(a, b):
    return ConcatFn.apply(a, b)


class FlattenFn(Function):
    """[N,C,H,W] -> [N, C*H*W] in NCHW element order (what nn.Flatten computes)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        x = to_nchw(x)
        return x.view(x.shape[0], -1)

    @staticmethod
    def backward(ctx, g):
        return g.reshape(ctx.shape)


def flatten(x):
    if x.dim() <= 2:
        return x
    n, c, h, w = x.shape
    if c == 1 or h * w == 1:
        # memory order is the same in both layouts: a pure view
        out = x.reshape(n, c * h * w)
    else:
        out = FlattenFn.apply(x)
    if getattr(x, "_vg_no_grad_sink", False):
        out._vg_no_grad_sink = True
    return out
