"""Whole-step HIP graphs for the hot loop body.

One iteration of the reference's inner loop (AbstractGenerativeModel.py:321-327: for every network, `steps[name]`
times: losses -> zero_grad -> backward -> optimizer step) launches several hundred short kernels.  Enqueueing them
one by one from Python takes about as long as the GPU needs to run them, so after a couple of eager iterations the
whole body is captured once into a HIP graph (stream capture through `torch.cuda.CUDAGraph`, which also pins every
intermediate tensor to a private memory pool) and later iterations replay it with a single launch.

What a replay does, in order, on the current stream:
  1. copy the caller's batch tensors into the graph's static input buffers;
  2. run the recorded host script: host-side random draws (the gradient penalty's alphas come from numpy's global
     generator, WassersteinGANGP.py:68) staged through pinned memory into their static device buffers, and host-side
     bookkeeping such as the optimizers' step counters — in the order the eager code performs them, so the random
     streams advance exactly as they do without a graph;
  3. launch the graph.

Under data parallelism the gradient all-reduce is not captured: at that point the capture is cut, so an iteration
becomes a chain  graph | all-reduce | graph | all-reduce | graph  whose pieces share one memory pool and whose
collectives are ordinary eager calls on the persistent gradient buckets between two graph launches (no dependence
on collectives being capturable; communication then no longer overlaps backward, which costs less than the host
time the graphs save).

Everything a kernel receives by value is frozen into the graph (learning rates, clip value, penalty weight ...), so
all of it is part of the cache key: a change makes a new graph.  Capture is skipped (the body simply runs eagerly)
for optimizers without a fused kernel, for batch norms with
`momentum=None`, while the benchmark's per-kernel event timer is installed, when the model's `use_step_graph`
attribute is set to False, or when VEGANS_HIP_GRAPH=0 (VEGANS_HIP_GRAPH_DP=0: only data-parallel runs stay eager).  A body that
cannot be captured (user modules that synchronise or branch on device values) falls back to eager for good, with a
warning.
"""
import gc
import os
import warnings
import weakref

import numpy as np
import torch

from .. import _hip

WARM_EAGER_CALLS = 2     # eager iterations of a given signature before it is captured
MAX_GRAPHS = 4           # graphs (and their memory pools) kept per model
HOST_INPUT_FLOATS = 1 << 20   # static device memory per graph for host-drawn inputs (alphas): 4 MiB

_capturing = None        # the _StepGraph being recorded, else None
_graphs = weakref.WeakKeyDictionary()   # model -> {signature: _StepGraph}; kept off the model so it still pickles


def capturing():
    return _capturing is not None


def _stage(array):
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float32)).pin_memory()


def host_input(shape, draw, device):
    """A float32 device tensor filled from `draw()` (a host function returning a numpy array of `shape`).
    Eagerly: draw now, asynchronous copy from pinned memory.  While a step is being captured: hand out a static buffer
    and remember the draw; it is executed before every replay instead."""
    if _capturing is None:
        # pinned staging + asynchronous copy: a pageable copy would block the host until the stream drains
        return _stage(draw()).to(device, non_blocking=True)
    buf = _capturing.carve(tuple(shape))
    _capturing.script.append((buf, draw))
    return buf


def cut(fn):
    """Only while capturing: end the graph recorded so far, schedule host function `fn` (e.g. a collective on
    persistent buffers) to run at this point of every replay, and start recording the next graph."""
    _capturing.cut(fn)


def host_effect(fn):
    """Host-side bookkeeping that belongs to one training iteration (e.g. optimizer step counters): run now when
    eager, once per replay when the iteration is being captured."""
    if _capturing is None:
        fn()
    else:
        _capturing.script.append((None, fn))


class _StepGraph:
    def __init__(self):
        self.eager_calls = 0
        self.failed = False
        self.graph = None
        self.segments = []
        self._cur = None
        self._pool = None
        self.script = []
        self.static = None
        self.arena = None
        self.arena_used = 0
        self.losses = None
        self.grads = None
        self.scratch = None
        self.replays = 0
        self.verify = False      # data parallel with captured collectives: check the ranks against each other once

    def carve(self, shape):
        """A static buffer for a host input. It must not come from the graph's own memory pool: the pool hands a block
        that an earlier kernel of the same graph has used and released to later allocations, so a value copied in
        before the launch would be overwritten by that earlier kernel during the replay."""
        n = int(np.prod(shape)) if len(shape) else 1
        start = self.arena_used
        self.arena_used = start + (n + 63) // 64 * 64          # 256-byte granules
        if self.arena_used > self.arena.numel():
            raise _hip.HipError("host inputs of one training step exceed %d floats" % self.arena.numel())
        return self.arena[start:start + n].view(shape)

    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        # "thread_local": only this thread is policed for capture-unsafe HIP calls. Communication back-ends keep helper
        # threads (RCCL watchdog, gloo workers) that poll events at any time; under the default global mode such a
        # call from another thread fails there and invalidates the capture here.
        self._cur.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _end(self):
        cur, self._cur = self._cur, None
        cur.capture_end()
        self.segments.append(cur)

    def cut(self, fn):
        self._end()
        self.segments.append(fn)
        self._begin()

    def capture(self, model, body, inputs):
        global _capturing
        self.static = [torch.empty_like(t) for t in inputs]      # preserves the (NHWC) strides
        for s, t in zip(self.static, inputs):
            s.copy_(t)
        self.script = []
        self.segments = []     # CUDAGraph objects and host callables, in replay order
        self.arena = torch.empty(HOST_INPUT_FLOATS, dtype=torch.float32, device=inputs[0].device)
        self.arena_used = 0
        self._pool = torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream()
        gc.collect()               # dead models must release their graphs now, not during the capture
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        _capturing = self
        _hip.freeze_workspace(True)
        # The recorded kernels hold raw addresses of the shared scratch buffers: keep the buffers alive as long as this
        # graph, whatever a later (larger) eager call replaces them with.
        self.scratch = _hip.workspace_buffers()
        # No cyclic garbage collection while the stream is capturing: collecting an old model would destroy its graphs
        # and free their memory pools, which HIP forbids during capture.
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(side):
                self._begin()
                try:
                    body(*self.static)
                finally:
                    if self._cur is not None:
                        self._end()
        finally:
            if gc_was_enabled:
                gc.enable()
            _capturing = None
            _hip.freeze_workspace(False)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = self.segments[-1]
        # values only: keeping grad_fn would keep this iteration's autograd graph (and its AccumulateGrad nodes, bound
        # to the capture stream) alive into later captures
        self.losses = {k: v.detach() for k, v in model._losses.items()}
        model._losses = dict(self.losses)
        self.grads = [(p, p.grad) for net in model.neural_nets.values() for p in net.parameters()]

    def replay(self, model, inputs):
        for s, t in zip(self.static, inputs):
            s.copy_(t)
        for buf, fn in self.script:
            if buf is None:
                fn()
            else:
                buf.copy_(_stage(fn()), non_blocking=True)
        for seg in self.segments:
            if isinstance(seg, torch.cuda.CUDAGraph):
                seg.replay()
            else:
                seg()
        model._losses = dict(self.losses)
        for p, g in self.grads:
            p.grad = g
        self.replays += 1
        if self.verify:
            self.verify = False
            from . import parallel
            parallel.verify_replicas(model)


_GRAPHABLE_OPTIMIZERS = (torch.optim.RMSprop, torch.optim.SGD, torch.optim.Adam, torch.optim.AdamW)


def _frozen_scalars(model):
    """Every host value the captured kernels receive by value."""
    sig = []
    for name, opt in model.optimizers.items():
        for group in opt.param_groups:
            sig.append((name,) + tuple((k, float(v) if isinstance(v, (int, float)) else repr(v))
                                       for k, v in sorted(group.items()) if k != "params"))
    for attr in ("_clip_val", "lmbda_grad", "lambda_l1", "lambda_z", "lambda_x", "lambda_KL"):
        if hasattr(model, attr):
            sig.append((attr, float(getattr(model, attr))))
    return tuple(sig)


def _ops_precision():
    from . import ops
    return ops.matmul_precision()


def _sharding(model):
    dp = getattr(model, "_dp", None)
    if dp is None or dp.world == 1:
        return (1,)
    return (dp.world, dp.rank, dp.local_rows, dp.global_rows)


def _eligible(model):
    from . import ops
    dp = getattr(model, "_dp", None)
    if dp is not None and dp.world > 1 and (os.environ.get("VEGANS_HIP_GRAPH_DP", "1") == "0"
                                            or getattr(model, "_needs_eager_dp", False)):
        # the second reason: the iteration exchanges rows between ranks in the middle of a forward pass
        # (parallel.gather_rows: the pairwise gradient penalty on vector data)
        return False
    if os.environ.get("VEGANS_HIP_GRAPH", "1") == "0" or ops.conv_timer is not None:
        return False
    if not getattr(model, "use_step_graph", True):
        return False
    if not all(type(o) in _GRAPHABLE_OPTIMIZERS for o in model.optimizers.values()):
        return False
    for net in model.neural_nets.values():
        if not net.training:
            return False
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.momentum is None:
                return False
    return True


def run_step(model, body, inputs):
    """Run `body(*inputs)` (one hot-loop iteration on device tensors `inputs`), from a HIP graph when possible."""
    if not _eligible(model):
        return body(*inputs)
    key = (tuple((tuple(t.shape), t.stride()) for t in inputs), tuple(sorted(model.steps.items())),
           _frozen_scalars(model), _sharding(model), _ops_precision(),
           tuple(p.requires_grad for net in model.neural_nets.values() for p in net.parameters()))
    cache = _graphs.setdefault(model, {})
    entry = cache.get(key)
    if entry is None:
        if len(cache) >= MAX_GRAPHS:
            cache.pop(next(iter(cache)))
        entry = cache[key] = _StepGraph()
    if entry.failed:
        return body(*inputs)
    if entry.graph is None:
        if entry.eager_calls < WARM_EAGER_CALLS:
            entry.eager_calls += 1
            return body(*inputs)
        from . import parallel
        dp = getattr(model, "_dp", None)
        sharded = dp is not None and dp.world > 1
        error = None
        try:
            entry.capture(model, body, inputs)
        except Exception as exc:  # not capturable
            error = exc
        if sharded and parallel.collectives_captured():
            # the graph carries its own all-reduces: every rank must have captured it, or none keeps it
            if not parallel._agree(error is None):
                parallel.refuse_captured(error if error is not None else "the capture failed on another rank")
                for net in model.neural_nets.values():
                    for p in net.parameters():
                        p.grad = None
                torch.cuda.synchronize()
                entry = cache[key] = _StepGraph()
                entry.eager_calls = WARM_EAGER_CALLS
                error = None
                try:
                    entry.capture(model, body, inputs)       # second attempt, cut at the gradient exchanges
                except Exception as exc:
                    error = exc
            else:
                entry.verify = True
        if error is not None:      # stay eager for this signature
            exc = error
            entry.failed = True
            entry.graph = None
            for net in model.neural_nets.values():
                for p in net.parameters():
                    p.grad = None
            torch.cuda.synchronize()
            warnings.warn("the training step could not be captured into a HIP graph (%s: %s); running it eagerly" %
                          (type(exc).__name__, exc))
            return body(*inputs)
    entry.replay(model, inputs)


def replays(model):
    """How many training iterations of `model` were replayed from captured HIP graphs so far."""
    return sum(e.replays for e in _graphs.get(model, {}).values())


def reset(model):
    """Drop the model's captured graphs (after changing network structure by hand, for example)."""
    _graphs.pop(model, None)
