"""Engine-owned accumulation of parameter gradients.

In the reference a parameter that is used by several forward passes of one sub-step (the critic of
WassersteinGANGP.py:88-107 sees the fake batch, the real batch and the interpolates, and the penalty reaches every
weight twice more through the double backward) gets its gradient summed by torch's `AccumulateGrad` nodes, one
`add` kernel per contribution, in whatever order the autograd engine happens to run the producers.  Floating-point
addition of more than two terms depends on that order, and the order is not a property of the program: nodes built
during a `create_graph` backward take their sequence numbers from the autograd worker thread's private counter.

Here the operators hand every parameter-gradient contribution to a `Ledger` instead of returning it to autograd.
Each contribution carries a label derived from program order on the calling thread (a ticket drawn when the forward
operator ran, extended by a fixed child index for operators created inside a backward pass), and when the backward
pass has finished the ledger adds the contributions of every parameter in label order with one multi-tensor kernel
(`vg_grad_sum_multi`) and binds the result to `param.grad`.  The sum is therefore a function of the program alone:
bitwise reproducible between runs, between eager and graph-replayed iterations, and between ranks.

Under data parallelism the ledger writes the (row-weighted) sum straight into the parameter's slot of its all-reduce
bucket and tells the bucket owner (`sink`) as soon as a parameter is complete, so the collective of a bucket can start
while the rest of the backward pass is still running.
"""
import itertools
from ctypes import c_float, c_int, c_size_t, c_void_p

import torch

from .. import _hip
from .._hip import _stream, call
from . import streams

MAX_SOURCES = 8          # contributions one kernel launch adds per tensor (more are chained)

_tickets = itertools.count(1)
_active = None           # the Ledger that is collecting, else None


def ticket():
    """Label of a forward operator call: program order on the thread that builds the graph."""
    return (next(_tickets),)


def collecting():
    return _active is not None


# Stream-order primitives of the ledger (host-logic tests replace them with recorders that need no GPU).
def _needs_events(grad):
    """Whether contributions of the current backward pass may be produced on several streams (engine/streams.py)."""
    return grad.is_cuda and streams.forked()


def _record_event():
    ev = torch.cuda.Event()
    ev.record()
    return ev


def _wait_event(ev):
    torch.cuda.current_stream().wait_event(ev)


def accepts(param):
    """Whether a ledger is collecting for `param` (an operator may then leave it work instead of a tensor)."""
    return _active is not None and id(param) in _active.index


def route(param, label, grad):
    """Give `grad` (a contribution to d loss / d param, or None) to the collecting ledger and return None, or return it
    unchanged when no ledger is collecting for `param` (plain autograd use of the operators: autograd then
    accumulates)."""
    led = _active
    if grad is None or led is None or id(param) not in led.index:
        return grad
    led.deposit(param, label, grad)
    return None


def _same_order(t, p):
    """`t` in the memory order of parameter `p` (the kernels walk raw memory)."""
    if t.stride() == p.stride() or t.numel() <= 1:
        return t
    if p.dim() == 4:
        from . import ops
        t = ops.to_nhwc(t) if p.is_contiguous(memory_format=torch.channels_last) else ops.to_nchw(t)
    elif not t.is_contiguous():
        from . import ops
        t = ops._plain_contiguous(t)
    if t.stride() != p.stride():
        raise _hip.HipError("gradient strides %s differ from parameter strides %s" % (t.stride(), p.stride()))
    return t


def sum_into(items, scale=1.0):
    """items: [(param, sources, out)] — write scale * (sources[0] + sources[1] + ...) (added left to right) into `out`
    (None: in place into sources[0]) for every item with ONE kernel launch per MAX_SOURCES sources; returns the
    tensors holding the sums."""
    results = []
    work = []
    for p, sources, out in items:
        sources = [_same_order(s, p) for s in sources]
        if out is None:
            if len(sources) == 1 and scale == 1.0:
                results.append(sources[0])
                continue
            out = sources[0]
        results.append(out)
        work.append((sources, out))
    while work:
        count = len(work)
        outs = (c_void_p * count)()
        srcs = (c_void_p * (count * MAX_SOURCES))()
        nsrc = (c_int * count)()
        sizes = (c_size_t * count)()
        scales = (c_float * count)()
        rest = []
        for i, (sources, out) in enumerate(work):
            head, tail = sources[:MAX_SOURCES], sources[MAX_SOURCES:]
            for t in head + [out]:
                if not t.is_cuda or t.dtype != torch.float32 or t.numel() != out.numel():
                    raise _hip.HipError("gradient contribution %s %s does not match its parameter" %
                                        (t.dtype, tuple(t.shape)))
            outs[i] = out.data_ptr()
            for k, t in enumerate(head):
                srcs[i * MAX_SOURCES + k] = t.data_ptr()
            nsrc[i] = len(head)
            sizes[i] = out.numel()
            # the scale belongs to the finished sum: a chain of launches applies it in its last one
            scales[i] = 1.0 if tail else scale
            if tail:
                rest.append(([out] + tail, out))
        call("vg_grad_sum_multi", count, outs, srcs, nsrc, sizes, scales, _stream())
        work = rest
    return results


class Ledger:
    """Collects the gradient contributions of `params` during one backward pass (context manager)."""

    def __init__(self, params, sink=None, summer=None):
        self.params = [p for p in params if p.requires_grad]
        self.index = {id(p): p for p in self.params}
        self.entries = {}
        self.counts = {}     # id(param) -> contributions received (a gradient left from an earlier pass included); the
                             # rows of `entries` are fewer once pending contractions have been run together
        self.sink = sink
        self.summer = summer if summer is not None else sum_into    # host-logic tests pass a CPU stand-in
        self.done = set()
        # Stream order while the passes of a sub-step run on several streams: a contribution leaves an event under ITS
        # parameter when it is deposited, the sum of a parameter waits for that parameter's events only (and consumes
        # them: nobody else needs them), and an early sum (data-parallel sink) leaves an event of its own under the
        # parameter, which the bucket's all-reduce waits for, whichever stream each of the three ran on.
        self.events = {}     # id(param) -> [event per contribution produced while side streams are in use]
        self.summed = {}     # id(param) -> event behind the kernel that wrote its sum into the bucket slot

    def __enter__(self):
        global _active
        if _active is not None:
            raise _hip.HipError("a gradient ledger is already collecting (nested backward passes of the training step)")
        for p in self.params:
            if p.grad is not None:
                # a gradient left from an earlier backward pass is the first term, as it is for torch's accumulation
                self.entries[id(p)] = [((), p.grad)]
                self.counts[id(p)] = 1
        _active = self
        return self

    def __exit__(self, exc_type, exc, tb):
        global _active
        _active = None
        if exc_type is None:
            self.close()
        return False

    def deposit(self, param, label, grad):
        key = id(param)
        if key in self.done:
            raise _hip.HipError("a gradient contribution arrived after its parameter's bucket had been handed to the "
                                "all-reduce: the model's backward graph changed between two iterations")
        self.entries.setdefault(key, []).append((label, grad.detach()))
        self.counts[key] = self.counts.get(key, 0) + 1
        if _needs_events(grad):
            # produced on whatever stream this backward node runs on; the summation waits for it (flush)
            self.events.setdefault(key, []).append(_record_event())
        if self.sink is not None:
            self.sink.deposited(self, param, self.counts[key])

    def materialize(self, param):
        """Run the contractions the operators left behind for `param` (ops.PendingWgrad): the pending pairs of one
        geometry in label order, up to ops.MAX_WGRAD_PAIRS per launch; each launch's result takes the place of its
        first pair."""
        rows = self.entries.get(id(param))
        if not rows or not any(getattr(t, "pending", False) for _, t in rows):
            return
        from . import ops
        rows = sorted(rows, key=lambda r: r[0])
        done = [r for r in rows if not getattr(r[1], "pending", False)]
        groups = {}
        for lab, t in rows:
            if getattr(t, "pending", False):
                groups.setdefault(t.key, []).append((lab, t))
        for members in groups.values():
            for i in range(0, len(members), ops.MAX_WGRAD_PAIRS):
                part = members[i:i + ops.MAX_WGRAD_PAIRS]
                done.append((part[0][0], ops.wgrad_of_pairs([t for _, t in part])))
        self.entries[id(param)] = done

    def sources(self, param):
        rows = sorted(self.entries.get(id(param), ()), key=lambda r: r[0])
        return [t for _, t in rows]

    def flush(self, params):
        """Sum what has arrived for `params` now (the data-parallel sink calls this for complete parameters)."""
        todo = [p for p in params if id(p) not in self.done and id(p) in self.entries]
        if not todo:
            return
        waited = False
        for p in todo:
            for ev in self.events.pop(id(p), ()):
                _wait_event(ev)
                waited = True
        for p in todo:
            self.materialize(p)
        if self.sink is None:
            outs, scale = [None] * len(todo), 1.0
        else:
            outs, scale = [self.sink.slot(p) for p in todo], self.sink.weight
        sums = self.summer([(p, self.sources(p), o) for p, o in zip(todo, outs)], scale)
        for p, s in zip(todo, sums):
            p.grad = s if s.shape == p.shape and s.stride() == p.stride() else s.as_strided(p.shape, p.stride())
            self.done.add(id(p))
        if self.sink is not None and (waited or (sums and _needs_events(sums[0]))):
            ev = _record_event()
            for p in todo:
                self.summed[id(p)] = ev

    def wait_summed(self, params):
        """Order the current stream behind the kernels that wrote the sums of `params` (a bucket about to be handed to
        its collective: its slots may have been written from several streams)."""
        seen = set()
        for p in params:
            ev = self.summed.get(id(p))
            if ev is not None and id(ev) not in seen:
                seen.add(id(ev))
                _wait_event(ev)

    def joined(self):
        """The launch stream has been ordered behind every side stream of the sub-step (streams.join): whatever is
        summed or exchanged on it from here on needs no event."""
        self.events = {}
        self.summed = {}

    def close(self):
        self.flush(self.params)
        if self.sink is not None:
            self.sink.closed(self)
