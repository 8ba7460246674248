"""Independent pieces of one sub-step on separate HIP streams.

Inside an adversary sub-step the critic is applied to the fake batch, to the real batch and (gradient penalty) to the
interpolates: three chains that share nothing but read-only weights until their losses are combined, forward and
backward. Each chain alternates matrix-core-bound convolutions with bandwidth-bound normalisation kernels and ends every
kernel with a tail in which most compute units idle; run on separate streams, one chain's bandwidth-bound kernels and
tails overlap another chain's convolutions (measured on the DCGAN-64 critic at batch 256: two independent
forward + backward passes take 2.93 ms on two streams against 3.24 ms on one, eagerly and replayed from a HIP graph).

    with streams.region() as fork:      # remembers this point of the launch stream
        ...                              # first chain: stays on the launch stream
        with fork.branch(0): ...         # second chain: side stream 0, ordered after the remembered point only
        with fork.branch(1): ...         # third chain: side stream 1
                                         # leaving the region: the launch stream waits for the side streams it used

torch's autograd runs each backward node on the stream its forward ran on, so the backward chains fork the same way by
themselves; `join()` after `backward()` orders the launch stream behind them. Results do not depend on how the streams
interleave: no kernel uses atomics, every stream role has its own scratch buffer (`_hip.workspace`), parameter
gradients are added in label order by the gradient ledger, and writers of a shared buffer (BatchNorm running
statistics, updated once per pass in the reference's order fake, real, interpolates) are chained by events
(`ordered_write`). Under stream capture the forks and joins become branches of the captured graph.
"""
import os

import torch

N_SIDE = 2      # with the launch stream and the upload stream of fit() that makes four: the hardware queues of a process
_side = {}        # device index -> [torch.cuda.Stream] * N_SIDE
_used = {}        # device index -> set of side-stream indices used since the last join
_touched = {}     # device index -> set of side-stream indices whose forward work may still get backward work
_writers = {}     # device index -> {buffer address: (stream, event)} since the last join
_override = None  # tests / benchmarks: True or False overrides the environment


def enabled():
    if _override is not None:
        return _override
    if os.environ.get("VEGANS_HIP_STREAMS", "1") == "0":
        return False
    from . import ops
    # the operand-plane cache of the bf16_compute / split modes is shared by all passes of an iteration: one stream
    # there. In the bf16 data path only the bf16 copies of the weights are shared, and those carry an event.
    return ops.matmul_precision() in ("fp32", "bf16")


def set_enabled(flag):
    """Force branches on (True) / off (False), or back to the default rule (None). Returns the previous override."""
    global _override
    previous, _override = _override, flag
    return previous


def _device():
    return torch.cuda.current_device()


def _streams(dev):
    got = _side.get(dev)
    if got is None:
        got = _side[dev] = [torch.cuda.Stream(device=dev) for _ in range(N_SIDE)]
    return got


def role():
    """Which stream role the calling thread's current stream plays: None for the launch stream (whatever stream that
    is: torch's default stream, or the stream a graph is being captured on), else the side stream's index."""
    dev = _device()
    got = _side.get(dev)
    if not got:
        return None
    raw = torch._C._cuda_getCurrentRawStream(dev)
    for i, s in enumerate(got):
        if s.cuda_stream == raw:
            return i
    return None


class region:
    """Context manager of a fork region on the launch stream (a no-op when branches are disabled, or inside a branch)."""

    def __init__(self):
        self.event = None
        self.dev = None

    def __enter__(self):
        if enabled() and role() is None:
            self.dev = _device()
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(self.dev))
            # from here on writers of shared buffers and producers of gradient contributions leave events behind
            _touched.setdefault(self.dev, set()).add(-1)
        return self

    def branch(self, index):
        return _Branch(self, index % N_SIDE)

    def __exit__(self, exc_type, exc, tb):
        if self.event is not None:
            join()
        return False


class _Branch:
    def __init__(self, owner, index):
        self.owner, self.index, self.ctx = owner, index, None

    def __enter__(self):
        if self.owner.event is None:
            return self
        dev = self.owner.dev
        side = _streams(dev)[self.index]
        side.wait_event(self.owner.event)
        _used.setdefault(dev, set()).add(self.index)
        _touched.setdefault(dev, set()).add(self.index)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def join(after_backward=False):
    """The current (launch) stream waits for every side stream used since the last join. With `after_backward` it
    waits for every side stream any forward pass of the sub-step ran on: torch's autograd has just run the backward
    nodes of those passes on the same streams."""
    dev = _device()
    used = _touched.get(dev) if after_backward else _used.get(dev)
    if used:
        if role() is not None:
            raise RuntimeError("streams.join() called from inside a branch")
        main = torch.cuda.current_stream(dev)
        for i in sorted(used):
            if i >= 0:
                main.wait_stream(_side[dev][i])
    if _used.get(dev):
        _used[dev].clear()
    if after_backward and _touched.get(dev):
        _touched[dev].clear()
    _writers.pop(dev, None)


def forked():
    """Whether work of the current sub-step has been put on side streams (forward or, later, its backward)."""
    return bool(_touched.get(_device()))


class ordered_write:
    """Context manager around a kernel launch that updates `buffer` in place (read-modify-write): while side streams are
    in use the launch is ordered behind the previous such launch on the same buffer, whichever stream that ran on."""

    def __init__(self, buffer):
        self.buffer = buffer

    def __enter__(self):
        if self.buffer is None or not forked():
            self.buffer = None
            return self
        dev = _device()
        cur = torch.cuda.current_stream(dev)
        prev = _writers.get(dev, {}).get(self.buffer.data_ptr())
        if prev is not None and prev[0] != cur:
            cur.wait_event(prev[1])
        return self

    def __exit__(self, *exc):
        if self.buffer is not None:
            dev = _device()
            cur = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event()
            ev.record(cur)
            _writers.setdefault(dev, {})[self.buffer.data_ptr()] = (cur, ev)
        return False
