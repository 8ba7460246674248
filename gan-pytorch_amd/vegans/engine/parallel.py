"""Data-parallel training: one process per GPU, batch sharded by contiguous rows, gradients summed over RCCL.

The reference's multi-GPU path is `torch.nn.DataParallel` inside one process (vegans/utils/networks.py:39-41):
every forward it re-broadcasts the weights, scatters contiguous row chunks of the batch, and its backward reduces
gradients to device 0.  Here each rank keeps its own resident replica; per batch it takes the same contiguous
row chunk DataParallel's scatter would give it (`torch.chunk` sizes), computes losses as means over its rows,
and after backward the weight gradients of the network being trained are all-reduced (sum) in a few buckets.
The gradient ledger (engine/grads.py) adds each parameter's contributions in a fixed order straight into its bucket slot,
with the rank's weight rows/global_rows applied; a bucket's all-reduce is issued as soon as its last parameter is
complete, so it overlaps with the rest of backward, and the fused optimizer kernel then applies the identical update on
every rank.  BatchNorm statistics stay per rank, as they
are per replica under DataParallel.
"""
import os
import torch
import torch.distributed as dist

from . import grads, graph

BUCKETS_PER_NET = 3


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def is_main():
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0


def chunk_bounds(n_rows, world, rank):
    """Row range [lo, hi) that `torch.chunk(batch, world)` (DataParallel's scatter) gives to `rank`."""
    size = (n_rows + world - 1) // world
    lo = min(rank * size, n_rows)
    hi = min(lo + size, n_rows)
    return lo, hi


class _Single:
    world, rank = 1, 0
    global_rows = 0

    def shard(self, rows):
        return rows


class _Sharded:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.global_rows = 0
        self.local_rows = 0

    def shard(self, rows):
        self.global_rows = len(rows)
        lo, hi = chunk_bounds(len(rows), self.world, self.rank)
        if hi <= lo:
            raise RuntimeError("batch of %d rows leaves rank %d of %d without data; use a batch size that is a "
                               "multiple of the number of GPUs" % (len(rows), self.rank, self.world))
        self.local_rows = hi - lo
        return rows[lo:hi]


def needs_workers(model):
    """ngpu > 1 asked of a process that is not part of a process group: `fit` starts one worker process per GPU itself
    (engine/spawn.py)."""
    ngpu = getattr(model, "ngpu", 0) or 0
    return ngpu > 1 and not (dist.is_available() and dist.is_initialized())


def context(model):
    """Decide how `model.fit` shards its batches. ngpu > 1 needs an initialised process group (one rank per GPU)."""
    ngpu = getattr(model, "ngpu", 0) or 0
    if ngpu > 1:
        if not _dist_on():
            raise RuntimeError(
                "ngpu=%d, but this process belongs to a process group of a single rank: launch one rank per GPU "
                "(`python -m torch.distributed.run --nproc-per-node %d ...`), or leave the process group "
                "uninitialised and let fit() start the workers itself." % (ngpu, ngpu))
        ctx = _Sharded(dist.get_world_size(), dist.get_rank())
        model._dp = ctx
        probe_captured_collectives()
        for name, net in model.neural_nets.items():
            if not hasattr(net, "_vg_reducer"):
                net._vg_reducer = GradReducer(list(net.parameters()))
        return ctx
    model._dp = _Single()
    return model._dp


class GradReducer:
    """Bucketed gradient all-reduce for one network, fed by the gradient ledger (engine/grads.py).

    `open(weight)` starts a backward pass and returns the ledger that collects the network's parameter gradients. The
    ledger writes weight * (sum of a parameter's contributions, in label order) into the parameter's slot of its bucket;
    eagerly it does so as soon as the parameter has received as many contributions as in the previous backward pass, and
    a bucket whose parameters are all complete is all-reduced at once, while the rest of the backward pass is still
    running. `finish()` waits for the collectives (or, while an iteration is being captured into HIP graphs, cuts the
    capture and schedules them between two graph launches). `summer` replaces the HIP summation kernel in host-logic
    tests."""

    def __init__(self, params, n_buckets=BUCKETS_PER_NET, summer=None):
        self._summer = summer
        params = [p for p in params if p.requires_grad]
        self.params = params
        # gradients become available roughly in reverse parameter order during backward
        order = list(reversed(params))
        total = sum(p.numel() for p in order)
        cap = max((total + n_buckets - 1) // n_buckets, 1)
        self.buckets = []
        cur, cur_n = [], 0
        for p in order:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        # the buckets are consecutive 256-byte aligned pieces of one buffer: eager iterations all-reduce them one by
        # one as they fill up (overlapping backward), replayed iterations all-reduce the whole buffer in one collective
        sizes = [sum(p.numel() for p in b) for b in self.buckets]
        starts, end = [], 0
        for n in sizes:
            starts.append(end)
            end += (n + 63) // 64 * 64
        self.whole = torch.zeros(max(end, 1), dtype=torch.float32, device=order[0].device if order else "cpu")
        self.flat = [self.whole[s0:s0 + n] for s0, n in zip(starts, sizes)]
        self.where = {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self.where[id(p)] = (bi, off)
                off += p.numel()
        self.pending = [0] * len(self.buckets)
        self.touched = [False] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.next = 0           # first bucket not yet handed to the collective
        self.weight = 1.0
        self.counts = None      # id(param) -> contributions it received in the previous backward pass
        self.armed = False

    def open(self, local_weight):
        """Start collecting the next backward pass; `local_weight` scales this rank's gradients before the sum."""
        self.pending = [len(b) for b in self.buckets]
        self.touched = [False] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.next = 0
        self.weight = float(local_weight)
        self.armed = True
        return grads.Ledger(self.params, sink=self, summer=self._summer)

    # -- ledger sink protocol ------------------------------------------------------------------------------------
    def slot(self, p):
        bi, off = self.where[id(p)]
        return self.flat[bi][off:off + p.numel()]

    def deposited(self, ledger, p, n_so_far):
        if _cut_for_collectives() or self.counts is None or self.counts.get(id(p)) != n_so_far:
            return
        ledger.flush([p])
        self._completed(p, ledger)

    def _completed(self, p, ledger):
        bi, _ = self.where[id(p)]
        self.touched[bi] = True
        self.pending[bi] -= 1
        # Buckets are handed to the collective strictly in bucket order: the order in which parameters complete depends
        # on how autograd schedules the backward pass, which may differ between ranks, and ranks that issue their
        # collectives in different orders pair up the wrong buffers.
        while self.next < len(self.buckets) and self.pending[self.next] == 0 and not _cut_for_collectives():
            self.launched[self.next] = True
            # the slots of this bucket were written by sum kernels on whatever stream each parameter completed on; the
            # collective is issued from the stream of the LAST one and must be ordered behind all of them
            ledger.wait_summed(self.buckets[self.next])
            self.works.append(_all_reduce_async(self.flat[self.next]))
            self.next += 1

    def closed(self, ledger):
        """The backward pass is over and the ledger has written every parameter that received a gradient."""
        seen = dict(ledger.counts)
        for p in self.params:
            if id(p) in seen:
                self.touched[self.where[id(p)][0]] = True
        self.counts = seen
        if not _cut_for_collectives():
            # buckets that got gradients but were not launched from `deposited` (first pass, changed graph, parameters
            # without a gradient this time)
            for bi in range(len(self.buckets)):
                if self.touched[bi] and not self.launched[bi]:
                    self.launched[bi] = True
                    self.works.append(_all_reduce_async(self.flat[bi]))

    def finish(self):
        """Wait for every all-reduce of the pass (eager), or schedule them between two graph launches (capturing)."""
        if not self.armed:
            return
        self.armed = False
        if _cut_for_collectives():
            flats = [self.flat[bi] for bi in range(len(self.buckets)) if self.touched[bi]]
            if len(flats) == len(self.flat) and len(flats) > 1:
                flats = [self.whole]          # one collective instead of one per bucket (padding between them is zero)
            if flats:
                graph.cut(lambda: _all_reduce_now(flats))
            return
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []


def _gloo():
    return dist.get_backend() == "gloo"


_captured_refused = None     # why this process does not capture its collectives (None: it does, or has not tried yet)
_probed = False


def collectives_captured():
    """Whether a captured iteration carries its gradient all-reduces INSIDE the HIP graph. Captured, a bucket's
    all-reduce is a branch of the graph on RCCL's stream that starts as soon as the bucket's last gradient exists and
    runs beside the rest of the backward pass, in replayed iterations as in eager ones. That needs a communication
    library that can be stream-captured: RCCL, the default; not gloo, whose collectives run on the host. Otherwise the
    capture is cut at every gradient exchange and the collectives run between two graph launches as blocking
    stream-ordered calls (a chain graph | all-reduce | graph ...: no overlap). VEGANS_DP_CAPTURE_COLLECTIVES=0 forces the
    chain; a failed first contact (probe_captured_collectives, or a capture that raises) selects it for the rest of the
    process, on all ranks together."""
    if _gloo() or os.environ.get("VEGANS_DP_CAPTURE_COLLECTIVES", "1") == "0":
        return False
    return _captured_refused is None


def refuse_captured(reason):
    global _captured_refused
    if _captured_refused is None:
        _captured_refused = str(reason)
        import warnings
        warnings.warn("gradient all-reduces will not be captured into the HIP graph (%s): replayed iterations run as a "
                      "chain graph | all-reduce | graph without overlap" % reason)


def overlap_mode(model=None):
    """How a steady-state data-parallel iteration exchanges gradients: "captured" (all-reduces inside the replayed
    graph, overlapping backward), "chain" (between graph launches, no overlap), "eager" (no graph: bucket all-reduces
    start during backward from the host)."""
    if os.environ.get("VEGANS_HIP_GRAPH", "1") == "0" or os.environ.get("VEGANS_HIP_GRAPH_DP", "1") == "0":
        return "eager"
    if model is not None and (getattr(model, "_needs_eager_dp", False) or not getattr(model, "use_step_graph", True)):
        return "eager"
    return "captured" if collectives_captured() else "chain"


def _agree(ok):
    """True when `ok` holds on every rank (an eager all-reduce: every rank takes the same branch afterwards)."""
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def probe_captured_collectives():
    """First contact with stream-captured collectives, once per process, before any training graph depends on them: a
    HIP graph holding one all-reduce of known values (rank + 1 in every element) is captured and replayed three times
    on fresh inputs, and the results are checked against world * (world + 1) / 2. Any exception or wrong value, on any
    rank, makes ALL ranks fall back to the chain."""
    global _probed
    if _probed or not _dist_on() or not collectives_captured():
        return
    _probed = True
    world, rank = dist.get_world_size(), dist.get_rank()
    ok, why, g, buf = True, "", None, None
    try:
        buf = torch.zeros(4096, device="cuda")
        side = torch.cuda.Stream()
        dist.all_reduce(buf)                                  # communicator set-up happens outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="thread_local")
            try:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
                work.wait()
            finally:
                g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
    except Exception as exc:   # noqa: BLE001 - whatever the communication library or the capture raises
        ok, why = False, "%s: %s" % (type(exc).__name__, exc)
    # Nothing has been exchanged yet (a capture executes nothing). Only if EVERY rank holds a graph are the graphs
    # replayed: a rank that replayed alone would wait in its all-reduce for ranks that never enter theirs.
    if not _agree(ok):
        refuse_captured(why or "another rank could not capture an all-reduce")
        return
    try:
        for k in range(3):
            buf.fill_(float(rank + 1 + k))
            g.replay()
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0 + k * world
            if ok and not bool((buf == want).all()):
                ok, why = False, "a replayed all-reduce returned %r instead of %r" % (float(buf[0]), want)
    except Exception as exc:   # noqa: BLE001
        ok, why = False, "%s: %s" % (type(exc).__name__, exc)
    if not _agree(ok):
        refuse_captured(why or "another rank's replayed all-reduce returned wrong values")


def _cut_for_collectives():
    """While an iteration is being captured: whether the capture is cut at every gradient exchange (see
    collectives_captured)."""
    if not graph.capturing():
        return False
    return not collectives_captured()


def verify_replicas(model):
    """After the first replay of a graph that carries its own collectives: the reduced gradient buckets must hold the
    same bits on every rank (an all-reduce hands every rank the same sum). One MAX and one MIN all-reduce per network,
    once per captured graph; a difference means the captured collectives did not exchange what the eager ones do and is
    raised, not papered over."""
    if not _dist_on():
        return
    for name, net in model.neural_nets.items():
        red = getattr(net, "_vg_reducer", None)
        if red is None:
            continue
        hi, lo = red.whole.clone(), red.whole.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        if not torch.equal(hi, lo):
            raise RuntimeError("data-parallel replicas disagree on the reduced gradients of %s after the first replayed "
                               "iteration with captured collectives; set VEGANS_DP_CAPTURE_COLLECTIVES=0" % name)


def _all_reduce_async(flat):
    """Start the sum of one bucket over all ranks; returns a work handle, or None when it already completed.

    gloo (CPU rehearsals and the two-ranks-on-one-GPU test; production is RCCL) is only ever handed HOST tensors: the
    bucket is copied to the host (which waits for the stream to reach this point), summed there, and copied back in
    stream order. No gloo helper thread touches the GPU that way, so nothing gloo does can interleave with graph
    launches on the device, and no device-wide wait is needed around the exchange."""
    if _gloo() and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def _fenced():
    """VEGANS_DP_FENCE=1 brackets the all-reduces between two graph launches by device-wide host waits (a diagnostic: it
    drains the GPU three times per iteration)."""
    return os.environ.get("VEGANS_DP_FENCE", "0") == "1"


def _all_reduce_now(flats):
    """All-reduce gradient buckets between two graph launches of a replayed iteration, in stream order on the current
    stream (blocking-collective semantics: the next graph launch is ordered behind them)."""
    fenced = _fenced()
    if fenced:
        torch.cuda.synchronize()
    for f in flats:
        if _gloo() and f.is_cuda:
            _all_reduce_async(f)
        else:
            dist.all_reduce(f, op=dist.ReduceOp.SUM)
    if fenced:
        torch.cuda.synchronize()


def reduce_logged(model, losses):
    """Logged losses under data parallelism: every rank holds the mean over ITS rows; the reported value is the mean over
    the global batch, i.e. the row-weighted sum over ranks - one small all-reduce per logging event. `losses` is a dict
    name -> 0-d device tensor; returns name -> python float. `RealFakeRatio` is re-formed from the reduced terms."""
    names = list(losses)
    stacked = torch.stack([losses[n].detach().reshape(()).float() for n in names])
    ctx = getattr(model, "_dp", None)
    if ctx is not None and ctx.world > 1 and _dist_on():
        stacked = stacked * (ctx.local_rows / float(ctx.global_rows))
        if _gloo() and stacked.is_cuda:
            host = stacked.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            stacked = host
        else:
            dist.all_reduce(stacked, op=dist.ReduceOp.SUM)
    values = dict(zip(names, stacked.tolist()))
    if "RealFakeRatio" in values and ctx is not None and ctx.world > 1:
        fake = values.get("Adversariat_fake")
        values["RealFakeRatio"] = values["Adversariat_real"] / fake if fake else float("nan")
    return values


def gather_rows(model, local):
    """The row shards of all ranks of a [rows, D] device tensor, concatenated in rank order: the whole batch, on every
    rank. No gradient flows through (the callers gather data and detached samples). Shards may differ in size
    (`chunk_bounds`); they travel padded to the largest. The exchange is an eager collective in the middle of a forward
    pass, so a model that needs it stays out of HIP-graph capture under data parallelism (`_needs_eager_dp`)."""
    ctx = getattr(model, "_dp", None)
    if ctx is None or ctx.world == 1 or not _dist_on():
        return local
    model._needs_eager_dp = True
    if graph.capturing():
        raise RuntimeError("a row gather was reached while the iteration was being captured")
    local = local.detach()
    size = (ctx.global_rows + ctx.world - 1) // ctx.world
    if local.shape[0] != ctx.local_rows:
        raise RuntimeError("gather_rows: %d rows, but this rank's shard has %d" % (local.shape[0], ctx.local_rows))
    padded = local.new_zeros((size,) + tuple(local.shape[1:]))
    padded[:local.shape[0]].copy_(local)
    if _gloo() and padded.is_cuda:
        host = padded.cpu()
        pieces = [torch.empty_like(host) for _ in range(ctx.world)]
        dist.all_gather(pieces, host)
        pieces = [t.to(local.device) for t in pieces]
    else:
        pieces = [torch.empty_like(padded) for _ in range(ctx.world)]
        dist.all_gather(pieces, padded)
    rows = []
    for rank, t in enumerate(pieces):
        lo, hi = chunk_bounds(ctx.global_rows, ctx.world, rank)
        rows.append(t[:hi - lo])
    return torch.cat(rows, dim=0)


def open_ledger(model, who):
    """The gradient ledger of one backward pass of network `who` (None: every network): collects the parameter
    gradients the operators produce and adds them in a fixed order; under data parallelism it writes them into the
    all-reduce buckets of the network's reducer."""
    names = [who] if who is not None else list(model.neural_nets)
    ctx = getattr(model, "_dp", None)
    if ctx is None or ctx.world == 1:
        return grads.Ledger([p for name in names for p in model.neural_nets[name].parameters()])
    if len(names) != 1:
        raise NotImplementedError("under data parallelism the networks are stepped one at a time (who=None is "
                                  "not supported with ngpu > 1)")
    red = getattr(model.neural_nets[names[0]], "_vg_reducer", None)
    if red is None:
        raise RuntimeError("network %s has no gradient reducer: parallel.context(model) was not called" % names[0])
    return red.open(ctx.local_rows / float(ctx.global_rows))


def reduce_gradients(model, name):
    """Wait for the all-reduces of network `name`; returns the factor the optimizer applies to the gradients
    (1.0: the per-rank weights rows/global_rows were applied before the sum)."""
    ctx = getattr(model, "_dp", None)
    if ctx is None or ctx.world == 1:
        return 1.0
    red = getattr(model.neural_nets[name], "_vg_reducer", None)
    if red is not None:
        red.finish()
    return 1.0
