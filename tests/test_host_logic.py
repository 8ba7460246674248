"""CPU-side checks: the C-ABI library loads and exports exactly what include/vegans_hip.h declares, host-side
geometry / planning / sharding logic, and the data-parallel gradient reducer over gloo with two ranks.
No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vegans_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vegans import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), "libvegans_hip.so does not export %s" % name
    # the python binding table covers the header one to one
    assert sorted(_hip.SIGNATURES) == declared
    assert _hip.load().vg_abi_version() == 1


def test_argument_errors_come_back_as_error_codes_not_crashes():
    from vegans import _hip
    lib = _hip.load()
    g = _hip.ConvGeom(1, 4, 4, 3, 8, 3, 3, 9, 9, 1, 1, 1, 1)      # Ho/Wo inconsistent with H, k, s, p
    rc = lib.vg_conv2d_fwd(ctypes.byref(g), None, None, None, None, None)
    assert rc == -1 and b"inconsistent" in lib.vg_last_error()
    assert lib.vg_conv2d_wgrad_workspace_bytes(ctypes.byref(g)) == 0
    good = _hip.ConvGeom(256, 32, 32, 64, 128, 4, 4, 16, 16, 2, 2, 1, 1)
    rc = lib.vg_conv2d_fwd(ctypes.byref(good), None, None, None, None, None)
    assert rc == -1 and b"null" in lib.vg_last_error()
    ws = lib.vg_conv2d_wgrad_workspace_bytes(ctypes.byref(good))
    assert ws > 0 and ws % (128 * 1024 * 4) == 0                  # whole split-K slabs of Co x kh*kw*Ci floats
    assert lib.vg_bn_workspace_bytes(65536, 128) >= 5 * 128 * 4
    assert lib.vg_reduce_workspace_bytes(10) == 4


def test_operands_of_two_gib_are_refused_before_any_launch():
    """The MFMA kernels address their operands with 31-bit byte offsets; a larger operand must come back as an error
    (checked on the host, ahead of the launch: the dummy pointers are never dereferenced)."""
    from vegans import _hip
    lib = _hip.load()
    g = _hip.ConvGeom(4096, 64, 64, 64, 64, 3, 3, 64, 64, 1, 1, 1, 1)      # x and y: 2^30 elements = 4 GiB each
    dummy = ctypes.c_void_p(256)
    assert lib.vg_conv2d_fwd(ctypes.byref(g), dummy, dummy, None, dummy, None) == -1
    assert b"31-bit byte offsets" in lib.vg_last_error()
    assert lib.vg_conv2d_dgrad(ctypes.byref(g), dummy, dummy, None, dummy, None) == -1
    assert b"31-bit byte offsets" in lib.vg_last_error()
    # the timing slot is per thread and starts empty; setting and withdrawing a pair launches nothing
    assert lib.vg_time_pending() == 0
    assert lib.vg_time_next_conv(dummy, dummy) == 0 and lib.vg_time_pending() == 1
    assert lib.vg_time_next_conv(None, None) == 0 and lib.vg_time_pending() == 0


def test_ops_refuse_cpu_tensors():
    from vegans import _hip
    from vegans.engine import ops
    with pytest.raises(_hip.HipError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, (1, 1), (1, 1))
    with pytest.raises(_hip.HipError):
        ops.loss_mean(torch.zeros(4, 1), 1.0)


def test_geometry():
    from vegans.engine.ops import Geom
    g = Geom(256, 64, 64, 3, 64, 4, 4, 2, 2, 1, 1)
    assert (g.Ho, g.Wo) == (32, 32) and g.y_shape() == (256, 64, 32, 32) and g.w_numel() == 64 * 16 * 3
    lin = Geom(32, 1, 1, 784, 512, 1, 1, 1, 1, 0, 0, flat=True)
    assert lin.x_shape() == (32, 784) and lin.y_shape() == (32, 512)
    with pytest.raises(ValueError):
        Geom(1, 2, 2, 3, 4, 4, 4, 1, 1, 0, 0)


def test_periods_steps_and_input_dim():
    from vegans.models.base import _period
    from vegans.utils.utils import get_input_dim, iterate_batches, number_of_batches
    assert _period("1e", 50) == 50 and _period("0.25e", 50) == 12 and _period("0.001e", 50) == 1
    assert _period(7, 50) == 7 and _period(None, 50) is None
    assert get_input_dim([1, 4, 4], [10]) == [11, 4, 4]
    assert get_input_dim(4, [3, 8, 8]) == [7, 8, 8]
    assert get_input_dim([3, 8, 8], [3, 8, 8]) == [6, 8, 8]
    with pytest.raises(AssertionError):
        get_input_dim([3, 8, 8], [3, 4, 4])
    X = np.arange(10)
    got = [b.tolist() for b in iterate_batches(X, None, 4)]
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and number_of_batches(10, 4) == 3
    pairs = list(iterate_batches(X, X * 2, 6))
    assert pairs[1][0].tolist() == [6, 7, 8, 9] and pairs[1][1].tolist() == [12, 14, 16, 18]


def test_chunk_bounds_match_torch_chunk():
    from vegans.engine.parallel import chunk_bounds
    for n in (1, 7, 8, 31, 32, 256, 257):
        for world in (1, 2, 3, 4, 8):
            chunks = torch.arange(n).chunk(world)
            for rank in range(world):
                lo, hi = chunk_bounds(n, world, rank)
                want = chunks[rank].tolist() if rank < len(chunks) else []
                assert list(range(lo, hi)) == want, (n, world, rank)


_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "gan-pytorch_amd"))
import torch, torch.distributed as dist
from vegans.engine import grads
from vegans.engine.parallel import GradReducer, chunk_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Tanh(), torch.nn.Linear(3, 1))
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Tanh(), torch.nn.Linear(3, 1))
ref.load_state_dict(net.state_dict())
X = torch.randn(10, 6)                       # global batch of 10 rows: shards of 5 and 5; then 7 rows: 4 and 3
launches = []

def cpu_summer(items, scale):                # stands in for the HIP kernel: same contract as grads.sum_into
    launches.append(len(items))
    out = []
    for p, sources, dst in items:
        acc = sources[0].reshape(-1).clone()
        for t in sources[1:]:
            acc = acc + t.reshape(-1)
        acc = acc * scale
        if dst is None:
            dst = acc.reshape(p.shape)
        else:
            dst.copy_(acc)
        out.append(dst)
    return out

params = list(net.parameters())
red = GradReducer(params, n_buckets=2, summer=cpu_summer)
assert len(red.buckets) >= 2
for rows in (10, 7, 7):
    lo, hi = chunk_bounds(rows, world, rank)
    mid = (lo + hi) // 2
    for p in params:
        p.grad = None
    n_local = hi - lo
    # two contributions per parameter (two halves of the shard; their sum is the gradient of the local mean), handed
    # over in a different order on every rank and every pass: the ledger's label order must decide the sum
    parts = []
    for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
        g = torch.autograd.grad(net(X[a:b]).sum() / n_local, params)
        parts += [((k + 1,), p, t) for p, t in zip(params, g)]
    if (rank + rows) %% 2:
        parts.reverse()
    with red.open(n_local / float(rows)) as ledger:
        assert grads.collecting()
        for label, p, t in parts:
            assert grads.route(p, label, t) is None
        for p in params:                                     # label order, whatever the arrival order
            got = ledger.sources(p) if id(p) not in ledger.done else None
            if got is not None:
                assert len(got) == 2
    assert not grads.collecting()
    red.finish()
    for p in ref.parameters():
        p.grad = None
    ref(X[:rows]).mean().backward()          # global mean on one process
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), (rows, (p.grad - q.grad).abs().max())
        assert p.grad.shape == q.shape
        bi, off = red.where[id(p)]
        assert p.grad.data_ptr() == red.flat[bi][off:].data_ptr()      # the optimizer reads the bucket itself
# the third pass knew every parameter's contribution count from the second: parameters were summed one by one as they
# became complete (so their buckets could be reduced during the backward pass), not in one launch at the end
assert launches[-len(params):] == [1] * len(params), launches
# a contribution that arrives after its parameter was handed over must fail loudly, not be dropped
for p in params:
    p.grad = None
with red.open(1.0) as ledger:
    for label, p, t in parts:
        grads.route(p, label, t)
    try:
        grads.route(params[0], (9,), parts[0][2])
        raise SystemExit("late contribution was accepted")
    except Exception as exc:
        assert "after its parameter" in str(exc), exc
red.finish()

# ---- the guards around captured collectives, on two ranks -----------------------------------------------------------
from vegans.engine import parallel
assert parallel.overlap_mode() == "chain"              # gloo collectives run on the host: never captured
assert parallel._agree(True) is True
assert parallel._agree(rank == 0) is False             # one rank failed: every rank learns it and takes the same branch
class _Net:
    pass
class _Model:
    pass
model = _Model()
holder = _Net()
holder._vg_reducer = red
model.neural_nets = {"Generator": holder}
red.whole.fill_(1.25)
parallel.verify_replicas(model)                        # equal buckets on both ranks: passes
red.whole[3] = 1.25 + rank                             # rank 1 holds another value in one slot
try:
    parallel.verify_replicas(model)
    raise SystemExit("unequal replicas were accepted")
except RuntimeError as exc:
    assert "replicas disagree" in str(exc), exc        # raised on BOTH ranks: the check itself is a collective
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_ledger_sums_in_label_order_on_cpu():
    """The ledger's ordering logic without a GPU: contributions deposited in any order are summed in label order, a
    gradient left from an earlier pass comes first, parameters outside the ledger go back to autograd."""
    from vegans.engine import grads
    p = torch.nn.Parameter(torch.zeros(3))
    q = torch.nn.Parameter(torch.zeros(2))
    seen = []

    def summer(items, scale):
        out = []
        for par, sources, dst in items:
            seen.append([float(t[0]) for t in sources])
            acc = sources[0].clone()
            for t in sources[1:]:
                acc = acc + t
            out.append(acc * scale)
        return out

    p.grad = torch.full((3,), 7.0)
    with grads.Ledger([p], summer=summer):
        assert grads.route(p, (5, 1), torch.full((3,), 3.0)) is None
        assert grads.route(p, (5,), torch.full((3,), 2.0)) is None
        assert grads.route(p, (2, 0, 1), torch.full((3,), 1.0)) is None
        t = torch.ones(2)
        assert grads.route(q, (1,), t) is t            # not this ledger's parameter: autograd keeps it
        assert grads.route(p, (9,), None) is None
    assert seen == [[7.0, 1.0, 2.0, 3.0]]
    assert torch.equal(p.grad, torch.full((3,), 13.0))
    assert not grads.collecting()
    a, b = grads.ticket(), grads.ticket()
    assert a < b and a + (0,) < b


def test_ledger_runs_pending_weight_gradients_grouped_by_geometry(monkeypatch):
    """Contractions the operators leave with the ledger (ops.PendingWgrad): at the sum they run grouped by geometry, in
    label order, at most ops.MAX_WGRAD_PAIRS per launch, and each launch's result stands where its first pair stood."""
    from vegans.engine import grads, ops

    class Pending:
        pending = True
        is_cuda = False

        def __init__(self, key, value):
            self.key, self.value = key, value

        def detach(self):
            return self

    launches = []

    def wgrad_of_pairs(items):
        launches.append([(t.key, t.value) for t in items])
        return torch.full((2,), float(sum(t.value for t in items)))

    monkeypatch.setattr(ops, "wgrad_of_pairs", wgrad_of_pairs)
    seen = []

    def summer(items, scale):
        out = []
        for par, sources, dst in items:
            seen.append([float(t[0]) for t in sources])
            out.append(sum(sources[1:], sources[0].clone()) * scale)
        return out

    p = torch.nn.Parameter(torch.zeros(2))
    with grads.Ledger([p], summer=summer):
        assert grads.accepts(p)
        for lab, key, v in (((7,), "a", 1.0), ((3,), "a", 2.0), ((5,), "b", 4.0), ((9,), "a", 8.0), ((1,), "a", 16.0),
                            ((8,), "a", 32.0), ((2,), "a", 64.0)):
            assert grads.route(p, lab, Pending(key, v)) is None
        assert grads.route(p, (4,), torch.full((2,), 0.5)) is None
    # geometry "a" in label order 1, 2, 3, 7 | 8, 9; geometry "b" alone; results stand at labels 1, 8 and 5
    assert launches == [[("a", 16.0), ("a", 64.0), ("a", 2.0), ("a", 1.0)], [("a", 32.0), ("a", 8.0)], [("b", 4.0)]]
    assert seen == [[83.0, 0.5, 4.0, 40.0]]
    assert torch.equal(p.grad, torch.full((2,), 127.5))
    assert not grads.accepts(p)

    # a data-parallel sink counts contributions as they arrive, not the rows left after pairs were run together
    class Sink:
        weight = 1.0

        def __init__(self):
            self.seen, self.final = [], None

        def deposited(self, ledger, param, n_so_far):
            self.seen.append(n_so_far)

        def slot(self, param):
            return torch.zeros(2)

        def closed(self, ledger):
            self.final = dict(ledger.counts)

    sink = Sink()
    q = torch.nn.Parameter(torch.zeros(2))
    with grads.Ledger([q], sink=sink, summer=summer):
        for lab in ((1,), (2,), (3,)):
            grads.route(q, lab, Pending("a", 1.0))
    assert sink.seen == [1, 2, 3] and sink.final == {id(q): 3}


def test_ledger_orders_sums_and_collectives_across_streams(monkeypatch):
    """Parameters that complete on alternating streams (the critic's fake / real / penalty chains run their backward
    nodes on three streams): the sum of a parameter must wait for the contributions of THAT parameter from every stream
    and must not consume another parameter's events; the all-reduce of a bucket must wait for the sum kernels of every
    parameter in it, whichever streams they ran on. Streams and events are recorders here (no GPU): a stream is a name,
    an event remembers where it was recorded, a wait is logged as (waiting stream, event)."""
    from vegans.engine import grads, parallel
    now = {"stream": "main"}
    log = []

    class Ev:
        def __init__(self, what):
            self.stream, self.what = now["stream"], what

    monkeypatch.setattr(grads, "_needs_events", lambda grad: True)
    monkeypatch.setattr(grads, "_record_event", lambda: Ev(len(log)))
    monkeypatch.setattr(grads, "_wait_event", lambda ev: log.append(("wait", now["stream"], ev)))
    reduced = []
    monkeypatch.setattr(parallel, "_all_reduce_async", lambda flat: (log.append(("reduce", now["stream"], flat)),
                                                                     reduced.append(flat))[1])
    monkeypatch.setattr(parallel, "_cut_for_collectives", lambda: False)

    params = [torch.nn.Parameter(torch.zeros(4)) for _ in range(4)]
    sums = []

    def summer(items, scale):
        out = []
        for p, sources, dst in items:
            acc = sources[0].clone()
            for t in sources[1:]:
                acc = acc + t
            dst.copy_(acc * scale)
            sums.append((now["stream"], id(p), len(log)))
            out.append(dst)
        return out

    red = parallel.GradReducer(params, n_buckets=2, summer=summer)
    assert len(red.buckets) == 2 and [len(b) for b in red.buckets] == [2, 2]
    order = [p for b in red.buckets for p in b]           # bucket 0: order[0], order[1]; bucket 1: order[2], order[3]
    streams_of = {id(order[0]): ("side0", "main"), id(order[1]): ("side1", "side0"),
                  id(order[2]): ("main", "side1"), id(order[3]): ("side0", "side1")}

    def backward_pass():
        # two contributions per parameter, from different streams; a parameter completes on its second contribution
        for p in params:
            p.grad = None
        with red.open(1.0) as ledger:
            for k in (0, 1):
                for p in order:
                    now["stream"] = streams_of[id(p)][k]
                    grads.route(p, (k + 1,), torch.full((4,), float(k + 1)))
            now["stream"] = "main"
            ledger.joined()
        red.finish()

    backward_pass()                 # first pass: contribution counts unknown, everything summed at the end on "main"
    assert all(s == "main" for s, _, _ in sums)
    del log[:], sums[:], reduced[:]
    backward_pass()                 # second pass: early sums on the completing stream, buckets reduced as they fill
    assert [s for s, _, _ in sums] == [streams_of[id(p)][1] for p in order]
    for p in order:
        first, second = streams_of[id(p)]
        at = next(i for s, key, i in sums if key == id(p))
        waited = [ev for kind, stream, ev in log[:at] if kind == "wait" and stream == second]
        # the sum, launched on `second`, is ordered behind the parameter's contribution from `first`
        assert any(ev.stream == first for ev in waited), (first, second)
    # every bucket's collective is ordered behind the sum kernels of both its parameters
    for bi, bucket in enumerate(red.buckets):
        at = next(i for i, row in enumerate(log) if row[0] == "reduce" and row[2] is red.flat[bi])
        stream = log[at][1]
        behind = {ev.stream for kind, s, ev in log[:at] if kind == "wait" and s == stream}
        for p in bucket:
            done_on = streams_of[id(p)][1]
            assert done_on == stream or done_on in behind, (bi, done_on, stream, behind)
    assert len(reduced) == 2
    for p in order:
        assert torch.equal(p.grad, torch.full((4,), 3.0))


def test_worker_supervision_stops_siblings_on_first_failure():
    """fit()'s worker supervision (engine/spawn.py): one worker dies, the other would wait in a collective for the
    communication library's timeout; the parent must notice at once, stop the survivor and report the dead one."""
    import time
    from vegans.engine import spawn
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"])
    dying = subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"])
    t0 = time.monotonic()
    assert spawn.wait_all([sleeper, dying]) == (1, 3)
    assert time.monotonic() - t0 < 30
    assert sleeper.poll() is not None and dying.poll() == 3          # nobody left running
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert spawn.wait_all(ok) is None
    # an interrupt in the parent while it waits leaves no worker behind either
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"])
    plain_sleep = spawn._sleep

    def interrupted(_):
        raise KeyboardInterrupt

    spawn._sleep = interrupted
    try:
        with pytest.raises(KeyboardInterrupt):
            spawn.wait_all([sleeper])
    finally:
        spawn._sleep = plain_sleep
    assert sleeper.poll() is not None


def test_grad_reducer_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
        assert "ok" in out
