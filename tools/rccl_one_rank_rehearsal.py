"""Rehearsal of the data-parallel training loop on real RCCL with the one GPU a development box has.

A one-rank `nccl` process group is created and the engine is told it is rank 0 of 2, so every piece of the multi-GPU
path runs: row sharding, the gradient buckets fed by the gradient ledger, the known-answer probe of stream-captured
collectives, the capture of an iteration while RCCL's watchdog thread is alive with the bucket all-reduces inside the
graph (the default with RCCL) or between the launches of a chain of graphs (VEGANS_DP_CAPTURE_COLLECTIVES=0), the
agreement of the ranks about the capture, and the check of the replicas against each other after the first replay.
Runs from the same seeds (captured, chain, eager, fenced) must end with bit-identical parameters.
What it cannot show is the exchange between ranks itself.
    python tools/rccl_one_rank_rehearsal.py [batches] [nccl|gloo]
"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402
from vegans.engine import parallel  # noqa: E402

ROWS = 512       # global batch: 256 rows per rank, the benchmark's per-GPU batch


def run(batches, graphs, rows=ROWS):
    os.environ["VEGANS_HIP_GRAPH_DP"] = "1" if graphs else "0"
    if not dist.is_initialized():
        os.environ["VEGANS_HIP_GRAPH"] = "1" if graphs else "0"
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(ngpu=2 if dist.is_initialized() else None)
    X = cfg.synthetic_images(rows * batches, (3, 64, 64), seed=1)
    kw = dict(batch_size=rows, print_every=None, save_losses_every=None, enable_tensorboard=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X[:rows * 4], epochs=1, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(X, epochs=1, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    params = torch.cat([p.detach().flatten() for net in model.neural_nets.values() for p in net.parameters()])
    return 1e3 * dt / batches, params.clone()


def compare(what, a, b):
    diff = (a - b).abs()
    off = (diff > 1e-4).float().mean().item()
    print("%s: largest parameter difference %.3g, share beyond 1e-4: %.2e" % (what, diff.max().item(), off), flush=True)
    return diff.max().item(), off


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    if backend == "none":       # control: the same four runs in one process without data parallelism
        os.environ["VEGANS_HIP_GRAPH_DP"] = "1"
    elif backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=0, world_size=1)
    if dist.is_initialized():
        print("backend", dist.get_backend(), "ranks", dist.get_world_size(), flush=True)
        sharded = parallel._Sharded
        parallel._dist_on = lambda: True
        parallel._Sharded = lambda world, rank: sharded(2, 0)      # rank 0 of 2: trains on the first half of every batch
    if dist.is_initialized():
        os.environ["VEGANS_DP_CAPTURE_COLLECTIVES"] = "0"
    ms_graph, p_graph = run(batches, True)
    print("graph chain + all-reduce between launches (%s): %.2f ms per iteration (256 rows per rank)"
          % (parallel.overlap_mode(), ms_graph), flush=True)
    ms_eager, p_eager = run(batches, False)
    print("eager + all-reduce as buckets complete: %.2f ms per iteration" % ms_eager, flush=True)
    _, p_graph2 = run(batches, True)
    compare("graph vs graph again", p_graph, p_graph2)
    if dist.is_initialized():
        os.environ["VEGANS_DP_FENCE"] = "1"
        ms_fenced, p_fenced = run(batches, True)
        os.environ["VEGANS_DP_FENCE"] = "0"
        print("graph chain with device-wide waits around the all-reduces: %.2f ms per iteration" % ms_fenced, flush=True)
        compare("graph vs fenced graph", p_graph, p_fenced)
        if backend == "nccl":
            # the collectives captured into the iteration's graph (one graph, the all-reduces as branches on RCCL's stream)
            os.environ.pop("VEGANS_DP_CAPTURE_COLLECTIVES", None)        # the default with RCCL
            try:
                ms_cap, p_cap = run(batches, True)
                print("one graph with the all-reduces captured (%s; refused: %s): %.2f ms per iteration"
                      % (parallel.overlap_mode(), parallel._captured_refused, ms_cap), flush=True)
                worst_cap, _ = compare("graph vs graph with captured collectives", p_graph, p_cap)
                if parallel.overlap_mode() != "captured" or worst_cap != 0.0:
                    raise SystemExit("captured collectives were refused or disagree with the chain")
            finally:
                os.environ["VEGANS_DP_CAPTURE_COLLECTIVES"] = "0"
    worst, off = compare("graph vs eager", p_graph, p_eager)
    if dist.is_initialized():
        dist.destroy_process_group()
    # every kernel is deterministic and the gradient ledger adds contributions in program order, so the very first run
    # of the process already agrees bit for bit with later ones and with replayed graphs
    if not torch.isfinite(p_graph).all() or worst != 0.0:
        raise SystemExit("graph and eager data-parallel runs disagree")
    print("ok")


if __name__ == "__main__":
    main()
