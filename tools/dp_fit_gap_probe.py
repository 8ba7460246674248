"""Where the host spends an iteration of `fit()` under data parallelism with the gloo rehearsal transport (two ranks on
one GPU): wall-clock time of the host inside each piece of the loop (batch staging + upload, latent draw, the training
step with its host-side gradient exchange), per rank, over the replayed iterations.

    VEGANS_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29577 tools/dp_fit_gap_probe.py

The gloo exchange copies every bucket to the host and waits for the stream (engine/parallel.py), so the host cannot
run ahead of the GPU and whatever it does between two training steps leaves the GPU idle; with RCCL the exchange is
stream-ordered and the host stays ahead as in the single-GPU run."""
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


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(os.environ.get("VEGANS_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(ngpu=world)
    spent = {}

    def timed(name, fn):
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            spent.setdefault(name, []).append(time.perf_counter() - t0)
            return out
        return wrapper

    model._to_device = timed("stage + upload (host)", model._to_device)
    model._draw_latent = timed("latent draw (host)", model._draw_latent)
    model.train_on_batch = timed("train_on_batch (host, incl. gloo exchange)", model.train_on_batch)
    from vegans.engine import parallel
    parallel._all_reduce_now = timed("  of which gradient exchange on the host", parallel._all_reduce_now)
    n, skip = 16, 6
    X = np.concatenate([cfg.synthetic_images(256 * world, (3, 64, 64), seed=100 + i % 4) for i in range(n)], axis=0)
    torch.cuda.synchronize()
    dist.barrier()
    stamps = []
    plain = model.train_on_batch
    model.train_on_batch = lambda *a: (stamps.append(time.perf_counter()), plain(*a))[1]
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X, epochs=1, batch_size=256 * world, print_every=None, save_losses_every=None, enable_tensorboard=False)
    torch.cuda.synchronize()
    per_iter = (stamps[-1] - stamps[skip]) / (len(stamps) - 1 - skip)
    lines = ["rank %d: %.2f ms per iteration through fit() (iterations %d..%d)" % (rank, 1e3 * per_iter, skip, n - 1)]
    for name, values in spent.items():
        per = len(values) // n if len(values) >= n else 1
        tail = values[skip * per:]
        lines.append("   %-48s %.2f ms per iteration" % (name, 1e3 * sum(tail) / (n - skip)))
    for r in range(world):
        if r == rank:
            print("\n".join(lines), flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
