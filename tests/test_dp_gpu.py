"""Data-parallel path on the GPU: two ranks (both on cuda:0, gloo transport, since one box has one GPU) train a
BatchNorm-free WGAN-GP on a global batch sharded by rows; the result must equal a single-process run on the whole
batch (the per-rank means weighted by rows/global_rows sum to the global mean), and both ranks must end with
identical weights.  The run is long enough for the iteration to be captured and replayed as HIP graphs on both
sides."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, io, contextlib
ROOT = %(root)r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np, torch, torch.distributed as dist
from torch import nn
import bench_configs as cfg
import vegans.GAN as gan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

def nets():
    G = nn.Sequential(nn.ConvTranspose2d(8, 16, 4, 1, 0), nn.ReLU(), nn.ConvTranspose2d(16, 8, 4, 2, 1), nn.ReLU(),
                      nn.ConvTranspose2d(8, 3, 4, 2, 1), nn.Tanh())
    D = nn.Sequential(nn.Conv2d(3, 8, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(8, 16, 4, 2, 1), nn.LeakyReLU(0.2),
                      nn.Conv2d(16, 1, 4, 1, 0), nn.Flatten(), nn.Identity())
    return cfg.seeded_init(G, 1, 0.1), cfg.seeded_init(D, 2, 0.1)

def run(ngpu, rows):
    torch.manual_seed(5); np.random.seed(5)
    G, D = nets()
    with contextlib.redirect_stdout(io.StringIO()):
        m = gan.WassersteinGANGP(G, D, x_dim=(3, 16, 16), z_dim=(8, 1, 1), folder=None, fixed_noise_size=2, ngpu=ngpu,
                                 optim_kwargs={"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 1e-3}})
    X = cfg.synthetic_images(rows * 6 - 2, (3, 16, 16), seed=3)       # last batch is partial (uneven shards)
    zgen = torch.Generator().manual_seed(11)
    zs = iter([torch.randn(rows, 8, 1, 1, generator=zgen) for _ in range(6)])
    m.sample = lambda n: next(zs)[:n].to("cuda").requires_grad_(True)
    np.random.seed(17)
    with contextlib.redirect_stdout(io.StringIO()):
        m.fit(X, epochs=1, batch_size=rows, print_every=None, save_losses_every=None, enable_tensorboard=False)
    # batches 1-2 run kernel by kernel, batch 3 is recorded and batches 3-5 replayed: under data parallelism as a
    # chain of graphs with the bucket all-reduces in between
    from vegans.engine import graph
    full = [e for e in graph._graphs[m].values() if e.graph is not None]
    assert len(full) == 1 and not full[0].failed, [(e.eager_calls, e.failed) for e in graph._graphs[m].values()]
    n_graphs = sum(isinstance(s, torch.cuda.CUDAGraph) for s in full[0].segments)
    n_host = len(full[0].segments) - n_graphs
    assert (n_graphs, n_host) == ((3, 2) if ngpu else (1, 0)), (n_graphs, n_host)
    return [p.detach().cpu().clone() for net in (G, D) for p in net.parameters()]

dp = run(world, 8)
ref = run(None, 8)
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-3)) for a, b in zip(dp, ref))
flat = torch.cat([p.reshape(-1) for p in dp]).cuda()
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
same = all(torch.equal(both[0], b) for b in both)
print("rank %%d worst_rel %%.3e identical_across_ranks %%s" %% (rank, worst, same))
assert same, "ranks diverged"
assert worst < 2e-4, worst
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_data_parallel_matches_single_process(gpu, tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
        assert "ok" in out
