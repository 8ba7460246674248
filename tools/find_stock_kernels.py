"""Which stock torch kernels still run inside one training iteration, and from where? Runs one eager WGAN-GP iteration
of a small DCGAN under torch.profiler with Python stacks and prints every aten op that launches a device kernel.
    python tools/find_stock_kernels.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench_configs as cfg  # noqa: E402
import vegans.GAN as gan  # noqa: E402

G = cfg.seeded_init(cfg.dcgan_generator(nz=16, ngf=16, nc=3, size=32), 1, 0.08)
D = cfg.seeded_init(cfg.dcgan_critic(ndf=16, nc=3, size=32), 2, 0.08)
m = gan.WassersteinGANGP(G, D, x_dim=(3, 32, 32), z_dim=(16, 1, 1), folder=None, fixed_noise_size=4)
m.use_step_graph = False
m._create_steps(None)
m.train()
X = m._to_device(cfg.synthetic_images(8, (3, 32, 32), seed=2))
Z = torch.randn(8, 16, 1, 1, device="cuda")
for _ in range(2):
    m.train_on_batch(X, Z)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.train_on_batch(X, Z)
    torch.cuda.synchronize()
# metadata-only ops (no device work)
QUIET = {"empty", "empty_like", "empty_strided", "view", "as_strided", "reshape", "detach", "detach_", "alias", "expand",
         "permute", "transpose", "t", "select", "slice", "unsqueeze", "squeeze", "_unsafe_view", "contiguous", "to",
         "_to_copy", "resolve_conj", "resolve_neg", "lift_fresh", "is_nonzero", "item", "_local_scalar_dense", "stride",
         "size", "result_type", "set_", "requires_grad_", "_reshape_alias", "view_as", "unbind", "narrow", "flatten"}
seen = {}
names = {}
for ev in prof.events():
    names[ev.name] = names.get(ev.name, 0) + 1
for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::fill_", "aten::copy_", "aten::add_"):
        chain, p = [], ev.cpu_parent
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        key = (ev.name, tuple(chain))
        seen[key] = seen.get(key, 0) + 1
for (name, stack), n in sorted(seen.items()):
    print("%3d x %s" % (n, name))
    for s in stack:
        print("        " + s)
