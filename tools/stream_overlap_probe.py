"""Does running independent critic passes on separate HIP streams buy anything on this chip?

Two independent forward + backward passes of the DCGAN-64 BatchNorm critic at batch 256 (as the fake and the real pass of
an adversary sub-step are), timed (a) one after the other on one stream, (b) on two streams, each both eagerly and
replayed from a captured HIP graph. Prints milliseconds per pair of passes.
    python tools/stream_overlap_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

import bench_configs as cfg  # noqa: E402
from vegans.engine import modules, ops  # noqa: E402

B = 256
D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64), 2).to("cuda")
modules.bind(D)
modules.prepare_parameters(D)
D.train()
xs = [ops.to_nhwc(torch.randn(B, 3, 64, 64, device="cuda")) for _ in range(2)]
seed = ops.unit_scalar("cuda")
side = torch.cuda.Stream()


def one_pass(x):
    loss = ops.loss_mean(D(x), 1.0)
    return torch.autograd.grad(loss, list(D.parameters()), seed)


def sequential():
    return one_pass(xs[0]), one_pass(xs[1])


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = one_pass(xs[1])
    a = one_pass(xs[0])
    main.wait_stream(side)
    return a, b


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def captured(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g.capture_begin(capture_error_mode="thread_local")
        keep = fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    return g, keep


for name, fn in (("one stream", sequential), ("two streams", forked)):
    eager = timed(fn)
    g, keep = captured(fn)
    replay = timed(g.replay)
    print("%-12s eager %.3f ms   graph replay %.3f ms" % (name, eager, replay), flush=True)
