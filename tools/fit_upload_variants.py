"""Compare ways of getting host batches to the GPU inside `fit()`: time per batch of 256 with each.
    python tools/fit_upload_variants.py [batches]
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

import bench  # noqa: E402
import bench_configs as cfg  # noqa: E402
from vegans.engine import ops  # noqa: E402
from vegans.utils import utils  # noqa: E402


def same_stream(model):
    """Upload on the training stream, no second stream at all."""
    def batches(X_train, y_train, batch_size, dp):
        for batch in utils.iterate_batches(X_train, y_train, batch_size):
            yield model._to_device(dp.shard(batch))
    return batches


def side_stream(model, wait_main, fixed_pinned):
    stream = torch.cuda.Stream()
    pinned = [None, None, None]
    free_at = [None, None, None]

    def to_device(array, slot):
        if not fixed_pinned:
            return model._to_device(array)
        if pinned[slot] is None or pinned[slot].shape != array.shape:
            pinned[slot] = torch.empty(array.shape, dtype=torch.float32).pin_memory()
        if free_at[slot] is not None:
            free_at[slot].synchronize()
        pinned[slot].numpy()[...] = array
        t = pinned[slot].to(model.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        free_at[slot] = ev
        return ops.to_nhwc(t)

    def batches(X_train, y_train, batch_size, dp):
        pending = None
        for i, batch in enumerate(utils.iterate_batches(X_train, y_train, batch_size)):
            if wait_main:
                stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                t = to_device(dp.shard(batch), i % 3)
                done = torch.cuda.Event()
                done.record(stream)
            if pending is not None:
                out, ev = pending
                torch.cuda.current_stream().wait_event(ev)
                out.record_stream(torch.cuda.current_stream())
                yield out
            pending = (t, done)
        if pending is not None:
            out, ev = pending
            torch.cuda.current_stream().wait_event(ev)
            out.record_stream(torch.cuda.current_stream())
            yield out
    return batches


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench.build_model(None)
    X = cfg.synthetic_images(256 * batches, (3, 64, 64), seed=1)
    kw = dict(batch_size=256, print_every=None, save_losses_every=None, enable_tensorboard=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(X[:256 * 6], epochs=1, **kw)
    torch.cuda.synchronize()
    shipped = model._device_batches
    X_dev = torch.from_numpy(X).cuda()
    variants = [
        ("as shipped (side stream, host-side bound on run-ahead)", shipped, X),
        ("training stream only", same_stream(model), X),
        ("side stream, waits for the training stream", side_stream(model, True, False), X),
        ("side stream, no wait, unbounded run-ahead", side_stream(model, False, False), X),
        ("side stream, waits, three fixed pinned buffers", side_stream(model, True, True), X),
        ("side stream, no wait, three fixed pinned buffers", side_stream(model, False, True), X),
        ("as shipped, input already a device tensor", shipped, X_dev),
    ]
    for name, fn, data in variants:
        model._device_batches = fn
        best = None
        for _ in range(2):
            with contextlib.redirect_stdout(io.StringIO()):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.fit(data, epochs=1, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("%-60s %.3f ms per batch" % (name, 1e3 * best / batches), flush=True)


if __name__ == "__main__":
    main()
