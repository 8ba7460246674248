"""`ngpu=N` inside ONE `fit()` call of ONE process.

The reference fans a single process out over the visible GPUs with `torch.nn.DataParallel` (vegans/utils/networks.py:39-41).
The engine's multi-GPU path is one process per GPU (engine/parallel.py); when the caller's process is not part of a
process group, `fit` starts those processes itself:

  parent   writes the model object (torch.save), the training arrays, the fit() arguments and the three random
           generator states to a scratch directory and starts N fresh Python processes (`python -m
           vegans.engine.spawn <dir> <rank> <world> <backend>`; new interpreters, never a re-exec of this process),
           then watches all of them: the first worker that exits with an error ends the others (they would otherwise
           sit in a collective until the communication library's timeout) and its log becomes the exception text;
  worker r selects GPU r, joins the process group through a file in the scratch directory (RCCL;
           VEGANS_DP_BACKEND=gloo for rehearsals), loads the model onto its
           GPU, restores the generator states (so every rank draws the same latent batches and penalty alphas and
           keeps its rows) and calls the ordinary `fit`, which shards every batch;
  rank 0   writes parameters, BatchNorm buffers, optimizer state, logged losses, hyper-parameters and the advanced
           generator states back; the parent loads them INTO the caller's own module, optimizer and model objects.
"""
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

_ARRAYS = ("X_train", "y_train", "X_test", "y_test")


POLL_SECONDS = 0.2
_sleep = time.sleep


def _stop(procs, grace=5.0):
    """End every worker that is still running: SIGTERM, then SIGKILL after `grace` seconds."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        p.terminate()
    deadline = time.monotonic() + grace
    for p in alive:
        try:
            p.wait(timeout=max(deadline - time.monotonic(), 0.1))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def wait_all(procs):
    """Wait for all `procs` (subprocess.Popen) together. Returns None when every one exited with status 0; as soon as
    one exits with another status the rest are stopped and (index, status) of that first failure is returned. Whatever
    ends the wait (an exception in the caller's thread, KeyboardInterrupt), no worker is left running."""
    try:
        while True:
            running = False
            for i, p in enumerate(procs):
                rc = p.poll()
                if rc is None:
                    running = True
                elif rc != 0:
                    return i, rc
            if not running:
                return None
            _sleep(POLL_SECONDS)
    finally:
        _stop(procs)


def fit_in_workers(model, fit_kwargs):
    """Run `model.fit(**fit_kwargs)` data-parallel over `model.ngpu` worker processes and bring the result back into
    `model`."""
    world = int(model.ngpu)
    backend = os.environ.get("VEGANS_DP_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise RuntimeError("ngpu=%d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = tempfile.mkdtemp(prefix="vegans_dp_", dir=base)
    try:
        kwargs = dict(fit_kwargs)
        for name in _ARRAYS:
            arr = kwargs.pop(name, None)
            if arr is not None:
                np.save(os.path.join(work, name + ".npy"), np.asarray(arr), allow_pickle=False)
        torch.cuda.synchronize()
        torch.save(model, os.path.join(work, "model.pt"))
        state = {"kwargs": kwargs, "torch_cpu": torch.get_rng_state(), "torch_cuda": torch.cuda.get_rng_state(),
                 "numpy": np.random.get_state(), "class": type(model).__name__}
        with open(os.path.join(work, "call.pkl"), "wb") as f:
            pickle.dump(state, f)
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([p for p in sys.path if p] + [env.get("PYTHONPATH", "")])
        procs, logs = [], []
        try:
            for rank in range(world):
                logs.append(open(os.path.join(work, "rank%d.log" % rank), "wb"))
                procs.append(subprocess.Popen([sys.executable, "-m", "vegans.engine.spawn", work, str(rank), str(world),
                                               backend], env=env, stdout=logs[-1], stderr=subprocess.STDOUT))
            failed = wait_all(procs)
        finally:
            _stop(procs)
            for log in logs:
                log.close()
        if failed is not None:
            rank, rc = failed
            tail = open(os.path.join(work, "rank%d.log" % rank), "rb").read()[-4000:].decode(errors="replace")
            raise RuntimeError("data-parallel worker %d of %d exited with status %d (the other workers were stopped):\n%s"
                               % (rank, world, rc, tail))
        sys.stdout.write(open(os.path.join(work, "rank0.log"), "rb").read().decode(errors="replace"))
        result = torch.load(os.path.join(work, "result.pt"), weights_only=False, map_location="cuda")
    finally:
        shutil.rmtree(work, ignore_errors=True)
    _adopt(model, result)


def _adopt(model, result):
    """Write rank 0's trained state into the caller's own objects (Parameters and buffers keep their identity)."""
    for name, net in model.neural_nets.items():
        net.network.load_state_dict(result["nets"][name])
        model.optimizers[name].load_state_dict(result["optimizers"][name])
    for key in ("logged_losses", "fixed_labels", "steps", "total_training_time", "batch_training_times"):
        if key in result["attributes"]:
            setattr(model, key, result["attributes"][key])
    for key, value in result["hyperparameters"].items():
        if key != "optimizers":
            model.hyperparameters[key] = value
    torch.set_rng_state(result["torch_cpu"].cpu())
    torch.cuda.set_rng_state(result["torch_cuda"].cpu())
    np.random.set_state(result["numpy"])
    model.eval()


def _worker(work, rank, world, backend):
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("worker %d sees no GPU" % rank)
    device = rank % ndev           # more ranks than GPUs only happens in gloo rehearsals on a one-GPU box
    torch.cuda.set_device(device)
    # rendezvous through a file of the scratch directory: no TCP port that another process could take in between
    rendezvous = "file://" + os.path.join(work, "rendezvous")
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=rendezvous, rank=rank, world_size=world,
                                device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, init_method=rendezvous, rank=rank, world_size=world)
    with open(os.path.join(work, "call.pkl"), "rb") as f:
        state = pickle.load(f)
    model = torch.load(os.path.join(work, "model.pt"), weights_only=False, map_location=torch.device("cuda", device))
    torch.set_rng_state(state["torch_cpu"])
    torch.cuda.set_rng_state(state["torch_cuda"].cpu())
    np.random.set_state(state["numpy"])
    kwargs = dict(state["kwargs"])
    for name in _ARRAYS:
        path = os.path.join(work, name + ".npy")
        if os.path.exists(path):
            kwargs[name] = np.load(path, mmap_mode="r", allow_pickle=False)
    model.fit(**kwargs)
    torch.cuda.synchronize()
    if rank == 0:
        attributes = {k: getattr(model, k) for k in ("logged_losses", "fixed_labels", "steps", "total_training_time",
                                                     "batch_training_times") if hasattr(model, k)}
        result = {
            "nets": {n: net.network.state_dict() for n, net in model.neural_nets.items()},
            "optimizers": {n: opt.state_dict() for n, opt in model.optimizers.items()},
            "attributes": attributes,
            "hyperparameters": {k: v for k, v in model.hyperparameters.items() if k != "optimizers"},
            "torch_cpu": torch.get_rng_state(), "torch_cuda": torch.cuda.get_rng_state(), "numpy": np.random.get_state(),
        }
        torch.save(result, os.path.join(work, "result.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    _worker(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
