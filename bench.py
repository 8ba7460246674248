"""Benchmark of the VeGANs adversarial training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (the north-star configuration): WassersteinGANGP, DCGAN-64 generator / BatchNorm critic (ngf = ndf = 64),
synthetic 3x64x64 images, z_dim = [100,1,1], batch 256 per GPU, fp32, steps G:1 / D:1, RMSprop.  One "step" is one
iteration of the reference's training loop (AbstractGenerativeModel.py:316-327): the batch's host-to-device copy, one
latent draw, the G sub-step and the D sub-step with gradient penalty.  `value` is measured through the public `fit()`
on a numpy array (logging off): W warm-up batches in one `fit` call, then a second `fit` call over exactly K batches
inside the clock.  The same K iterations on batches already resident in HBM with pre-drawn latents are reported beside
it (`resident_batches`).  Both timed regions must have replayed the captured HIP graph K times (asserted).
With N > 1 the global batch is 256*N (weak scaling), sharded by rows, one rank per GPU, with bucketed RCCL all-reduce
of the gradients; the ranks come from `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or,
when bench.py is started from a plain shell (no WORLD_SIZE in the environment), from this process itself, which then
starts that launcher as a child before anything has touched the GPU and passes the ranks' output through.
`config.overlap` says how the steady state exchanged gradients: "captured" (all-reduces inside the replayed HIP graph,
overlapping backward), "chain" (between graph launches), "eager".

Prints ONE JSON line on rank 0: metric/value (whole-job images/s), plus
  roofline     — achieved TFLOP/s of the dominant convolution kernel (largest total duration): algorithmic conv FLOPs
                 of its launches divided by their HIP-event-timed durations, against the fp32 MFMA peak. The events
                 are taken in eager iterations run right after the timed region (which replays a HIP graph);
  hbm_kernels  — achieved HBM GB/s of the normalisation / activation / penalty / optimizer kernels against the 8 TB/s
                 peak, from the newest committed rocprofv3 counter passes of this benchmark;
  cpu_baseline — the CPU oracle (oracle/restated.py, stock torch CPU kernels) timed on the host cores on a bounded
                 sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench_configs as cfg  # noqa: E402

BATCH_PER_GPU = 256
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16), without sparsity
SETUP_STEPS = 3                    # untimed, before the warm-up: eager, eager, graph capture
ROOFLINE_STEPS = 5                 # instrumented eager iterations after the timed region
OPTIM_KWARGS = {"Generator": {"lr": 1e-4}, "Adversariat": {"lr": 5e-5}}


def build_model(ngpu):
    import vegans.GAN as gan
    G = cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 64), 1)
    D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64, batch_norm=True), 2)
    with contextlib.redirect_stdout(io.StringIO()):
        model = gan.WassersteinGANGP(G, D, x_dim=(3, 64, 64), z_dim=(100, 1, 1), optim_kwargs=OPTIM_KWARGS,
                                     fixed_noise_size=16, device="cuda", folder=None, ngpu=ngpu)
    return model


def cpu_baseline(iterations=2, warmup=1):
    """The oracle on the host cores: same networks, batch 256, fp32, steps 1:1."""
    from oracle.restated import OracleGAN
    torch.manual_seed(0)
    np.random.seed(0)
    G = cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 64), 1)
    D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64, batch_norm=True), 2)
    oracle = OracleGAN("wgangp", G, D, (3, 64, 64), (100, 1, 1), optim_kwargs=OPTIM_KWARGS, fixed_noise_size=16)
    X = cfg.synthetic_images(BATCH_PER_GPU * (iterations + warmup), (3, 64, 64), seed=0)
    oracle.fit(X[:BATCH_PER_GPU * warmup], None, epochs=1, batch_size=BATCH_PER_GPU)
    t0 = time.perf_counter()
    oracle.fit(X[BATCH_PER_GPU * warmup:], None, epochs=1, batch_size=BATCH_PER_GPU)
    dt = time.perf_counter() - t0
    return {"value": round(BATCH_PER_GPU * iterations / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(),
            "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": "%d WGAN-GP iterations of batch %d (DCGAN-64, fp32) after %d warm-up, oracle/restated.py on torch "
                      "CPU kernels" % (iterations, BATCH_PER_GPU, warmup)}


HBM_PEAK_GBPS = 8000.0             # MI355X HBM3E


def newest_pmc_summary():
    """(path, kernels) of the newest committed counter summary of the fp32 benchmark (profiles/round<N>_*_pmc_summary.json,
    written by tools/pmc_summary.py from rocprofv3 FETCH_SIZE / WRITE_SIZE runs of this benchmark): highest round
    number first, then the latest modification time; (None, {}) when there is none."""
    import glob
    import re

    def age(path):
        m = re.match(r"round(\d+)_", os.path.basename(path))
        return (int(m.group(1)) if m else -1, os.path.getmtime(path))

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), key=age)
    for path in reversed(found):
        try:
            return path, json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
    return None, {}


# the bandwidth-bound kernels of an iteration whose achieved HBM rate the line reports against the chip's peak
HBM_KERNELS = ("bn_apply_kernel", "bn_bwd_apply_kernel", "bn_dbwd_apply_kernel", "act_bwd_kernel", "colreduce_partial",
               "gp_interp_kernel", "gp_rownorm_kernel", "gp_penalty_bwd_kernel", "rmsprop_multi_kernel",
               "grad_sum_multi_kernel", "slab_reduce_kernel", "dgrad_scatter42_kernel", "dgrad_scatter_kernel", "stem_fwd_kernel",
               "stem_wgrad_kernel")


def hbm_kernels():
    """Achieved HBM GB/s of the elementwise, normalisation, penalty and optimizer kernels from the newest committed
    counter passes (bytes from FETCH_SIZE / WRITE_SIZE, duration from the kernel trace of the same run), each against
    the 8 TB/s peak."""
    path, kernels = newest_pmc_summary()
    rows = []
    for name, entry in kernels.items():
        if entry.get("gbps") is None:
            continue
        key = next((k for k in HBM_KERNELS if k in name), None)
        if key is None:
            continue
        # the kernel's name with its template arguments (balanced brackets), namespaces dropped
        at = name.index(key) + len(key)
        short, depth = key, 0
        for ch in name[at:]:
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif depth == 0:
                break
            short += ch
            if depth == 0:
                break
        short = short.replace("(anonymous namespace)::", "").replace(" >", ">")
        rows.append({"kernel": short, "launches": entry["launches"], "avg_us": entry["avg_us"],
                     "mb_per_launch": round(entry["hbm_bytes_per_launch"] / 1e6, 1), "gbps": entry["gbps"],
                     "frac_of_peak": round(entry["gbps"] / HBM_PEAK_GBPS, 3)})
    rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
    return {"peak_gbps": HBM_PEAK_GBPS, "source": os.path.relpath(path, ROOT) if path else None, "kernels": rows}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed counter summary, or None."""
    _, kernels = newest_pmc_summary()
    for name, entry in kernels.items():
        if kernel in name:
            return entry["hbm_bytes_per_launch"]
    stem = kernel.rsplit(",", 1)[0]           # summaries made before the kernel got its last template argument
    for name, entry in kernels.items():
        if stem in name:
            return entry["hbm_bytes_per_launch"]
    return None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell: start N fresh rank processes through torch.distributed.run as a
    CHILD of this process, which has not touched the GPU and never will, and exit with their status (replacing this
    process instead would be an exec; none is made). Of what the ranks write to stdout only the benchmark line reaches
    this process's stdout; everything else (gloo announces its connections there) goes to stderr."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()
    finally:
        proc.stdout.close()
        status = proc.wait()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", choices=["fp32", "fp32_split", "bf16", "bf16_compute"], default="fp32",
                    help="precision mode of the timed region (fp32 is the parity-checked mode; bf16 = feature maps stored "
                         "as bf16; bf16_compute = f32 tensors, bf16 matrix operands)")
    ap.add_argument("--no-bf16-extra", action="store_true", help="skip the extra bf16-mode and split-mode measurements of an fp32 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--allow-eager", action="store_true",
                    help="do not insist that the timed regions replay captured graphs (counter-collection runs set "
                         "VEGANS_HIP_GRAPH=0 so that every kernel is a dispatch of its own)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d, but the launcher started %d ranks" % (args.gpus, world))
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VEGANS_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 flow on one GPU
        import datetime
        limit = datetime.timedelta(seconds=300)      # a rank that never arrives ends the run in minutes, not half an hour
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=limit,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)

    from vegans.engine import ops, parallel, streams

    ops.set_matmul_precision(args.dtype)
    torch.manual_seed(0)
    np.random.seed(0)
    model = build_model(ngpu=world if world > 1 else None)
    model._create_steps(None)
    model.train()
    dp = parallel.context(model)
    global_batch = BATCH_PER_GPU * world

    from vegans.engine import graph

    # synthetic data: a few distinct global batches; `fit` gets them as one numpy array of K (resp. set-up + W) batches
    host_pool = [cfg.synthetic_images(global_batch, (3, 64, 64), seed=100 + i) for i in range(4)]

    def host_batches(n):
        return np.concatenate([host_pool[i % 4] for i in range(n)], axis=0)

    fit_kw = dict(epochs=1, batch_size=global_batch, print_every=None, save_losses_every=None, enable_tensorboard=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def fit_quietly(X):
        with contextlib.redirect_stdout(io.StringIO()):
            model.fit(X, **fit_kw)

    # Set-up, like building the model: the first iterations of a given shape run eagerly and the third records the HIP
    # graphs that later iterations replay (data-parallel runs: a chain of graphs with the gradient all-reduces in
    # between); then the warm-up iterations, all inside one untimed `fit` call.
    X_timed = host_batches(args.steps)              # made ahead of the warm-up, so the GPU does not sit idle for the
    try:                                            # time it takes to assemble it right before the timed region
        fit_quietly(host_batches(SETUP_STEPS + args.warmup))
    except RuntimeError as exc:
        # First contact of captured collectives with several ranks: the engine checks the replicas against each other
        # after the first replay (parallel.verify_replicas; every rank reaches the same verdict). Should they disagree,
        # the benchmark starts over with the collectives between the graph launches and says so in `config.overlap`.
        if world == 1 or "replicas disagree" not in str(exc):
            raise
        parallel.refuse_captured(exc)
        graph.reset(model)
        torch.manual_seed(0)
        np.random.seed(0)
        model = build_model(ngpu=world)
        model._create_steps(None)
        model.train()
        fit_quietly(host_batches(SETUP_STEPS + args.warmup))
    replayed = graph.replays(model)
    fence()
    t0 = time.perf_counter()
    fit_quietly(X_timed)
    fence()
    elapsed = time.perf_counter() - t0
    if graph.replays(model) - replayed != args.steps and not args.allow_eager:
        raise SystemExit("the timed fit() replayed %d graph iterations, expected %d: the headline would not be the "
                         "captured path" % (graph.replays(model) - replayed, args.steps))
    del X_timed

    # the same iterations on batches resident in HBM with pre-drawn latents (no upload, no latent draw in the clock)
    model.train()
    dp = model._dp                                  # the sharding context the last fit() call left on the model
    pool = [model._to_device(dp.shard(host_pool[i])) for i in range(4)]
    zgen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    latents = [torch.randn(len(pool[0]), 100, 1, 1, device="cuda", generator=zgen) for _ in range(4)]

    def step(i):
        model.train_on_batch(pool[i % 4], latents[i % 4])

    for i in range(SETUP_STEPS + args.warmup):     # a batch tensor made outside fit() may differ in strides: own graph
        step(i)
    replayed = graph.replays(model)
    fence()
    t_res = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    resident_elapsed = time.perf_counter() - t_res
    if graph.replays(model) - replayed != args.steps and not args.allow_eager:
        raise SystemExit("the resident-batch region did not replay the captured graph")

    # The north-star configuration names a bf16 mode next to fp32: measure it too (same model object, same batches;
    # the changed precision makes train_on_batch record new graphs), reported beside the fp32 figure, never as `value`.
    # Likewise the split mode, which computes f32-grade products on the bf16 matrix cores.
    extras = {}
    notes = {"bf16": "bf16 data path: feature maps and their gradients stored as bf16 in HBM and fed to the bf16 matrix "
                     "cores as they lie in memory, f32 accumulation; parameters, their gradients, optimizer state, "
                     "BatchNorm statistics, losses, images and scores f32; losses within 5e-2 of the fp32 reference "
                     "(tests/test_models_gpu.py, tests/test_bf16_storage_gpu.py)",
             "bf16_compute": "f32 tensors in HBM; operands of the convolutions rounded to bf16 ahead of the matrix "
                             "cores, f32 accumulation; everything else f32; losses within 2e-2 of the fp32 reference "
                             "(tests/test_models_gpu.py)",
             "fp32_split": "convolution operands split exactly into three bf16 terms, six bf16 MFMA products per f32 "
                           "product, f32 accumulation; error against float64 no larger than the f32 MFMA kernels' "
                           "(tests/test_ops_gpu.py) and the fp32 parity tolerances hold (tests/test_models_gpu.py)"}
    if args.dtype == "fp32" and not args.no_bf16_extra and world == 1:     # extras on the single-GPU line only
        for mode in ("bf16", "bf16_compute", "fp32_split"):
            ops.set_matmul_precision(mode)
            for i in range(SETUP_STEPS + args.warmup):
                step(i)
            fence()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            fence()
            mode_elapsed = time.perf_counter() - t1
            ops.set_matmul_precision("fp32")
            if world > 1:
                t = torch.tensor([mode_elapsed], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                mode_elapsed = float(t.item())
            extras[mode + "_mode"] = {"value": round(global_batch * args.steps / mode_elapsed, 2), "unit": "images/s",
                                      "ms_per_step": round(1e3 * mode_elapsed / args.steps, 3), "note": notes[mode]}
            if mode == "bf16" and not args.no_kernel_timing:
                # the bf16 mode's own roofline entry: its dominant contraction against the dense bf16 matrix peak
                ops.set_matmul_precision(mode)
                bt = ops.ConvTimer()
                ops.conv_timer = bt
                for i in range(3):
                    step(i)
                ops.conv_timer = None
                fence()
                ops.set_matmul_precision("fp32")
                # every launch of a fat layer (17 GFLOP per pass or more: pgemm_kernel on bf16 tensors), all three
                # contractions together and each by itself; a weight-gradient launch may cover several passes
                by_mode = {}
                for _, g, start, stop, cmode in bt.records:
                    flops = 2.0 * g.N * g.Ho * g.Wo * g.Co * g.kh * g.kw * g.Ci
                    if flops < 1.5e10:
                        continue
                    acc = by_mode.setdefault(("fwd", "dgrad", "wgrad")[cmode], [0, 0.0, 0.0])
                    acc[0] += 1
                    acc[1] += start.elapsed_time(stop)
                    acc[2] += flops
                n = sum(v[0] for v in by_mode.values())
                ms = sum(v[1] for v in by_mode.values())
                fl = sum(v[2] for v in by_mode.values())
                tf = fl / (ms * 1e-3) / 1e12
                extras["bf16_mode"]["roofline"] = {
                    "bound": "mfma", "kernel": "pgemm_kernel<..., NP = 1> (bf16 implicit GEMM on bf16 tensors): all launches "
                                               "on the fat layers",
                    "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 4), "launches": n, "avg_launch_us": round(1e3 * ms / n, 2),
                    "by_contraction": {k: {"launches": v[0], "avg_launch_us": round(1e3 * v[1] / v[0], 2),
                                           "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(by_mode.items())},
                    "traffic": None,
                    "note": "operands are read as bf16 from HBM / L2 as they lie in memory; at 340 flop per HBM byte these "
                            "layers sit at the ridge of the bf16 roofline, and the implicit GEMM re-reads each "
                            "activation kh*kw/s^2 times through L2: bound by operand traffic into the CUs, not by the "
                            "matrix cores"}

    # Per-kernel durations for the roofline entry. Kernels inside a replayed graph cannot be timed one by one, so
    # the same iterations run again eagerly with a start/stop event pair attached to every convolution launch, on the
    # launch stream. Each instrumented iteration is queued behind a replayed (untimed) one: the host enqueues the eager
    # launches while the GPU works through the replay, so the kernels find their successors already enqueued and the
    # chip in the state the timed region keeps it in. (Behind an idle stretch or a spin kernel the same kernels run
    # about a tenth slower for some ten milliseconds: the trace of such a run shows it per iteration.)
    timer = None
    if not args.no_kernel_timing:
        timer = ops.ConvTimer()
        # one stream here: a kernel's begin-to-end time is its own cost only while nothing else runs beside it
        # (the timed regions above run the critic's independent passes on separate streams)
        streams_were = streams.set_enabled(False)
        for i in range(SETUP_STEPS):                # single-stream iterations are a different program: own graph
            step(i)
        for i in range(ROOFLINE_STEPS):
            step(2 * i)                             # replayed
            ops.conv_timer = timer
            step(2 * i + 1)                         # eager, instrumented
            ops.conv_timer = None
        streams.set_enabled(streams_were)
        fence()
    if world > 1:
        t = torch.tensor([elapsed, resident_elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, resident_elapsed = float(t[0].item()), float(t[1].item())

    if rank == 0:
        g_macs, d_macs = cfg.dcgan_forward_macs(64)
        flops_per_image = 2.0 * (4 * g_macs + 14 * d_macs)     # necessary conv work of one WGAN-GP iteration
        value = global_batch * args.steps / elapsed
        out = {
            "metric": "training images/sec (G+D step) DCGAN-64 WGAN-GP",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16_compute": "bf16 operands, f32 tensors", "fp32_split": "f32 (3xbf16 split)"}[args.dtype], "data": "synthetic",
            "config": {"workload": "WassersteinGANGP DCGAN-64 (ngf=ndf=64, BatchNorm critic), 3x64x64, z=100, "
                                   "batch 256 per GPU, %s, steps G1:D1, RMSprop" % args.dtype,
                       "global_batch": global_batch, "parallelism": "dp%d" % world if world > 1 else "single",
                       "overlap": parallel.overlap_mode(model) if world > 1 else None,
                       "backend": dist.get_backend() if world > 1 else None},
            "whole_step_conv_tflops": round(value * flops_per_image / 1e12, 2),
            "measured_through": "fit() on a numpy array: pinned upload, latent draw and the replayed HIP graph inside the "
                                "clock (%d graph replays asserted)" % args.steps,
            "resident_batches": {"value": round(global_batch * args.steps / resident_elapsed, 2), "unit": "images/s",
                                 "ms_per_step": round(1e3 * resident_elapsed / args.steps, 3),
                                 "note": "train_on_batch on batches already in HBM with pre-drawn latents"},
        }
        if timer is not None:
            summary = timer.summary()
            name, (n, ms, fl) = max(summary.items(), key=lambda kv: kv[1][1])
            achieved = fl / (ms * 1e-3) / 1e12
            all_ms = sum(v[1] for v in summary.values())
            all_fl = sum(v[2] for v in summary.values())
            out["roofline"] = {
                "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": measured_traffic(name) if args.dtype == "fp32" else None,
                "traffic_source": "HBM bytes per launch from the newest committed rocprofv3 counter passes "
                                  "(profiles/*_pmc_summary.json: FETCH_SIZE and WRITE_SIZE runs of this benchmark)",
                "launches": n, "avg_launch_us": round(1e3 * ms / n, 2),
                "ms_per_step": round(ms / ROOFLINE_STEPS, 3),
                "timed": "HIP events attached to each launch's dispatch (kernel begin to kernel end), %d eager "
                         "single-stream iterations after the timed region, each queued behind a replayed iteration "
                         "(the timed region replays a HIP graph)" % ROOFLINE_STEPS,
            }
            out["conv_kernels"] = {"ms_per_step": round(all_ms / ROOFLINE_STEPS, 3),
                                   "aggregate_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2)}
        out.update(extras)
        if args.dtype == "fp32":
            out["hbm_kernels"] = hbm_kernels()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
