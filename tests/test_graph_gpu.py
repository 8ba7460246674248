"""The captured training step (engine/graph.py) against the same step run kernel by kernel.

A replayed HIP graph launches the very kernels the eager path launches on the same data, and every kernel of the
library is deterministic (no atomics), so for the clipped Wasserstein GAN everything must agree BIT FOR BIT:
parameters, BatchNorm buffers, optimizer state, published losses and gradients.

With the gradient penalty a critic weight receives four gradient contributions (fake, real and two from the
double-backward pass).  The engine's gradient ledger adds them in an order fixed by the program (engine/grads.py), not
in the order autograd's scheduler happens to reach them, so the penalty variant is bit-reproducible as well; the
host random stream (numpy alphas) must have advanced identically.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(algo, steps=None, seed=4, width=8, **extra):
    import bench_configs as cfg
    import vegans.GAN as gan
    torch.manual_seed(seed)
    np.random.seed(seed)
    G = cfg.seeded_init(cfg.dcgan_generator(nz=16, ngf=width, nc=3, size=16), 1, 0.08)
    D = cfg.seeded_init(cfg.dcgan_critic(ndf=width, nc=3, size=16, sigmoid=algo in ("LSGAN", "VanillaGAN")), 2, 0.08)
    okw = {"Generator": {"lr": 1e-3}, "Adversariat": {"lr": 5e-4}}
    m = getattr(gan, algo)(G, D, x_dim=(3, 16, 16), z_dim=(16, 1, 1), optim_kwargs=okw, folder=None,
                           fixed_noise_size=4, **extra)
    m._create_steps(copy.deepcopy(steps))
    m.train()
    return m


def _run(m, batches, latents, use_graph):
    from vegans.engine import graph
    m.use_step_graph = use_graph
    np.random.seed(77)
    losses = []
    for X, Z in zip(batches, latents):
        m.train_on_batch(X, Z)
        losses.append({k: v.detach().clone() for k, v in m._losses.items()})
    torch.cuda.synchronize()
    entries = list(graph._graphs.get(m, {}).values())
    return losses, np.random.random(), entries


def _state(m):
    out = {}
    for name, net in m.neural_nets.items():
        for k, v in net.state_dict().items():
            out[name + "." + k] = v.detach().clone()
        for i, p in enumerate(net.parameters()):
            for k, v in m.optimizers[name].state[p].items():
                out["%s.opt%d.%s" % (name, i, k)] = v.detach().clone() if isinstance(v, torch.Tensor) else v
            out["%s.grad%d" % (name, i)] = None if p.grad is None else p.grad.detach().clone()
    return out


def _same(a, b, exact, what):
    if exact:
        assert torch.equal(a, b), what
    else:
        scale = max(float(b.abs().max()), 1e-3)
        err = float((a.double() - b.double()).abs().max())
        assert err <= 1e-5 * scale, "%s: |diff| %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("algo,steps,extra,exact", [
    ("WassersteinGAN", {"Adversariat": 3}, {"clip_val": 0.02}, True),
    ("WassersteinGAN", None, {}, True),
    ("WassersteinGANGP", None, {}, True),
    ("LSGAN", {"Generator": 2}, {}, True),          # Adam: step-dependent scalars refreshed per replay
])
def test_replayed_step_matches_eager(gpu, algo, steps, extra, exact):
    import bench_configs as cfg
    iters, B = 7, 8
    data = cfg.synthetic_images(B * iters, (3, 16, 16), seed=8)
    g = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 16, 1, 1, generator=g) for _ in range(iters)]

    results = []
    for use_graph in (False, True):
        m = _model(algo, steps, **extra)
        batches = [m._to_device(data[i * B:(i + 1) * B]) for i in range(iters)]
        latents = [z.to("cuda") for z in zs]
        losses, next_draw, entries = _run(m, batches, latents, use_graph)
        if use_graph:
            assert len(entries) == 1 and entries[0].graph is not None and not entries[0].failed
            assert entries[0].eager_calls == 2          # two eager calls, the rest replayed
        else:
            assert entries == []
        results.append((losses, next_draw, _state(m)))

    (l0, r0, s0), (l1, r1, s1) = results
    assert r0 == r1, "host random stream advanced differently under replay"
    for i, (a, b) in enumerate(zip(l0, l1)):
        assert set(a) == set(b)
        for k in a:
            if k == "RealFakeRatio" and not exact:
                continue                                  # a quotient of two losses, logged only
            _same(a[k].reshape(1), b[k].reshape(1), exact, "iteration %d loss %s" % (i, k))
    assert set(s0) == set(s1)
    for k in s0:
        if isinstance(s0[k], torch.Tensor):
            _same(s0[k].float(), s1[k].float(), exact or ".opt" in k and k.endswith("step"), k)
        else:
            assert s0[k] == s1[k], k


def test_replayed_step_matches_eager_in_split_mode(gpu):
    """The split mode keeps the bf16 terms of operands for the rest of an iteration (ops.plane_cache); a captured
    graph must contain exactly the splits the eager iteration runs and read no planes made outside itself: replayed
    and eager iterations of the clipped Wasserstein GAN agree bit for bit, as in f32. Width 32 puts several layers on
    the plane kernels."""
    import bench_configs as cfg
    from vegans.engine import ops
    iters, B = 6, 8
    data = cfg.synthetic_images(B * iters, (3, 16, 16), seed=8)
    g = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 16, 1, 1, generator=g) for _ in range(iters)]
    previous = ops.set_matmul_precision("fp32_split")
    try:
        results = []
        for use_graph in (False, True):
            m = _model("WassersteinGAN", {"Adversariat": 2}, width=32, clip_val=0.02)
            batches = [m._to_device(data[i * B:(i + 1) * B]) for i in range(iters)]
            losses, next_draw, entries = _run(m, batches, [z.to("cuda") for z in zs], use_graph)
            if use_graph:
                assert len(entries) == 1 and entries[0].graph is not None and not entries[0].failed
            results.append((losses, _state(m)))
            assert not ops.plane_cache.active and not ops.plane_cache.entries
    finally:
        ops.set_matmul_precision(previous)
    (l0, s0), (l1, s1) = results
    for i, (a, b) in enumerate(zip(l0, l1)):
        for k in a:
            _same(a[k].reshape(1), b[k].reshape(1), True, "iteration %d loss %s" % (i, k))
    for k in s0:
        if isinstance(s0[k], torch.Tensor):
            _same(s0[k].float(), s1[k].float(), True, k)


@pytest.mark.parametrize("algo,steps,extra", [("WassersteinGANGP", None, {}), ("LSGAN", {"Generator": 2}, {})])
def test_replayed_step_matches_eager_in_the_bf16_data_path(gpu, algo, steps, extra):
    """The bf16 data path (bf16 feature maps, the critic's passes of a sub-step on separate streams sharing the bf16
    copies of the weights): replayed and eager iterations agree bit for bit, the first run of the pair included, as in
    f32. Width 32 puts the inner layers on the bf16 contractions."""
    import bench_configs as cfg
    from vegans.engine import ops
    iters, B = 6, 8
    data = cfg.synthetic_images(B * iters, (3, 16, 16), seed=8)
    g = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 16, 1, 1, generator=g) for _ in range(iters)]
    previous = ops.set_matmul_precision("bf16")
    try:
        results = []
        for use_graph in (False, True, False):
            m = _model(algo, steps, width=32, **extra)
            batches = [m._to_device(data[i * B:(i + 1) * B]) for i in range(iters)]
            losses, next_draw, entries = _run(m, batches, [z.to("cuda") for z in zs], use_graph)
            if use_graph:
                assert len(entries) == 1 and entries[0].graph is not None and not entries[0].failed
            results.append((losses, _state(m)))
            assert not ops.plane_cache.active and not ops.plane_cache.entries
    finally:
        ops.set_matmul_precision(previous)
    for other in (1, 2):
        (l0, s0), (l1, s1) = results[0], results[other]
        for i, (a, b) in enumerate(zip(l0, l1)):
            for k in a:
                _same(a[k].reshape(1), b[k].reshape(1), True, "run %d iteration %d loss %s" % (other, i, k))
        for k in s0:
            if isinstance(s0[k], torch.Tensor):
                _same(s0[k].float(), s1[k].float(), True, "run %d %s" % (other, k))


def test_graph_follows_hyperparameter_and_shape_changes(gpu):
    """Values frozen into a graph are part of its signature: a new learning rate or batch shape takes a new graph
    (after its own eager calls) instead of replaying stale constants."""
    import bench_configs as cfg
    from vegans.engine import graph
    m = _model("WassersteinGANGP")
    X = m._to_device(cfg.synthetic_images(8, (3, 16, 16), seed=1))
    Z = torch.randn(8, 16, 1, 1, device="cuda")
    for _ in range(4):
        m.train_on_batch(X, Z)
    assert len(graph._graphs[m]) == 1
    m.optimizers["Generator"].param_groups[0]["lr"] = 5e-3
    before = [p.detach().clone() for p in m.generator.parameters()]
    for _ in range(4):
        m.train_on_batch(X, Z)
    assert len(graph._graphs[m]) == 2
    # an RMSprop step moves a weight by at most lr / (sqrt(1 - alpha) + eps) = 10 lr; four steps at the new rate move
    # further than four steps at the old rate ever could
    moved = max(float((p.detach() - b).abs().max()) for p, b in zip(m.generator.parameters(), before))
    assert moved > 4 * 10 * 1e-3 * 0.5, moved
    m.train_on_batch(X[:5].contiguous(memory_format=torch.channels_last), Z[:5].contiguous())   # ragged last batch
    assert len(graph._graphs[m]) == 3
    m.eval()
    assert not graph._eligible(m)                # not in training mode: never captured or replayed
    m.train()
    m.use_step_graph = False
    assert not graph._eligible(m)
    graph.reset(m)
    assert m not in graph._graphs


def test_fit_through_graph_matches_fit_without(gpu):
    """The public `fit` (logging and a ragged last batch in between) gives bit-identical weights either way
    (gradient-penalty variant: four gradient contributions per critic weight, added in ledger order)."""
    import bench_configs as cfg
    X = cfg.synthetic_images(8 * 6 + 3, (3, 16, 16), seed=3)
    finals = []
    for use_graph in (False, True):
        m = _model("WassersteinGANGP", seed=9)
        m.use_step_graph = use_graph
        torch.manual_seed(21)        # latent draws
        np.random.seed(22)           # alpha draws
        m.fit(X, epochs=2, batch_size=8, print_every=None, save_losses_every=3, enable_tensorboard=False)
        finals.append(({k: v.clone() for k, v in m.generator.state_dict().items()},
                       {k: v.clone() for k, v in m.adversariat.state_dict().items()}, m.get_losses()))
    for a, b in zip(finals[0][:2], finals[1][:2]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert finals[0][2] == finals[1][2]


def test_three_runs_are_bit_identical(gpu):
    """Same seeds, three runs in one process (the first run of a process included), eager and replayed: the gradient
    ledger adds the four contributions to every critic weight in program order, so the gradients every optimizer step
    receives, the losses and the final weights are the same bit patterns every time."""
    import bench_configs as cfg
    iters, B = 6, 8
    data = cfg.synthetic_images(B * iters, (3, 16, 16), seed=8)
    g = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 16, 1, 1, generator=g) for _ in range(iters)]
    runs = []
    for use_graph in (False, False, True, False):
        m = _model("WassersteinGANGP", width=16)
        seen = []
        plain = m._step

        def step(who=None, plain=plain, m=m, seen=seen):
            seen.append([p.grad.detach().clone() for p in m.neural_nets[who].parameters()])
            plain(who=who)

        if not use_graph:
            m._step = step
        batches = [m._to_device(data[i * B:(i + 1) * B]) for i in range(iters)]
        losses, _, _ = _run(m, batches, [z.to("cuda") for z in zs], use_graph)
        runs.append((seen, losses, _state(m)))
    for other in runs[1:]:
        if other[0] and runs[0][0]:
            for k, (a, b) in enumerate(zip(runs[0][0], other[0])):
                for i, (ga, gb) in enumerate(zip(a, b)):
                    assert torch.equal(ga, gb), "optimizer step %d parameter %d: gradients differ between runs" % (k, i)
        for a, b in zip(runs[0][1], other[1]):
            for k in a:
                assert torch.equal(a[k], b[k]) or (torch.isnan(a[k]) and torch.isnan(b[k])), k
        for k in runs[0][2]:
            if isinstance(runs[0][2][k], torch.Tensor):
                assert torch.equal(runs[0][2][k], other[2][k]), k


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_merged_weight_gradients_agree_with_one_launch_per_contribution(gpu, monkeypatch, mode):
    """The weight-gradient contributions of a layer run as one contraction when the ledger sums them (the default), or
    each in its own launch (VEGANS_HIP_MERGED_WGRAD=0): the same sums in another order. The gradients the first two
    optimizer steps receive agree to f32 summation noise, and the merged path really ran."""
    import bench_configs as cfg
    from vegans.engine import ops
    B = 8
    data = cfg.synthetic_images(B, (3, 16, 16), seed=8)
    z = torch.randn(B, 16, 1, 1, generator=torch.Generator().manual_seed(12))
    previous = ops.set_matmul_precision(mode)
    try:
        seen, merged_launches = {}, []
        plain_pairs = ops.wgrad_of_pairs

        def counting(items):
            merged_launches.append(len(items))
            return plain_pairs(items)

        monkeypatch.setattr(ops, "wgrad_of_pairs", counting)
        for flag in ("1", "0"):
            monkeypatch.setenv("VEGANS_HIP_MERGED_WGRAD", flag)
            m = _model("WassersteinGANGP", width=16)
            grads_seen = []
            plain = m._step

            def step(who=None, plain=plain, m=m, grads_seen=grads_seen):
                grads_seen.append([p.grad.detach().float().clone() for p in m.neural_nets[who].parameters()])
                plain(who=who)

            m._step = step
            _run(m, [m._to_device(data)], [z.to("cuda")], False)
            seen[flag] = grads_seen
        assert merged_launches and max(merged_launches) >= 3, merged_launches
        tol = 2e-5 if mode == "fp32" else 2e-3
        for a, b in zip(seen["1"], seen["0"]):
            for ga, gb in zip(a, b):
                scale = max(float(gb.abs().max()), 1e-12)
                assert float((ga - gb).abs().max()) <= tol * scale
    finally:
        ops.set_matmul_precision(previous)


def test_captured_graph_survives_a_larger_scratch_buffer(gpu):
    """The recorded kernels hold the address of the shared scratch buffer. A later eager call that needs more scratch
    replaces the buffer; the graph must keep the one it recorded alive and still replay bit for bit."""
    import bench_configs as cfg
    from vegans import _hip
    from vegans.engine import graph, ops
    iters, B = 6, 8
    data = cfg.synthetic_images(B * iters, (3, 16, 16), seed=8)
    g = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 16, 1, 1, generator=g).to("cuda") for _ in range(iters)]
    finals = []
    for grow in (False, True):
        m = _model("WassersteinGANGP", width=16)
        batches = [m._to_device(data[i * B:(i + 1) * B]) for i in range(iters)]
        np.random.seed(77)
        for i in range(4):
            m.train_on_batch(batches[i], zs[i])
        entry = next(iter(graph._graphs[m].values()))
        assert entry.graph is not None and entry.scratch
        if grow:
            before = _hip.workspace(1, torch.device("cuda", 0))
            bigger = before.numel() * 2 + (64 << 20)
            # a weight gradient whose split-K slabs exceed the current scratch buffer
            big = _hip.workspace(bigger, torch.device("cuda", 0))
            assert big.data_ptr() != before.data_ptr() or big.numel() >= bigger
            ops.fill_(big.view(torch.float32), float("nan"))        # whoever reads the NEW buffer by mistake sees NaN
            assert any(b.data_ptr() == before.data_ptr() for b in entry.scratch)
            del before
            torch.cuda.empty_cache()
        for i in range(4, iters):
            m.train_on_batch(batches[i], zs[i])
        torch.cuda.synchronize()
        finals.append(_state(m))
    for k in finals[0]:
        if isinstance(finals[0][k], torch.Tensor):
            assert torch.equal(finals[0][k], finals[1][k]), k
