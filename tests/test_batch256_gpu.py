"""Parity at the size the benchmark times: WGAN-GP on the full-width DCGAN-64 (ngf = ndf = 64, BatchNorm critic,
3x64x64, z = 100, RMSprop 1e-4 / 5e-5) at BATCH 256.

At this batch the kernels take other plans than at the batches the rest of the suite uses: the weight gradient's
split-K count and chunk length, up to 2048 per-tile statistics records merged per BatchNorm layer, 65536- and
262144-row grids, the 32 MiB slab set. What is compared here:

1. two whole iterations, teacher-forced before every sub-step, against the CPU oracle (oracle/restated.py): losses at
   1e-4 of their own value, every parameter tensor's gradient (l2) before the optimizer touches it, with the same
   sub-step evaluated by the oracle's code in float64 as the yardstick that tells whose rounding a difference is; in
   the f32 mode and in the split mode (f32-grade products from bf16 terms);
2. a replayed HIP graph of the iteration against the same iteration run kernel by kernel: bit identity of parameters,
   buffers, optimizer state, gradients and losses.
The operator-level cases at 256 images (the fat layers, the stem, the statistics / backward-sum / slope-mask
epilogues) are in test_ops_gpu.py.
"""
import numpy as np
import pytest
import torch

import golden_util as gu  # noqa: F401  (path set-up of the test package)
from derivative_check import _float64_sub_step, _Forced, _MaskTap, _torch_step_on  # noqa: F401
from test_models_gpu import DRIFT, GRAD_L2_EVENT, REL, _build_model, _oracle_and_model, _sync_from_oracle

pytestmark = pytest.mark.gpu

B = 256
OKW = {"Generator": {"lr": 1e-4}, "Adversariat": {"lr": 5e-5}}


def _nets():
    import bench_configs as cfg
    return (cfg.dcgan_init(cfg.dcgan_generator(100, 64, 3, 64), 1), cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64), 2))


@pytest.mark.parametrize("mode", ["fp32", "fp32_split"])
def test_wgangp_batch256_teacher_forced_vs_oracle(gpu, capsys, mode):
    """Two iterations at batch 256, synchronised with the oracle before every sub-step.

    Losses: each within 1e-4 of ITS OWN value of the fp32 oracle's (floor 1e-2; the critic's total, a sum of cancelling
    terms, at 1e-4 of its largest term).

    Gradients, per parameter tensor, before the optimizer touches them (l2 norm of the difference over the l2 norm of
    the gradient). Two fp32 implementations do not agree to 1e-4 here, and neither agrees with float64: at this batch a
    pass evaluates 6e7 ReLU / LeakyReLU units, a handful of which have a pre-activation within rounding of zero and get
    the other slope in one implementation; one such unit in the generator's first layer (2e6 units) moves that layer's
    weight gradient by 1e-3 of its norm. Measured: engine and fp32 oracle 2e-3 apart, each up to 3e-3 from float64, in
    some sub-steps 3e-5. The slope decisions are therefore taken out of the comparison: the engine's own masks are
    recorded (_MaskTap) and the float64 evaluation is repeated with those masks forced (_float64_sub_step). Against
    THAT gradient, which is the exact derivative of the function the engine evaluated, every tensor must be within
    1e-4; a term that is wrong by a percent, a lost digit in a long sum or a wrong tile plan at this batch shows up in
    full. The distances from the fp32 oracle and from the unforced float64 gradient are printed and bounded by
    GRAD_L2_EVENT.

    Parameters after the RMSprop step: compared with torch's own RMSprop (CPU) applied to the oracle's state and the
    ENGINE's gradient, element by element at 1e-6 of the tensor's largest magnitude (floor 1e-2): gradient and update
    are so each held to their own exact counterpart. Against the oracle's parameters (its own gradient, i.e. other slope
    decisions) the share of elements beyond 1e-4 of the tensor's scale is printed, and DRIFT bounds every element.
    BatchNorm buffers of both networks after every sub-step: 1e-4 of the tensor's largest magnitude."""
    import bench_configs as cfg
    from vegans.engine import ops
    previous = ops.set_matmul_precision(mode)
    try:
        oracle, model = _oracle_and_model("wgangp", "WassersteinGANGP", _nets, (3, 64, 64), (100, 1, 1), OKW)
        oracle._set_mode(True)
        model.use_step_graph = False
        model._create_steps(None)
        model.train()
        zgen = torch.Generator().manual_seed(10)
        floor = 1e-2
        problems, worst_loss, worst = [], {}, {}
        shares, worst_step = {}, 0.0
        for it in range(2):
            Xn = cfg.synthetic_images(B, (3, 64, 64), seed=90 + it)
            X, Xd = torch.from_numpy(Xn), model._to_device(Xn)
            Z = torch.randn(B, 100, 1, 1, generator=zgen)
            A = np.random.default_rng(2000 + it).random((B, 1, 1, 1))
            model._draw_alpha = lambda n, A=A: torch.from_numpy(A.astype(np.float32)).to("cuda")
            for who in ("Generator", "Adversariat"):
                alpha = A if who == "Adversariat" else None
                _sync_from_oracle(oracle, model)
                truth, g64 = _float64_sub_step(oracle, who, X, Z, alpha)
                # the engine's sub-step, stopped between backward and the optimizer for the gradients
                model._losses = {}
                with _MaskTap(model) as tap, model._training_only(who):
                    model.calculate_losses(X_batch=Xd, Z_batch=Z.to("cuda"), who=who)
                    model._zero_grad(who=who)
                    model._backward(who=who)
                got = {k: float(v.detach()) for k, v in model._losses.items()}
                gm = [p.grad.detach().cpu().double() for p in model.neural_nets[who].network.parameters()]
                _, g64m = _float64_sub_step(oracle, who, X, Z, alpha, masks=tap.masks)
                shadow = _torch_step_on(oracle, who, gm)
                want = oracle.sub_step(who, X, Z.clone(), None, alpha)
                g32 = [p.grad.detach().clone() for p in oracle.nets[who].parameters()]
                model._step(who=who)
                tag = "iteration %d %s" % (it, who)
                for k, t in truth.items():
                    if k == "RealFakeRatio":
                        continue
                    scale = max(abs(t), floor)
                    if k == "Adversariat":
                        scale = max(scale, 0.5 * abs(truth["Adversariat_fake"]), 0.5 * abs(truth["Adversariat_real"]),
                                    10 * abs(truth["Adversariat_grad"]))
                    direct = abs(got[k] - want[k]) / scale
                    worst_loss[k] = max(worst_loss.get(k, 0.0), direct)
                    if direct > REL:
                        problems.append("%s loss %s: engine %.8g, fp32 oracle %.8g, float64 %.8g" % (tag, k, got[k], want[k], t))
                for (k, _), a, b, c, cm in zip(oracle.nets[who].named_parameters(), gm, g32, g64, g64m):
                    norm = max(float(c.norm()), 1e-30)
                    direct = float((a - b.double()).norm()) / norm
                    e64, o64 = float((a - c).norm()) / norm, float((b.double() - c).norm()) / norm
                    forced = float((a - cm).norm()) / max(float(cm.norm()), 1e-30)
                    key = who + "." + k
                    worst[key] = tuple(max(x, y) for x, y in zip(worst.get(key, (0.0,) * 4), (forced, direct, e64, o64)))
                    if forced > REL:
                        problems.append("%s gradient of %s: %.2e (l2) from the float64 gradient under the engine's own "
                                        "masks" % (tag, key, forced))
                    if max(direct, e64) > GRAD_L2_EVENT:
                        problems.append("%s gradient of %s: %.2e (l2) from the fp32 oracle's, %.2e from float64 (oracle "
                                        "%.2e)" % (tag, key, direct, e64, o64))
                for name in ("Generator", "Adversariat"):
                    state = model.neural_nets[name].network.state_dict()
                    params = set(k for k, _ in oracle.nets[name].named_parameters())
                    for k, v in oracle.nets[name].state_dict().items():
                        if k.endswith("num_batches_tracked"):
                            assert int(state[k]) == int(v), (tag, name, k)
                            continue
                        a, b = state[k].detach().cpu().double(), v.detach().double()
                        scale = max(float(b.abs().max()), 1e-2)
                        d = (a - b).abs()
                        key = name + "." + k
                        if k not in params:
                            if float(d.max()) > REL * scale:
                                problems.append("%s buffer %s: off by %.3e of its scale" % (tag, key, float(d.max()) / scale))
                            continue
                        if name != who:
                            assert float(d.max()) == 0.0, (tag, key)       # not this sub-step's network: untouched
                            continue
                        shares[key] = shares.get(key, 0.0) + float((d > REL * scale).sum()) / b.numel() / 2
                        if float(d.max()) > DRIFT:
                            problems.append("%s %s: an element is off by %.3e" % (tag, key, float(d.max())))
                        e = shadow[k].detach().double()
                        off = float((a - e).abs().max()) / max(float(e.abs().max()), 1e-2)
                        worst_step = max(worst_step, off)
                        if off > 1e-6:
                            problems.append("%s %s: %.2e of its scale from torch's RMSprop step on the engine's gradient" %
                                            (tag, key, off))
    finally:
        ops.set_matmul_precision(previous)
    with capsys.disabled():
        print("\n%s, DCGAN-64 WGAN-GP full width at batch %d, 2 teacher-forced iterations:" % (mode, B))
        print("   worst loss distance from the fp32 oracle (of the value): " +
              ", ".join("%s %.1e" % kv for kv in sorted(worst_loss.items())))
        print("   gradients (l2, worst over the sub-steps): from float64 under the engine's masks | from the fp32 oracle "
              "| from float64 | the fp32 oracle's own distance from float64")
        for key in sorted(worst):
            print("      %-28s %.1e | %.1e | %.1e | %.1e" % ((key,) + worst[key]))
        k = max(shares, key=shares.get)
        print("   parameters after the step: at most %.1e of a tensor's scale from torch's RMSprop on the engine's gradient; "
              "against the oracle's parameters, largest average share of elements beyond 1e-4 of the scale: %s %.1e"
              % (worst_step, k, shares[k]))
    assert not problems, "\n".join(problems[:40])


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


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_replayed_iteration_matches_eager_at_batch256(gpu, mode):
    """The iteration the benchmark replays (full-width DCGAN-64 WGAN-GP, batch 256) against the same iteration run
    kernel by kernel: five iterations each way (two eager calls, then the capture and replays); parameters, BatchNorm
    buffers, RMSprop state, the last gradients and every loss must be the same bit patterns, and the host random stream
    must have advanced identically."""
    import bench_configs as cfg
    from vegans.engine import graph, ops
    iters = 5
    data = [cfg.synthetic_images(B, (3, 64, 64), seed=70 + i) for i in range(iters)]
    zgen = torch.Generator().manual_seed(12)
    zs = [torch.randn(B, 100, 1, 1, generator=zgen) for _ in range(iters)]
    case = dict(algo="wgangp", cls="WassersteinGANGP", x_dim=(3, 64, 64), z_dim=(100, 1, 1), optim_kwargs=OKW,
                fixed_noise_size=4)
    previous = ops.set_matmul_precision(mode)
    try:
        results = []
        for use_graph in (False, True):
            torch.manual_seed(4)
            np.random.seed(4)
            m = _build_model(case, _nets() + (None,))
            m._create_steps(None)
            m.train()
            m.use_step_graph = use_graph
            np.random.seed(77)
            losses = []
            for Xn, Z in zip(data, zs):
                m.train_on_batch(m._to_device(Xn), Z.to("cuda"))
                losses.append({k: v.detach().clone() for k, v in m._losses.items()})
            torch.cuda.synchronize()
            entries = list(graph._graphs.get(m, {}).values())
            if use_graph:
                assert len(entries) == 1 and entries[0].graph is not None and not entries[0].failed
                assert entries[0].eager_calls == 2
            else:
                assert entries == []
            results.append((losses, np.random.random(), _state(m)))
            graph.reset(m)
            del m
            torch.cuda.empty_cache()
    finally:
        ops.set_matmul_precision(previous)
    (l0, r0, s0), (l1, r1, s1) = results
    assert r0 == r1, "host random stream advanced differently under replay"
    for i, (a, b) in enumerate(zip(l0, l1)):
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]) or (torch.isnan(a[k]) and torch.isnan(b[k])), "iteration %d loss %s" % (i, k)
    assert set(s0) == set(s1)
    for k in s0:
        if isinstance(s0[k], torch.Tensor):
            assert torch.equal(s0[k], s1[k]), k
        else:
            assert s0[k] == s1[k], k


def test_fit_runs_are_bit_identical_at_batch256(gpu):
    """Three `fit` calls from the same seeds on fresh models (full-width DCGAN-64 WGAN-GP, batch 256, the critic's
    passes on three streams): kernel by kernel, kernel by kernel again, and replayed from a HIP graph. Parameters and
    BatchNorm buffers must be the same bits. At this size BatchNorm backward sums come from convolution epilogues, handed
    from one backward node to another; which norm gets them must not depend on how autograd interleaves the three
    chains (the interleaving differs between the first and later backward passes of a process)."""
    import contextlib
    import io

    import bench_configs as cfg
    from vegans.engine import graph
    X = np.concatenate([cfg.synthetic_images(B, (3, 64, 64), seed=30 + i) for i in range(5)], axis=0)
    case = dict(algo="wgangp", cls="WassersteinGANGP", x_dim=(3, 64, 64), z_dim=(100, 1, 1), optim_kwargs=OKW,
                fixed_noise_size=4)
    finals = []
    for use_graph in (False, False, True):
        torch.manual_seed(0)
        np.random.seed(0)
        m = _build_model(case, _nets() + (None,))
        m.use_step_graph = use_graph
        with contextlib.redirect_stdout(io.StringIO()):
            m.fit(X, epochs=1, batch_size=B, print_every=None, save_losses_every=None, enable_tensorboard=False)
        torch.cuda.synchronize()
        assert graph.replays(m) == (3 if use_graph else 0)
        finals.append({name + "." + k: v.detach().clone() for name, net in m.neural_nets.items()
                       for k, v in net.network.state_dict().items()})
        graph.reset(m)
        del m
        torch.cuda.empty_cache()
    for other in finals[1:]:
        for k, v in finals[0].items():
            assert torch.equal(v, other[k]), k
