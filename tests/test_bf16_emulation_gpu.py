"""The engine's two bf16 modes against a CPU restatement of WHERE they round (oracle/bf16_emulation.py), teacher-forced
at the full DCGAN-64 width.

The comparison with the fp32 reference (tests/test_models_gpu.py: losses within 5e-2, gradients within 8e-2 / 2.5e-1)
states how far a bf16 run is from the fp32 one; it is an envelope a kernel that is wrong by a few percent would fit in.
Here the oracle's own float32 networks are evaluated with bf16 roundings at the points the data path has them (stored
feature maps and their gradients; operands of the contractions) and with the ENGINE's ReLU / LeakyReLU slope decisions
forced (test_batch256_gpu._MaskTap), so what is left between the two is summation order, the few places where the
engine fuses two roundings into one, and what those seed: a value whose f32 sum lands within summation noise of a bf16
rounding boundary is stored one bf16 step (2^-8 of its size) apart in the two implementations. That happens to about
5e-5 of the elements of a layer (measured: 4e-5 relative l2 after the first bf16 convolution on identical inputs,
tools/bf16_layer_probe.py), and every following convolution + BatchNorm pair multiplies the difference by 3 to 5, so the
critic's scores differ by 1e-3 (l2) after five layers and gradients by a few 1e-3. The tolerances below sit at about
twice what was measured: losses within 1e-2 of their value (measured 5e-3), gradients within 2e-2 in the l2 norm per
parameter tensor (measured 4e-4 ... 9e-3), where the same run is 7e-2 ... 1.7e-1 from the fp32 oracle. A term that is
wrong by a few percent in any kernel of the bf16 path fails here.
"""
import copy
import ctypes

import numpy as np
import pytest
import torch
from torch import nn

import golden_util as gu  # noqa: F401
from derivative_check import _torch_step_on
from test_batch256_gpu import OKW, _Forced, _MaskTap, _nets
from test_models_gpu import _oracle_and_model, _sync_from_oracle

pytestmark = pytest.mark.gpu

LOSS_TOL = {"bf16": 1e-2, "bf16_compute": 1e-2}     # of the loss's own value (floor 1e-1; the critic total at its largest term)
GRAD_TOL = {"bf16": 2e-2, "bf16_compute": 2e-2}     # l2, per parameter tensor
STEP_TOL = 1e-6     # parameters after the optimizer step, of a tensor's scale, from torch's step on the engine's gradient


def _emulated_sub_step(oracle, who, X, Z, alpha, mode, masks):
    """Losses and parameter gradients of one sub-step from the oracle's code on copies of its float32 networks, with
    the roundings of engine mode `mode` inserted and the engine's slope masks forced."""
    from oracle.bf16_emulation import emulate
    twin = object.__new__(type(oracle))
    twin.__dict__.update(oracle.__dict__)
    twin.nets = {}
    for name, net in oracle.nets.items():
        net = copy.deepcopy(net)
        for i, m in enumerate(net):
            if isinstance(m, (nn.ReLU, nn.LeakyReLU)):
                net[i] = _Forced(masks[name].get(i, []), getattr(m, "negative_slope", 0.0))
        twin.nets[name] = emulate(net, mode)
    twin._set_mode(True)
    out = twin.losses(who, X, Z, None, alpha)
    grads = torch.autograd.grad(out[who], list(twin.nets[who].parameters()))
    left = sum(len(q) for per_net in masks.values() for q in per_net.values())
    assert left == 0, "%d recorded masks were not consumed" % left
    names = [k for k, _ in twin.nets[who].named_parameters()]
    return {k: float(v.detach()) for k, v in out.items()}, grads, names


def test_weight_rounding_rule_matches_the_library(gpu):
    """oracle/bf16_emulation.rounds_weights against the library's own dispatch (vg_conv2d_h_applies) for every
    convolution of the DCGAN-64 and DCGAN-128 networks and each of the three contractions."""
    import bench_configs as cfg
    from oracle.bf16_emulation import rounds_weights
    from vegans import _hip
    from vegans.engine import ops
    lib = _hip.load()
    for size in (64, 128):
        G, D = cfg.dcgan_generator(100, 64, 3, size), cfg.dcgan_critic(64, 3, size)
        for net, x in ((G, torch.zeros(4, 100, 1, 1)), (D, torch.zeros(4, 3, size, size))):
            shapes = {}
            hooks = [m.register_forward_hook(lambda m, i, o, shapes=shapes: shapes.__setitem__(m, (i[0].shape, o.shape)))
                     for m in net if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
            net(x)
            for h in hooks:
                h.remove()
            first = True
            for m, (xin, yout) in shapes.items():
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                if isinstance(m, nn.Conv2d):
                    g = ops.Geom(xin[0], xin[2], xin[3], m.in_channels, m.out_channels, k, k, s, s, p, p)
                    served = [lib.vg_conv2d_h_applies(mode, ctypes.byref(g.c)) for mode in (0, 1, 2)]
                    needed = [True, m.in_channels % 8 == 0, True]        # no input gradient into an image
                else:
                    g = ops.Geom(xin[0], yout[2], yout[3], m.out_channels, m.in_channels, k, k, s, s, p, p)
                    served = [lib.vg_conv2d_h_applies(mode, ctypes.byref(g.c)) for mode in (1, 0, 2)]
                    needed = [True, not first, True]
                want = rounds_weights(m) and not (first and isinstance(m, nn.ConvTranspose2d))
                for mode, (got, need) in enumerate(zip(served, needed)):
                    if need:
                        assert bool(got) == want, (size, type(m).__name__, m.in_channels, m.out_channels, mode, got, want)
                first = False


@pytest.mark.parametrize("mode", ["bf16", "bf16_compute"])
def test_bf16_modes_teacher_forced_vs_their_cpu_restatement(gpu, capsys, mode):
    """WGAN-GP, DCGAN-64 at ngf = ndf = 64, batch 64, RMSprop, three iterations; before every sub-step the engine takes
    over the fp32 oracle's parameters, buffers and optimizer state. Every loss within LOSS_TOL of its own value and every
    parameter tensor's gradient within GRAD_TOL (l2) of the restatement's; the distances from the plain fp32 oracle (the
    mode's reported tolerance) are printed beside them."""
    import bench_configs as cfg
    from vegans.engine import ops
    B, iters = 64, 3
    previous = ops.set_matmul_precision(mode)
    try:
        oracle, model = _oracle_and_model("wgangp", "WassersteinGANGP", _nets, (3, 64, 64), (100, 1, 1), OKW)
        oracle._set_mode(True)
        model.use_step_graph = False
        model._create_steps(None)
        model.train()
        zgen = torch.Generator().manual_seed(10)
        problems, worst_loss, worst_grad, worst_step = [], {}, {}, 0.0
        for it in range(iters):
            Xn = cfg.synthetic_images(B, (3, 64, 64), seed=40 + it)
            X, Xd = torch.from_numpy(Xn), model._to_device(Xn)
            Z = torch.randn(B, 100, 1, 1, generator=zgen)
            A = np.random.default_rng(1000 + it).random((B, 1, 1, 1))
            model._draw_alpha = lambda n, A=A: torch.from_numpy(A.astype(np.float32)).to("cuda")
            for who in ("Generator", "Adversariat"):
                alpha = A if who == "Adversariat" else None
                _sync_from_oracle(oracle, model)
                model._losses = {}
                with _MaskTap(model) as tap, model._training_only(who):
                    model.calculate_losses(X_batch=Xd, Z_batch=Z.to("cuda"), who=who)
                    model._zero_grad(who=who)
                    model._backward(who=who)
                got = {k: float(v.detach()) for k, v in model._losses.items()}
                gm = [p.grad.detach().cpu().double() for p in model.neural_nets[who].network.parameters()]
                want, ge, names = _emulated_sub_step(oracle, who, X, Z, alpha, mode, tap.masks)
                shadow = _torch_step_on(oracle, who, gm)
                plain = oracle.sub_step(who, X, Z.clone(), None, alpha)
                g32 = [p.grad.detach().double() for p in oracle.nets[who].parameters()]
                model._step(who=who)
                tag = "iteration %d %s" % (it, who)
                for k, v in want.items():
                    if k == "RealFakeRatio":
                        continue
                    scale = max(abs(v), 1e-1)
                    if k == "Adversariat":
                        scale = max(scale, 0.5 * abs(want["Adversariat_fake"]), 0.5 * abs(want["Adversariat_real"]),
                                    10 * abs(want["Adversariat_grad"]))
                    e, f = abs(got[k] - v) / scale, abs(got[k] - plain[k]) / scale
                    worst_loss[k] = tuple(max(a, b) for a, b in zip(worst_loss.get(k, (0.0, 0.0)), (e, f)))
                    if e > LOSS_TOL[mode]:
                        problems.append("%s loss %s: engine %.7g, restatement %.7g (fp32 oracle %.7g)" % (tag, k, got[k], v, plain[k]))
                for k, a, b, c in zip(names, gm, ge, g32):
                    norm = max(float(b.double().norm()), 1e-30)
                    e = float((a - b.double()).norm()) / norm
                    f = float((a - c).norm()) / max(float(c.norm()), 1e-30)
                    key = who + "." + k
                    worst_grad[key] = tuple(max(x, y) for x, y in zip(worst_grad.get(key, (0.0, 0.0)), (e, f)))
                    if e > GRAD_TOL[mode]:
                        problems.append("%s gradient of %s: %.2e (l2) from the restatement's" % (tag, key, e))
                # the parameters are f32 masters in every mode: after the step they are what torch's own RMSprop makes
                # of the oracle's state and the ENGINE's gradient
                state = model.neural_nets[who].network.state_dict()
                for k, e in shadow.items():
                    a, e = state[k].detach().cpu().double(), e.detach().double()
                    off = float((a - e).abs().max()) / max(float(e.abs().max()), 1e-2)
                    worst_step = max(worst_step, off)
                    if off > STEP_TOL:
                        problems.append("%s %s.%s: %.2e of its scale from torch's RMSprop step on the engine's gradient"
                                        % (tag, who, k, off))
    finally:
        ops.set_matmul_precision(previous)
    with capsys.disabled():
        print("\n%s mode, DCGAN-64 WGAN-GP full width, batch %d, %d teacher-forced iterations" % (mode, B, iters))
        print("   losses, worst distance (of the value): from the CPU restatement of the mode | from the fp32 oracle")
        for k in sorted(worst_loss):
            print("      %-28s %.1e | %.1e" % ((k,) + worst_loss[k]))
        print("   gradients (l2), worst over the sub-steps: from the restatement | from the fp32 oracle")
        for k in sorted(worst_grad):
            print("      %-28s %.1e | %.1e" % ((k,) + worst_grad[k]))
        print("   parameters after the step: at most %.1e of a tensor's scale from torch's RMSprop on the engine's gradient"
              % worst_step)
    assert not problems, "\n".join(problems[:40])
