"""Helpers that take the ReLU / LeakyReLU slope decisions out of a gradient comparison.

Two float32 implementations of the same network do not produce gradients that agree to 1e-4, and neither agrees with a
float64 evaluation: a pre-activation within rounding of zero gets the other slope in one of them, and a single such unit
in a small early layer moves that layer's weight gradient by 1e-3 of its norm. `_MaskTap` records the slope masks the
ENGINE applied, `_float64_sub_step(..., masks=...)` evaluates the oracle's code in float64 with exactly those masks
forced: the exact derivative of the function the engine evaluated, against which its gradients can be held to 1e-4
(measured: 1e-7 ... 4e-6). `_torch_step_on` applies torch's own optimizer to the oracle's state and a given gradient.
"""
import copy
from unittest import mock

import torch


def _float64_sub_step(oracle, who, X, Z, alpha, masks=None):
    """Losses and parameter gradients of one sub-step, evaluated by the oracle's own code in float64 on copies of its
    current networks. With `masks` (what _MaskTap recorded from the engine's run of the same sub-step) every ReLU /
    LeakyReLU of the copies applies the ENGINE's slope mask instead of deciding by the sign of its own input."""
    twin = object.__new__(type(oracle))
    twin.__dict__.update(oracle.__dict__)
    twin.nets = {k: copy.deepcopy(v).double() for k, v in oracle.nets.items()}
    if masks is not None:
        for name, net in twin.nets.items():
            for i, m in enumerate(net):
                if isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU)):
                    net[i] = _Forced(masks[name].get(i, []), getattr(m, "negative_slope", 0.0))
    twin._set_mode(True)
    with mock.patch.object(torch.Tensor, "float", lambda self: self.double()):
        out = twin.losses(who, X.double(), Z.double(), None, alpha)
        grads = torch.autograd.grad(out[who], list(twin.nets[who].parameters()))
    if masks is not None:
        left = sum(len(q) for per_net in masks.values() for q in per_net.values())
        assert left == 0, "%d recorded masks were not consumed: the two sides ran different passes" % left
    return {k: float(v.detach()) for k, v in out.items()}, grads


def _torch_step_on(oracle, who, gradients):
    """The parameters torch's own optimizer (CPU) produces from the oracle's current parameters and optimizer state of
    network `who` and the given gradients; the oracle itself is left untouched. Returns name -> tensor."""
    net = copy.deepcopy(oracle.nets[who])
    opt_o = oracle.optimizers[who]
    opt = type(opt_o)(net.parameters(), **{k: v for k, v in opt_o.param_groups[0].items() if k != "params"})
    opt.load_state_dict(copy.deepcopy(opt_o.state_dict()))
    for p, g in zip(net.parameters(), gradients):
        p.grad = g.to(p.dtype).reshape(p.shape).clone()
    opt.step()
    return dict(net.named_parameters())


class _Forced(torch.nn.Module):
    """Stands in for a ReLU / LeakyReLU of the float64 copy: multiplies by a recorded slope mask (one per call, in call
    order)."""

    def __init__(self, queue, slope):
        super().__init__()
        self.queue, self.slope = queue, slope

    def forward(self, x):
        m = self.queue.pop(0)
        assert m.shape == x.shape, (m.shape, x.shape)
        m = m.to(x.dtype)
        return x * (m + (1.0 - m) * self.slope)


class _MaskTap:
    """Records the slope mask of every ReLU / LeakyReLU the engine applies, without changing which kernels run: the
    layer classes' forward methods are wrapped (forward hooks would make the Sequential container run BatchNorm and
    activation as separate operators). A fused BatchNorm + activation or convolution + activation call records under
    the activation module that follows it in its Sequential. masks[network name][module index] = [bool tensor per call]."""

    def __init__(self, model):
        from vegans.engine import modules
        self.modules = modules
        self.where = {}
        for name, wrapper in model.neural_nets.items():
            for i, m in enumerate(wrapper.network):
                self.where[id(m)] = (name, i)
        self.masks = {name: {} for name in model.neural_nets}
        self.saved = []

    def _record(self, module, y, fused):
        name, i = self.where[id(module)]
        mask = (y.detach() > 0).contiguous().cpu()
        self.masks[name].setdefault(i + 1 if fused else i, []).append(mask)

    def __enter__(self):
        from vegans import _hip
        tap = self

        def wrap_layer(cls):
            plain = cls.forward

            def forward(self, x, act=_hip.ACT_NONE, slope=0.0):
                y = plain(self, x, act, slope)
                if act != _hip.ACT_NONE:
                    tap._record(self, y, True)
                return y

            self.saved.append((cls, plain))
            cls.forward = forward

        def wrap_act(cls):
            plain = cls.forward

            def forward(self, x):
                y = plain(self, x)
                tap._record(self, y, False)
                return y

            self.saved.append((cls, plain))
            cls.forward = forward

        for cls in (self.modules.HipBatchNorm2d, self.modules.HipConv2d):
            wrap_layer(cls)
        for cls in (self.modules.HipReLU, self.modules.HipLeakyReLU):
            wrap_act(cls)
        return self

    def __exit__(self, *exc):
        for cls, plain in self.saved:
            cls.forward = plain
        return False
