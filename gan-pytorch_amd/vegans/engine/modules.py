"""Re-binding of the user's torch.nn layers to the HIP operators.

The reference hands the user's `nn.Module` straight to torch (vegans/utils/networks.py:36-47).  Here every leaf
layer of the network is switched, in place, to a subclass whose `forward` calls the engine operator while the
module object, its `Parameter`s, buffers and `state_dict` keys stay exactly the same.  Containers and arbitrary
user `forward()` code keep working because only leaves change.  A leaf layer type the engine has no kernel for
raises at construction time instead of silently running a stock kernel.
"""
import torch
from torch import nn

from . import ops
from .. import _hip


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class HipConv2d(nn.Conv2d):
    def forward(self, x, act=_hip.ACT_NONE, slope=0.0):
        return ops.conv2d(x, self.weight, self.bias, _pair(self.stride), _pair(self.padding), act, slope)


class HipConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, output_size=None):
        if output_size is not None:
            raise NotImplementedError("ConvTranspose2d(output_size=...) is not supported by the HIP engine")
        return ops.conv_transpose2d(x, self.weight, self.bias, _pair(self.stride), _pair(self.padding),
                                    _pair(self.output_padding))


class HipLinear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class _HipBatchNorm:
    def _hip_forward(self, x, act=_hip.ACT_NONE, slope=0.0):
        # same bookkeeping as torch.nn.modules.batchnorm._BatchNorm.forward; the batch counter is bumped by the
        # statistics kernel unless its value is needed on the host first (momentum=None: cumulative average)
        counter = None
        if self.momentum is None:
            factor = 0.0
        else:
            factor = self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if self.momentum is None:
                self.num_batches_tracked.add_(1)
                factor = 1.0 / float(self.num_batches_tracked)
            else:
                counter = self.num_batches_tracked
        use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        return ops.batch_norm(x, self.weight, self.bias, rm, rv, use_batch_stats, factor, self.eps, act, slope,
                              counter if use_batch_stats else None)


class HipBatchNorm2d(nn.BatchNorm2d, _HipBatchNorm):
    def forward(self, x, act=_hip.ACT_NONE, slope=0.0):
        if x.dim() != 4:
            raise ValueError("expected 4D input (got {}D input)".format(x.dim()))
        return self._hip_forward(x, act, slope)


class HipBatchNorm1d(nn.BatchNorm1d, _HipBatchNorm):
    def forward(self, x, act=_hip.ACT_NONE, slope=0.0):
        if x.dim() != 2:
            raise NotImplementedError("BatchNorm1d on %d-d input is not supported by the HIP engine" % x.dim())
        return self._hip_forward(x, act, slope)


class HipReLU(nn.ReLU):
    def forward(self, x):
        return ops.activation(x, _hip.ACT_RELU)


class HipLeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return ops.activation(x, _hip.ACT_LRELU, self.negative_slope)


class HipTanh(nn.Tanh):
    def forward(self, x):
        return ops.activation(x, _hip.ACT_TANH)


class HipSigmoid(nn.Sigmoid):
    def forward(self, x):
        return ops.activation(x, _hip.ACT_SIGMOID)


class HipMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        return ops.max_pool2d(x, _pair(self.kernel_size), _pair(self.stride if self.stride is not None else self.kernel_size),
                              _pair(self.padding))


class HipFlatten(nn.Flatten):
    def forward(self, x):
        if self.start_dim != 1 or self.end_dim not in (-1, x.dim() - 1):
            raise NotImplementedError("only Flatten(start_dim=1, end_dim=-1) is supported by the HIP engine")
        return ops.flatten(x)


def _has_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def _fused_act(norm, act):
    """(code, slope) when `act` directly behind `norm` (a BatchNorm, or a convolution: ops.conv2d decides per geometry
    whether its epilogue takes the activation) can run inside that layer's kernels, else None."""
    if not isinstance(norm, (HipBatchNorm2d, HipBatchNorm1d, HipConv2d)) or _has_hooks(norm) or _has_hooks(act):
        return None
    if type(act) is HipReLU:
        return _hip.ACT_RELU, 0.0
    if type(act) is HipLeakyReLU:
        # the fused backward reads the slope mask off the sign of the OUTPUT, which needs a non-negative slope
        return (_hip.ACT_LRELU, act.negative_slope) if act.negative_slope >= 0 else None
    return None


def _wants_stats(conv, norm):
    """A training-mode BatchNorm2d directly behind a (transposed) convolution, neither with hooks."""
    return (isinstance(conv, (HipConv2d, HipConvTranspose2d)) and type(norm) is HipBatchNorm2d and norm.training
            and not _has_hooks(conv) and not _has_hooks(norm))


class HipSequential(nn.Sequential):
    """nn.Sequential that runs a BatchNorm (or, where the library has a fused kernel, a convolution) followed by
    ReLU / LeakyReLU as one fused operator: the
    pre-activation tensor is never written, and the backward kernels apply the slope mask on the fly. The pairing
    is decided per call from the children actually present, so edits to the container keep working."""

    def forward(self, x):
        mods = list(self._modules.values())
        i, n = 0, len(mods)
        handed = None        # (act, slope, token): x is a fused activation's output that only mods[i] will see
        while i < n:
            m = mods[i]
            if handed is not None:
                ops.act_grad.private_input(x, *handed)
                handed = None
            fused = _fused_act(m, mods[i + 1]) if i + 1 < n else None
            if i + 1 < n and _wants_stats(m, mods[i + 1]):
                # the norm behind this convolution takes its batch statistics from the convolution's epilogue
                ops.norm_stats.request()
                x = m(x)
                ops.norm_stats.cancel()
                i += 1
            elif fused is not None:
                # conv + activation in front of another convolution: that one's backward can carry its input gradient
                # through the activation's slope mask in its own kernel
                pair = (type(m) is HipConv2d and i + 2 < n and type(mods[i + 2]) is HipConv2d
                        and not _has_hooks(mods[i + 2]) and torch.is_grad_enabled())
                if pair:
                    ops.act_grad.announce()
                x = m(x, fused[0], fused[1])
                token = ops.act_grad.withdraw()
                if token is not None:
                    handed = (fused[0], fused[1], token)
                i += 2
            else:
                if (type(m) in (HipReLU, HipLeakyReLU) and i + 1 < n and type(mods[i + 1]) is HipConv2d and x.dim() == 4
                        and not _has_hooks(m) and not _has_hooks(mods[i + 1]) and torch.is_grad_enabled()):
                    ops.act_grad.announce()
                x = m(x)
                token = ops.act_grad.withdraw()
                if token is not None:
                    handed = ((_hip.ACT_RELU, 0.0) if type(m) is HipReLU else (_hip.ACT_LRELU, m.negative_slope)) + (token,)
                i += 1
            ops.act_grad.cancel()
        return x


_SWAP = {
    nn.Sequential: HipSequential,
    nn.Conv2d: HipConv2d,
    nn.ConvTranspose2d: HipConvTranspose2d,
    nn.Linear: HipLinear,
    nn.BatchNorm2d: HipBatchNorm2d,
    nn.BatchNorm1d: HipBatchNorm1d,
    nn.ReLU: HipReLU,
    nn.LeakyReLU: HipLeakyReLU,
    nn.Tanh: HipTanh,
    nn.Sigmoid: HipSigmoid,
    nn.Flatten: HipFlatten,
    nn.MaxPool2d: HipMaxPool2d,
}
_STOCK = {v: k for k, v in _SWAP.items()}

# leaves that launch no kernel (pure metadata / identity)
_PASS_THROUGH = (nn.Identity, nn.Unflatten)


def _validate(m):
    if isinstance(m, nn.MaxPool2d):
        if _pair(m.dilation) != (1, 1) or m.ceil_mode or m.return_indices:
            raise NotImplementedError("MaxPool2d with dilation / ceil_mode / return_indices is not supported by the HIP engine")
        return
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        if _pair(m.dilation) != (1, 1) or m.groups != 1:
            raise NotImplementedError("%s with dilation/groups is not supported by the HIP engine" % type(m).__name__)
        if isinstance(m.padding, str) or m.padding_mode != "zeros":
            raise NotImplementedError("%s padding=%r/%s is not supported by the HIP engine" %
                                      (type(m).__name__, m.padding, m.padding_mode))


def bind(network):
    """Switch every supported leaf of `network` to its HIP subclass (idempotent). Returns the network."""
    from ..utils.layers import LayerReshape, LayerPrintSize
    for m in network.modules():
        t = type(m)
        if t in _STOCK:
            continue
        if t in _SWAP:
            _validate(m)
            m.__class__ = _SWAP[t]
            continue
        if isinstance(m, _PASS_THROUGH) or isinstance(m, (LayerReshape, LayerPrintSize)):
            continue
        if len(list(m.children())) == 0 and len(list(m.parameters(recurse=False))) + len(list(m.buffers(recurse=False))) >= 0:
            if t.__module__.startswith("torch.nn.modules"):
                raise NotImplementedError(
                    "layer %s has no MI355X kernel in this engine (supported: %s)" %
                    (t.__name__, ", ".join(sorted(k.__name__ for k in _SWAP))))
    return network


def unbind(network):
    """Give the user their stock torch classes back (parameters are untouched)."""
    for m in network.modules():
        if type(m) in _STOCK:
            m.__class__ = _STOCK[type(m)]
    return network


def prepare_parameters(network):
    """Store 4-d weights as [Co][kh][kw][Ci] (channels_last) so kernels read them without a per-call re-layout.
    Logical values, shapes and state_dict contents are unchanged."""
    for p in network.parameters():
        if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
            p.data = ops.to_nhwc(p.data)
            if p.grad is not None:
                p.grad = None
    return network
