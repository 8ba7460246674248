"""Fused optimizer steps on the HIP kernels, driven from the user's own torch.optim object.

The reference calls `self.optimizers[who].step()` (AbstractGenerativeModel.py:374-378) and, for WGAN, clamps the
critic afterwards (WassersteinGAN.py:74-81).  Here the hyper-parameters and the state tensors stay inside the
torch optimizer (`optimizer.state_dict()` keeps working and can be loaded into stock torch), but the arithmetic
runs in one kernel per parameter tensor: gradient pre-scale (data-parallel mean), moment updates, parameter
update and the optional clamp in a single pass over memory.
"""
import math
from ctypes import byref, c_float, c_size_t, c_void_p

import numpy as np
import torch

from .. import _hip
from . import graph
from .._hip import AdamCfg, RmspropCfg, SgdCfg, _ptr, _stream, call

_NO_CLAMP = (1.0, -1.0)  # lo > hi disables the clamp inside the kernels


def _zeros_like(p):
    t = torch.empty_like(p, memory_format=torch.preserve_format)
    call("vg_fill", _ptr(t), c_float(0.0), c_size_t(t.numel()), _stream())
    return t


def _as_float(v):
    return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


def _check_flags(group, kind):
    for flag in ("capturable", "differentiable"):
        if group.get(flag, False):
            raise NotImplementedError("%s(%s=True) is not supported by the HIP engine" % (kind, flag))


def _grad_mem(p):
    g = p.grad
    if g.stride() != p.stride():
        # elementwise kernels walk raw memory: gradient and parameter must share one memory order
        if p.dim() == 4:
            from . import ops
            g = ops.to_nhwc(g) if p.is_contiguous(memory_format=torch.channels_last) else ops.to_nchw(g)
        if g.stride() != p.stride():
            raise _hip.HipError("gradient strides %s differ from parameter strides %s" % (g.stride(), p.stride()))
    return g


def fused_step(optimizer, grad_scale=1.0, clamp=None):
    """One optimizer step for every parameter that has a gradient. `clamp=(lo, hi)` clamps after the update."""
    from . import ops
    # the kernels below rewrite parameters in place, behind torch's version counters
    ops.plane_cache.drop([p for group in optimizer.param_groups for p in group["params"]])
    lo, hi = clamp if clamp is not None else _NO_CLAMP
    kind = type(optimizer)
    if kind is torch.optim.RMSprop:
        _rmsprop(optimizer, grad_scale, lo, hi)
    elif kind in (torch.optim.Adam, torch.optim.AdamW):
        _adam(optimizer, grad_scale, lo, hi, decoupled=kind is torch.optim.AdamW)
    elif kind is torch.optim.SGD:
        _sgd(optimizer, grad_scale, lo, hi)
    else:
        raise NotImplementedError(
            "optimizer %s has no fused MI355X kernel (supported: RMSprop, Adam, AdamW, SGD)" % kind.__name__)
    if clamp is not None:
        # the reference clamps EVERY critic parameter after the step (WassersteinGAN.py:78-79); the fused kernels only
        # visit parameters that have a gradient
        clamp_([p for group in optimizer.param_groups for p in group["params"] if p.grad is None], lo, hi)


def _pointer_array(tensors):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _rmsprop(opt, grad_scale, lo, hi):
    for group in opt.param_groups:
        _check_flags(group, "RMSprop")
        sign = -1.0 if group.get("maximize", False) else 1.0
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            continue
        for p in params:
            state = opt.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0)
                state["square_avg"] = _zeros_like(p)
                if group["momentum"] > 0:
                    state["momentum_buffer"] = _zeros_like(p)
                if group["centered"]:
                    state["grad_avg"] = _zeros_like(p)
        states = [opt.state[p] for p in params]
        graph.host_effect(lambda states=states: [st["step"].add_(1) for st in states])
        cfg = RmspropCfg(_as_float(group["lr"]), group["alpha"], group["eps"], group["weight_decay"],
                         group["momentum"], 1 if group["centered"] else 0, sign * grad_scale, lo, hi)
        grads = [_grad_mem(p) for p in params]
        for t in grads:
            _ptr(t)      # device / dtype check
        sizes = (c_size_t * len(params))(*[p.numel() for p in params])
        # one launch for the whole group
        call("vg_rmsprop_step_multi", len(params), _pointer_array([p.data for p in params]), _pointer_array(grads),
             _pointer_array([st["square_avg"] for st in states]),
             _pointer_array([st.get("momentum_buffer") for st in states]),
             _pointer_array([st.get("grad_avg") for st in states]), sizes, byref(cfg), _stream())


def _adam(opt, grad_scale, lo, hi, decoupled):
    for group in opt.param_groups:
        _check_flags(group, "Adam")
        sign = -1.0 if group.get("maximize", False) else 1.0
        beta1, beta2 = group["betas"]
        lr = _as_float(group["lr"])
        params = [p for p in group["params"] if p.grad is not None]
        for p in params:
            state = opt.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0)
                state["exp_avg"] = _zeros_like(p)
                state["exp_avg_sq"] = _zeros_like(p)
                if group["amsgrad"]:
                    state["max_exp_avg_sq"] = _zeros_like(p)
        if not params:
            continue
        states = [opt.state[p] for p in params]

        def scalars_of(step):
            # host-side doubles, exactly as torch's single-tensor Adam forms them
            return lr / (1 - beta1 ** step), math.sqrt(1 - beta2 ** step)

        shared = None
        if graph.capturing():
            # a replayed graph cannot carry step-dependent values: the kernels read them from two device floats that
            # every replay refreshes, at which moment the step counters advance too
            if len({float(st["step"]) for st in states}) != 1:
                raise _hip.HipError("Adam parameters of one group are at different steps")

            def advance(states=states):
                for st in states:
                    st["step"] += 1
                return np.asarray(scalars_of(float(states[0]["step"])), dtype=np.float64)

            shared = graph.host_input((2,), advance, params[0].device)
        def config(step_size, bc2_sqrt):
            return AdamCfg(lr, beta1, beta2, group["eps"], group["weight_decay"], step_size, bc2_sqrt,
                           1 if group["amsgrad"] else 0, 1 if decoupled else 0, sign * grad_scale, lo, hi,
                           shared.data_ptr() if shared is not None else None)

        if shared is None:
            for st in states:
                st["step"] += 1
        steps = {float(st["step"]) for st in states}
        if shared is not None or len(steps) == 1:
            # all tensors of the group are at the same step (the usual case): one launch for the group
            cfg = config(0.0, 1.0) if shared is not None else config(*scalars_of(steps.pop()))
            grads = [_grad_mem(p) for p in params]
            for t in grads:
                _ptr(t)      # device / dtype check
            sizes = (c_size_t * len(params))(*[p.numel() for p in params])
            call("vg_adam_step_multi", len(params), _pointer_array([p.data for p in params]), _pointer_array(grads),
                 _pointer_array([st["exp_avg"] for st in states]), _pointer_array([st["exp_avg_sq"] for st in states]),
                 _pointer_array([st.get("max_exp_avg_sq") for st in states]), sizes, byref(cfg), _stream())
            continue
        for p, state in zip(params, states):
            cfg = config(*scalars_of(float(state["step"])))
            call("vg_adam_step", _ptr(p.data), _ptr(_grad_mem(p)), _ptr(state["exp_avg"]), _ptr(state["exp_avg_sq"]),
                 _ptr(state.get("max_exp_avg_sq")), c_size_t(p.numel()), byref(cfg), _stream())


def _sgd(opt, grad_scale, lo, hi):
    for group in opt.param_groups:
        _check_flags(group, "SGD")
        sign = -1.0 if group.get("maximize", False) else 1.0
        for p in group["params"]:
            if p.grad is None:
                continue
            state = opt.state[p]
            first = 0
            if group["momentum"] != 0 and "momentum_buffer" not in state:
                state["momentum_buffer"] = _zeros_like(p)
                first = 1
            cfg = SgdCfg(_as_float(group["lr"]), group["momentum"], group["dampening"], group["weight_decay"],
                         1 if group["nesterov"] else 0, first, sign * grad_scale, lo, hi)
            call("vg_sgd_step", _ptr(p.data), _ptr(_grad_mem(p)), _ptr(state.get("momentum_buffer")),
                 c_size_t(p.numel()), byref(cfg), _stream())


def clamp_(params, lo, hi):
    from . import ops
    params = list(params)
    ops.plane_cache.drop(params)
    for p in params:
        call("vg_clamp", _ptr(p.data), c_size_t(p.numel()), c_float(lo), c_float(hi), _stream())
