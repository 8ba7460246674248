"""ctypes binding of libvegans_hip.so (C ABI declared in include/vegans_hip.h).

Every function here takes torch tensors that already live on the MI355X, checks dtype / device / contiguity
on the host (a kernel that faults can take the whole GPU down), and enqueues the kernel on torch's current
stream.  There is no CPU or stock-torch fallback: if the library is missing or the tensors are not on a GPU
the call raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libvegans_hip.so")

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


class ConvGeom(Structure):
    """Mirror of vg_conv_geom."""
    _fields_ = [(n, c_int32) for n in ("N", "H", "W", "Ci", "Co", "kh", "kw", "Ho", "Wo", "sh", "sw", "ph", "pw")]


class RmspropCfg(Structure):
    _fields_ = [("lr", c_float), ("alpha", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("momentum", c_float), ("centered", c_int32), ("grad_scale", c_float),
                ("clamp_lo", c_float), ("clamp_hi", c_float)]


class AdamCfg(Structure):
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("weight_decay", c_float), ("step_size", c_float), ("bias_correction2_sqrt", c_float),
                ("amsgrad", c_int32), ("decoupled_wd", c_int32), ("grad_scale", c_float),
                ("clamp_lo", c_float), ("clamp_hi", c_float), ("step_scalars", c_void_p)]


class SgdCfg(Structure):
    _fields_ = [("lr", c_float), ("momentum", c_float), ("dampening", c_float), ("weight_decay", c_float),
                ("nesterov", c_int32), ("first_step", c_int32), ("grad_scale", c_float),
                ("clamp_lo", c_float), ("clamp_hi", c_float)]


_P = c_void_p  # device pointers and the stream travel as void*
_G = POINTER(ConvGeom)

# name -> (restype, argtypes); kept in one table so tests can compare it with the header.
SIGNATURES = {
    "vg_last_error": (ctypes.c_char_p, []),
    "vg_abi_version": (c_int, []),
    "vg_time_next_conv": (c_int, [c_void_p, c_void_p]),
    "vg_time_pending": (c_int, []),
    "vg_conv2d_fwd": (c_int, [_G, _P, _P, _P, _P, _P]),
    "vg_conv2d_dgrad": (c_int, [_G, _P, _P, _P, _P, _P]),
    "vg_conv2d_wgrad_workspace_bytes": (c_size_t, [_G]),
    "vg_conv2d_wgrad": (c_int, [_G, _P, _P, _P, _P, c_size_t, _P]),
    "vg_conv2d_wgrad_noalloc": (c_int, [_G, _P, _P, _P, _P]),
    "vg_conv2d_wgrad_partial": (c_int, [_G, _P, _P, _P, _P, c_size_t, _P]),
    "vg_conv2d_wgrad_reduce": (c_int, [_G, _P, _P, _P]),
    "vg_conv2d_wgrad_multi_applies": (c_int, [_G]),
    "vg_conv2d_wgrad_multi_workspace_bytes": (c_size_t, [_G, c_int]),
    "vg_conv2d_wgrad_partial_multi": (c_int, [_G, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vg_conv2d_wgrad_reduce_multi": (c_int, [_G, c_int, _P, _P, _P]),
    "vg_conv2d_variant": (c_int, [c_int, _G, ctypes.c_char_p, c_size_t]),
    "vg_conv2d_fwd_act_applies": (c_int, [_G, c_int]),
    "vg_conv2d_fwd_act": (c_int, [_G, _P, _P, _P, _P, c_int, c_float, _P]),
    "vg_conv2d_prec_workspace_bytes": (c_size_t, [c_int, _G, c_int]),
    "vg_conv2d_fwd_prec": (c_int, [_G, _P, _P, _P, _P, c_int, _P, c_size_t, _P, _P, _P]),
    "vg_conv2d_dgrad_prec": (c_int, [_G, _P, _P, _P, _P, c_int, _P, c_size_t, _P, _P, _P]),
    "vg_conv2d_wgrad_partial_prec": (c_int, [_G, _P, _P, _P, _P, c_size_t, c_int, _P, _P, _P]),
    "vg_split_planes_bytes": (c_size_t, [ctypes.c_longlong, c_int]),
    "vg_split_planes_f32": (c_int, [_P, ctypes.c_longlong, c_int, _P, _P]),
    "vg_conv2d_split_applies": (c_int, [c_int, _G, c_int]),
    "vg_maxpool2d_fwd": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P]),
    "vg_maxpool2d_bwd": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P]),
    "vg_maxpool2d_pick": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P]),
    "vg_nchw_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vg_nchw_to_nhwc_f64": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vg_cast_f64_to_f32": (c_int, [_P, _P, c_size_t, _P]),
    "vg_nhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vg_concat_channels": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "vg_slice_channels": (c_int, [_P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "vg_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vg_colsum": (c_int, [_P, _P, c_int, c_int, _P, c_size_t, _P]),
    "vg_bcast_rows": (c_int, [_P, _P, c_int, c_int, _P]),
    "vg_add": (c_int, [_P, _P, _P, c_size_t, _P]),
    "vg_scale": (c_int, [_P, c_float, _P, c_size_t, _P]),
    "vg_fill": (c_int, [_P, c_float, c_size_t, _P]),
    "vg_scalar_combine": (c_int, [_P, c_float, _P, c_float, _P, c_float, _P, _P, _P]),
    "vg_bn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vg_bn_stats": (c_int, [_P, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vg_bn_invstd_from_var": (c_int, [_P, _P, c_int, c_float, _P]),
    "vg_conv2d_bnstats_workspace_bytes": (c_size_t, [c_int, _P]),
    "vg_conv2d_fwd_bnstats": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "vg_conv2d_dgrad_bnstats": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "vg_conv2d_fwd_bnbwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_float, _P, c_size_t, _P, _P]),
    "vg_conv2d_dgrad_bnbwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_float, _P, c_size_t, _P, _P]),
    "vg_conv2d_dgrad_actmask": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, _P, _P]),
    "vg_bn_bwd_sums_from_conv": (c_int, [c_int, _P, _P, _P, _P]),
    "vg_bn_stats_from_conv": (c_int, [c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "vg_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "vg_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, c_int, c_float, _P, _P, _P]),
    "vg_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_float, _P, _P]),
    "vg_bn_bwd_eval": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "vg_bn_dbwd_reduce": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, c_int, c_float, _P, _P, _P]),
    "vg_bn_dbwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_float,
                                 _P, _P]),
    "vg_act_fwd": (c_int, [_P, _P, c_size_t, c_int, c_float, _P]),
    "vg_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, c_float, _P]),
    "vg_act_dbwd_ref": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    "vg_reduce_workspace_bytes": (c_size_t, [c_size_t]),
    "vg_loss_mean_fwd": (c_int, [_P, c_size_t, c_float, _P, _P, c_size_t, _P]),
    "vg_loss_mean_bwd": (c_int, [_P, c_size_t, c_float, _P, _P]),
    "vg_loss_bce_fwd": (c_int, [_P, _P, c_float, c_size_t, _P, _P, c_size_t, _P]),
    "vg_loss_bce_bwd": (c_int, [_P, _P, _P, c_float, c_size_t, _P, _P]),
    "vg_loss_mse_fwd": (c_int, [_P, _P, c_float, c_size_t, _P, _P, c_size_t, _P]),
    "vg_loss_mse_bwd": (c_int, [_P, _P, _P, c_float, c_size_t, _P, _P]),
    "vg_loss_l1_fwd": (c_int, [_P, _P, c_size_t, _P, _P, c_size_t, _P]),
    "vg_loss_l1_bwd": (c_int, [_P, _P, _P, c_size_t, _P, _P, _P]),
    "vg_gp_interp": (c_int, [_P, _P, _P, _P, c_int, c_size_t, _P]),
    "vg_gp_rownorm": (c_int, [_P, _P, c_int, c_size_t, _P]),
    "vg_gp_penalty_fwd": (c_int, [_P, c_int, _P, _P]),
    "vg_gp_penalty_bwd": (c_int, [_P, _P, _P, _P, c_int, c_size_t, _P]),
    "vg_rmsprop_step": (c_int, [_P, _P, _P, _P, _P, c_size_t, POINTER(RmspropCfg), _P]),
    "vg_rmsprop_step_multi": (c_int, [c_int, _P, _P, _P, _P, _P, _P, POINTER(RmspropCfg), _P]),
    "vg_adam_step": (c_int, [_P, _P, _P, _P, _P, c_size_t, POINTER(AdamCfg), _P]),
    "vg_adam_step_multi": (c_int, [c_int, _P, _P, _P, _P, _P, _P, POINTER(AdamCfg), _P]),
    "vg_grad_sum_multi": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "vg_sgd_step": (c_int, [_P, _P, _P, c_size_t, POINTER(SgdCfg), _P]),
    "vg_clamp": (c_int, [_P, c_size_t, c_float, c_float, _P]),
    # activations stored as bf16
    "vg_conv2d_h_applies": (c_int, [c_int, _G]),
    "vg_conv2d_fwd_h": (c_int, [_G, _P, _P, _P, _P, _P]),
    "vg_conv2d_dgrad_h": (c_int, [_G, _P, _P, _P, _P, _P]),
    "vg_conv2d_dgrad_h_actmask": (c_int, [_G, _P, _P, _P, _P, c_int, c_float, _P]),
    "vg_conv2d_wgrad_h_workspace_bytes": (c_size_t, [_G]),
    "vg_conv2d_wgrad_partial_h": (c_int, [_G, _P, _P, _P, _P, c_size_t, _P]),
    "vg_conv2d_wgrad_reduce_h": (c_int, [_G, _P, _P, _P]),
    "vg_conv2d_wgrad_h_multi_workspace_bytes": (c_size_t, [_G, c_int]),
    "vg_conv2d_wgrad_partial_h_multi": (c_int, [_G, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vg_conv2d_wgrad_reduce_h_multi": (c_int, [_G, c_int, _P, _P, _P]),
    "vg_conv2d_stem_h_applies": (c_int, [c_int, _G]),
    "vg_conv2d_stem_fwd_h": (c_int, [_G, _P, _P, _P, _P, c_int, c_float, _P]),
    "vg_conv2d_stem_dgrad_h": (c_int, [_G, _P, _P, _P, _P, _P]),
    "vg_conv2d_stem_wgrad_partial_h": (c_int, [_G, _P, _P, _P, c_size_t, _P]),
    "vg_cast_f32_to_bf16": (c_int, [_P, _P, c_size_t, _P]),
    "vg_cast_bf16_to_f32": (c_int, [_P, _P, c_size_t, _P]),
    "vg_add_h": (c_int, [_P, _P, _P, c_size_t, _P]),
    "vg_fill_h": (c_int, [_P, c_float, c_size_t, _P]),
    "vg_colsum_h": (c_int, [_P, _P, c_int, c_int, _P, c_size_t, _P]),
    "vg_bn_stats_h": (c_int, [_P, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vg_bn_apply_h": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "vg_bn_bwd_reduce_h": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, c_int, c_float, _P, _P, _P]),
    "vg_bn_bwd_apply_h": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_float, _P, _P]),
    "vg_bn_bwd_eval_h": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "vg_bn_dbwd_reduce_h": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, c_int, c_float, _P, _P, _P]),
    "vg_bn_dbwd_apply_h": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_float,
                                   _P, _P]),
    "vg_act_fwd_h": (c_int, [_P, _P, c_size_t, c_int, c_float, _P]),
    "vg_act_bwd_h": (c_int, [_P, _P, _P, c_size_t, c_int, c_float, _P]),
    "vg_act_dbwd_ref_h": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
}

_lib = None


def load():
    """Load the shared library once; raises if it has not been built (run `python __graft_entry__.py build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvegans_hip.so not found at %s: build it with `make -C gan-pytorch_amd/csrc` "
                "(this engine has no CPU or stock-kernel fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class HipError(RuntimeError):
    pass


def _check(rc, name):
    if rc != 0:
        raise HipError("%s failed (%d): %s" % (name, rc, load().vg_last_error().decode()))


def _stream():
    """Raw handle of torch's current HIP stream on the current device."""
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    """Device pointer of a dense float32 GPU tensor (None -> null)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("tensor is not on the GPU: this engine only runs on MI355X (device='cuda')")
    if t.dtype != torch.float32:
        raise HipError("tensor dtype %s: the kernels compute in float32" % t.dtype)
    return c_void_p(t.data_ptr())


def _dense(t):
    """True if the tensor's memory is one gap-free block (any permutation of a contiguous layout)."""
    return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last) or t.numel() <= 1 or \
        t.is_non_overlapping_and_dense()


_workspace = {}
_workspace_frozen = False


def freeze_workspace(frozen):
    """While frozen (a HIP graph is being captured) the scratch buffer must not be replaced: a buffer allocated during
    capture would belong to that graph's memory pool and die with it."""
    global _workspace_frozen
    _workspace_frozen = bool(frozen)


def workspace_buffers():
    """The scratch buffers currently in use (one per device): a captured HIP graph keeps these references so the
    addresses it recorded stay valid after the shared buffer has been replaced by a larger one."""
    return list(_workspace.values())


def workspace(nbytes, device):
    """Grow-only scratch buffer per device and stream role (the launch stream and each side stream of
    engine/streams.py have one of their own): kernels that may run concurrently never share scratch."""
    from .engine import streams
    key = (device.index if device.index is not None else torch.cuda.current_device(), streams.role())
    buf = _workspace.get(key)
    if buf is None or buf.numel() < nbytes:
        if _workspace_frozen:
            raise HipError("scratch buffer would have to grow to %d bytes during graph capture" % nbytes)
        nbytes = max(int(nbytes), 1 << 20)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspace[key] = buf
    return buf


def call(name, *args):
    rc = getattr(load(), name)(*args)
    _check(rc, name)
