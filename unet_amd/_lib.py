"""ctypes binding of libsolpex_hip.so (the C ABI declared in include/solpex_hip.h).

The product path has NO CPU fallback: if the shared library is missing this module raises at first
use, and every wrapper converts a negative status into ValueError (bad argument / unsupported
shape, matching the reference's error conventions) or RuntimeError (HIP failure).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsolpex_hip.so")

SX_F32, SX_BF16 = 0, 1
SX_CONV3, SX_CONV1, SX_CONVT_FWD, SX_CONVT_BWD = 0, 1, 2, 3
SX_ERR_INVALID, SX_ERR_UNSUPPORTED, SX_ERR_HIP = -1, -2, -3
(SX_T_INPUT, SX_T_SKIP, SX_T_BOTTLENECK, SX_T_PRED, SX_T_DPRED, SX_T_DSKIP, SX_T_DBOTTLENECK,
 SX_T_FILM_DPARAMS, SX_T_DINPUT) = range(9)
SX_BWD_WEIGHT_GRADS, SX_BWD_ACCUMULATE, SX_BWD_ZERO_DPARAMS, SX_BWD_DEFER_FILM = 1, 2, 4, 8
SX_BWD_ONLY_BOTTLENECK, SX_BWD_AFTER_BOTTLENECK, SX_BWD_NO_DPARAMS, SX_BWD_INPUT_GRAD = 16, 32, 64, 128
SX_BWD_SUMS_CLEARED, SX_BWD_HEAD_BIAS_READY = 256, 512
SX_T_HEAD_BIAS_SUMS, SX_HEAD_BIAS_SLOTS = 9, 64
SX_NORM_LOG, SX_NORM_LINEAR, SX_NORM_SYMLOG = 0, 1, 2
SX_WGRAD_STATIC_BYTES = 32 << 20
SX_GN_BWD_SUMS_READY = 4
PX_KINDS = {"huber": 0, "l1": 1, "mse": 2}


class UNetDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_ch", "out_ch", "base", "depth", "groups_gn", "P", "film_hidden",
                                         "B", "H", "W", "dtype", "use_film")]


class View(C.Structure):
    _fields_ = [("offset_bytes", C.c_int64), ("ld", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


_i, _l, _f, _p = C.c_int, C.c_int64, C.c_float, C.c_void_p
_pp = C.POINTER(C.c_void_p)
_D = C.POINTER(UNetDesc)

# name -> (restype, argtypes).  Every symbol declared in include/solpex_hip.h is listed here.
SIGNATURES = {
    "sx_version": (_i, []),
    "sx_last_error": (C.c_char_p, []),
    "sx_packed_weight_elems": (_l, [_i, _i, _i, _i]),
    "sx_set_fp32_exact": (_i, [_i]),
    "sx_set_fp32_planes": (_i, [_i]),
    "sx_pack_conv_weight": (_i, [_i, _i, _i, _p, _i, _i, _p, _p]),
    "sx_conv_forward": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "sx_conv_forward_gn_bwd_sums": (_i, [_i] * 7 + [_p, _i, _p, _p, _i, _i, _p, _i, _p, _i, _p, _i, _f, _p, _p, _p, _p, _p, _p]),
    "sx_conv_wgrad_workspace_bytes": (_l, [_i] * 6),
    "sx_conv_wgrad": (_i, [_i] * 9 + [_p, _i, _p, _i, _i, _i, _p, _i, _p, _l, _p]),
    "sx_conv_wgrad_static": (_i, [_i] * 9 + [_p, _i, _p, _i, _i, _i, _p, _i, _p]),
    "sx_gn_silu_forward": (_i, [_i, _i, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _i, _p]),
    "sx_block_out_forward": (_i, [_i, _i, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p]),
    "sx_block_out_pool_forward": (_i, [_i, _i, _i, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p]),
    "sx_block_out_pool_idx_forward": (_i, [_i, _i, _i, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "sx_gn_silu_backward": (_i, [_i, _i, _i, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _i,
                                 _p, _i, _p, _p, _i, _p]),
    "sx_pack_input": (_i, [_i, _p, _l, _l, _l, _l, _i, _i, _i, _i, _p, _i, _p]),
    "sx_unpack_output": (_i, [_i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "sx_maxpool2_forward": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "sx_maxpool2_backward_add": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "sx_maxpool2_backward_add_idx": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "sx_channel_sum": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "sx_global_avg_pool": (_i, [_i, _i, _i, _i, _p, _i, _p, _p]),
    "sx_gemm_f32": (_i, [_i, _i, _i, _p, _l, _l, _p, _l, _l, _p, _l, _p, _i, _p, _l, _p, _l, _p, _l, _p, _i, _p]),
    "sx_layernorm_forward": (_i, [_i, _i, _p, _p, _p, _f, _p, _p, _p]),
    "sx_layernorm_backward": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sx_smooth_l1": (_i, [_l, _p, _p, _f, _p, _p, _f, _i, _p]),
    "sx_mul": (_i, [_l, _p, _p, _p, _p]),
    "sx_gather_rows": (_i, [_p, _l, _l, _p, _i, _l, _p, _l, _p]),
    "sx_zero_border": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "sx_loss_workspace_bytes": (_l, [_i] * 6),
    "sx_loss_prepare": (_i, [_i, _i, _i, _i, _p, _p, _i, _i, _f, _p, _p, _p]),
    "sx_loss_forward_backward": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _f, _f, _f, _i, _i, _i, _f, _i,
                                      _p, _p, _p, _p, _i, _f, _p]),
    "sx_loss_forward_backward_bias": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _f, _f, _f, _i, _i, _i, _f, _i,
                                           _p, _p, _p, _p, _i, _f, _p, _p, _p]),
    "sx_norm_apply": (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sx_masked_err_sums": (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sx_loss_ev_l1": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _f, _p]),
    "sx_grad_sumsq": (_i, [_p, _l, _p, _p, _p]),
    "sx_grad_sumsq_partial": (_i, [_p, _l, _p, _p]),
    "sx_adamw_step_fused": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _p, _i, _f, _p, _p]),
    "sx_conv3_shortcut_forward": (_i, [_i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p]),
    "sx_conv3_shortcut_dgrad": (_i, [_i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p, _i, _p]),
    "sx_inverse_fit": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _f, _p]),
    "sx_inverse_adam_step": (_i, [_i, _i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _f, _f, _i, _f, _p, _p, _p, _p, _p]),
    "sx_adamw_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _p, _f, _p, _p, _p]),
    "sx_unet_num_params": (_i, [_D]),
    "sx_unet_wpack_bytes": (_l, [_D]),
    "sx_unet_workspace_bytes": (_l, [_D]),
    "sx_unet_view": (_i, [_D, _i, _i, C.POINTER(View)]),
    "sx_unet_pack_weights": (_i, [_D, _pp, _p, _p]),
    "sx_unet_encode": (_i, [_D, _pp, _p, _p, _p, _l, _l, _l, _l, _p, _p]),
    "sx_unet_pack_encode": (_i, [_D, _pp, _p, _p, _p, _l, _l, _l, _l, _p, _p]),
    "sx_unet_decode": (_i, [_D, _pp, _p, _p, _p, _pp, _p]),
    "sx_unet_decode_backward": (_i, [_D, _pp, _pp, _p, _p, _p, _pp, _i, _p]),
    "sx_unet_encode_backward": (_i, [_D, _pp, _pp, _p, _p, _p, _i, _p]),
}

_lib = None


def lib():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C unet_amd/csrc`.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    """Map a C status to the reference's Python error conventions."""
    if rc >= 0:
        return rc
    msg = lib().sx_last_error().decode("utf-8", "replace")
    if what:
        msg = f"{what}: {msg}"
    if rc in (SX_ERR_INVALID, SX_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_table(tensors):
    """Host array of device pointers (entries may be None)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
