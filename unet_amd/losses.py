"""MI355X-native mirror of `solpex.losses` (reference: solpex/losses.py).

`masked_weighted_loss`, `mae_norm` and `batch_error_sums_ev` keep the reference signatures and error
behaviour; on device they run the fused HIP kernels of libsolpex_hip.so (loss forward AND its
gradient in one pass).  `edge_weights` stays host-side scipy exactly as in the reference (called once
per run, solpex/train.py:132-134).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

_BASES = ("huber", "l1", "mse")


def _require_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs CUDA (ROCm) tensors; this build has no CPU path")


def _nhwc(t):
    """(B,C,H,W) tensor -> (tensor, ld, sx dtype) readable by the NHWC kernels without a copy if possible."""
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    ok = (t.dtype in (torch.float32, torch.bfloat16) and (sc == 1 or Cc == 1) and sh == W * sw and sb == H * W * sw
          and sw >= Cc)
    if not ok:
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.contiguous(memory_format=torch.channels_last)
        if Cc == 1:
            t = t.contiguous()
        sw = Cc
    return t, int(sw), (L.SX_BF16 if t.dtype == torch.bfloat16 else L.SX_F32)


def _prep_mask(mask, B, H, W, who):
    if mask.dim() != 4 or mask.shape[0] != B or mask.shape[1] != 1 or tuple(mask.shape[2:]) != (H, W):
        raise ValueError(f"{who}: mask must be (B,1,H,W) = ({B},1,{H},{W}), got {tuple(mask.shape)}")
    return mask.float().contiguous()


def _prep_w(w, H, W, device, who):
    if w is None:
        return None
    if not torch.is_tensor(w):
        w = torch.as_tensor(w)
    if w.numel() != H * W:
        raise ValueError(f"{who}: w must broadcast from (H,W) = ({H},{W}); per-sample weight maps are not supported, "
                         f"got {tuple(w.shape)}")
    return w.to(device=device, dtype=torch.float32).reshape(H, W).contiguous()


def _prep_cw(cw, Cc, device):
    if cw is None:
        return None
    if not torch.is_tensor(cw):
        cw = torch.as_tensor(np.asarray(cw, dtype=np.float32))
    cw = cw.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    if cw.numel() == 1 and Cc != 1:
        cw = cw.expand(Cc).contiguous()
    if cw.numel() != Cc:
        raise ValueError(f"channel_weights length={cw.numel()} but C={Cc}")
    return cw


class LossBuffers:
    """Device scratch for one loss shape (workspace, denominators, outputs)."""

    def __init__(self, B, Cc, H, W, multiscale, grad_ds, device):
        lib = L.lib()
        nbytes = L.check(lib.sx_loss_workspace_bytes(B, Cc, H, W, int(multiscale), int(grad_ds)))
        self.ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        self.den = torch.empty(9, dtype=torch.float32, device=device)
        self.out = torch.empty(4, dtype=torch.float32, device=device)    # loss, sum|p-t|*m, sum m, sum m


def loss_prepare(bufs, B, Cc, H, W, mask, w, multiscale, grad_ds, lam_grad):
    L.check(L.lib().sx_loss_prepare(B, Cc, H, W, L.ptr(mask), L.ptr(w), int(multiscale), int(grad_ds), float(lam_grad),
                                    L.ptr(bufs.ws), L.ptr(bufs.den), L.stream_ptr()), "sx_loss_prepare")


def loss_forward_backward(bufs, sx_dtype, B, Cc, H, W, pred, pred_ld, target, mask, w, cw, base, huber_beta, lam_w, lam_grad,
                          grad_mode, grad_base, multiscale, ms_weight, grad_ds, d_pred, dpred_ld, grad_scale=1.0, out=None,
                          bias_sums=None):
    """`bias_sums` (optional, zeroed [SX_HEAD_BIAS_SLOTS][C] fp32): also form the output head's bias gradient from d_pred;
    returns True when the kernels took it (then nothing may add to d_pred afterwards)."""
    import ctypes
    done = ctypes.c_int(0)
    L.check(L.lib().sx_loss_forward_backward_bias(
        sx_dtype, B, Cc, H, W, L.ptr(pred), pred_ld, L.ptr(target), L.ptr(mask), L.ptr(w), L.ptr(cw),
        L.PX_KINDS[base], float(huber_beta), float(lam_w), float(lam_grad), 1 if grad_mode == "magnitude" else 0,
        L.PX_KINDS[grad_base], int(multiscale), float(ms_weight), int(grad_ds), L.ptr(bufs.den), L.ptr(bufs.ws),
        L.ptr(bufs.out if out is None else out), L.ptr(d_pred), dpred_ld, float(grad_scale),
        L.ptr(bias_sums), ctypes.cast(ctypes.byref(done), ctypes.c_void_p) if bias_sums is not None else None, L.stream_ptr()),
        "sx_loss_forward_backward")
    return bool(done.value)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, w, cw, opts):
        B, Cc, H, W = pred.shape
        p, ld, dt = _nhwc(pred.detach())
        need_grad = ctx.needs_input_grad[0]
        bufs = LossBuffers(B, Cc, H, W, opts["multiscale"], opts["grad_ds"], pred.device)
        loss_prepare(bufs, B, Cc, H, W, mask, w, opts["multiscale"], opts["grad_ds"], opts["lam_grad"])
        d_pred = None
        if need_grad:
            sess = getattr(pred, "_sx_sess", None)
            if sess is not None and sess.view(L.SX_T_PRED).data_ptr() == pred.data_ptr() and ld == sess.view(L.SX_T_PRED).stride(3):
                d_pred = sess.view(L.SX_T_DPRED)        # written in place: decode backward reads it without a copy
            else:
                d_pred = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
        loss_forward_backward(bufs, dt, B, Cc, H, W, p, ld, target, mask, w, cw, opts["base"], opts["huber_beta"],
                              opts["lam_w"], opts["lam_grad"], opts["grad_mode"], opts["grad_base"], opts["multiscale"],
                              opts["ms_weight"], opts["grad_ds"], d_pred, ld)
        ctx.d_pred = d_pred
        ctx.pred_dtype = pred.dtype
        return bufs.out[0].clone()

    @staticmethod
    def backward(ctx, g):
        d = ctx.d_pred
        d.mul_(g.to(d.dtype))      # in place: keeps the zero-copy hand-off to the decoder's backward
        if d.dtype != ctx.pred_dtype:
            d = d.to(ctx.pred_dtype)
        return d, None, None, None, None, None


def masked_weighted_loss(pred, target, mask, *, w=None, lam_grad=0.2, lam_w=1.0, base="huber", huber_beta=0.05,
                         grad_mode="vector", grad_base="l1", multiscale=0, ms_weight=0.5, grad_ds: int = 1,
                         channel_weights=None):
    """Reference: solpex/losses.py:82-213 (same arguments, same ValueErrors)."""
    if base not in _BASES:
        raise ValueError(f"Unknown base={base}")
    if lam_grad != 0.0 and grad_base not in _BASES:
        raise ValueError(f"Unknown grad_base={grad_base}")
    _require_cuda(pred, "masked_weighted_loss")
    if pred.dim() != 4 or tuple(target.shape) != tuple(pred.shape):
        raise ValueError(f"masked_weighted_loss: pred/target must both be (B,C,H,W), got {tuple(pred.shape)} / {tuple(target.shape)}")
    B, Cc, H, W = pred.shape
    if grad_ds is None or grad_ds < 1:
        grad_ds = 1
    opts = dict(lam_grad=float(lam_grad), lam_w=float(lam_w), base=base, huber_beta=float(huber_beta),
                grad_mode="magnitude" if grad_mode == "magnitude" else "vector",
                grad_base=grad_base if grad_base in _BASES else "l1",
                multiscale=int(min(max(int(multiscale or 0), 0), 2)), ms_weight=float(ms_weight), grad_ds=int(grad_ds))
    m = _prep_mask(mask, B, H, W, "masked_weighted_loss")
    wt = _prep_w(w, H, W, pred.device, "masked_weighted_loss")
    cw = _prep_cw(channel_weights, Cc, pred.device)
    tgt = target.detach().float().contiguous()
    return _LossFn.apply(pred, tgt, m, wt, cw, opts)


def edge_weights(mask_np: np.ndarray, sigma_px: float = 3.0) -> np.ndarray:
    """Reference: solpex/losses.py:215-224.  Host-side (scipy), once per run."""
    from scipy.ndimage import binary_erosion, distance_transform_edt
    inside = np.asarray(mask_np).astype(bool)
    if not inside.any():
        return np.zeros_like(mask_np, dtype=np.float32)
    rim = inside & ~binary_erosion(inside, iterations=1, border_value=0)
    dist = distance_transform_edt(~rim)
    return (np.exp(-(dist ** 2) / (2.0 * sigma_px ** 2)) * inside).astype(np.float32)


def _err_sums(pred, target, mask, table):
    _require_cuda(pred, "masked metrics")
    B, Cc, H, W = pred.shape
    p = pred.detach()
    if p.dtype not in (torch.float32, torch.bfloat16):
        p = p.float()
    dt = L.SX_BF16 if p.dtype == torch.bfloat16 else L.SX_F32
    sb, sc, sh, sw = p.stride()
    tgt = target.detach().float().contiguous()
    m = _prep_mask(mask, B, H, W, "masked metrics")
    out = torch.zeros(3, dtype=torch.float32, device=p.device)
    if table is None:
        args = (None, None, None, None, None)
        denorm = 0
    else:
        args = table.c_args()
        denorm = 1
    L.check(L.lib().sx_masked_err_sums(dt, denorm, B, Cc, H, W, L.ptr(p), sb, sc, sh, sw, L.ptr(tgt), L.ptr(m), *args,
                                       L.ptr(out), L.stream_ptr()), "sx_masked_err_sums")
    return out


def mae_norm(pred, target, mask):
    """Reference: solpex/losses.py:226-227 (sum over channels / number of masked PIXELS)."""
    s = _err_sums(pred, target, mask, None)
    return s[0] / s[2].clamp_min(1e-8)


def batch_error_sums_ev(pred, target, mask, norm):
    """Reference: solpex/losses.py:229-253.  Returns (abs_sum, sq_sum, px) device scalars."""
    from .data import norm_table
    with torch.no_grad():
        s = _err_sums(pred, target, mask, norm_table(norm, pred.shape[1]))
        return s[0], s[1], s[2]
