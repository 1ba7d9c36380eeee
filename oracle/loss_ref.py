"""CPU restatement of the mask-aware loss and metrics.

Follows /root/reference/solpex/losses.py:
  _masked_mean            :60-79    sum(px*m*wf) / max(sum(m*wf), 1e-8), batch-global
  masked_weighted_loss    :82-213   base + lam_w*edge + lam_grad*sobel (+ multiscale)
  _sobel_kernels_cached   :16-45    kx = [[-1,0,1],[-2,0,2],[-1,0,1]], ky = kx^T (cross-correlation, zero pad)
  edge_weights            :215-224  exp(-EDT(~boundary)^2 / (2 sigma^2)) * mask
  mae_norm                :226-227
  batch_error_sums_ev     :229-253
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

_KX = [[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]


def sobel_xy(x):
    """Depthwise 3x3 Sobel cross-correlation with zero padding (losses.py:48-57)."""
    C = x.shape[1]
    kx = torch.tensor(_KX, dtype=x.dtype, device=x.device)
    ky = kx.t().contiguous()
    wx = kx.expand(C, 1, 3, 3).contiguous()
    wy = ky.expand(C, 1, 3, 3).contiguous()
    return F.conv2d(x, wx, padding=1, groups=C), F.conv2d(x, wy, padding=1, groups=C)


def masked_mean(px, mask, w=None, channel_weights=None):
    """losses.py:60-79."""
    wf = torch.ones_like(px)
    if w is not None:
        if w.ndim == 2:
            w = w[None, None]
        wf = wf * w
    if channel_weights is not None:
        wf = wf * channel_weights.reshape(1, -1, 1, 1).to(px.dtype)
    m = mask.expand_as(px) if mask.shape[1] == 1 else mask
    num = (px * m * wf).sum()
    den = (m * wf).sum().clamp_min(1e-8)
    return num / den


def _pix(a, b, kind, beta):
    if kind == "huber":
        return F.smooth_l1_loss(a, b, beta=beta, reduction="none")
    if kind == "l1":
        return (a - b).abs()
    if kind == "mse":
        return (a - b) ** 2
    raise ValueError(f"Unknown base={kind}")


def masked_weighted_loss(pred, target, mask, *, w=None, lam_grad=0.2, lam_w=1.0, base="huber",
                         huber_beta=0.05, grad_mode="vector", grad_base="l1", multiscale=0,
                         ms_weight=0.5, grad_ds=1, channel_weights=None):
    """losses.py:82-213."""
    if base not in ("huber", "l1", "mse"):
        raise ValueError(f"Unknown base={base}")
    base_px = _pix(pred, target, base, huber_beta)
    L = masked_mean(base_px, mask, None, channel_weights)
    if w is not None and lam_w != 0.0:
        L = L + lam_w * masked_mean(base_px, mask, w, channel_weights)
    if lam_grad != 0.0:
        ds = 1 if (grad_ds is None or grad_ds < 1) else int(grad_ds)
        if ds > 1:
            p_g = F.avg_pool2d(pred, ds, ds)
            t_g = F.avg_pool2d(target, ds, ds)
            m_g = (F.avg_pool2d(mask, ds, ds) > 0.5).to(pred.dtype)
        else:
            p_g, t_g, m_g = pred, target, mask
        pgx, pgy = sobel_xy(p_g)
        tgx, tgy = sobel_xy(t_g)
        if grad_mode == "magnitude":
            p = torch.sqrt(pgx * pgx + pgy * pgy + 1e-12)
            t = torch.sqrt(tgx * tgx + tgy * tgy + 1e-12)
            if grad_base not in ("huber", "l1", "mse"):
                raise ValueError(f"Unknown grad_base={grad_base}")
            g_px = _pix(p, t, grad_base, huber_beta)
        else:
            if grad_base not in ("huber", "l1", "mse"):
                raise ValueError(f"Unknown grad_base={grad_base}")
            g_px = _pix(pgx, tgx, grad_base, huber_beta) + _pix(pgy, tgy, grad_base, huber_beta)
        L = L + lam_grad * masked_mean(g_px, m_g, None, channel_weights)
    if multiscale and multiscale > 0:
        def half(t):
            return F.avg_pool2d(t, 2, 2)
        kw = dict(lam_grad=lam_grad, lam_w=lam_w, base=base, huber_beta=huber_beta, grad_mode=grad_mode,
                  grad_base=grad_base, multiscale=0, channel_weights=channel_weights)
        p2, t2, m2 = half(pred), half(target), half(mask)
        w2 = half(w) if w is not None else None
        L = L + ms_weight * masked_weighted_loss(p2, t2, m2, w=w2, grad_ds=max(1, int(grad_ds // 2)), **kw)
        if multiscale >= 2:
            p4, t4, m4 = half(p2), half(t2), half(m2)
            w4 = half(w2) if w2 is not None else None
            L = L + 0.5 * ms_weight * masked_weighted_loss(p4, t4, m4, w=w4, grad_ds=max(1, int(grad_ds // 4)), **kw)
    return L


def edge_weights(mask_np, sigma_px=3.0):
    """losses.py:215-224 (host-side scipy, once per run)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt
    m = np.asarray(mask_np).astype(bool)
    if not m.any():
        return np.zeros_like(mask_np, dtype=np.float32)
    boundary = m & ~binary_erosion(m, iterations=1, border_value=0)
    d = distance_transform_edt(~boundary)
    return (np.exp(-(d ** 2) / (2.0 * sigma_px ** 2)) * m).astype(np.float32)


def mae_norm(pred, target, mask):
    """losses.py:226-227: numerator over all channels, denominator pixels only."""
    return ((pred - target).abs() * mask).sum() / mask.sum().clamp_min(1e-8)


def batch_error_sums_ev(pred, target, mask, inverse_fn):
    """losses.py:229-253.  `inverse_fn(z, mask)` is the normalizer inverse."""
    p = torch.nan_to_num(inverse_fn(pred, mask), nan=0.0, posinf=0.0, neginf=0.0)
    y = torch.nan_to_num(inverse_fn(target, mask), nan=0.0, posinf=0.0, neginf=0.0)
    m = mask.expand_as(p) if mask.shape[1] == 1 else mask
    d = (p - y) * m
    return d.abs().sum(), (d * d).sum(), m.sum()


def weighted_masked_mse(pred, target, mask, cw):
    """scripts/eval_inverse_cycle_conditional_unet.py:105-115 (whole-batch denominator)."""
    m = mask.expand_as(pred)
    w = cw.view(1, -1, 1, 1).to(pred.dtype)
    return ((pred - target) ** 2 * m * w).sum() / (m * w).sum().clamp_min(1e-8)


def weighted_masked_mse_per_sample(pred, target, mask, cw):
    """Per-sample form of the above (B independent problems)."""
    m = mask.expand_as(pred)
    w = cw.view(1, -1, 1, 1).to(pred.dtype)
    num = ((pred - target) ** 2 * m * w).sum(dim=(1, 2, 3))
    den = (m * w).sum(dim=(1, 2, 3)).clamp_min(1e-8)
    return num / den
