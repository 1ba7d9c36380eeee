"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's per-case inverse loop.

Follows scripts/eval_inverse_cycle_conditional_unet.py: weighted_masked_mse :105-115, masked_mae :125-129,
build_x_from_mask :75-77, the Adam + CosineAnnealingLR loop :332-411 (clamp :388-390, early stop :398-411) and the
final evaluation :413-416.  Pinned by tests/golden/inverse_case.npz (generated from the live reference).
"""
from __future__ import annotations

import torch

from . import unet_ref


def weighted_masked_mse(pred, target, mask, channel_weights):
    m = mask.expand_as(pred) if (mask.shape[1] == 1 and pred.shape[1] != 1) else mask
    w = channel_weights.view(1, -1, 1, 1).to(pred.dtype)
    num = (((pred - target) ** 2) * m * w).sum()
    den = (m * w).sum().clamp_min(1e-8)
    return num / den


def masked_mae(pred, target, mask):
    m = mask.expand_as(pred) if (mask.shape[1] == 1 and pred.shape[1] != 1) else mask
    return ((pred - target).abs() * m).sum() / m.sum().clamp_min(1e-8)


def inverse_case(sd, cfg, mask_1, y_true_norm_1, p0_1, field_idx, cw_sel, *, steps, lr, reg, clip, patience, tol):
    """One case (B = 1).  Returns (p_final (P,), final fit, final masked MAE, per-step losses list)."""
    sd = {k: v.detach() for k, v in sd.items()}
    y_sel = y_true_norm_1[:, field_idx]
    p_var = torch.nn.Parameter(p0_1.clone())
    opt = torch.optim.Adam([p_var], lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=1e-5)
    best, wait, losses = float("inf"), 0, []
    for _ in range(steps):
        y_pred = unet_ref.unet_forward(sd, cfg, mask_1, p_var)          # x = the mask itself (:75-77)
        loss_fit = weighted_masked_mse(y_pred[:, field_idx], y_sel, mask_1, cw_sel)
        loss = loss_fit + reg * (p_var ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        item = float(loss.item())
        losses.append(item)
        if clip is not None and clip > 0:
            with torch.no_grad():
                p_var.clamp_(-float(clip), float(clip))
        if patience > 0:
            # (inf - x) / inf is NaN: while `best` is still inf nothing counts as an improvement (the reference's behaviour)
            rel = (best - item) / max(abs(best), 1e-12)
            if rel > tol:
                best, wait = item, 0
            else:
                wait += 1
                if wait >= patience:
                    break
    with torch.no_grad():
        y_pred = unet_ref.unet_forward(sd, cfg, mask_1, p_var)
        fit = float(weighted_masked_mse(y_pred[:, field_idx], y_sel, mask_1, cw_sel))
        mae = float(masked_mae(y_pred[:, field_idx], y_sel, mask_1))
    return p_var.detach()[0].clone(), fit, mae, losses


def inverse_case_lbfgs(sd, cfg, mask_1, y_true_norm_1, p0_1, field_idx, cw_sel, *, steps, reg, clip):
    """The script's default optimiser path (:336-342 torch.optim.LBFGS(lr 1, max_iter 20, strong Wolfe), closure :359-370,
    clamp :388-390, final evaluation :413-416) for one case.  Returns (p_final, final fit, per-step last-closure losses,
    per-step closure counts).  Pinned by tests/golden/inverse_lbfgs.npz."""
    sd = {k: v.detach() for k, v in sd.items()}
    y_sel = y_true_norm_1[:, field_idx]
    p_var = torch.nn.Parameter(p0_1.clone())
    opt = torch.optim.LBFGS([p_var], lr=1.0, max_iter=20, line_search_fn="strong_wolfe")
    losses, counts = [], []
    for _ in range(steps):
        last, n = [None], [0]

        def closure():
            opt.zero_grad(set_to_none=True)
            y_pred = unet_ref.unet_forward(sd, cfg, mask_1, p_var)
            loss = weighted_masked_mse(y_pred[:, field_idx], y_sel, mask_1, cw_sel) + reg * (p_var ** 2).mean()
            loss.backward()
            last[0] = float(loss.item())
            n[0] += 1
            return loss
        opt.step(closure)
        if clip is not None and clip > 0:
            with torch.no_grad():
                p_var.clamp_(-float(clip), float(clip))
        losses.append(last[0])
        counts.append(n[0])
    with torch.no_grad():
        y_pred = unet_ref.unet_forward(sd, cfg, mask_1, p_var)
        fit = float(weighted_masked_mse(y_pred[:, field_idx], y_sel, mask_1, cw_sel))
    return p_var.detach()[0].clone(), fit, losses, counts
