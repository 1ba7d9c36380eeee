"""Batched inverse problems through the frozen conditional U-Net.

Reference: scripts/eval_inverse_cycle_conditional_unet.py:105-162 (losses, forward_from_scaled) and :316-444 (the per-case
optimisation loop and the forward-cycle metrics).  The reference solves one case at a time (B = 1: mask (1,1,H,W),
p (1,P)); here N cases run as one batch per call (SURVEY 8 row a20 / BASELINE config 4) with every per-case quantity kept
per SAMPLE: loss denominators, the L2 regulariser, Adam state, clamp and early stopping.  Only the data gradient of the
network is computed (`SX_BWD_WEIGHT_GRADS` off): the reference's `loss.backward()` also fills every weight gradient and
throws it away.

`optimizer="lbfgs"` (the reference script's default, :336-342) keeps the reference's structure: torch.optim.LBFGS is host
control logic over five numbers per case (two-loop recursion, strong-Wolfe line search), so cases run one after the other
and only the closure -- network forward, per-sample fit, data-gradient backward -- runs on the device (B = 1 session).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib as L
from . import dist as sxdist


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _weights(model, field_idx, channel_weights, device):
    """Full-length channel weight / selection vectors: a --fields subset is a zero weight outside the subset."""
    Cc = model._cfg["out_ch"]
    idx = list(range(Cc)) if field_idx is None else [int(i) for i in field_idx]
    cw = torch.ones(len(idx)) if channel_weights is None else torch.as_tensor(channel_weights, dtype=torch.float32).reshape(-1).cpu()
    if cw.numel() != len(idx):
        raise ValueError(f"channel_weights length={cw.numel()} but {len(idx)} fields are selected")
    w = torch.zeros(Cc)
    s = torch.zeros(Cc)
    for k, i in enumerate(idx):
        if not 0 <= i < Cc:
            raise ValueError(f"field index {i} outside 0..{Cc - 1}")
        w[i] = cw[k]
        s[i] = 1.0
    return w.to(device), s.to(device)


def _mask3(mask):
    m = mask
    if m.dim() == 4:
        if m.shape[1] != 1:
            raise ValueError("mask must be (N,1,H,W) or (N,H,W)")
        m = m[:, 0]
    if m.dim() != 3:
        raise ValueError("mask must be (N,1,H,W) or (N,H,W)")
    return _f32c(m)


def weighted_masked_mse(pred, target, mask, channel_weights):
    """Per-sample weighted masked MSE, reference :105-115 applied to every sample of the batch on its own.
    pred/target (N,C,H,W), mask (N,1,H,W) or (N,H,W), channel_weights (C,) -> (N,) fp32."""
    if not pred.is_cuda:
        raise RuntimeError("unet_amd.inverse: tensors must be CUDA (ROCm) tensors; there is no CPU fallback")
    N, Cc, H, W = pred.shape
    p = pred.detach().float().permute(0, 2, 3, 1).contiguous()          # NHWC fp32
    w = torch.as_tensor(channel_weights, dtype=torch.float32, device=pred.device).reshape(-1).contiguous()
    fit = torch.empty(N, dtype=torch.float32, device=pred.device)
    L.check(L.lib().sx_inverse_fit(L.SX_F32, N, Cc, H, W, L.ptr(p), Cc, L.ptr(_f32c(target)), L.ptr(_mask3(mask)), L.ptr(w), None,
                                   L.ptr(fit), None, None, 0, 1.0, L.stream_ptr()), "sx_inverse_fit")
    return fit


def cosine_lr(lr0, step, t_max, eta_min=1e-5):
    """torch.optim.lr_scheduler.CosineAnnealingLR in closed form: the rate used by optimiser step `step` (0-based)."""
    return eta_min + (lr0 - eta_min) * (1.0 + math.cos(math.pi * step / t_max)) / 2.0


class _Problem:
    """Device state of one chunk of B inverse problems."""

    def __init__(self, model, B, H, W, dtype):
        from ._engine import UNetSession
        params = model._ordered_params()
        dev = params[0].device
        self.sess = UNetSession(model._cfg, B, H, W, dtype, True, dev, film_shared=True)
        self.sess.set_params(params)
        self.sess.pack_weights()
        self.B = B
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        P = model.P
        self.m, self.v = z(B, P), z(B, P)
        self.fit, self.mae, self.loss = z(B), z(B), z(B)
        self.best = torch.full((B,), float("inf"), dtype=torch.float32, device=dev)
        self.wait, self.active = z(B, dt=torch.int32), torch.ones(B, dtype=torch.int32, device=dev)

    def forward_fit(self, x, p, target, mask3, w, sel, want_grad):
        s = self.sess
        s.encode(x, p)
        s.decode(p, None)
        pred = s.view(L.SX_T_PRED)
        ld = pred.stride(3)
        dpred = s.view(L.SX_T_DPRED) if want_grad else None
        L.check(L.lib().sx_inverse_fit(s.dtype, self.B, pred.shape[1], s.H, s.W, L.ptr(pred), ld, L.ptr(target), L.ptr(mask3),
                                       L.ptr(w), L.ptr(sel), L.ptr(self.fit), L.ptr(self.mae), L.ptr(dpred), ld, 1.0,
                                       L.stream_ptr()), "sx_inverse_fit")

    def backward_params(self, p):
        s = self.sess
        s.decode_backward(p, None, L.SX_BWD_ZERO_DPARAMS | L.SX_BWD_DEFER_FILM)        # data gradients only
        s.encode_backward(p, L.SX_BWD_DEFER_FILM)
        return s.view(L.SX_T_FILM_DPARAMS)


@torch.no_grad()
def batched_inverse(model, mask, y_true_norm, p0, *, field_idx=None, channel_weights=None, steps=300, lr=5e-2,
                    optimizer="adam", reg=1e-4, clip_scaled=4.0, early_stop_patience=100, early_stop_tol=1e-6,
                    batch=1024, dtype=None, return_history=False):
    """Recover scaled parameters for N cases at once (reference loop :332-416, Adam + CosineAnnealingLR path).

    mask (N,1,H,W) or (N,H,W) - also the network input (`build_x_from_mask`, :75-77);  y_true_norm (N,C,H,W) targets in
    normalised space;  p0 (N,P) initial scaled parameters.  Returns dict: p_scaled (N,P), fit (N,) final
    weighted_masked_mse (:413-416), cycle_mae_norm (N,) (:439), steps_run (N,), and loss_history (steps,N) if requested
    (NaN where a case had already stopped)."""
    if optimizer not in ("adam", "lbfgs"):
        raise ValueError(f"unet_amd.inverse: unknown optimizer {optimizer!r} (adam | lbfgs)")
    if model.P <= 0:
        raise ValueError("batched_inverse needs a FiLM-conditioned model (P > 0)")
    if not p0.is_cuda:
        raise RuntimeError("unet_amd.inverse: tensors must be CUDA (ROCm) tensors; there is no CPU fallback")
    was_training = model.training
    model.eval()                                   # :284 (dropout off)
    try:
        dev = p0.device
        mask3 = _mask3(mask.to(dev))
        N, H, W = mask3.shape
        tgt = _f32c(y_true_norm.to(dev))
        w, sel = _weights(model, field_idx, channel_weights, dev)
        sx_dtype = model._pick_dtype() if dtype is None else (L.SX_BF16 if dtype == torch.bfloat16 else L.SX_F32)
        p_all = _f32c(p0).clone()
        fit_all = torch.empty(N, dtype=torch.float32, device=dev)
        mae_all = torch.empty(N, dtype=torch.float32, device=dev)
        steps_all = torch.zeros(N, dtype=torch.int32, device=dev)
        hist = torch.full((steps, N), float("nan"), dtype=torch.float32, device=dev) if return_history else None
        lib = L.lib()
        prob = None
        if optimizer == "lbfgs":
            return _lbfgs_cases(model, mask3, tgt, p_all, w, sel, sx_dtype, steps, reg, clip_scaled, early_stop_patience,
                                early_stop_tol, fit_all, mae_all, steps_all, hist)
        for lo in range(0, N, batch):
            hi = min(N, lo + batch)
            B = hi - lo
            if prob is None or prob.B != B:
                prob = _Problem(model, B, H, W, sx_dtype)
            else:
                prob.m.zero_(); prob.v.zero_(); prob.best.fill_(float("inf")); prob.wait.zero_(); prob.active.fill_(1)
            x = mask3[lo:hi].unsqueeze(1)
            p = p_all[lo:hi]                       # contiguous row block, updated in place
            t_, m_ = tgt[lo:hi], mask3[lo:hi]
            for step in range(steps):
                prob.forward_fit(x, p, t_, m_, w, sel, True)
                g = prob.backward_params(p)
                steps_all[lo:hi] += prob.active
                act_before = prob.active.bool() if hist is not None else None
                L.check(lib.sx_inverse_adam_step(B, model.P, L.ptr(p), L.ptr(g), L.ptr(prob.m), L.ptr(prob.v), L.ptr(prob.fit),
                                                 float(cosine_lr(lr, step, steps)), 0.9, 0.999, 1e-8, step + 1, float(reg),
                                                 float(clip_scaled or 0.0), int(early_stop_patience), float(early_stop_tol),
                                                 L.ptr(prob.best), L.ptr(prob.wait), L.ptr(prob.active), L.ptr(prob.loss),
                                                 L.stream_ptr()), "sx_inverse_adam_step")
                if hist is not None:
                    hist[step, lo:hi] = torch.where(act_before, prob.loss, hist[step, lo:hi])
                # every case of the chunk stopped: the reference would have left all of their loops
                if early_stop_patience > 0 and (step + 1) % 16 == 0 and int(prob.active.sum()) == 0:
                    break
            prob.forward_fit(x, p, t_, m_, w, sel, False)          # :413-416 / :437-439 at the recovered parameters
            fit_all[lo:hi] = prob.fit
            mae_all[lo:hi] = prob.mae
        out = {"p_scaled": p_all, "fit": fit_all, "cycle_mae_norm": mae_all, "steps_run": steps_all}
        if hist is not None:
            out["loss_history"] = hist
        return out
    finally:
        model.train(was_training)


def _lbfgs_cases(model, mask3, tgt, p_all, w, sel, sx_dtype, steps, reg, clip, patience, tol, fit_all, mae_all, steps_all, hist):
    """Reference loop :336-416 with use_lbfgs: per case, torch.optim.LBFGS(lr 1, max_iter 20, strong Wolfe) around a device
    closure.  As in the reference, the value logged / early-stopped on is the LAST closure evaluation of each step."""
    N, H, W = mask3.shape
    P = model.P
    prob = _Problem(model, 1, H, W, sx_dtype)
    for n in range(N):
        x, t_, m_ = mask3[n:n + 1].unsqueeze(1), tgt[n:n + 1], mask3[n:n + 1]
        p_var = torch.nn.Parameter(p_all[n:n + 1].clone())
        opt = torch.optim.LBFGS([p_var], lr=1.0, max_iter=20, line_search_fn="strong_wolfe")
        last = [0.0]

        def closure():
            p = p_var.data
            prob.forward_fit(x, p, t_, m_, w, sel, True)
            g = prob.backward_params(p)
            # loss = fit + reg * mean(p^2)  (:364-366); its gradient in p
            p_var.grad = g + (2.0 * reg / P) * p
            loss = prob.fit[0] + reg * (p * p).mean()
            last[0] = float(loss)
            return loss
        best, wait, ran = float("inf"), 0, 0
        for step in range(steps):
            opt.step(closure)
            if clip is not None and clip > 0:
                p_var.data.clamp_(-float(clip), float(clip))
            ran += 1
            if hist is not None:
                hist[step, n] = last[0]
            if patience > 0:
                rel = (best - last[0]) / max(abs(best), 1e-12)       # NaN while best is inf, as in the reference
                if rel > tol:
                    best, wait = last[0], 0
                else:
                    wait += 1
                    if wait >= patience:
                        break
        p_all[n:n + 1] = p_var.data
        prob.forward_fit(x, p_all[n:n + 1], t_, m_, w, sel, False)
        fit_all[n:n + 1] = prob.fit
        mae_all[n:n + 1] = prob.mae
        steps_all[n] = ran
    out = {"p_scaled": p_all, "fit": fit_all, "cycle_mae_norm": mae_all, "steps_run": steps_all}
    if hist is not None:
        out["loss_history"] = hist
    return out


def inverse_with_restarts(model, mask, y_true_norm, p0_restarts, **kw):
    """Best-of-R restarts (:316-422): p0_restarts (R,N,P); keeps, per case, the restart with the smallest final fit."""
    best = None
    for r in range(p0_restarts.shape[0]):
        res = batched_inverse(model, mask, y_true_norm, p0_restarts[r], **kw)
        if best is None:
            best = res
        else:
            better = res["fit"] < best["fit"]
            for k in ("p_scaled", "fit", "cycle_mae_norm", "steps_run"):
                sel = better if best[k].dim() == 1 else better[:, None]
                best[k] = torch.where(sel, res[k], best[k])
    return best


def batched_inverse_sharded(model, mask, y_true_norm, p0, **kw):
    """config 4 across ranks: cases are independent, so each rank takes a contiguous slice and the rows are all-gathered at the end
    (no data-path collective)."""
    rank, world = sxdist.rank(), sxdist.world_size()
    lo, hi = sxdist.shard_range(p0.shape[0], rank, world)
    res = batched_inverse(model, mask[lo:hi], y_true_norm[lo:hi], p0[lo:hi], **kw)
    out = {}
    for k in ("p_scaled", "fit", "cycle_mae_norm", "steps_run"):
        v = res[k] if res[k].dim() == 2 else res[k][:, None]
        g = sxdist.gather_rows(v.float(), p0.shape[0])          # every rank gets the full table
        out[k] = g if res[k].dim() == 2 else g[:, 0]
    return out


def masked_stats_np(y_pred, y_true, mask):
    """Host-side float64 masked MAE / RMSE of the reference's report (:131-141)."""
    m = np.asarray(mask) > 0.5
    if not np.any(m):
        return float("nan"), float("nan")
    d = (np.asarray(y_pred, dtype=np.float64) - np.asarray(y_true, dtype=np.float64))[:, m]
    return float(np.mean(np.abs(d))), float(np.sqrt(np.mean(d * d)))
