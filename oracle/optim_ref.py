"""CPU restatement of the optimizer step of the training loop.

Follows /root/reference/solpex/train.py:82 (torch.optim.AdamW, betas (0.9, 0.999),
eps 1e-8, decoupled weight decay) and :227-233 (clip_grad_norm_ to max-norm
`clip_grad` with torch's 1e-6 guard, then step), restated as explicit math.
"""
from __future__ import annotations

import torch


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: total L2 norm; scale = min(1, max_norm/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """One decoupled-weight-decay Adam update (torch.optim.AdamW, amsgrad=False).
    `step` is the 1-based step count.  Returns new (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
