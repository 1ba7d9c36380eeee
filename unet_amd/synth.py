"""Seeded synthetic SOLPS-shaped inputs: annular mask, standardised
control parameters, masked normalised targets.  Host-side torch only; used by
bench.py, the golden-vector tool and the tests so that all three see the same data.

Mask shape follows the reference's edge annulus `inside_edge & ~inside_core`
(scripts/qc_build_dataset.py:233-241); x = mask when no geometry channels are
configured (scripts/run_conditional_unet_pipeline.py:217); cfg-5 style inputs
follow scripts/run_source_from_plasma_pipeline.py:118-123.
"""
from __future__ import annotations

import torch

CHANNEL_WEIGHTS_8 = [1.0, 1.0, 1.2, 1.5, 1.2, 1.2, 1.2, 1.2]   # scripts/run_full_workflow.sh:28


def annulus_mask(H, W, inner=0.45, outer=1.0, dtype=torch.float32):
    i = torch.arange(H, dtype=torch.float32)[:, None]
    j = torch.arange(W, dtype=torch.float32)[None, :]
    rho = torch.sqrt(((i - H / 2) / (0.47 * H)) ** 2 + ((j - W / 2) / (0.45 * W)) ** 2)
    return ((rho >= inner) & (rho <= outer)).to(dtype)


def make_batch(B, H, W, in_ch=1, out_ch=8, P=5, seed=1234, jitter_masks=False, empty_last=False):
    """Returns dict(x, y, mask, params) of CPU fp32 tensors in the loader's layout
    (solpex/data.py:386-391): x (B,in_ch,H,W), y (B,out_ch,H,W), mask (B,1,H,W), params (B,P)."""
    g = torch.Generator().manual_seed(seed)
    masks = []
    for b in range(B):
        inner = 0.45
        if jitter_masks:
            inner = 0.30 + 0.30 * float(torch.rand((), generator=g))
        masks.append(annulus_mask(H, W, inner=inner))
    mask = torch.stack(masks)[:, None]
    if empty_last and B > 1:
        mask[-1] = 0.0
    y = torch.randn(B, out_ch, H, W, generator=g) * mask
    params = torch.randn(B, max(P, 1), generator=g).clamp_(-3, 3)[:, :P] if P > 0 else torch.zeros(B, 0)
    chans = [mask]
    if in_ch > 1:
        n_field = in_ch - 1
        extra = torch.randn(B, n_field, H, W, generator=g) * mask
        chans.append(extra)
    x = torch.cat(chans, dim=1)
    return {"x": x.contiguous(), "y": y.contiguous(), "mask": mask.contiguous(), "params": params.contiguous()}
