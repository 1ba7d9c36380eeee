"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's latent MLPs and trainers.

Follows solpex/latent.py: ParamToZ :14-30 / ZToParam :31-48 (Linear -> SiLU [-> LayerNorm] [-> Dropout] per hidden
width, then Linear), train_param2z :51-121, train_z2param :124-199, train_cycle_consistent :202-310 (SmoothL1 beta
0.05, clip_grad_norm_ 1.0, AdamW, shuffled DataLoader, sample-weighted epoch means).  Functional and state-dict driven:
keys `net.{i}.weight/bias` exactly as nn.Sequential numbers them.  Pinned by tests/golden/latent_mlp.npz (generated
from the live reference by tools/make_golden.py latent_mlp_case).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, TensorDataset

from . import optim_ref

BETA = 0.05


def layer_plan(sd):
    """[(linear index, layernorm index or None)] in forward order, from the state-dict keys alone."""
    idx = sorted({int(k.split(".")[1]) for k in sd})
    lin = [i for i in idx if sd[f"net.{i}.weight"].dim() == 2]
    plan = []
    for i in lin:
        ln = i + 2 if (f"net.{i + 2}.weight" in sd and sd[f"net.{i + 2}.weight"].dim() == 1) else None
        plan.append((i, ln))
    return plan


def mlp_forward(sd, x, drop_masks=None):
    """drop_masks: optional list (one per hidden block) of already scaled keep masks."""
    plan = layer_plan(sd)
    h = x
    for j, (i, ln) in enumerate(plan):
        h = F.linear(h, sd[f"net.{i}.weight"], sd[f"net.{i}.bias"])
        if j == len(plan) - 1:
            break
        h = F.silu(h)
        if ln is not None:
            h = F.layer_norm(h, (h.shape[-1],), sd[f"net.{ln}.weight"], sd[f"net.{ln}.bias"], 1e-5)
        if drop_masks is not None and drop_masks[j] is not None:
            h = h * drop_masks[j]
    return h


def smooth_l1(pred, tgt):
    d = pred - tgt
    ad = d.abs()
    return torch.where(ad < BETA, 0.5 * d * d / BETA, ad - 0.5 * BETA).mean()


class _Opt:
    """clip_grad_norm_(1.0) + AdamW over a dict of leaf tensors (restated in optim_ref)."""

    def __init__(self, sds, lr, wd):
        self.t = [v for sd in sds for v in sd.values()]
        self.m = [torch.zeros_like(v) for v in self.t]
        self.v = [torch.zeros_like(v) for v in self.t]
        self.lr, self.wd, self.k = lr, wd, 0

    def step(self):
        self.k += 1
        gs, _ = optim_ref.clip_grad_norm([t.grad for t in self.t], 1.0)
        with torch.no_grad():
            for i, (t, g) in enumerate(zip(self.t, gs)):
                pn, self.m[i], self.v[i] = optim_ref.adamw_step(t, g, self.m[i], self.v[i], self.k, self.lr, weight_decay=self.wd)
                t.copy_(pn)
                t.grad = None


def _leaf(sd):
    return {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}


def train_single(sd, a_tr, b_tr, a_va, b_va, *, lr=1e-3, epochs=200, batch_size=256, wd=1e-4):
    """train_param2z (a = P, b = Z) / train_z2param (a = Z, b = P).  Returns (final sd, hist)."""
    sd = _leaf(sd)
    tr = DataLoader(TensorDataset(torch.from_numpy(a_tr), torch.from_numpy(b_tr)), batch_size=batch_size, shuffle=True)
    va = DataLoader(TensorDataset(torch.from_numpy(a_va), torch.from_numpy(b_va)), batch_size=batch_size, shuffle=False)
    opt = _Opt([sd], lr, wd)
    hist = {"tr": [], "va": []}
    for _ in range(epochs):
        s, n = 0.0, 0
        for a, b in tr:
            loss = smooth_l1(mlp_forward(sd, a), b)
            loss.backward()
            opt.step()
            s += float(loss.item()) * a.shape[0]
            n += a.shape[0]
        hist["tr"].append(s / max(n, 1))
        s, n = 0.0, 0
        with torch.no_grad():
            for a, b in va:
                s += float(smooth_l1(mlp_forward(sd, a), b).item()) * a.shape[0]
                n += a.shape[0]
        hist["va"].append(s / max(n, 1))
    return {k: v.detach() for k, v in sd.items()}, hist


def train_cycle(sd_p2z, sd_z2p, Z_tr, P_tr, Z_va, P_va, *, lr=1e-3, epochs=400, batch_size=64, wd=1e-4, lam_cycle=0.1):
    p2z, z2p = _leaf(sd_p2z), _leaf(sd_z2p)
    tr = DataLoader(TensorDataset(torch.from_numpy(Z_tr), torch.from_numpy(P_tr)), batch_size=batch_size, shuffle=True)
    va = DataLoader(TensorDataset(torch.from_numpy(Z_va), torch.from_numpy(P_va)), batch_size=batch_size, shuffle=False)
    opt = _Opt([p2z, z2p], lr, wd)                     # parameter order: p2z first (latent.py:227)
    hist = {"tr": [], "va": [], "tr_fwd": [], "tr_inv": [], "tr_cyc_pzp": [], "tr_cyc_zpz": []}

    def terms(z, p):
        return (smooth_l1(mlp_forward(z2p, z), p), smooth_l1(mlp_forward(p2z, p), z),
                smooth_l1(mlp_forward(z2p, mlp_forward(p2z, p)), p), smooth_l1(mlp_forward(p2z, mlp_forward(z2p, z)), z))
    for _ in range(epochs):
        sums, n = [0.0] * 5, 0
        for z, p in tr:
            t = terms(z, p)
            loss = t[0] + t[1] + lam_cycle * (t[2] + t[3])
            loss.backward()
            opt.step()
            bs = z.shape[0]
            for j, v in enumerate((loss,) + t):
                sums[j] += float(v.item()) * bs
            n += bs
        for j, k in enumerate(("tr", "tr_fwd", "tr_inv", "tr_cyc_pzp", "tr_cyc_zpz")):
            hist[k].append(sums[j] / max(n, 1))
        s, n = 0.0, 0
        with torch.no_grad():
            for z, p in va:
                t = terms(z, p)
                s += float((t[0] + t[1] + lam_cycle * (t[2] + t[3])).item()) * z.shape[0]
                n += z.shape[0]
        hist["va"].append(s / max(n, 1))
    return {k: v.detach() for k, v in p2z.items()}, {k: v.detach() for k, v in z2p.items()}, hist
