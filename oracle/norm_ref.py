"""CPU restatement of the output normalizers' pointwise maps.

Follows /root/reference/solpex/data.py:
  MaskedLogStandardizer.transform/inverse     :96-110
  MaskedLinearStandardizer.transform/inverse  :139-143
  MaskedSymLogStandardizer.transform/inverse  :157-163, :187-199
  MultiChannelNormalizer.transform/inverse    :212-223  (per-channel dispatch)
"""
from __future__ import annotations

from dataclasses import dataclass
import torch

KIND_LOG, KIND_LINEAR, KIND_SYMLOG = 0, 1, 2


@dataclass
class NormSpec:
    kind: int
    mu: float
    sigma: float
    eps: float = 1.0
    scale: float = 1.0


def _finite_or_zero(t):
    return torch.where(torch.isfinite(t), t, torch.zeros_like(t))


def _transform_1(s: NormSpec, y, m):
    if s.kind == KIND_LOG:
        y = torch.nan_to_num(y, nan=0.0, posinf=0.0, neginf=0.0)
        z = torch.log(torch.clamp(y, min=0) + s.eps)
        z = (z - torch.tensor(s.mu, dtype=torch.float32)) / (torch.tensor(s.sigma, dtype=torch.float32) + 1e-8)
        return _finite_or_zero(z) * m
    if s.kind == KIND_LINEAR:
        return (y - torch.tensor(s.mu, dtype=torch.float32)) / torch.tensor(s.sigma, dtype=torch.float32)
    if s.kind == KIND_SYMLOG:
        y = torch.nan_to_num(y)
        z = torch.sign(y) * torch.log1p(torch.abs(y) / s.scale)
        z = (z - torch.tensor(s.mu, dtype=torch.float32)) / torch.tensor(s.sigma, dtype=torch.float32)
        return _finite_or_zero(z) * m
    raise ValueError(s.kind)


def _inverse_1(s: NormSpec, z, m):
    mu = torch.tensor(s.mu, dtype=torch.float32)
    sg = torch.tensor(s.sigma, dtype=torch.float32)
    if s.kind == KIND_LOG:
        y = torch.exp(z * (sg + 1e-8) + mu) - s.eps
        y = _finite_or_zero(y)
        return y if m is None else y * m
    if s.kind == KIND_LINEAR:
        return z * sg + mu
    if s.kind == KIND_SYMLOG:
        t = z * sg + mu
        y = torch.sign(t) * s.scale * torch.expm1(torch.abs(t))
        y = _finite_or_zero(y)
        return y if m is None else y * m
    raise ValueError(s.kind)


def norm_transform(specs, y, m):
    return torch.cat([_transform_1(s, y[:, i:i + 1], m) for i, s in enumerate(specs)], dim=1)


def norm_inverse(specs, z, m):
    return torch.cat([_inverse_1(s, z[:, i:i + 1], m) for i, s in enumerate(specs)], dim=1)
