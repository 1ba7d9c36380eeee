"""MI355X-native mirror of `solpex.predict` (reference: solpex/predict.py).

`load_checkpoint`, `predict_fields`, `predict_te`, `scale_params` keep the reference signatures, return
values, printed warnings and exceptions.  New: `predict_fields_batch` runs N
(mask, params) pairs in one HIP forward and de-normalises on the device.
"""
from __future__ import annotations

import numpy as np
import torch

from .data import (MaskedLinearStandardizer, MaskedLogStandardizer, MaskedSymLogStandardizer, MultiChannelNormalizer,
                   apply_param_transform, invert_param_transform)  # noqa: F401  (re-exported like the reference)


def scale_params(params, mu, std):
    """Reference: solpex/predict.py:17-19."""
    p = np.asarray(params, dtype=np.float32)
    if mu is None or std is None:
        return p
    return (p - np.asarray(mu, dtype=np.float32)) / np.asarray(std, dtype=np.float32)


def _norm_from_ckpt(norm_ckpt):
    """Rebuild a normalizer from its checkpoint form (reference: solpex/predict.py:21-54), including the
    legacy (mu, sigma, eps) tuple."""
    if isinstance(norm_ckpt, (tuple, list)) and len(norm_ckpt) == 3:
        mu, sigma, eps = norm_ckpt
        n = MaskedLogStandardizer(eps=float(eps))
        n.mu, n.sigma = mu, sigma
        return n
    if not isinstance(norm_ckpt, dict):
        raise ValueError("Unsupported norm checkpoint format.")
    kind = norm_ckpt.get("kind", "")
    if kind == "MultiChannelNormalizer":
        subs = {k: _norm_from_ckpt(v) for k, v in norm_ckpt.get("norms", {}).items()}
        return MultiChannelNormalizer(y_keys=norm_ckpt.get("y_keys", []), norms_by_name=subs)
    if kind == "MaskedLogStandardizer":
        n = MaskedLogStandardizer(eps=float(norm_ckpt.get("eps", 1.0)))
    elif kind == "MaskedLinearStandardizer":
        n = MaskedLinearStandardizer(eps=float(norm_ckpt.get("eps", 1e-12)))
    elif kind == "MaskedSymLogStandardizer":
        n = MaskedSymLogStandardizer(scale=float(norm_ckpt.get("scale", 1.0)), eps=float(norm_ckpt.get("eps", 1e-12)))
    else:
        raise ValueError(f"Unsupported normalizer kind: {kind!r}")
    n.mu, n.sigma = norm_ckpt.get("mu"), norm_ckpt.get("sigma")
    return n


def _get_model_in_channels(model):
    """Reference: solpex/predict.py:56-62 looks for a `model.enc1` that current models do not have and so
    returns None; mirrored (the in-channel check of predict_fields is therefore performed by the UNet itself)."""
    enc1 = getattr(model, "enc1", None)
    if enc1 is not None and hasattr(enc1, "net") and len(enc1.net) > 0 and hasattr(enc1.net[0], "in_channels"):
        return int(enc1.net[0].in_channels)
    return None


def load_checkpoint(path, device):
    """Reference: solpex/predict.py:64-94.  NB: like the reference, `depth` is NOT read from the checkpoint."""
    ckpt = torch.load(path, map_location=device, weights_only=False)
    from .models import UNet
    model = UNet(in_ch=ckpt.get("in_ch", 1), out_ch=ckpt.get("out_ch", 1), base=ckpt.get("base", 2),
                 dropout=ckpt.get("dropout", 0.0), P=ckpt.get("P", 0), film_hidden=ckpt.get("film_hidden", 128),
                 z_dim=ckpt.get("z_dim", 0)).to(device)
    missing, unexpected = model.load_state_dict(ckpt["model"], strict=False)
    if missing:
        print(f"[load_checkpoint] WARNING: missing keys: {missing}")
    if unexpected:
        print(f"[load_checkpoint] WARNING: unexpected keys: {unexpected}")
    norm = _norm_from_ckpt(ckpt["norm"])
    if "param_mu" in ckpt and "param_std" in ckpt:
        param_scaler = (ckpt.get("param_mu"), ckpt.get("param_std"))
    else:
        param_scaler = ckpt.get("param_scaler", (None, None))
    return model, norm, param_scaler, ckpt.get("param_transform", None), ckpt.get("param_keys", None)


def _build_inputs(mask, Rn, Zn, device):
    m = torch.from_numpy(mask).float() if isinstance(mask, np.ndarray) else mask.float()
    if m.dim() == 2:
        m = m[None, None]
    elif m.dim() == 3:
        m = m.unsqueeze(0)
    m = (m > 0.5).float().to(device)
    H, W = m.shape[-2:]
    chans = [m]
    if Rn is not None and Zn is not None:
        R = torch.from_numpy(np.asarray(Rn, np.float32))[None, None].to(device)
        Z = torch.from_numpy(np.asarray(Zn, np.float32))[None, None].to(device)
        if R.shape[-2:] != (H, W) or Z.shape[-2:] != (H, W):
            raise ValueError(f"Rn/Zn shape mismatch: got {R.shape[-2:]},{Z.shape[-2:]}, expected {(H,W)}")
        chans += [R.expand(m.shape[0], 1, H, W), Z.expand(m.shape[0], 1, H, W)]
    return m, torch.cat(chans, dim=1)


def predict_fields(model, norm, mask, params, Rn=None, Zn=None, device=None, as_numpy=True):
    """Reference: solpex/predict.py:96-141.  Single (mask, params) -> (C,H,W) float32 in physical units."""
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    m, x = _build_inputs(mask, Rn, Zn, device)
    p = None
    if params is not None:
        p = torch.from_numpy(np.asarray(params, dtype=np.float32).ravel()).unsqueeze(0).to(device)
    expected = _get_model_in_channels(model)
    if expected is not None and x.shape[1] != expected:
        raise RuntimeError(f"Model expects in_ch={expected}, but you built C={x.shape[1]}.")
    with torch.no_grad():
        y = norm.inverse(model(x, params=p), m)
    if not as_numpy:
        return y
    out = y.squeeze(0).detach().cpu().numpy().astype(np.float32)
    return out[None, ...] if out.ndim == 2 else out


def predict_te(model, norm, mask, params, Rn=None, Zn=None, device=None, as_numpy=True):
    """Reference: solpex/predict.py:144-155: channel 0 of predict_fields."""
    out = predict_fields(model=model, norm=norm, mask=mask, params=params, Rn=Rn, Zn=Zn, device=device, as_numpy=as_numpy)
    return out[0] if as_numpy else out[:, 0:1]


def predict_fields_batch(model, norm, masks, params, Rn=None, Zn=None, device=None, chunk=1024, as_numpy=True):
    """N cases in one go: masks (N,H,W) or (H,W) shared, params (N,P) scaled.  Returns (N,C,H,W) float32
    physical fields.  Cases are independent (GroupNorm is per-sample), so this equals N predict_fields calls."""
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    pr = torch.as_tensor(np.asarray(params, dtype=np.float32)) if not torch.is_tensor(params) else params.float()
    N = pr.shape[0]
    mk = torch.as_tensor(np.asarray(masks, dtype=np.float32)) if not torch.is_tensor(masks) else masks.float()
    if mk.dim() == 2:
        mk = mk[None].expand(N, -1, -1)
    outs = []
    with torch.no_grad():
        for s in range(0, N, chunk):
            m, x = _build_inputs_batch(mk[s:s + chunk], Rn, Zn, device)
            z = model(x, params=pr[s:s + chunk].to(device))
            y = norm.inverse(z, m)
            outs.append(y.cpu() if as_numpy else y)
    y = torch.cat(outs, dim=0)
    return y.numpy().astype(np.float32) if as_numpy else y


def _build_inputs_batch(mk, Rn, Zn, device):
    m = (mk[:, None] > 0.5).float().to(device).contiguous()
    n, _, H, W = m.shape
    chans = [m]
    if Rn is not None and Zn is not None:
        R = torch.from_numpy(np.asarray(Rn, np.float32))[None, None].to(device)
        Z = torch.from_numpy(np.asarray(Zn, np.float32))[None, None].to(device)
        if R.shape[-2:] != (H, W) or Z.shape[-2:] != (H, W):
            raise ValueError(f"Rn/Zn shape mismatch: got {R.shape[-2:]},{Z.shape[-2:]}, expected {(H,W)}")
        chans += [R.expand(n, 1, H, W), Z.expand(n, 1, H, W)]
    return m, torch.cat(chans, dim=1)
