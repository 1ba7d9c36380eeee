"""Shared helpers for the parity tests (oracle side lives here, never in the product)."""
import numpy as np
import torch

from oracle.unet_ref import UNetCfg, init_state_dict


def cfg_from_golden(g):
    return UNetCfg(in_ch=int(g["cfg_in_ch"]), out_ch=int(g["cfg_out_ch"]), base=int(g["cfg_base"]),
                   groups_gn=int(g["cfg_groups_gn"]), P=int(g["cfg_P"]), film_hidden=int(g["cfg_film_hidden"]),
                   z_dim=int(g["cfg_z_dim"]), depth=int(g["cfg_depth"]))


def sd_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def sd_from_golden(g, cfg):
    """Weights: stored in the fixture when small, else regenerated from the seed and checksum-verified."""
    keys = [k for k in g.files if k.startswith("w::")]
    if keys:
        return {k[3:]: torch.from_numpy(g[k]).clone() for k in keys}
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    assert abs(sd_checksum(sd) - float(g["sd_checksum"])) <= 1e-6 * float(g["sd_checksum"])
    return sd


def loss_kw_from_golden(g):
    kw = {}
    for k in g.files:
        if k.startswith("losskw_"):
            v = g[k]
            kw[k[7:]] = v.item() if v.dtype.kind in "fiu" else str(v)
    return kw


def t(a):
    return torch.from_numpy(np.asarray(a)).clone()


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_err(a, b, floor=1e-6):
    """max |a-b| / (|b| + floor*max|b|)-style scaled error used for elementwise checks."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    scale = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / scale)
