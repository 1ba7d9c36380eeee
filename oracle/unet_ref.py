"""Functional CPU restatement of the FiLM-conditioned residual U-Net.

Follows /root/reference/solpex/models.py:
  DoubleConv        :61-80   (conv3-GN-SiLU-conv3-GN-SiLU + 1x1 shortcut)
  FiLMGenerator     :38-58   (Linear-SiLU-Linear -> (1+gamma, beta))
  UNet.__init__     :98-150  (channel plan, parameter names)
  UNet.encode       :173-184 (pool between levels, FiLM before the skip is stored)
  UNet.decode       :186-203 (convT, pad/crop, cat[h, skip], DoubleConv, dropout, FiLM, 1x1 head)
  latent helpers    :10-12, :32-35, :158-171

The network is expressed as pure functions of a state dict (same key names and
shapes as the reference's `state_dict()`), so one set of weights can be pushed
through this oracle and through the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass
import math
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class UNetCfg:
    in_ch: int = 1
    out_ch: int = 1
    base: int = 32
    groups_gn: int = 8
    dropout: float = 0.0
    P: int = 0
    film_hidden: int = 128
    z_dim: int = 0
    depth: int = 3

    @property
    def enc_channels(self):
        return [self.base * (2 ** i) for i in range(self.depth)]

    @property
    def bot_ch(self):
        return self.base * (2 ** self.depth)


def _block_names(prefix):
    return [f"{prefix}.net.0.weight", f"{prefix}.net.1.weight", f"{prefix}.net.1.bias",
            f"{prefix}.net.3.weight", f"{prefix}.net.4.weight", f"{prefix}.net.4.bias",
            f"{prefix}.shortcut.weight"]


def param_shapes(cfg: UNetCfg):
    """Ordered {name: shape}, identical to the reference state_dict (models.py:108-150)."""
    shapes = {}

    def block(prefix, cin, cout):
        n = _block_names(prefix)
        shapes[n[0]] = (cout, cin, 3, 3)
        shapes[n[1]] = (cout,)
        shapes[n[2]] = (cout,)
        shapes[n[3]] = (cout, cout, 3, 3)
        shapes[n[4]] = (cout,)
        shapes[n[5]] = (cout,)
        if cin != cout:
            shapes[n[6]] = (cout, cin, 1, 1)

    enc = cfg.enc_channels
    cin = cfg.in_ch
    for i, c in enumerate(enc):
        block(f"encoders.{i}", cin, c)
        cin = c
    block("bot", enc[-1], cfg.bot_ch)
    dec_in = cfg.bot_ch
    ups, decs = {}, {}
    for j, i in enumerate(reversed(range(cfg.depth))):
        c = enc[i]
        shapes[f"up_convs.{j}.weight"] = (dec_in, c, 2, 2)
        shapes[f"up_convs.{j}.bias"] = (c,)
        dec_in = c
    for j, i in enumerate(reversed(range(cfg.depth))):
        block(f"decoders.{j}", 2 * enc[i], enc[i])
    shapes["out.weight"] = (cfg.out_ch, enc[0], 1, 1)
    shapes["out.bias"] = (cfg.out_ch,)
    if cfg.P > 0:
        def film(prefix, c):
            shapes[f"{prefix}.net.0.weight"] = (cfg.film_hidden, cfg.P)
            shapes[f"{prefix}.net.0.bias"] = (cfg.film_hidden,)
            shapes[f"{prefix}.net.2.weight"] = (2 * c, cfg.film_hidden)
            shapes[f"{prefix}.net.2.bias"] = (2 * c,)
        for i, c in enumerate(enc):
            film(f"film_enc.{i}", c)
        film("film_bot", cfg.bot_ch)
        for j, i in enumerate(reversed(range(cfg.depth))):
            film(f"film_dec.{j}", enc[i])
    if cfg.z_dim > 0:
        shapes["z_proj.weight"] = (cfg.z_dim, cfg.bot_ch)
        shapes["z_proj.bias"] = (cfg.z_dim,)
        shapes["z_unproj.weight"] = (cfg.bot_ch, cfg.z_dim)
        shapes["z_unproj.bias"] = (cfg.bot_ch,)
    return shapes


def init_state_dict(cfg: UNetCfg, seed: int = 0, live_film: bool = True, dtype=torch.float32):
    """Seeded random weights of the right shapes.  Not the reference's init
    distribution bit-for-bit (that needs the reference's module construction
    order); fan-in scaled uniform like torch's default, GroupNorm affine and the
    FiLM output layer perturbed away from identity when `live_film`."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("net.1.weight") or name.endswith("net.4.weight"):
            if "film" in name:
                pass
            else:
                t = torch.ones(shp, dtype=dtype)
                if live_film:
                    t = t + 0.1 * torch.randn(shp, generator=g, dtype=dtype)
                sd[name] = t
                continue
        if (name.endswith("net.1.bias") or name.endswith("net.4.bias")) and "film" not in name:
            t = torch.zeros(shp, dtype=dtype)
            if live_film:
                t = 0.1 * torch.randn(shp, generator=g, dtype=dtype)
            sd[name] = t
            continue
        if "film" in name and ".net.2." in name:
            sd[name] = (0.05 * torch.randn(shp, generator=g, dtype=dtype)) if live_film else torch.zeros(shp, dtype=dtype)
            continue
        if len(shp) == 1:
            fan_in = None
            # bias of conv/linear: bound 1/sqrt(fan_in) of the matching weight
            wname = name[:-4] + "weight"
            wshape = param_shapes(cfg)[wname]
            if name.startswith("up_convs"):
                fan_in = wshape[1] * wshape[2] * wshape[3]
            else:
                fan_in = int(torch.tensor(wshape[1:]).prod().item())
            bound = 1.0 / math.sqrt(fan_in)
        else:
            if name.startswith("up_convs"):
                fan_in = shp[1] * shp[2] * shp[3]
            else:
                fan_in = int(torch.tensor(shp[1:]).prod().item())
            bound = 1.0 / math.sqrt(fan_in)
        sd[name] = (torch.rand(shp, generator=g, dtype=dtype) * 2 - 1) * bound
    return sd


# ---------------------------------------------------------------- blocks
def film_generator(sd, prefix, params, C):
    """models.py:52-58 -> (1+gamma, beta), each (B,C,1,1)."""
    h = F.silu(F.linear(params, sd[f"{prefix}.net.0.weight"], sd[f"{prefix}.net.0.bias"]))
    o = F.linear(h, sd[f"{prefix}.net.2.weight"], sd[f"{prefix}.net.2.bias"])
    gamma, beta = o[:, :C], o[:, C:]
    return (1.0 + gamma)[:, :, None, None], beta[:, :, None, None]


def double_conv(sd, prefix, x, groups_gn):
    """models.py:61-80."""
    w1 = sd[f"{prefix}.net.0.weight"]
    cout = w1.shape[0]
    g = min(groups_gn, cout)
    h = F.conv2d(x, w1, None, padding=1)
    h = F.silu(F.group_norm(h, g, sd[f"{prefix}.net.1.weight"], sd[f"{prefix}.net.1.bias"], eps=1e-5))
    h = F.conv2d(h, sd[f"{prefix}.net.3.weight"], None, padding=1)
    h = F.silu(F.group_norm(h, g, sd[f"{prefix}.net.4.weight"], sd[f"{prefix}.net.4.bias"], eps=1e-5))
    sk = f"{prefix}.shortcut.weight"
    s = F.conv2d(x, sd[sk], None) if sk in sd else x
    return h + s


def _film(sd, cfg, prefix, h, params):
    if params is None or cfg.P <= 0:
        return h
    g1, b = film_generator(sd, prefix, params, h.shape[1])
    return h * g1 + b


def unet_encode(sd, cfg: UNetCfg, x, params=None):
    """models.py:173-184."""
    skips = []
    h = x
    for i in range(cfg.depth):
        if i > 0:
            h = F.max_pool2d(h, 2)
        h = double_conv(sd, f"encoders.{i}", h, cfg.groups_gn)
        h = _film(sd, cfg, f"film_enc.{i}", h, params)
        skips.append(h)
    b = double_conv(sd, "bot", F.max_pool2d(h, 2), cfg.groups_gn)
    b = _film(sd, cfg, "film_bot", b, params)
    return b, skips


def unet_decode(sd, cfg: UNetCfg, b, skips, params=None, drop_masks=None):
    """models.py:186-203.  `drop_masks[j]` (B,C,1,1), already scaled by
    1/(1-p), stands in for Dropout2d in training mode (None = eval / p == 0)."""
    h = b
    for j in range(cfg.depth):
        skip = skips[cfg.depth - 1 - j]
        h = F.conv_transpose2d(h, sd[f"up_convs.{j}.weight"], sd[f"up_convs.{j}.bias"], stride=2)
        dh = skip.shape[2] - h.shape[2]
        dw = skip.shape[3] - h.shape[3]
        if dh > 0 or dw > 0:
            h = F.pad(h, [0, max(dw, 0), 0, max(dh, 0)])
        if dh < 0 or dw < 0:
            h = h[:, :, :skip.shape[2], :skip.shape[3]]
        h = double_conv(sd, f"decoders.{j}", torch.cat([h, skip], dim=1), cfg.groups_gn)
        if drop_masks is not None and drop_masks[j] is not None:
            h = h * drop_masks[j]
        h = _film(sd, cfg, f"film_dec.{j}", h, params)
    return F.conv2d(h, sd["out.weight"], sd["out.bias"])


def unet_forward(sd, cfg: UNetCfg, x, params=None, return_bottleneck=False, drop_masks=None):
    """models.py:205-208."""
    b, skips = unet_encode(sd, cfg, x, params)
    y = unet_decode(sd, cfg, b, skips, params, drop_masks)
    return (y, b) if return_bottleneck else y


# ---------------------------------------------------------------- latent helpers
def bottleneck_to_z(b):
    """models.py:10-12."""
    return b.mean(dim=(2, 3))


def project_z(sd, cfg, b):
    """models.py:158-163."""
    z = bottleneck_to_z(b)
    if cfg.z_dim > 0:
        z = F.linear(z, sd["z_proj.weight"], sd["z_proj.bias"])
    return z


def unproject_z(sd, cfg, z_low, b_shape):
    """models.py:165-171 / :32-35."""
    z = F.linear(z_low, sd["z_unproj.weight"], sd["z_unproj.bias"]) if cfg.z_dim > 0 else z_low
    B, C, h, w = b_shape
    return z.view(B, C, 1, 1).expand(B, C, h, w)
