"""Host-side driver of the whole-network C entry points (sx_unet_*).

`UNetSession` owns what one forward/backward of a given shape needs: the descriptor, the workspace
(all saved activations), the packed weights and the parameter / gradient pointer tables.  torch is
used here only as the device allocator and stream provider.
"""
from __future__ import annotations

import ctypes as C
import torch

from . import _lib as L


def param_names(in_ch, out_ch, base, depth, P, film_hidden=128):
    """Parameter names in the reference's state_dict order (solpex/models.py:108-142); the order of the
    pointer tables the C side expects (include/solpex_hip.h, "Whole-network entry points")."""
    def block(prefix):
        return [f"{prefix}.net.0.weight", f"{prefix}.net.1.weight", f"{prefix}.net.1.bias", f"{prefix}.net.3.weight",
                f"{prefix}.net.4.weight", f"{prefix}.net.4.bias", f"{prefix}.shortcut.weight"]
    names = []
    for i in range(depth):
        names += block(f"encoders.{i}")
    names += block("bot")
    for j in range(depth):
        names += [f"up_convs.{j}.weight", f"up_convs.{j}.bias"]
    for j in range(depth):
        names += block(f"decoders.{j}")
    names += ["out.weight", "out.bias"]
    if P > 0:
        def film(prefix):
            return [f"{prefix}.net.0.weight", f"{prefix}.net.0.bias", f"{prefix}.net.2.weight", f"{prefix}.net.2.bias"]
        for i in range(depth):
            names += film(f"film_enc.{i}")
        names += film("film_bot")
        for j in range(depth):
            names += film(f"film_dec.{j}")
    return names


def encoder_param_mask(names):
    """True for parameters the encode half owns (encoders, bot, film_enc, film_bot)."""
    return [n.startswith(("encoders.", "bot.", "film_enc.", "film_bot.")) for n in names]


_TORCH_DT = {L.SX_F32: torch.float32, L.SX_BF16: torch.bfloat16}


class UNetSession:
    """One (config, batch shape, dtype) instance of the HIP U-Net: workspace + packed weights."""

    def __init__(self, cfg, B, H, W, dtype, use_film, device, wpack=None, film_shared=False):
        self.lib = L.lib()
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.dtype = dtype
        self.tdtype = _TORCH_DT[dtype]
        self.esz = 2 if dtype == L.SX_BF16 else 4
        d = L.UNetDesc(in_ch=cfg["in_ch"], out_ch=cfg["out_ch"], base=cfg["base"], depth=cfg["depth"],
                       groups_gn=cfg["groups_gn"], P=cfg["P"], film_hidden=cfg["film_hidden"], B=B, H=H, W=W,
                       dtype=dtype, use_film=(2 if film_shared else 1) if use_film else 0)
        self.desc = d
        self.B, self.H, self.W = B, H, W
        self.use_film = bool(use_film)
        n = self.lib.sx_unet_num_params(C.byref(d))
        if n < 0:
            L.check(n, "UNet (HIP)")
        self.nparams = n
        ws_bytes = L.check(self.lib.sx_unet_workspace_bytes(C.byref(d)))
        wp_bytes = L.check(self.lib.sx_unet_wpack_bytes(C.byref(d)))
        self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        if wpack is not None and wpack.numel() == wp_bytes and wpack.device == self.device:
            self.wpack = wpack
        else:
            self.wpack = torch.empty(wp_bytes, dtype=torch.uint8, device=self.device)
        assert self.ws.data_ptr() % 256 == 0 and self.wpack.data_ptr() % 256 == 0
        self._views = {}
        # gradient buffers the caller may fill by a strided copy must have zero padding channels
        self.view(L.SX_T_DPRED, padded=True).zero_()
        self.names = param_names(cfg["in_ch"], cfg["out_ch"], cfg["base"], cfg["depth"], cfg["P"], cfg["film_hidden"])
        assert len(self.names) == n, (len(self.names), n)
        self._ptable = None
        self._ptable_key = None
        self._gtable = None
        self._gtable_key = None

    # ------------------------------------------------------------------ tensor views into the workspace
    def view(self, which, index=0, padded=False, fresh=False):
        """(B, C, H, W)-shaped strided torch view (NHWC memory) of a workspace tensor.
        `fresh` returns a new tensor object (for values handed to callers) instead of the cached one."""
        key = (which, index, padded)
        v = self._views.get(key)
        if v is not None:
            return v.view(v.shape) if fresh else v
        vw = L.View()
        L.check(self.lib.sx_unet_view(C.byref(self.desc), which, index, C.byref(vw)))
        if which == L.SX_T_FILM_DPARAMS:
            n = self.B * max(1, self.cfg["P"])
            t = self.ws[vw.offset_bytes: vw.offset_bytes + 4 * n].view(torch.float32).view(self.B, max(1, self.cfg["P"]))
            t = t[:, :self.cfg["P"]]
        else:
            ld, Cc, H, W = vw.ld, vw.C, vw.H, vw.W
            # the view may be a channel slice of a wider buffer: build it from the base with as_strided
            base = self.ws[vw.offset_bytes:].view(self.tdtype)
            Cv = ld if (padded and which in (L.SX_T_PRED, L.SX_T_DPRED, L.SX_T_INPUT, L.SX_T_DINPUT)) else Cc
            t = base.as_strided((self.B, Cv, H, W), (H * W * ld, 1, W * ld, ld))
        self._views[key] = t
        return t.view(t.shape) if fresh else t

    def head_bias_sums(self):
        """fp32 [SX_HEAD_BIAS_SLOTS * out_ch] slots of the output head's bias gradient (SX_T_HEAD_BIAS_SUMS)."""
        v = self._views.get("head_bias")
        if v is None:
            vw = L.View()
            L.check(self.lib.sx_unet_view(C.byref(self.desc), L.SX_T_HEAD_BIAS_SUMS, 0, C.byref(vw)))
            n = L.SX_HEAD_BIAS_SLOTS * vw.C
            v = self.ws[vw.offset_bytes: vw.offset_bytes + 4 * n].view(torch.float32)
            self._views["head_bias"] = v
        return v

    # ------------------------------------------------------------------ pointer tables
    def set_params(self, tensors):
        key = tuple(t.data_ptr() for t in tensors)
        if key != self._ptable_key:
            for t in tensors:
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                    raise ValueError("UNet (HIP): parameters must be contiguous fp32 tensors on the model's device")
            self._ptable = L.ptr_table(tensors)
            self._ptable_key = key
        return self._ptable

    def set_grads(self, tensors):
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        if key != self._gtable_key:
            self._gtable = L.ptr_table(tensors)
            self._gtable_key = key
        return self._gtable

    # ------------------------------------------------------------------ calls
    def pack_weights(self):
        L.check(self.lib.sx_unet_pack_weights(C.byref(self.desc), self._ptable, L.ptr(self.wpack), L.stream_ptr()),
                "sx_unet_pack_weights")

    def encode(self, x, film_params):
        sb, sc, sh, sw = x.stride()
        L.check(self.lib.sx_unet_encode(C.byref(self.desc), self._ptable, L.ptr(self.wpack), L.ptr(self.ws), L.ptr(x),
                                        sb, sc, sh, sw, L.ptr(film_params), L.stream_ptr()), "sx_unet_encode")

    def pack_encode(self, x, film_params):
        """pack_weights() + encode() in one call: one launch for the weight packs, the input pack and the FiLM generators."""
        sb, sc, sh, sw = x.stride()
        L.check(self.lib.sx_unet_pack_encode(C.byref(self.desc), self._ptable, L.ptr(self.wpack), L.ptr(self.ws), L.ptr(x),
                                             sb, sc, sh, sw, L.ptr(film_params), L.stream_ptr()), "sx_unet_pack_encode")

    def decode(self, film_params, drop_table=None):
        L.check(self.lib.sx_unet_decode(C.byref(self.desc), self._ptable, L.ptr(self.wpack), L.ptr(self.ws),
                                        L.ptr(film_params), drop_table, L.stream_ptr()), "sx_unet_decode")

    def decode_backward(self, film_params, drop_table, flags):
        L.check(self.lib.sx_unet_decode_backward(C.byref(self.desc), self._ptable, self._gtable, L.ptr(self.wpack),
                                                 L.ptr(self.ws), L.ptr(film_params), drop_table, flags, L.stream_ptr()),
                "sx_unet_decode_backward")

    def encode_backward(self, film_params, flags):
        L.check(self.lib.sx_unet_encode_backward(C.byref(self.desc), self._ptable, self._gtable, L.ptr(self.wpack),
                                                 L.ptr(self.ws), L.ptr(film_params), flags, L.stream_ptr()),
                "sx_unet_encode_backward")
