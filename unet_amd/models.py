"""MI355X-native mirror of `solpex.models` (reference: solpex/models.py).

Same classes, constructor signatures, attribute names and state_dict keys/shapes as the reference, so
checkpoints move both ways; the arithmetic of `UNet.encode/decode/forward` runs in hand-written HIP
kernels through libsolpex_hip.so (no ATen convolution / GroupNorm / pooling calls, no CPU fallback).
The torch modules below are only parameter containers (they give identical names, shapes and the
reference's default initialisation).
"""
from __future__ import annotations

import ctypes as C
import torch
from torch import nn

from . import _lib as L
from ._engine import UNetSession, param_names, encoder_param_mask


def bottleneck_to_z(b: torch.Tensor) -> torch.Tensor:
    """(B,C,h,w) -> (B,C) global average (solpex/models.py:10-12) via sx_global_avg_pool."""
    if not b.is_cuda:
        raise RuntimeError("bottleneck_to_z: the HIP path needs a CUDA (ROCm) tensor")
    if b.requires_grad and torch.is_grad_enabled():
        return _AvgPoolFn.apply(b)
    return _avg_pool_impl(b)


def _nhwc_args(t):
    """Pass a (B,C,H,W) tensor to an NHWC kernel: zero-copy when it is already pixel-major."""
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if not (sc == 1 and sh == W * sw and sb == H * W * sw and sw >= Cc and t.dtype in (torch.float32, torch.bfloat16)):
        t = t.contiguous(memory_format=torch.channels_last)
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        sw = t.stride(3)
    return t, int(sw)


def _avg_pool_impl(b):
    t, ld = _nhwc_args(b.detach())
    B, Cc, H, W = t.shape
    z = torch.empty(B, Cc, dtype=torch.float32, device=t.device)
    dt = L.SX_BF16 if t.dtype == torch.bfloat16 else L.SX_F32
    L.check(L.lib().sx_global_avg_pool(dt, B, H * W, Cc, L.ptr(t), ld, L.ptr(z), L.stream_ptr()), "sx_global_avg_pool")
    return z.to(b.dtype)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b):
        ctx.shape = b.shape
        return _avg_pool_impl(b)

    @staticmethod
    def backward(ctx, dz):
        B, Cc, H, W = ctx.shape
        return (dz / float(H * W)).view(B, Cc, 1, 1).expand(B, Cc, H, W)


def z_to_bottleneck(z: torch.Tensor, b_shape) -> torch.Tensor:
    """(B,C) -> (B,C,h,w) broadcast view (solpex/models.py:32-35)."""
    B, Cc, h, w = b_shape
    return z.view(B, Cc, 1, 1).expand(B, Cc, h, w)


class ParamToZ(nn.Module):
    """Small MLP params -> z (solpex/models.py:14-30).  Outside the accelerated path: plain torch."""

    def __init__(self, P: int, z_dim: int, hidden=(256, 256), dropout=0.0, use_layernorm=False):
        super().__init__()
        layers, d = [], P
        for h in hidden:
            layers += [nn.Linear(d, h), nn.SiLU()]
            if use_layernorm:
                layers.append(nn.LayerNorm(h))
            if dropout and dropout > 0:
                layers.append(nn.Dropout(dropout))
            d = h
        layers.append(nn.Linear(d, z_dim))
        self.net = nn.Sequential(*layers)

    def forward(self, p):
        return self.net(p)


class FiLMGenerator(nn.Module):
    """One FiLM generator (solpex/models.py:38-58).  Inside UNet.encode/decode all generators of a network half are
    evaluated by one launch of the film kernels; a direct call runs the same two-layer MLP on its own (blocks.py)."""

    def __init__(self, P: int, n_channels: int, hidden: int = 128):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(P, hidden), nn.SiLU(inplace=True), nn.Linear(hidden, n_channels * 2))
        nn.init.zeros_(self.net[-1].weight)     # identity modulation at init (models.py:47-49)
        nn.init.zeros_(self.net[-1].bias)
        self.n_channels = n_channels

    def forward(self, params):
        """params (B, P) -> (1 + gamma, beta), each (B, C, 1, 1)"""
        from .blocks import film_forward
        return film_forward(self, params)


class DoubleConv(nn.Module):
    """conv3-GN-SiLU-conv3-GN-SiLU + 1x1 shortcut (solpex/models.py:61-80).  Inside UNet.encode/decode the block runs fused
    in the network plan; a direct call `block(x)` takes the same kernels one C-ABI call at a time (blocks.py)."""

    def __init__(self, in_ch, out_ch, groups_gn=8):
        super().__init__()
        g = min(groups_gn, out_ch)
        self.net = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False), nn.GroupNorm(g, out_ch), nn.SiLU(inplace=True),
            nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False), nn.GroupNorm(g, out_ch), nn.SiLU(inplace=True))
        self.shortcut = nn.Conv2d(in_ch, out_ch, 1, bias=False) if in_ch != out_ch else nn.Identity()

    def forward(self, x):
        from .blocks import double_conv_forward
        return double_conv_forward(self, x)


def _put(dst, src):
    """Make `dst` (a workspace view) hold `src`; free when `src` already is that memory."""
    if src is None:
        dst.zero_()
    elif not (src.data_ptr() == dst.data_ptr() and src.stride() == dst.stride() and src.dtype == dst.dtype):
        dst.copy_(src)


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sess, owner, x, film_params, *weights):
        sess.encode(x, film_params)
        ctx.sess, ctx.owner, ctx.film_params = sess, owner, film_params
        D = sess.cfg["depth"]
        # fresh tensor objects: outputs must not be the session's cached views (no reference cycles)
        return (sess.view(L.SX_T_BOTTLENECK, fresh=True),) + tuple(sess.view(L.SX_T_SKIP, l, fresh=True) for l in range(D))

    @staticmethod
    def backward(ctx, db, *dskips):
        sess, owner = ctx.sess, ctx.owner
        D = sess.cfg["depth"]
        _put(sess.view(L.SX_T_DBOTTLENECK), db)
        for l in range(D):
            _put(sess.view(L.SX_T_DSKIP, l), dskips[l])
        want_w = any(ctx.needs_input_grad[4:])
        want_x = ctx.needs_input_grad[2]
        flags = L.SX_BWD_ZERO_DPARAMS | (L.SX_BWD_WEIGHT_GRADS if want_w else 0) | (L.SX_BWD_INPUT_GRAD if want_x else 0)
        gviews = owner._grad_views(sess) if want_w else None
        sess.encode_backward(ctx.film_params, flags)
        dx = sess.view(L.SX_T_DINPUT).float().contiguous() if want_x else None
        dparams = None
        if ctx.film_params is not None and ctx.needs_input_grad[3]:
            dparams = sess.view(L.SX_T_FILM_DPARAMS).clone()
        wgrads = [None] * (len(ctx.needs_input_grad) - 4)
        if want_w:
            enc_idx = owner._enc_idx
            for k, i in enumerate(enc_idx):
                if ctx.needs_input_grad[4 + k]:
                    wgrads[k] = gviews[i]
        return (None, None, dx, dparams, *wgrads)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sess, owner, film_params, drop_masks, b, *rest):
        D = sess.cfg["depth"]
        # b and the skips are already in place inside sess.ws (UNet.decode guarantees it)
        drop_table = None
        if drop_masks is not None:
            drop_table = L.ptr_table(drop_masks)
        sess.decode(film_params, drop_table)
        ctx.sess, ctx.owner, ctx.film_params = sess, owner, film_params
        ctx.drop_masks, ctx.drop_table = drop_masks, drop_table
        return sess.view(L.SX_T_PRED, fresh=True)

    @staticmethod
    def backward(ctx, dy):
        sess, owner = ctx.sess, ctx.owner
        D = sess.cfg["depth"]
        _put(sess.view(L.SX_T_DPRED), dy)
        nfixed = 5
        want_w = any(ctx.needs_input_grad[nfixed + D:])
        flags = L.SX_BWD_ZERO_DPARAMS | (L.SX_BWD_WEIGHT_GRADS if want_w else 0)
        gviews = owner._grad_views(sess) if want_w else None
        sess.decode_backward(ctx.film_params, ctx.drop_table, flags)
        dparams = None
        if ctx.film_params is not None and ctx.needs_input_grad[2]:
            dparams = sess.view(L.SX_T_FILM_DPARAMS).clone()
        db = sess.view(L.SX_T_DBOTTLENECK)
        dskips = [sess.view(L.SX_T_DSKIP, l) for l in range(D)]
        wgrads = [None] * (len(ctx.needs_input_grad) - nfixed - D)
        if want_w:
            for k, i in enumerate(owner._dec_idx):
                if ctx.needs_input_grad[nfixed + D + k]:
                    wgrads[k] = gviews[i]
        return (None, None, dparams, None, db, *dskips, *wgrads)


class UNet(nn.Module):
    """FiLM-conditioned residual U-Net (reference: solpex/models.py:82-213), HIP-backed.

    Constructor arguments, attributes (`P`, `z_dim`, `depth`, `dropout`, `z_proj`, `z_unproj`, `encoders`, `bot`,
    `up_convs`, `decoders`, `out`, `film_*`) and methods match the reference.  Extra, optional:
    `compute_dtype` (None = fp32, or bf16 when called under `torch.autocast(dtype=torch.bfloat16)`;
    set to torch.bfloat16 / torch.float32 to force a tier).
    """

    def __init__(self, in_ch=1, out_ch=1, base=32, groups_gn=8, dropout=0.0, P: int = 0, film_hidden: int = 128,
                 z_dim: int = 0, depth: int = 3):
        super().__init__()
        self.dropout, self.P, self.z_dim, self.depth = dropout, P, z_dim, depth
        self.pool = nn.MaxPool2d(2)
        chans = [base * (2 ** i) for i in range(depth)]
        self.encoders = nn.ModuleList()
        c_prev = in_ch
        for c in chans:
            self.encoders.append(DoubleConv(c_prev, c, groups_gn))
            c_prev = c
        bot_ch = base * (2 ** depth)
        self.bot = DoubleConv(chans[-1], bot_ch, groups_gn)
        self.up_convs, self.decoders, self.dropouts = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        c_prev = bot_ch
        for i in reversed(range(depth)):
            self.up_convs.append(nn.ConvTranspose2d(c_prev, chans[i], 2, 2))
            self.decoders.append(DoubleConv(2 * chans[i], chans[i], groups_gn))
            self.dropouts.append(nn.Dropout2d(dropout) if dropout > 0 else nn.Identity())
            c_prev = chans[i]
        self.out = nn.Conv2d(chans[0], out_ch, 1)
        if P > 0:
            self.film_enc = nn.ModuleList([FiLMGenerator(P, c, film_hidden) for c in chans])
            self.film_bot = FiLMGenerator(P, bot_ch, film_hidden)
            self.film_dec = nn.ModuleList([FiLMGenerator(P, chans[i], film_hidden) for i in reversed(range(depth))])
        if z_dim > 0:
            self.z_proj = nn.Linear(bot_ch, z_dim)
            self.z_unproj = nn.Linear(z_dim, bot_ch)
        else:
            self.z_proj = None
            self.z_unproj = None
        self._cfg = dict(in_ch=in_ch, out_ch=out_ch, base=base, depth=depth, groups_gn=groups_gn, P=P,
                         film_hidden=film_hidden)
        self.compute_dtype = None
        self._names = param_names(in_ch, out_ch, base, depth, P, film_hidden)
        mask = encoder_param_mask(self._names)
        self._enc_idx = [i for i, m in enumerate(mask) if m]
        self._dec_idx = [i for i, m in enumerate(mask) if not m]
        self._wcache = {}
        self._wgen = 0
        # in_ch == base makes the first block's shortcut an nn.Identity (reference: solpex/models.py:73-77): no parameter, no
        # state_dict entry.  The fused block kernels always run a 1x1 shortcut GEMM next to conv1, so the identity is fed to
        # them as a constant identity-matrix weight (exact in bf16 and fp32); its "gradient" is discarded.
        self._const_names = []
        if in_ch == base:
            self.register_buffer("_identity_shortcut", torch.eye(base, dtype=torch.float32).view(base, base, 1, 1).contiguous(),
                                 persistent=False)
            self._const_names.append("encoders.0.shortcut.weight")

    # ------------------------------------------------------------------ plumbing
    def _const_tensors(self):
        """Structural constants that sit in the kernels' parameter table but are no parameters of the module."""
        return {n: self._identity_shortcut for n in self._const_names}

    def _ordered_params(self):
        named = dict(self.named_parameters())
        named.update(self._const_tensors())
        return [named[n] for n in self._names]

    def _pick_dtype(self):
        cd = self.compute_dtype
        if cd is None and torch.is_autocast_enabled():
            cd = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            if cd != torch.bfloat16:
                raise ValueError("UNet (HIP): only bfloat16 autocast is supported (fp16 has no kernels)")
        if cd in (None, torch.float32):
            return L.SX_F32
        if cd == torch.bfloat16:
            return L.SX_BF16
        raise ValueError(f"UNet (HIP): unsupported compute dtype {cd}")

    def _new_session(self, B, H, W, dtype, use_film, device):
        params = self._ordered_params()
        # Packed (MFMA-layout) weights are cached per dtype/device and rebuilt when a parameter's
        # storage or version counter changed; in training mode they are rebuilt every forward.
        # `_wgen` counts updates made through raw pointers (TrainEngine's AdamW kernel), which bump no version counter.
        # fp32 packs also depend on the arithmetic mode they were made in (split-bf16 / exact, sx_set_fp32_exact).
        key = (self._wgen, L.lib().sx_set_fp32_planes(-1)) + tuple((p.data_ptr(), p._version) for p in params)
        ckey = (dtype, str(device))
        ent = self._wcache.get(ckey)
        sess = UNetSession(self._cfg, B, H, W, dtype, use_film, device, wpack=None if ent is None else ent[1])
        sess.set_params(params)
        if ent is None or ent[0] != key or self.training:
            sess.pack_weights()
            self._wcache[ckey] = (key, sess.wpack)
        sess._params_ref = params
        return sess

    def _grad_views(self, sess):
        """fp32 gradient buffers the kernels write; one flat allocation per session."""
        gv = getattr(sess, "_gviews", None)
        if gv is None:
            params = sess._params_ref
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, dtype=torch.float32, device=sess.device)
            gv, off = [], 0
            for p in params:
                gv.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            sess._gflat, sess._gviews = flat, gv
            sess.set_grads(gv)
        return gv

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("UNet (HIP): inputs must be CUDA (ROCm) tensors; this build has no CPU path")
        if x.dim() != 4 or x.shape[1] != self._cfg["in_ch"]:
            raise ValueError(f"UNet (HIP): expected input (B,{self._cfg['in_ch']},H,W), got {tuple(x.shape)}")
        return x if x.dtype == torch.float32 else x.float()

    def _film_params(self, params, B):
        if params is None or self.P <= 0:
            return None
        if params.dim() != 2 or params.shape != (B, self.P):
            raise ValueError(f"UNet (HIP): params must be (B,{self.P}), got {tuple(params.shape)}")
        p = params if params.dtype == torch.float32 else params.float()
        return p.contiguous()

    # ------------------------------------------------------------------ public API (reference: models.py:152-213)
    @staticmethod
    def _apply_film(features, film_layer, params):
        """features * gamma + beta (solpex/models.py:152-156); the fused path applies it inside the block-output kernel."""
        from .blocks import apply_film
        return apply_film(features, film_layer, params)

    def project_z(self, b):
        z = bottleneck_to_z(b)
        if self.z_proj is not None:
            z = self.z_proj(z.to(self.z_proj.weight.dtype))
        return z

    def unproject_z(self, z_low, b_shape):
        z_full = self.z_unproj(z_low) if self.z_unproj is not None else z_low
        return z_to_bottleneck(z_full, b_shape)

    def _encode_impl(self, x, params):
        x = self._check_input(x)
        B, _, H, W = x.shape
        fp = self._film_params(params, B)
        sess = self._new_session(B, H, W, self._pick_dtype(), fp is not None, x.device)
        plist = sess._params_ref
        outs = _EncodeFn.apply(sess, self, x, fp, *[plist[i] for i in self._enc_idx])
        b, skips = outs[0], list(outs[1:])
        for t in outs:
            t._sx_sess = sess
        return b, skips, sess

    def encode(self, x, params=None):
        b, skips, _ = self._encode_impl(x, params)
        return b, skips

    def _decode_impl(self, sess, b, skips, params):
        B = sess.B
        fp = self._film_params(params, B)
        if (fp is not None) != sess.use_film:
            raise ValueError("UNet (HIP): encode and decode must both be called with, or both without, params")
        drop_masks = None
        if self.training and self.dropout and self.dropout > 0:
            keep = 1.0 - float(self.dropout)
            drop_masks = []
            for j in range(self.depth):
                c = self._cfg["base"] * (2 ** (self.depth - 1 - j))
                m = torch.bernoulli(torch.full((B, c), keep, device=sess.device, dtype=torch.float32)) / keep
                drop_masks.append(m)
        plist = sess._params_ref
        return _DecodeFn.apply(sess, self, fp, drop_masks, b, *skips, *[plist[i] for i in self._dec_idx])

    def decode(self, b, skips, params=None):
        sess = getattr(b, "_sx_sess", None)
        in_place = sess is not None and all(getattr(s, "_sx_sess", None) is sess for s in skips)
        if in_place:
            in_place = (b.data_ptr() == sess.view(L.SX_T_BOTTLENECK).data_ptr() and
                        all(s.data_ptr() == sess.view(L.SX_T_SKIP, l).data_ptr() for l, s in enumerate(skips)))
        if not in_place:
            # foreign tensors (e.g. decode_from_bottleneck): stage them into a fresh session
            if not b.is_cuda:
                raise RuntimeError("UNet (HIP): inputs must be CUDA (ROCm) tensors; this build has no CPU path")
            B = b.shape[0]
            H, W = skips[0].shape[2], skips[0].shape[3]
            use_film = params is not None and self.P > 0
            dtype = sess.dtype if sess is not None else self._pick_dtype()
            new = self._new_session(B, H, W, dtype, use_film, b.device)
            staged = _StageFn.apply(new, b, *skips)
            b, skips = staged[0], list(staged[1:])
            sess = new
        return self._decode_impl(sess, b, skips, params)

    def forward(self, x, params=None, return_bottleneck: bool = False):
        b, skips, sess = self._encode_impl(x, params)
        y = self._decode_impl(sess, b, skips, params)
        return (y, b) if return_bottleneck else y

    def decode_from_bottleneck(self, x, b_pred, params=None):
        _, skips = self.encode(x, params=params)
        return self.decode(b_pred, skips, params=params)


class _StageFn(torch.autograd.Function):
    """Copy foreign (b, skips) into a session's workspace slots (differentiable identity)."""

    @staticmethod
    def forward(ctx, sess, b, *skips):
        D = sess.cfg["depth"]
        dst_b = sess.view(L.SX_T_BOTTLENECK)
        if tuple(b.shape) != tuple(dst_b.shape):
            raise ValueError(f"UNet (HIP): bottleneck shape {tuple(b.shape)} != {tuple(dst_b.shape)}")
        dst_b.copy_(b)
        outs = [sess.view(L.SX_T_BOTTLENECK, fresh=True)]
        for l in range(D):
            d = sess.view(L.SX_T_SKIP, l)
            if tuple(skips[l].shape) != tuple(d.shape):
                raise ValueError(f"UNet (HIP): skip {l} shape {tuple(skips[l].shape)} != {tuple(d.shape)}")
            d.copy_(skips[l])
            outs.append(sess.view(L.SX_T_SKIP, l, fresh=True))
        ctx.dtypes = [b.dtype] + [s.dtype for s in skips]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        return (None,) + tuple(None if g is None else g.to(dt) for g, dt in zip(grads, ctx.dtypes))
