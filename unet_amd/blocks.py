"""Sub-blocks of the U-Net called on their own (the reference allows `model.encoders[0](x)`, `film(params)`,
`UNet._apply_film(h, film, params)`: solpex/models.py:52-58, 79-80, 152-156).

Inside `UNet.encode/decode` these run fused in the network plan (unet_plan.hip); a direct call takes the same kernels one
C-ABI call at a time: conv1 + shortcut (`sx_conv3_shortcut_forward`), GroupNorm + SiLU (`sx_gn_silu_forward`), conv2
(`sx_conv_forward`), residual sum (`sx_block_out_forward`), and in the backward pass their gradient counterparts
(`sx_gn_silu_backward`, `sx_conv_wgrad`, `sx_conv3_shortcut_dgrad`).  Differentiable w.r.t. the input and every parameter.
The arithmetic tier follows the input: float32 -> exact fp32 MFMA, bfloat16 -> bf16 activations with fp32 accumulation."""
from __future__ import annotations

import torch

from . import _lib as L

GN_EPS = 1e-5


def _tier(x):
    if not x.is_cuda:
        raise RuntimeError("DoubleConv (HIP): the input must be a CUDA (ROCm) tensor; this build has no CPU path")
    if x.dtype == torch.bfloat16:
        return L.SX_BF16, torch.bfloat16
    if x.dtype == torch.float32:
        return L.SX_F32, torch.float32
    raise ValueError(f"DoubleConv (HIP): float32 or bfloat16 input expected, got {x.dtype}")


def _to_nhwc(x, sx_dtype, tdt, ld):
    """(B,C,H,W) with any strides -> contiguous NHWC in the tier's dtype, channels zero padded to `ld`."""
    B, Cc, H, W = x.shape
    xf = x.detach()
    xf = xf if xf.dtype == torch.float32 else xf.float()
    out = torch.empty(B, H, W, ld, dtype=tdt, device=x.device)
    sb, sc, sh, sw = xf.stride()
    L.check(L.lib().sx_pack_input(sx_dtype, L.ptr(xf), sb, sc, sh, sw, B, Cc, H, W, L.ptr(out), ld, L.stream_ptr()), "sx_pack_input")
    return out


def _packed(sx_dtype, tdt, mode, tflip, w, rows, k, dev):
    """Packed (MFMA-operand) form of a conv weight; `rows`/`k` = the GEMM's output / reduction channel counts."""
    wp = torch.empty(L.lib().sx_packed_weight_elems(mode, sx_dtype, rows, k), dtype=tdt, device=dev)
    wf = w.detach().float().contiguous()
    L.check(L.lib().sx_pack_conv_weight(sx_dtype, mode, tflip, L.ptr(wf), w.shape[0], w.shape[1], L.ptr(wp), L.stream_ptr()),
            "sx_pack_conv_weight")
    return wp


class _DoubleConvFn(torch.autograd.Function):
    """out = SiLU(GN2(conv2(SiLU(GN1(conv1(x)))))) + shortcut(x)   (solpex/models.py:61-80)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, ws, groups):
        lib, st = L.lib(), L.stream_ptr()
        sx_dtype, tdt = _tier(x)
        B, Cin, H, W = x.shape
        Cc = w1.shape[0]
        if w1.shape[1] != Cin:
            raise ValueError(f"DoubleConv (HIP): expected {w1.shape[1]} input channels, got {Cin}")
        if Cc % 8:
            raise ValueError(f"DoubleConv (HIP): out_ch must be a multiple of 8 (got {Cc})")
        dev = x.device
        cin_p = (Cin + 7) // 8 * 8
        xin = _to_nhwc(x, sx_dtype, tdt, cin_p)
        # an Identity shortcut is a 1x1 convolution with the unit matrix: one code path, no gradient for it
        ws_eff = ws if ws is not None else torch.eye(Cc, device=dev).reshape(Cc, Cc, 1, 1)
        p1 = _packed(sx_dtype, tdt, L.SX_CONV3, 0, w1, Cc, Cin, dev)
        ps = _packed(sx_dtype, tdt, L.SX_CONV1, 0, ws_eff, Cc, Cin, dev)
        p2 = _packed(sx_dtype, tdt, L.SX_CONV3, 0, w2, Cc, Cc, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        y1 = torch.empty(B, H, W, Cc, dtype=tdt, device=dev)
        s = torch.empty_like(y1)
        st1, st2 = torch.zeros(B, groups, 2, **f32), torch.zeros(B, groups, 2, **f32)
        L.check(lib.sx_conv3_shortcut_forward(sx_dtype, B, H, W, cin_p, Cc, L.ptr(xin), cin_p, L.ptr(p1), L.ptr(ps), L.ptr(y1), Cc,
                                              L.ptr(st1), groups, L.ptr(s), Cc, st), "sx_conv3_shortcut_forward")
        g1f, b1f, g2f, b2f = (t.detach().float().contiguous() for t in (g1, b1, g2, b2))
        a1 = torch.empty_like(y1)
        L.check(lib.sx_gn_silu_forward(sx_dtype, B, H * W, Cc, groups, GN_EPS, L.ptr(y1), Cc, L.ptr(st1), L.ptr(g1f), L.ptr(b1f),
                                       L.ptr(a1), Cc, st), "sx_gn_silu_forward")
        y2 = torch.empty_like(y1)
        L.check(lib.sx_conv_forward(sx_dtype, L.SX_CONV3, B, H, W, Cc, Cc, L.ptr(a1), Cc, H, W, L.ptr(p2), None, L.ptr(y2), Cc, H, W,
                                    0, L.ptr(st2), groups, st), "sx_conv_forward")
        out = torch.empty_like(y1)
        L.check(lib.sx_block_out_forward(sx_dtype, B, H * W, Cc, groups, GN_EPS, L.ptr(y2), Cc, L.ptr(st2), L.ptr(g2f), L.ptr(b2f),
                                         L.ptr(s), Cc, None, None, None, L.ptr(out), Cc, None, 0, st), "sx_block_out_forward")
        ctx.save_for_backward(xin, y1, st1, a1, y2, st2, out, g1f, b1f, g2f, b2f, w1, w2, ws_eff)
        ctx.meta = (sx_dtype, tdt, B, Cin, cin_p, Cc, H, W, groups, ws is not None)
        return out.permute(0, 3, 1, 2)          # (B, C, H, W) view of the NHWC result

    @staticmethod
    def backward(ctx, d_out):
        lib, st = L.lib(), L.stream_ptr()
        xin, y1, st1, a1, y2, st2, out, g1f, b1f, g2f, b2f, w1, w2, ws_eff = ctx.saved_tensors
        sx_dtype, tdt, B, Cin, cin_p, Cc, H, W, groups, has_ws = ctx.meta
        dev = xin.device
        f32 = dict(dtype=torch.float32, device=dev)
        need_x, need_w = ctx.needs_input_grad[0], any(ctx.needs_input_grad[1:8])
        do = _to_nhwc(d_out, sx_dtype, tdt, Cc)
        # through the residual sum / SiLU / GN2: dY2 and, next to where dY1 will go, dR = d(shortcut branch)
        dyr = torch.empty(B, H, W, 2 * Cc, dtype=tdt, device=dev)
        dy2 = torch.empty(B, H, W, Cc, dtype=tdt, device=dev)
        sums2, sums1 = torch.zeros(B, Cc, 4, **f32), torch.zeros(B, Cc, 4, **f32)
        dg1, db1, dg2, db2 = (torch.zeros(Cc, **f32) for _ in range(4))
        esz = dyr.element_size()
        L.check(lib.sx_gn_silu_backward(sx_dtype, 1, B, H * W, Cc, groups, GN_EPS, L.ptr(y2), Cc, L.ptr(st2), L.ptr(g2f), L.ptr(b2f),
                                        L.ptr(do), Cc, L.ptr(out), Cc, None, None, L.ptr(sums2), L.ptr(dy2), Cc,
                                        dyr.data_ptr() + Cc * esz, 2 * Cc, L.ptr(dg2), L.ptr(db2), 0, st), "sx_gn_silu_backward")
        p2t = _packed(sx_dtype, tdt, L.SX_CONV3, 1, w2, Cc, Cc, dev)
        da1 = torch.empty_like(dy2)
        L.check(lib.sx_conv_forward(sx_dtype, L.SX_CONV3, B, H, W, Cc, Cc, L.ptr(dy2), Cc, H, W, L.ptr(p2t), None, L.ptr(da1), Cc, H, W,
                                    0, None, 0, st), "sx_conv_forward")
        # through SiLU / GN1: dY1 into the first half of the joint buffer
        L.check(lib.sx_gn_silu_backward(sx_dtype, 0, B, H * W, Cc, groups, GN_EPS, L.ptr(y1), Cc, L.ptr(st1), L.ptr(g1f), L.ptr(b1f),
                                        L.ptr(da1), Cc, None, 0, None, None, L.ptr(sums1), L.ptr(dyr), 2 * Cc, None, 0,
                                        L.ptr(dg1), L.ptr(db1), 0, st), "sx_gn_silu_backward")
        dw1 = dw2 = dws = None
        if need_w:
            def wgrad(mode, ca, cb, cb_out, p_ptr, p_ld, q, q_ld, k):
                nb = lib.sx_conv_wgrad_workspace_bytes(mode, B, H, W, ca, cb)
                wsb = torch.empty(max(int(nb), 1), dtype=torch.uint8, device=dev)
                dw = torch.empty(ca, cb_out, k, k, **f32)
                L.check(lib.sx_conv_wgrad(sx_dtype, mode, B, H, W, ca, cb, ca, cb_out, p_ptr, p_ld, L.ptr(q), q_ld, H, W, L.ptr(dw), 0,
                                          L.ptr(wsb), nb, st), "sx_conv_wgrad")
                return dw
            dw2 = wgrad(L.SX_CONV3, Cc, Cc, Cc, L.ptr(dy2), Cc, a1, Cc, 3)
            dw1 = wgrad(L.SX_CONV3, Cc, cin_p, Cin, L.ptr(dyr), 2 * Cc, xin, cin_p, 3)
            if has_ws:
                dws = wgrad(L.SX_CONV1, Cc, cin_p, Cin, dyr.data_ptr() + Cc * esz, 2 * Cc, xin, cin_p, 1)
        dx = None
        if need_x:
            p1t = _packed(sx_dtype, tdt, L.SX_CONV3, 1, w1, Cin, Cc, dev)
            pst = _packed(sx_dtype, tdt, L.SX_CONV1, 1, ws_eff, Cin, Cc, dev)
            dxin = torch.empty(B, H, W, cin_p, dtype=tdt, device=dev)
            L.check(lib.sx_conv3_shortcut_dgrad(sx_dtype, B, H, W, Cc, cin_p, L.ptr(dyr), 2 * Cc, L.ptr(p1t), L.ptr(pst), L.ptr(dxin),
                                                cin_p, None, 0, st), "sx_conv3_shortcut_dgrad")
            dx = dxin[..., :Cin].permute(0, 3, 1, 2)
        cast = (lambda t, like: None if t is None else t.to(like.dtype))
        return (dx, cast(dw1, w1), cast(dg1, g1f), cast(db1, b1f), cast(dw2, w2), cast(dg2, g2f), cast(db2, b2f),
                cast(dws, ws_eff) if has_ws else None, None)


def double_conv_forward(block, x):
    """`DoubleConv.forward(x)` (solpex/models.py:79-80) on the HIP kernels; x (B, in_ch, H, W), float32 or bfloat16."""
    if x.dim() != 4:
        raise ValueError(f"DoubleConv (HIP): expected a (B, C, H, W) tensor, got {tuple(x.shape)}")
    n = block.net
    ws = block.shortcut.weight if isinstance(block.shortcut, torch.nn.Conv2d) else None
    return _DoubleConvFn.apply(x, n[0].weight, n[1].weight, n[1].bias, n[3].weight, n[4].weight, n[4].bias, ws, n[1].num_groups)


def film_forward(gen, params):
    """`FiLMGenerator.forward(params)` (solpex/models.py:52-58): (B, P) -> (1 + gamma, beta), each (B, C, 1, 1).  The
    two-layer MLP runs on the fp32 MFMA GEMM of the latent MLPs (`sx_gemm_f32`, bias + SiLU in its epilogue)."""
    from .latent import _MLP

    class _Net(_MLP):                      # the generator's own Sequential seen through the MLP kernels' module interface
        def __init__(self, net):
            torch.nn.Module.__init__(self)
            self.net = net
    mlp = _Net(gen.net)
    mlp.train(gen.training)
    out = mlp(params)
    gamma, beta = out.split(gen.n_channels, dim=1)
    return 1.0 + gamma[:, :, None, None], beta[:, :, None, None]


def apply_film(features, film_layer, params):
    """`UNet._apply_film` (solpex/models.py:152-156): features * gamma + beta with the generator's (B, C, 1, 1) outputs."""
    gamma, beta = film_layer(params)
    return features * gamma.to(features.dtype) + beta.to(features.dtype)
