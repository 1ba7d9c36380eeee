"""GPU: kernel-level parity of the C-ABI entry points against the CPU oracle / plain torch fp32.

Tolerances (BASELINE.json north_star): fp32 tier <= 1e-4 relative, bf16 tier <= 1e-2 relative.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, max_err, t

pytestmark = pytest.mark.gpu

TOL = {0: 1e-4, 1: 1e-2}


@pytest.fixture(scope="module")
def L():
    from unet_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module", autouse=True, params=["split", "exact", "x6"])
def fp32_mode(request, L):
    """Every kernel test runs in all three fp32 arithmetic modes: split-bf16 on two planes (the default: three bf16 MFMAs per
    product block, conv_x3.hip), exact fp32 MFMA, and split-bf16 on three planes ("x6": six MFMAs, what wide networks run);
    packed fp32 weights are made after the switch.  The bf16 cases do not depend on it."""
    planes = {"exact": 0, "split": 2, "x6": 3}[request.param]
    was = L.lib().sx_set_fp32_planes(planes)
    yield request.param
    assert L.lib().sx_set_fp32_planes(was) == planes


def dev(x, dt=None):
    x = x.cuda()
    return x if dt is None else x.to(dt)


def tdt(dtype):
    return torch.bfloat16 if dtype == 1 else torch.float32


def nhwc(x, dtype, ld=None):
    """(B,C,H,W) cpu fp32 -> device NHWC tensor [B,H,W,ld] of the tier's dtype."""
    B, Cc, H, W = x.shape
    ld = ld or Cc
    out = torch.zeros(B, H, W, ld, dtype=tdt(dtype), device="cuda")
    out[..., :Cc] = x.permute(0, 2, 3, 1).cuda().to(tdt(dtype))
    return out


def from_nhwc(y, Cc):
    return y[..., :Cc].float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, dtype):
    """Round a CPU fp32 tensor to the tier's storage dtype (so both sides see identical inputs)."""
    return x.to(tdt(dtype)).float()


def pack(L, dtype, mode, tflip, w, Cout, Cin, rows, k):
    n = L.lib().sx_packed_weight_elems(mode, dtype, rows, k)
    buf = torch.empty(n, dtype=tdt(dtype), device="cuda")
    wd = w.contiguous().cuda()
    L.check(L.lib().sx_pack_conv_weight(dtype, mode, tflip, L.ptr(wd), Cout, Cin, L.ptr(buf), L.stream_ptr()))
    return buf


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 8, 8, 16), (1, 13, 7, 16, 24), (2, 32, 16, 32, 64), (1, 8, 8, 64, 128),
                                             (3, 5, 3, 8, 8),
                                             # shapes that take the weights-stationary bf16 kernel (W >= 16), one per tile config
                                             (2, 32, 32, 64, 32), (2, 16, 16, 128, 128), (1, 16, 32, 256, 64), (3, 20, 24, 32, 32),
                                             (1, 24, 16, 256, 256), (2, 40, 48, 8, 32), (1, 16, 16, 128, 64), (1, 32, 16, 64, 24),
                                             # 8-wide tiles of the same kernel (8 <= W < 16)
                                             (2, 16, 8, 128, 256), (1, 12, 10, 256, 64), (2, 9, 15, 32, 32), (5, 16, 8, 256, 256),
                                             (70, 16, 8, 64, 128),
                                             # 512 input channels: two K halves, the second accumulating
                                             (1, 16, 16, 512, 128), (2, 16, 8, 512, 256)])
def test_conv3x3_forward_stats_and_dgrad(L, dtype, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H + Cin)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5), dtype)
    ref = F.conv2d(x, w, padding=1)
    xd = nhwc(x, dtype)
    y = torch.empty(B, H, W, Cout, dtype=tdt(dtype), device="cuda")
    G = min(8, Cout)
    stats = torch.zeros(B, G, 2, dtype=torch.float32, device="cuda")
    wp = pack(L, dtype, L.SX_CONV3, 0, w, Cout, Cin, Cout, Cin)
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONV3, B, H, W, Cin, Cout, L.ptr(xd), Cin, H, W, L.ptr(wp), None,
                                    L.ptr(y), Cout, H, W, 0, L.ptr(stats), G, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y, Cout), ref) < TOL[dtype]
    rg = ref.view(B, G, -1)
    s_ref = torch.stack([rg.sum(-1), (rg * rg).sum(-1)], dim=-1)
    assert rel_err(stats.cpu(), s_ref) < 1e-4 if dtype == 0 else rel_err(stats.cpu(), s_ref) < 1e-2
    # data gradient = same kernel on the transposed + flipped pack
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    dref = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1)
    wpd = pack(L, dtype, L.SX_CONV3, 1, w, Cout, Cin, Cin, Cout)
    dyd = nhwc(dy, dtype)
    dx = torch.empty(B, H, W, Cin, dtype=tdt(dtype), device="cuda")
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONV3, B, H, W, Cout, Cin, L.ptr(dyd), Cout, H, W, L.ptr(wpd), None,
                                    L.ptr(dx), Cin, H, W, 0, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx, Cin), dref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 32, 8, 32), (2, 24, 16, 32, 64), (1, 16, 32, 64, 128), (3, 16, 8, 128, 256),
                                             (1, 32, 16, 256, 128), (2, 16, 16, 128, 64), (2, 40, 16, 64, 32), (2, 16, 8, 256, 128),
                                             (2, 13, 7, 16, 24), (1, 20, 12, 128, 128), (2, 16, 8, 16, 8), (2, 16, 8, 8, 8),
                                             (2, 8, 8, 32, 16), (2, 16, 8, 1 * 8, 16)])
def test_conv3_shortcut_fused_forward_and_dgrad(L, dtype, B, H, W, Cin, Cout):
    """DoubleConv fusions: conv1 + 1x1 shortcut from one pass over the input, and their joint data gradient from the
    [dY1 | dR] buffer; shapes cover every tile configuration of the fused kernel variants and the unfused fallbacks."""
    g = torch.Generator().manual_seed(B * 100 + H + Cin + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w3 = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5), dtype)
    w1 = q(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5, dtype)
    xd = nhwc(x, dtype, ld=Cin + 8)
    xd[..., Cin:] = 5.0
    y = torch.empty(B, H, W, Cout, dtype=tdt(dtype), device="cuda")
    sh = torch.full((B, H, W, Cout + 8), 7.0, dtype=tdt(dtype), device="cuda")
    G = min(8, Cout)
    stats = torch.zeros(B, G, 2, dtype=torch.float32, device="cuda")
    p3 = pack(L, dtype, L.SX_CONV3, 0, w3, Cout, Cin, Cout, Cin)
    p1 = pack(L, dtype, L.SX_CONV1, 0, w1, Cout, Cin, Cout, Cin)
    L.check(L.lib().sx_conv3_shortcut_forward(dtype, B, H, W, Cin, Cout, L.ptr(xd), Cin + 8, L.ptr(p3), L.ptr(p1), L.ptr(y), Cout,
                                              L.ptr(stats), G, L.ptr(sh), Cout + 8, L.stream_ptr()))
    torch.cuda.synchronize()
    ref3, ref1 = F.conv2d(x, w3, padding=1), F.conv2d(x, w1)
    assert rel_err(from_nhwc(y, Cout), ref3) < TOL[dtype]
    assert rel_err(from_nhwc(sh, Cout), ref1) < TOL[dtype]
    assert float((sh[..., Cout:].float() - 7.0).abs().max()) == 0.0
    rg = ref3.view(B, G, -1)
    assert rel_err(stats.cpu(), torch.stack([rg.sum(-1), (rg * rg).sum(-1)], dim=-1)) < (1e-4 if dtype == 0 else 1e-2)
    # joint data gradient
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    dr = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    dref = torch.nn.grad.conv2d_input(x.shape, w3, dy, padding=1) + torch.nn.grad.conv2d_input(x.shape, w1, dr)
    dyr = torch.cat([nhwc(dy, dtype), nhwc(dr, dtype)], dim=-1).contiguous()
    p3t = pack(L, dtype, L.SX_CONV3, 1, w3, Cout, Cin, Cin, Cout)
    p1t = pack(L, dtype, L.SX_CONV1, 1, w1, Cout, Cin, Cin, Cout)
    dx = torch.full((B, H, W, Cin + 8), 3.0, dtype=tdt(dtype), device="cuda")
    nsum = Cin // 2 if (Cin // 2) % 8 == 0 else Cin        # channel sums of the first nsum output channels ride along
    chs = torch.zeros(nsum, dtype=torch.float32, device="cuda")
    L.check(L.lib().sx_conv3_shortcut_dgrad(dtype, B, H, W, Cout, Cin, L.ptr(dyr), 2 * Cout, L.ptr(p3t), L.ptr(p1t), L.ptr(dx), Cin + 8,
                                            L.ptr(chs), nsum, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx, Cin), dref) < TOL[dtype]
    got = from_nhwc(dx, Cin)
    assert rel_err(chs.cpu(), got[:, :nsum].sum(dim=(0, 2, 3))) < (1e-4 if dtype == 0 else 3e-3)
    assert float((dx[..., Cin:].float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout,off", [(2, 16, 8, 8, 16, 0), (2, 16, 8, 8, 16, 8), (1, 16, 16, 16, 32, 0), (2, 8, 8, 24, 8, 0),
                                                 (2, 16, 16, 40, 32, 0), (3, 16, 8, 8, 8, 8)])
def test_conv_channel_padding_does_not_read_what_follows_the_tensor(L, mode, B, H, W, Cin, Cout, off):
    """K is walked in chunks of 32 channels; with Cin % 32 != 0 the tail of a chunk lies in the next pixels -- or, at the
    last pixels of the last image, in whatever follows the tensor in memory.  That must never enter the sums (here: NaN
    right behind the tensor; `off` = the input is the upper channel half of a wider [A | B] buffer, as the joint data
    gradient's fallback reads it).  bf16 tier (the weights-stationary kernel); the generic kernel guards per element."""
    dtype = 1
    g = torch.Generator().manual_seed(B + H + Cin + Cout + mode)
    k = 3 if mode == 0 else 1
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5), dtype)
    ld = Cin + off
    n = B * H * W * ld
    big = torch.full((n + 256,), float("nan"), dtype=torch.bfloat16, device="cuda")
    buf = big[:n].view(B, H, W, ld)
    buf[...] = 1.0
    buf[..., off:] = x.permute(0, 2, 3, 1).to(torch.bfloat16).cuda()
    xin = buf[..., off:]
    wp = pack(L, dtype, mode, 0, w, Cout, Cin, Cout, Cin)
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().sx_conv_forward(dtype, mode, B, H, W, Cin, Cout, xin.data_ptr(), ld, H, W, L.ptr(wp), None, L.ptr(y), Cout, H, W,
                                    0, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, padding=k // 2)
    got = from_nhwc(y, Cout)
    assert bool(torch.isfinite(got).all()), torch.isnan(got).nonzero()[:4].tolist()
    assert rel_err(got, ref) < TOL[dtype]


@pytest.mark.parametrize("Cin,Cout", [(32, 64), (128, 32), (256, 128), (64, 256)])
def test_conv1x1_weights_stationary_accumulate(L, Cin, Cout):
    dtype, B, H, W = 1, 2, 24, 32
    g = torch.Generator().manual_seed(Cin + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5, dtype)
    base = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    ref = F.conv2d(x, w) + base
    xd = nhwc(x, dtype, ld=Cin + 16)
    yd = nhwc(base, dtype, ld=Cout + 8)
    wp = pack(L, dtype, L.SX_CONV1, 0, w, Cout, Cin, Cout, Cin)
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONV1, B, H, W, Cin, Cout, L.ptr(xd), Cin + 16, H, W, L.ptr(wp), None,
                                    L.ptr(yd), Cout + 8, H, W, 1, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(yd, Cout), ref) < TOL[dtype]
    assert float(yd[..., Cout:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [0, 1])
def test_conv1x1_bias_accumulate_and_ld(L, dtype):
    B, H, W, Cin, Cout = 2, 12, 10, 32, 16
    g = torch.Generator().manual_seed(7)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5, dtype)
    bias = torch.randn(Cout, generator=g)
    base = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    ref = F.conv2d(x, w, bias) + base
    xd = nhwc(x, dtype, ld=Cin + 16)            # input is a channel slice of a wider buffer
    yd = nhwc(base, dtype, ld=Cout + 8)
    wp = pack(L, dtype, L.SX_CONV1, 0, w, Cout, Cin, Cout, Cin)
    bd = bias.cuda()
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONV1, B, H, W, Cin, Cout, L.ptr(xd), Cin + 16, H, W, L.ptr(wp), L.ptr(bd),
                                    L.ptr(yd), Cout + 8, H, W, 1, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(yd, Cout), ref) < TOL[dtype]
    assert float(yd[..., Cout:].float().abs().max()) == 0.0     # padding channels untouched


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,h,w,Cin,Cout,Ho,Wo", [(2, 4, 4, 32, 16, 8, 8), (1, 3, 1, 16, 8, 7, 3), (2, 8, 4, 64, 32, 16, 8),
                                                   (2, 16, 8, 256, 128, 32, 16), (3, 32, 16, 128, 64, 64, 32), (2, 64, 32, 64, 32, 128, 64),
                                                   (1, 9, 17, 32, 16, 19, 35)])
def test_conv_transpose_forward_and_backward(L, dtype, B, h, w, Cin, Cout, Ho, Wo):
    g = torch.Generator().manual_seed(h * 10 + Cin)
    x = q(torch.randn(B, Cin, h, w, generator=g), dtype)
    wt = q(torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5, dtype)
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x, wt, bias, stride=2)
    xd = nhwc(x, dtype)
    # output lands in the lower channel half of a (Ho, Wo) "cat" buffer
    cat = torch.full((B, Ho, Wo, 2 * Cout), 7.0, dtype=tdt(dtype), device="cuda")
    wp = pack(L, dtype, L.SX_CONVT_FWD, 0, wt, Cout, Cin, Cout, Cin)
    bd = bias.cuda()
    L.check(L.lib().sx_zero_border(dtype, B, Ho, Wo, 2 * h, 2 * w, Cout, L.ptr(cat), 2 * Cout, L.stream_ptr()))
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONVT_FWD, B, h, w, Cin, Cout, L.ptr(xd), Cin, h, w, L.ptr(wp), L.ptr(bd),
                                    L.ptr(cat), 2 * Cout, Ho, Wo, 0, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    got = from_nhwc(cat, Cout)
    padded = F.pad(ref, [0, Wo - 2 * w, 0, Ho - 2 * h])
    assert rel_err(got, padded) < TOL[dtype]
    assert float((cat[..., Cout:].float() - 7.0).abs().max()) == 0.0
    # data gradient (gather form)
    dy = q(torch.randn(B, Cout, Ho, Wo, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    F.pad(F.conv_transpose2d(xr, wt, bias, stride=2), [0, Wo - 2 * w, 0, Ho - 2 * h]).backward(dy)
    dyd = nhwc(dy, dtype, ld=2 * Cout)
    wpb = pack(L, dtype, L.SX_CONVT_BWD, 0, wt, Cout, Cin, Cin, Cout)
    dx = torch.empty(B, h, w, Cin, dtype=tdt(dtype), device="cuda")
    L.check(L.lib().sx_conv_forward(dtype, L.SX_CONVT_BWD, B, h, w, Cout, Cin, L.ptr(dyd), 2 * Cout, Ho, Wo, L.ptr(wpb), None,
                                    L.ptr(dx), Cin, h, w, 0, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx, Cin), xr.grad) < TOL[dtype]
    # weight gradient of the transposed conv
    wr = wt.clone().requires_grad_(True)
    F.pad(F.conv_transpose2d(x, wr, bias, stride=2), [0, Wo - 2 * w, 0, Ho - 2 * h]).backward(dy)
    nb = L.lib().sx_conv_wgrad_workspace_bytes(L.SX_CONVT_FWD, B, h, w, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.empty(Cin, Cout, 2, 2, dtype=torch.float32, device="cuda")
    L.check(L.lib().sx_conv_wgrad(dtype, L.SX_CONVT_FWD, B, h, w, Cin, Cout, Cin, Cout, L.ptr(xd), Cin, L.ptr(dyd), 2 * Cout,
                                  Ho, Wo, L.ptr(dw), 0, L.ptr(ws), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), wr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode,B,H,W,Cin,Cout", [(0, 2, 16, 8, 8, 16), (0, 3, 13, 7, 16, 24), (0, 2, 32, 16, 32, 64),
                                                  (0, 1, 16, 16, 64, 128), (1, 2, 12, 10, 32, 16), (0, 4, 32, 32, 8, 32),
                                                  # 4-wide grids (32 x 4 pixel tiles: the largest halo), ragged 3-wide, 40 output rows
                                                  (0, 2, 40, 4, 16, 16), (1, 2, 8, 4, 8, 32), (0, 3, 6, 3, 8, 8), (0, 2, 24, 20, 24, 40)])
def test_conv_wgrad(L, dtype, mode, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H + Cin + mode)
    k = 3 if mode == 0 else 1
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, padding=k // 2)
    xd, dyd = nhwc(x, dtype), nhwc(dy, dtype)
    nb = L.lib().sx_conv_wgrad_workspace_bytes(mode, B, H, W, Cout, Cin)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.full((Cout, Cin, k, k), 1.0, dtype=torch.float32, device="cuda")
    L.check(L.lib().sx_conv_wgrad(dtype, mode, B, H, W, Cout, Cin, Cout, Cin, L.ptr(dyd), Cout, L.ptr(xd), Cin, H, W,
                                  L.ptr(dw), 1, L.ptr(ws), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu() - 1.0, ref) < TOL[dtype]


@pytest.mark.parametrize("mode,B,H,W,Cin,Cout", [
    # persistent bf16 tier (wgrad_ws.hip): one shape per (a fragments, b fragments, tile width) family
    (0, 3, 24, 32, 32, 32), (0, 2, 16, 16, 64, 64), (0, 5, 13, 20, 8, 32), (0, 2, 16, 48, 16, 16), (0, 1, 32, 16, 128, 64),
    (0, 2, 40, 16, 64, 128), (0, 2, 16, 8, 64, 64), (0, 3, 12, 9, 128, 32), (0, 1, 16, 16, 256, 128), (0, 2, 8, 8, 256, 256),
    (1, 3, 24, 32, 32, 32), (1, 2, 16, 16, 64, 8), (1, 2, 20, 18, 128, 64), (1, 2, 16, 8, 64, 32), (1, 2, 9, 17, 32, 128)])
def test_conv_wgrad_persistent_bf16(L, mode, B, H, W, Cin, Cout):
    """bf16 weight gradient on channel slices of wider buffers (ld > C)."""
    dtype = 1
    g = torch.Generator().manual_seed(H * 7 + Cin + mode)
    k = 3 if mode == 0 else 1
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, padding=k // 2)
    xd, dyd = nhwc(x, dtype, ld=Cin + 8), nhwc(dy, dtype, ld=2 * Cout)
    xd[..., Cin:] = 3.0          # neighbouring channels of the wider buffers must not leak in
    dyd[..., Cout:] = -2.0
    nb = L.lib().sx_conv_wgrad_workspace_bytes(mode, B, H, W, Cout, Cin)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.full((Cout, Cin, k, k), 1.0, dtype=torch.float32, device="cuda")
    L.check(L.lib().sx_conv_wgrad(dtype, mode, B, H, W, Cout, Cin, Cout, Cin, L.ptr(dyd), 2 * Cout, L.ptr(xd), Cin + 8, H, W,
                                  L.ptr(dw), 1, L.ptr(ws), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu() - 1.0, ref) < 2e-3       # exact bf16 products, fp32 accumulation: only summation order differs


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cc", [(2, 16, 8, 16), (3, 7, 5, 24), (2, 32, 16, 64),
                                      # shapes whose (sample, group) unit fits one block's registers: the backward is ONE launch
                                      # (act_bwd_group_kernel: 1, 2 and 4 slots per thread, a partly filled block, 64-channel groups)
                                      (3, 16, 8, 256), (2, 32, 16, 128), (2, 8, 8, 64), (2, 12, 10, 128), (1, 4, 4, 512),
                                      (2, 64, 32, 64)])      # one size up: two launches again
def test_groupnorm_silu_block_out_forward_backward(L, dtype, B, H, W, Cc):
    g = torch.Generator().manual_seed(Cc + H)
    G = min(8, Cc)
    y2 = q(torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.3, dtype)
    s = q(torch.randn(B, Cc, H, W, generator=g), dtype)
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.2 * torch.randn(Cc, generator=g)
    gf = 1 + 0.3 * torch.randn(B, Cc, generator=g)
    bf = 0.3 * torch.randn(B, Cc, generator=g)
    drop = (torch.rand(B, Cc, generator=g) > 0.3).float() / 0.7
    dO = q(torch.randn(B, Cc, H, W, generator=g), dtype)
    # oracle
    yr, sr, gr, br = y2.clone().requires_grad_(True), s.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    gfr, bfr = gf.clone().requires_grad_(True), bf.clone().requires_grad_(True)
    R = (F.silu(F.group_norm(yr, G, gr, br, eps=1e-5)) + sr) * drop[:, :, None, None]
    O = R * gfr[:, :, None, None] + bfr[:, :, None, None]
    O.backward(dO)
    # device: statistics from the exact inputs
    yg = y2.view(B, G, -1)
    stats = torch.stack([yg.sum(-1), (yg * yg).sum(-1)], -1).cuda().contiguous()
    yd, sd, dOd = nhwc(y2, dtype), nhwc(s, dtype), nhwc(dO, dtype)
    out = torch.empty(B, H, W, Cc, dtype=tdt(dtype), device="cuda")
    Rd = torch.empty_like(out)
    gmd, btd, gfd, bfd, drd = gamma.cuda(), beta.cuda(), gf.cuda().contiguous(), bf.cuda().contiguous(), drop.cuda().contiguous()
    lib = L.lib()
    L.check(lib.sx_block_out_forward(dtype, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gmd), L.ptr(btd), L.ptr(sd), Cc,
                                     L.ptr(gfd), L.ptr(bfd), L.ptr(drd), L.ptr(out), Cc, L.ptr(Rd), Cc, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cc), O.detach()) < TOL[dtype]
    assert rel_err(from_nhwc(Rd, Cc), R.detach()) < TOL[dtype]
    sums = torch.zeros(B, Cc, 4, dtype=torch.float32, device="cuda")
    dy = torch.empty_like(out)
    dR = torch.empty_like(out)
    dg = torch.empty(Cc, dtype=torch.float32, device="cuda")
    db = torch.empty(Cc, dtype=torch.float32, device="cuda")
    L.check(lib.sx_gn_silu_backward(dtype, 1, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gmd), L.ptr(btd), L.ptr(dOd), Cc,
                                    L.ptr(Rd), Cc, L.ptr(gfd), L.ptr(drd), L.ptr(sums), L.ptr(dy), Cc, L.ptr(dR), Cc, L.ptr(dg), L.ptr(db),
                                    0, L.stream_ptr()))
    torch.cuda.synchronize()
    tol = TOL[dtype] * (1 if dtype == 0 else 3)
    assert rel_err(from_nhwc(dy, Cc), yr.grad) < tol
    assert rel_err(from_nhwc(dR, Cc), sr.grad) < tol
    assert rel_err(dg.cpu(), gr.grad) < tol and rel_err(db.cpu(), br.grad) < tol
    assert rel_err(sums[..., 2].cpu(), gfr.grad) < tol and rel_err(sums[..., 3].cpu(), bfr.grad) < tol
    # mode 2: R rebuilt from the shortcut branch instead of read back
    sums2 = torch.zeros(B, Cc, 4, dtype=torch.float32, device="cuda")
    dy2, dR2 = torch.empty_like(out), torch.empty_like(out)
    L.check(lib.sx_gn_silu_backward(dtype, 2, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gmd), L.ptr(btd), L.ptr(dOd), Cc,
                                    L.ptr(sd), Cc, L.ptr(gfd), L.ptr(drd), L.ptr(sums2), L.ptr(dy2), Cc, L.ptr(dR2), Cc, None, None,
                                    0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(sums2[..., 2].cpu(), gfr.grad) < tol and rel_err(sums2[..., 3].cpu(), bfr.grad) < tol
    assert rel_err(from_nhwc(dy2, Cc), yr.grad) < tol and torch.equal(dR2, dR)
    # mode 0: plain SiLU(GN(y))
    yr2 = y2.clone().requires_grad_(True)
    A = F.silu(F.group_norm(yr2, G, gamma, beta, eps=1e-5))
    A.backward(dO)
    Ad = torch.empty_like(out)
    L.check(lib.sx_gn_silu_forward(dtype, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gmd), L.ptr(btd), L.ptr(Ad), Cc, L.stream_ptr()))
    sums.zero_()
    L.check(lib.sx_gn_silu_backward(dtype, 0, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gmd), L.ptr(btd), L.ptr(dOd), Cc,
                                    None, 0, None, None, L.ptr(sums), L.ptr(dy), Cc, None, 0, None, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(Ad, Cc), A.detach()) < TOL[dtype]
    assert rel_err(from_nhwc(dy, Cc), yr2.grad) < tol


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode,gn_mode,B,H,W,Cin,Cout", [
    # 3x3 producers of dA1 (GroupNorm-1 sums) at every tile family of the weights-stationary kernel, and shapes it does not
    # cover (two launches); 1x1 producers of a block's d_out (four sums, FiLM / dropout factors, shortcut branch)
    (0, 0, 3, 32, 16, 32, 32), (0, 0, 2, 64, 32, 64, 64), (0, 0, 2, 32, 16, 128, 128), (0, 0, 5, 16, 8, 256, 256),
    (0, 0, 2, 13, 7, 16, 24), (0, 0, 3, 20, 16, 64, 64), (0, 0, 2, 8, 8, 128, 128),
    (1, 2, 3, 32, 16, 8, 32), (1, 2, 2, 24, 32, 16, 64), (1, 2, 2, 16, 8, 32, 128), (1, 2, 2, 9, 5, 8, 16)])
def test_conv_forward_with_fused_groupnorm_backward_sums(L, dtype, mode, gn_mode, B, H, W, Cin, Cout):
    """sx_conv_forward_gn_bwd_sums == sx_conv_forward followed by phase 1 of sx_gn_silu_backward on its output."""
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout + mode)
    k = 3 if mode == 0 else 1
    G = min(8, Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    yn = q(torch.randn(B, Cout, H, W, generator=g) * 1.5 + 0.3, dtype)          # the tensor GroupNorm normalised
    sb = q(torch.randn(B, Cout, H, W, generator=g), dtype)                      # shortcut branch (gn_mode 2)
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
    gf = (1 + 0.3 * torch.randn(B, Cout, generator=g)).cuda()
    drop = ((torch.rand(B, Cout, generator=g) > 0.3).float() / 0.7).cuda()
    stats = torch.stack([yn.view(B, G, -1).sum(-1), (yn * yn).view(B, G, -1).sum(-1)], -1).contiguous().cuda()
    xd, ynd, sbd = nhwc(x, dtype), nhwc(yn, dtype), nhwc(sb, dtype)
    wp = pack(L, dtype, mode, 0, w, Cout, Cin, Cout, Cin)
    gd, bd = gamma.cuda(), beta.cuda()
    lib = L.lib()

    def run(fused):
        y = torch.empty(B, H, W, Cout, dtype=tdt(dtype), device="cuda")
        sums = torch.zeros(B, Cout, 4, device="cuda")
        dy = torch.empty_like(y)
        dr = torch.empty_like(y)
        fg, dp = (gf, drop) if gn_mode == 2 else (None, None)
        if fused:
            L.check(lib.sx_conv_forward_gn_bwd_sums(dtype, mode, B, H, W, Cin, Cout, L.ptr(xd), Cin, L.ptr(wp), L.ptr(y), Cout, gn_mode,
                                                    L.ptr(ynd), Cout, L.ptr(sbd) if gn_mode == 2 else None, Cout, L.ptr(stats), G, 1e-5,
                                                    L.ptr(gd), L.ptr(bd), L.ptr(fg), L.ptr(dp), L.ptr(sums), L.stream_ptr()))
            flag = L.SX_GN_BWD_SUMS_READY
        else:
            L.check(lib.sx_conv_forward(dtype, mode, B, H, W, Cin, Cout, L.ptr(xd), Cin, H, W, L.ptr(wp), None, L.ptr(y), Cout, H, W,
                                        0, None, 0, L.stream_ptr()))
            flag = 0
        L.check(lib.sx_gn_silu_backward(dtype, gn_mode | flag, B, H * W, Cout, G, 1e-5, L.ptr(ynd), Cout, L.ptr(stats), L.ptr(gd),
                                        L.ptr(bd), L.ptr(y), Cout, L.ptr(sbd) if gn_mode == 2 else None, Cout, L.ptr(fg), L.ptr(dp),
                                        L.ptr(sums), L.ptr(dy), Cout, L.ptr(dr) if gn_mode == 2 else None, Cout, None, None, 0,
                                        L.stream_ptr()))
        torch.cuda.synchronize()
        return y.float().cpu(), sums.cpu(), dy.float().cpu()
    y0, s0, d0 = run(False)
    y1, s1, d1 = run(True)
    assert torch.equal(y0, y1)
    ns = 4 if gn_mode == 2 else 2
    assert float(s1[..., ns:].abs().max()) == 0.0 if ns < 4 else True
    for kk in range(ns):
        scale = float(s0[..., kk].abs().max())
        assert float((s1[..., kk] - s0[..., kk]).abs().max()) < 2e-4 * max(scale, 1e-6), (kk, scale)
    assert rel_err(d1, d0) < (1e-5 if dtype == 0 else 4e-3)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cc,film", [(2, 16, 8, 16, True), (3, 6, 10, 24, False), (2, 32, 16, 64, True), (1, 8, 8, 512, True)])
def test_block_out_pool_forward_is_block_out_then_maxpool(L, dtype, B, H, W, Cc, film):
    """The fused encoder-block epilogue against the two calls it replaces: out, R and the pooled tensor bit-identical
    (strided `out`, as inside the virtual concat)."""
    g = torch.Generator().manual_seed(Cc + H)
    G = 8
    y2 = q(torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.3, dtype)
    s = q(torch.randn(B, Cc, H, W, generator=g), dtype)
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda()
    beta = (0.2 * torch.randn(Cc, generator=g)).cuda()
    gf = (1 + 0.3 * torch.randn(B, Cc, generator=g)).cuda().contiguous()
    bf = (0.3 * torch.randn(B, Cc, generator=g)).cuda().contiguous()
    drop = ((torch.rand(B, Cc, generator=g) > 0.3).float() / 0.7).cuda().contiguous()
    yg = y2.view(B, G, -1)
    stats = torch.stack([yg.sum(-1), (yg * yg).sum(-1)], -1).cuda().contiguous()
    yd, sd = nhwc(y2, dtype), nhwc(s, dtype)
    lib = L.lib()
    res = []
    for fused in (False, True):
        out = torch.zeros(B, H, W, 2 * Cc, dtype=tdt(dtype), device="cuda")      # left half of a concat buffer
        Rd = torch.zeros(B, H, W, Cc, dtype=tdt(dtype), device="cuda")
        pool = torch.zeros(B, H // 2, W // 2, Cc, dtype=tdt(dtype), device="cuda")
        fg, fb, dr = (L.ptr(gf), L.ptr(bf), L.ptr(drop)) if film else (None, None, None)
        if fused:
            L.check(lib.sx_block_out_pool_forward(dtype, B, H, W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                                                  L.ptr(sd), Cc, fg, fb, dr, L.ptr(out), 2 * Cc, L.ptr(Rd), Cc, L.ptr(pool), Cc,
                                                  L.stream_ptr()))
        else:
            L.check(lib.sx_block_out_forward(dtype, B, H * W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                                             L.ptr(sd), Cc, fg, fb, dr, L.ptr(out), 2 * Cc, L.ptr(Rd), Cc, L.stream_ptr()))
            L.check(lib.sx_maxpool2_forward(dtype, B, H, W, Cc, L.ptr(out), 2 * Cc, L.ptr(pool), Cc, L.stream_ptr()))
        torch.cuda.synchronize()
        res.append((out.float().cpu(), Rd.float().cpu(), pool.float().cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # odd sizes are refused (the plan then keeps the two calls)
    assert lib.sx_block_out_pool_forward(dtype, B, H + 1, W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                                         L.ptr(sd), Cc, None, None, None, L.ptr(out), 2 * Cc, None, Cc, L.ptr(pool), Cc,
                                         L.stream_ptr()) != 0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cc", [(2, 8, 6, 16), (2, 7, 5, 8)])
def test_maxpool_forward_backward_first_max_ties(L, dtype, B, H, W, Cc):
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(B, Cc, H, W, generator=g), dtype)
    x[:, :, :2, :2] = 0.25          # exact ties: the first maximum must take the gradient, as ATen does
    x[0, 0] = 0.0
    xr = x.clone().requires_grad_(True)
    P = F.max_pool2d(xr, 2)
    dP = q(torch.randn(P.shape, generator=g), dtype)
    dskip = q(torch.randn(x.shape, generator=g), dtype)
    P.backward(dP)
    ref_dx = xr.grad + dskip
    xd = nhwc(x, dtype, ld=2 * Cc)
    y = torch.empty(B, H // 2, W // 2, Cc, dtype=tdt(dtype), device="cuda")
    lib = L.lib()
    L.check(lib.sx_maxpool2_forward(dtype, B, H, W, Cc, L.ptr(xd), 2 * Cc, L.ptr(y), Cc, L.stream_ptr()))
    dPd, dsd = nhwc(dP, dtype), nhwc(dskip, dtype)
    dx = torch.empty(B, H, W, Cc, dtype=tdt(dtype), device="cuda")
    L.check(lib.sx_maxpool2_backward_add(dtype, B, H, W, Cc, L.ptr(xd), 2 * Cc, L.ptr(dPd), Cc, L.ptr(dsd), Cc, L.ptr(dx), Cc, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(y, Cc), P.detach())
    assert rel_err(from_nhwc(dx, Cc), ref_dx) < (1e-6 if dtype == 0 else 1e-2)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,H,W,Cc,film", [(2, 16, 8, 16, True), (3, 6, 10, 24, False), (2, 32, 16, 64, True), (1, 8, 8, 512, False)])
def test_pool_positions_route_the_gradient_like_the_plain_backward(L, dtype, B, H, W, Cc, film):
    """sx_block_out_pool_idx_forward records the window position of every pooled value; sx_maxpool2_backward_add_idx must give
    bit-identical dx to sx_maxpool2_backward_add reading the block output itself -- with exact ties (first maximum wins), NaNs
    (a NaN beats any number) and a strided output (the skip half of a concat buffer).  Outputs of the recording launch equal
    the plain fused launch's."""
    g = torch.Generator().manual_seed(Cc + 3 * H)
    G = 8
    y2 = q(torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.3, dtype)
    y2[:, :, :2, :4] = 0.5                       # exact ties inside windows
    s = q(torch.randn(B, Cc, H, W, generator=g), dtype)
    s[:, :, :2, :4] = -0.25
    s[0, 1, 2, 2] = float("nan")                 # a NaN in one window, second position
    s[0, 2, 3, 3] = float("nan")                 # ... and in the last position of another
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda()
    beta = (0.2 * torch.randn(Cc, generator=g)).cuda()
    gf = (1 + 0.3 * torch.randn(B, Cc, generator=g)).cuda().contiguous()
    bf = (0.3 * torch.randn(B, Cc, generator=g)).cuda().contiguous()
    yg = y2.view(B, G, -1)
    stats = torch.stack([yg.sum(-1), (yg * yg).sum(-1)], -1).cuda().contiguous()
    yd, sd = nhwc(y2, dtype), nhwc(s, dtype)
    lib = L.lib()
    fg, fb = (L.ptr(gf), L.ptr(bf)) if film else (None, None)
    outs = []
    for rec in (False, True):
        out = torch.zeros(B, H, W, 2 * Cc, dtype=tdt(dtype), device="cuda")
        pool = torch.zeros(B, H // 2, W // 2, Cc, dtype=tdt(dtype), device="cuda")
        idx = torch.full((B, H // 2, W // 2, Cc + 8), 9, dtype=torch.uint8, device="cuda")
        L.check(lib.sx_block_out_pool_idx_forward(dtype, B, H, W, Cc, G, 1e-5, L.ptr(yd), Cc, L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                                                  L.ptr(sd), Cc, fg, fb, None, L.ptr(out), 2 * Cc, None, 0, L.ptr(pool), Cc,
                                                  L.ptr(idx) if rec else None, Cc + 8, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((out, pool, idx))
    (o0, p0, _), (out, pool, idx) = outs
    assert torch.equal(o0.view(torch.int16 if dtype == 1 else torch.int32), out.view(torch.int16 if dtype == 1 else torch.int32))
    assert torch.equal(p0.view(torch.int16 if dtype == 1 else torch.int32), pool.view(torch.int16 if dtype == 1 else torch.int32))
    assert int(idx[..., :Cc].max()) <= 3 and int(idx[..., Cc:].min()) == 9        # positions 0..3, padding bytes untouched
    assert int(idx[:, 0, :2, :Cc].max()) == 0                                     # ties: the first element of the window
    dP = nhwc(q(torch.randn(B, Cc, H // 2, W // 2, generator=g), dtype), dtype)
    dskip = nhwc(q(torch.randn(B, Cc, H, W, generator=g), dtype), dtype)
    dx0 = torch.empty(B, H, W, Cc, dtype=tdt(dtype), device="cuda")
    dx1 = torch.empty_like(dx0)
    L.check(lib.sx_maxpool2_backward_add(dtype, B, H, W, Cc, L.ptr(out), 2 * Cc, L.ptr(dP), Cc, L.ptr(dskip), Cc, L.ptr(dx0), Cc, L.stream_ptr()))
    L.check(lib.sx_maxpool2_backward_add_idx(dtype, B, H, W, Cc, L.ptr(idx), Cc + 8, L.ptr(dP), Cc, L.ptr(dskip), Cc, L.ptr(dx1), Cc,
                                             L.stream_ptr()))
    torch.cuda.synchronize()
    it = torch.int16 if dtype == 1 else torch.int32
    assert torch.equal(dx0.view(it), dx1.view(it))
    # without a skip gradient, and the refusals
    L.check(lib.sx_maxpool2_backward_add(dtype, B, H, W, Cc, L.ptr(out), 2 * Cc, L.ptr(dP), Cc, None, 0, L.ptr(dx0), Cc, L.stream_ptr()))
    L.check(lib.sx_maxpool2_backward_add_idx(dtype, B, H, W, Cc, L.ptr(idx), Cc + 8, L.ptr(dP), Cc, None, 0, L.ptr(dx1), Cc, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dx0.view(it), dx1.view(it))
    assert lib.sx_maxpool2_backward_add_idx(dtype, B, H + 1, W, Cc, L.ptr(idx), Cc + 8, L.ptr(dP), Cc, None, 0, L.ptr(dx1), Cc, L.stream_ptr()) != 0
    assert lib.sx_maxpool2_backward_add_idx(dtype, B, H, W, Cc, L.ptr(idx), Cc + 4, L.ptr(dP), Cc, None, 0, L.ptr(dx1), Cc, L.stream_ptr()) != 0


def test_loss_cases_match_golden(L, golden):
    """Every option combination of masked_weighted_loss: value and gradient vs the reference's vectors."""
    from unet_amd.losses import masked_weighted_loss, mae_norm
    g = golden("loss_cases.npz")
    pred, target, mask, w, cw = (t(g[k]).cuda() for k in ("pred", "target", "mask", "w", "cw"))
    worst_l = worst_g = 0.0
    for i, d in enumerate(g["desc"]):
        base, gmode, gbase, gds, ms, use_w, use_cw, lam_w, lam_grad = str(d).split("|")
        p = pred.clone().requires_grad_(True)
        Lv = masked_weighted_loss(p, target, mask, w=w if int(use_w) else None, lam_grad=float(lam_grad), lam_w=float(lam_w),
                                  base=base, huber_beta=0.05, grad_mode=gmode, grad_base=gbase, multiscale=int(ms),
                                  ms_weight=0.5, grad_ds=int(gds), channel_weights=cw if int(use_cw) else None)
        Lv.backward()
        el = abs(float(Lv) - g["losses"][i]) / abs(g["losses"][i])
        eg = max_err(p.grad.cpu(), g["grads"][i])
        worst_l, worst_g = max(worst_l, el), max(worst_g, eg)
        assert el < 1e-4, (d, float(Lv), g["losses"][i])
        assert eg < 1e-4, d
    z = masked_weighted_loss(pred, target, torch.zeros_like(mask), w=w, lam_grad=0.1, lam_w=0.5, grad_ds=2, channel_weights=cw)
    assert float(z) == float(g["loss_zero_mask"]) == 0.0
    assert abs(float(mae_norm(pred, target, mask)) - float(g["mae_norm"])) < 1e-5
    with pytest.raises(ValueError):
        masked_weighted_loss(pred, target, mask, base="nope")
    with pytest.raises(ValueError):
        masked_weighted_loss(pred, target, mask, grad_base="nope")


def test_loss_bf16_nhwc_prediction(L, golden):
    from unet_amd.losses import masked_weighted_loss
    from oracle import loss_ref
    g = golden("loss_cases.npz")
    pred, target, mask, w, cw = (t(g[k]) for k in ("pred", "target", "mask", "w", "cw"))
    pb = pred.to(torch.bfloat16)
    pr = pb.float().requires_grad_(True)
    ref = loss_ref.masked_weighted_loss(pr, target, mask, w=w, lam_grad=0.1, lam_w=0.5, grad_ds=2, channel_weights=cw)
    ref.backward()
    pd = pb.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = masked_weighted_loss(pd, target.cuda(), mask.cuda(), w=w.cuda(), lam_grad=0.1, lam_w=0.5, grad_ds=2, channel_weights=cw.cuda())
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_err(pd.grad.float().cpu(), pr.grad) < 1e-2


@pytest.mark.parametrize("tdtype", [torch.bfloat16, torch.float32])
def test_loss_forms_the_head_bias_gradient_on_the_way_out(L, tdtype):
    """sx_loss_forward_backward_bias: the slots sum to the per-channel sums of d_pred as stored, the gradient itself is the plain
    call's; configurations its kernels do not cover report 0 and leave the slots alone."""
    from unet_amd.losses import LossBuffers, loss_prepare, loss_forward_backward
    g = torch.Generator().manual_seed(11)
    B, Cc, H, W = 8, 8, 32, 16
    sx = 1 if tdtype == torch.bfloat16 else 0
    pred = torch.randn(B, H, W, 8, generator=g).to(tdtype).cuda()
    target = torch.randn(B, Cc, H, W, generator=g).cuda()
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.2).float().cuda()
    w = (1 + torch.rand(H, W, generator=g)).cuda()
    res = {}
    for grad_ds, ms in ((2, 0), (1, 0), (2, 2)):
        bufs = LossBuffers(B, Cc, H, W, ms, grad_ds, pred.device)
        loss_prepare(bufs, B, Cc, H, W, mask, w, ms, grad_ds, 0.1)
        d0, d1 = torch.empty_like(pred), torch.empty_like(pred)
        out0 = torch.zeros(4, device="cuda"); out1 = torch.zeros(4, device="cuda")
        args = (bufs, sx, B, Cc, H, W, pred, 8, target, mask, w, None, "huber", 0.05, 0.5, 0.1, "vector", "l1", ms, 0.5, grad_ds)
        assert loss_forward_backward(*args, d0, 8, 0.5, out=out0) is False
        slots = torch.full((L.SX_HEAD_BIAS_SLOTS * Cc,), 0.0, device="cuda")
        loss_prepare(bufs, B, Cc, H, W, mask, w, ms, grad_ds, 0.1)
        done = loss_forward_backward(*args, d1, 8, 0.5, out=out1, bias_sums=slots)
        torch.cuda.synchronize()
        # (the multi-scale denominators are fp32 atomics too: its two runs agree to the last bits only)
        assert torch.equal(d0, d1) if ms == 0 else torch.allclose(d0.float(), d1.float(), rtol=1e-2 if sx else 1e-5, atol=1e-9)
        assert torch.allclose(out0, out1, rtol=1e-5, atol=0)      # the numerators are fp32 atomics: equal up to order
        res[(grad_ds, ms)] = done
        if done:
            got = slots.view(L.SX_HEAD_BIAS_SLOTS, Cc).sum(0)
            ref = d1.float().sum((0, 1, 2))
            assert rel_err(got.cpu(), ref.cpu()) < 1e-5
        else:
            assert float(slots.abs().max()) == 0.0
    assert res[(2, 0)] is True and res[(1, 0)] is False and res[(2, 2)] is False


@pytest.mark.parametrize("tdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("Cc,H,W,gmode,gbase", [(8, 72, 136, "vector", "l1"), (5, 72, 136, "magnitude", "huber"),
                                               (8, 32, 16, "vector", "mse"), (3, 34, 66, "vector", "l1")])
def test_loss_tile_kernel_against_the_oracle_across_tile_borders(L, tdtype, Cc, H, W, gmode, gbase):
    """The one-kernel training loss (16 x 32 pooling windows per tile): grids of several tiles in both directions with partial
    tiles at the borders, channel counts that leave one channel group partly or wholly empty; value and d_pred against the
    oracle's autograd on the same (rounded) prediction."""
    from unet_amd.losses import LossBuffers, loss_prepare, loss_forward_backward
    from oracle import loss_ref
    g = torch.Generator().manual_seed(H + W + Cc)
    B = 3
    sx = 1 if tdtype == torch.bfloat16 else 0
    pred = torch.zeros(B, H, W, 8)
    pred[..., :Cc] = torch.randn(B, H, W, Cc, generator=g)
    pred = pred.to(tdtype)
    target = torch.randn(B, Cc, H, W, generator=g)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.25).float()
    w = 1 + torch.rand(H, W, generator=g)
    cw = 0.5 + torch.rand(Cc, generator=g)
    pr = pred.float()[..., :Cc].permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = loss_ref.masked_weighted_loss(pr, target, mask, w=w, lam_grad=0.1, lam_w=0.5, base="huber", huber_beta=0.05,
                                        grad_mode=gmode, grad_base=gbase, multiscale=0, grad_ds=2, channel_weights=cw)
    ref.backward()
    bufs = LossBuffers(B, Cc, H, W, 0, 2, "cuda")
    md, wd, cwd, pd, td = mask.cuda(), w.cuda(), cw.cuda(), pred.cuda(), target.cuda()
    loss_prepare(bufs, B, Cc, H, W, md, wd, 0, 2, 0.1)
    d = torch.full_like(pd, 7.0)
    out = torch.zeros(4, device="cuda")
    slots = torch.zeros(L.SX_HEAD_BIAS_SLOTS * Cc, device="cuda")
    done = loss_forward_backward(bufs, sx, B, Cc, H, W, pd, 8, td, md, wd, cwd, "huber", 0.05, 0.5, 0.1, gmode, gbase, 0, 0.5, 2,
                                 d, 8, 1.0, out=out, bias_sums=slots)
    torch.cuda.synchronize()
    assert done
    assert abs(float(out[0]) - float(ref)) < 1e-5 * abs(float(ref))
    got = d.float().cpu()
    assert float(got[..., Cc:].abs().max()) == 0.0 if Cc < 8 else True
    gref = pr.grad.permute(0, 2, 3, 1)
    tol = 1e-2 if sx else 1e-5
    assert rel_err(got[..., :Cc], gref) < tol
    assert rel_err(slots.view(L.SX_HEAD_BIAS_SLOTS, Cc).sum(0).cpu(), got[..., :Cc].sum((0, 1, 2))) < 1e-5


def test_normalizers_and_error_sums_match_golden(L, golden):
    from unet_amd import data as D
    from unet_amd.losses import batch_error_sums_ev, edge_weights
    g = golden("norm_cases.npz")
    cls = {0: D.MaskedLogStandardizer, 1: D.MaskedLinearStandardizer, 2: D.MaskedSymLogStandardizer}
    norms, keys = {}, []
    for i, (k, mu, sg, ep, sc) in enumerate(zip(g["kinds"], g["mus"], g["sigmas"], g["epss"], g["scales"])):
        n = cls[int(k)](scale=float(sc), eps=float(ep)) if int(k) == 2 else cls[int(k)](eps=float(ep))
        n.mu, n.sigma = torch.tensor(float(mu)), torch.tensor(float(sg))
        norms[f"c{i}"] = n
        keys.append(f"c{i}")
    norm = D.MultiChannelNormalizer(keys, norms)
    mask = t(g["mask"])
    for devname in ("cpu", "cuda"):
        z = norm.transform(t(g["y_phys"]).to(devname), mask.to(devname))
        np.testing.assert_allclose(z.cpu().numpy(), g["z"], rtol=2e-5, atol=2e-6)
        inv = norm.inverse(t(g["zz"]).to(devname), mask.to(devname))
        np.testing.assert_allclose(inv.cpu().numpy(), g["inv"], rtol=2e-5, atol=1e-5)
    a, s, px = batch_error_sums_ev(t(g["pred"]).cuda(), t(g["zz"]).cuda(), mask.cuda(), norm)
    assert abs(float(a) - float(g["abs_sum"])) < 1e-4 * float(g["abs_sum"])
    assert abs(float(s) - float(g["sq_sum"])) < 1e-4 * float(g["sq_sum"])
    assert float(px) == float(g["px"])
    np.testing.assert_allclose(edge_weights(g["edge_mask"], 3.0), g["edge_w"], rtol=1e-6, atol=1e-7)


def test_adamw_clip_matches_golden(L, golden):
    g = golden("adamw_clip.npz")
    n = int(g["n_tensors"])
    ps = [t(g[f"p0_{i}"]) for i in range(n)]
    sizes = [p.numel() for p in ps]
    flat_p = torch.cat([p.reshape(-1) for p in ps]).cuda()
    m = torch.zeros_like(flat_p)
    v = torch.zeros_like(flat_p)
    ss = torch.zeros(2, device="cuda")       # two sum-of-squares slots: each step's kernel clears the next step's slot
    gn = torch.zeros(1, device="cuda")
    parts = torch.zeros(256, device="cuda")  # SX_SUMSQ_PARTS: caller-owned block partials of the norm
    lib = L.lib()
    for step in range(3):
        flat_g = torch.cat([t(g[f"g{step}_{i}"]).reshape(-1) for i in range(n)]).cuda()
        cur, nxt = ss[step & 1:], ss[(step + 1) & 1:]
        L.check(lib.sx_grad_sumsq(L.ptr(flat_g), flat_g.numel(), L.ptr(cur), L.ptr(parts), L.stream_ptr()))
        L.check(lib.sx_adamw_step(L.ptr(flat_p), L.ptr(flat_g), L.ptr(m), L.ptr(v), flat_p.numel(), float(g["lr"]), 0.9, 0.999,
                                  1e-8, float(g["wd"]), step + 1, float(g["clip"]), L.ptr(cur), 1.0, L.ptr(gn), L.ptr(nxt),
                                  L.stream_ptr()))
        torch.cuda.synchronize()
        assert abs(float(gn) - float(g[f"gnorm_{step}"])) < 1e-5 * float(g[f"gnorm_{step}"])
        off = 0
        for i in range(n):
            assert max_err(flat_p[off:off + sizes[i]].cpu(), g[f"p{step + 1}_{i}"].reshape(-1)) < 2e-6
            off += sizes[i]
    # the two-launch form (partials + an AdamW launch that totals them itself) is bit-identical to the three-launch form,
    # rejects partials that belong to another size, and sx_adamw_step without a norm never clips
    p_a, p_b = flat_p.clone(), flat_p.clone()
    m_a, m_b, v_a, v_b = m.clone(), m.clone(), v.clone(), v.clone()
    ss.zero_()
    L.check(lib.sx_grad_sumsq(L.ptr(flat_g), flat_g.numel(), L.ptr(ss), L.ptr(parts), L.stream_ptr()))
    L.check(lib.sx_adamw_step(L.ptr(p_a), L.ptr(flat_g), L.ptr(m_a), L.ptr(v_a), p_a.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-4, 4,
                              float(g["clip"]), L.ptr(ss), 1.0, None, None, L.stream_ptr()))
    parts2 = torch.zeros(256, device="cuda")
    npart = lib.sx_grad_sumsq_partial(L.ptr(flat_g), flat_g.numel(), L.ptr(parts2), L.stream_ptr())
    assert npart >= 1
    L.check(lib.sx_adamw_step_fused(L.ptr(p_b), L.ptr(flat_g), L.ptr(m_b), L.ptr(v_b), p_b.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-4, 4,
                                    float(g["clip"]), L.ptr(parts2), npart, 1.0, L.ptr(gn), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(p_a, p_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b)
    with pytest.raises(ValueError):
        L.check(lib.sx_adamw_step_fused(L.ptr(p_b), L.ptr(flat_g), L.ptr(m_b), L.ptr(v_b), p_b.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-4,
                                        5, float(g["clip"]), L.ptr(parts2), npart + 1, 1.0, None, L.stream_ptr()))
    p_c, m_c, v_c = flat_p.clone(), m.clone(), v.clone()
    big = flat_g * 1e3                       # far above the clip threshold
    L.check(lib.sx_adamw_step(L.ptr(p_c), L.ptr(big), L.ptr(m_c), L.ptr(v_c), p_c.numel(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1,
                              float(g["clip"]), None, 1.0, None, None, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.allclose(m_c, 0.9 * m + 0.1 * big, rtol=1e-6, atol=0)          # unclipped gradient went into the moment


def test_fp32_split_arithmetic_error_against_fp64(L, fp32_mode):
    """What the split-bf16 mode costs in accuracy, measured against fp64 on one 3x3 convolution and its weight gradient
    (256 -> 256 channels, 16x8): <= 2e-5 relative L2 (three bf16 products of hi/lo planes, fp32 accumulation; bound per
    product 3 * 2^-18), exact mode <= 2e-6.  TF32, cuDNN's default for the reference's fp32 convolutions, sits at ~5e-4."""
    B, H, W, Cc = 4, 16, 8, 256
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5)
    dy = torch.randn(B, Cc, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    xd, dyd = nhwc(x, 0), nhwc(dy, 0)
    y = torch.empty(B, H, W, Cc, dtype=torch.float32, device="cuda")
    wp = pack(L, 0, L.SX_CONV3, 0, w, Cc, Cc, Cc, Cc)
    L.check(L.lib().sx_conv_forward(0, L.SX_CONV3, B, H, W, Cc, Cc, L.ptr(xd), Cc, H, W, L.ptr(wp), None, L.ptr(y), Cc, H, W, 0,
                                    None, 0, L.stream_ptr()))
    nb = L.lib().sx_conv_wgrad_workspace_bytes(L.SX_CONV3, B, H, W, Cc, Cc)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.empty(Cc, Cc, 3, 3, dtype=torch.float32, device="cuda")
    L.check(L.lib().sx_conv_wgrad(0, L.SX_CONV3, B, H, W, Cc, Cc, Cc, Cc, L.ptr(dyd), Cc, L.ptr(xd), Cc, H, W, L.ptr(dw), 0,
                                  L.ptr(ws), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    e_y = float((from_nhwc(y, Cc).double() - ref).norm() / ref.norm())
    e_w = float((dw.cpu().double() - wref).norm() / wref.norm())
    print(f"[fp32 {fp32_mode}] conv {e_y:.2e}  wgrad {e_w:.2e}")
    # three planes: fp32-grade convolutions; the weight gradients stay on two planes in that mode (their error does not
    # reach the network's gradients: tools/fp32_mix.py)
    assert e_y < (2e-5 if fp32_mode == "split" else 2e-6)
    assert e_w < (2e-6 if fp32_mode == "exact" else 2e-5)
    if fp32_mode == "split":
        assert e_y > 5e-7          # the split kernels really ran (the exact ones sit at ~2e-7)


def _rand_shapes(n, seed):
    """Seeded random problem shapes: every grid-width class (<= 4, <= 8, wider), ragged tiles, channel counts on and off the
    16 / 32 / 64 boundaries."""
    r = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        W = int(r.choice([2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 33]))
        H = int(r.choice([1, 2, 5, 8, 13, 16, 31, 32, 40]))
        Cin = int(r.choice([4, 8, 12, 16, 24, 32, 40, 64, 72, 128]))
        Cout = int(r.choice([4, 8, 12, 16, 24, 32, 48, 64, 80, 128]))
        B = int(r.choice([1, 2, 3, 5]))
        out.append((B, H, W, Cin, Cout))
    return out


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_conv_and_wgrad_random_shapes(L, fp32_mode, mode, dtype):
    """Both tiers (fp32 in both arithmetic modes), 40 seeded random shapes per kernel size: forward (+ GroupNorm statistics), data gradient and weight gradient
    of 3x3 / 1x1 convolutions against torch -- tile edges, 4- and 8-wide grids, K and row counts off the chunk sizes."""
    k = 3 if mode == 0 else 1
    md = L.SX_CONV3 if mode == 0 else L.SX_CONV1
    for (B, H, W, Cin, Cout) in _rand_shapes(40, 100 + mode):
        g = torch.Generator().manual_seed(B + 7 * H + 13 * W + Cin + 3 * Cout)
        if dtype == 1 and (Cin % 8 or Cout % 8):
            continue                                   # bf16 moves 8 channels per 16-byte piece
        x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
        w = q(torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5), dtype)
        dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
        tag = (dtype, mode, B, H, W, Cin, Cout)
        pad_c = 4 if dtype == 0 else 8
        tol = TOL[dtype]
        ref = F.conv2d(x, w, padding=k // 2)
        xd, dyd = nhwc(x, dtype, ld=Cin + pad_c), nhwc(dy, dtype)
        y = torch.full((B, H, W, Cout + pad_c), 7.0, dtype=tdt(dtype), device="cuda")
        G = 4 if Cout % 4 == 0 else 1
        stats = torch.zeros(B, G, 2, dtype=torch.float32, device="cuda")
        wp = pack(L, dtype, md, 0, w, Cout, Cin, Cout, Cin)
        L.check(L.lib().sx_conv_forward(dtype, md, B, H, W, Cin, Cout, L.ptr(xd), Cin + pad_c, H, W, L.ptr(wp), None, L.ptr(y), Cout + pad_c,
                                        H, W, 0, L.ptr(stats), G, L.stream_ptr()))
        wpd = pack(L, dtype, md, 1, w, Cout, Cin, Cin, Cout)
        dx = torch.empty(B, H, W, Cin, dtype=tdt(dtype), device="cuda")
        L.check(L.lib().sx_conv_forward(dtype, md, B, H, W, Cout, Cin, L.ptr(dyd), Cout, H, W, L.ptr(wpd), None, L.ptr(dx), Cin, H, W,
                                        0, None, 0, L.stream_ptr()))
        nb = L.lib().sx_conv_wgrad_workspace_bytes(md, B, H, W, Cout, Cin)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device="cuda")
        L.check(L.lib().sx_conv_wgrad(dtype, md, B, H, W, Cout, Cin, Cout, Cin, L.ptr(dyd), Cout, L.ptr(xd), Cin + pad_c, H, W, L.ptr(dw), 0,
                                      L.ptr(ws), nb, L.stream_ptr()))
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(y, Cout), ref) < tol, tag
        assert float((y[..., Cout:].float() - 7.0).abs().max()) == 0.0, tag
        rg = ref.view(B, G, -1)
        assert rel_err(stats.cpu(), torch.stack([rg.sum(-1), (rg * rg).sum(-1)], dim=-1)) < tol, tag
        assert rel_err(from_nhwc(dx, Cin), torch.nn.grad.conv2d_input(x.shape, w, dy, padding=k // 2)) < tol, tag
        assert rel_err(dw.cpu(), torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=k // 2)) < tol, tag


def test_fp32_conv_transpose_random_shapes(L, fp32_mode):
    """fp32 tier: ConvTranspose2d(2, 2) forward (+ bias), data gradient and weight gradient on 24 seeded random shapes."""
    r = np.random.RandomState(5)
    for _ in range(24):
        B = int(r.choice([1, 2, 3])); h = int(r.choice([1, 3, 4, 8, 9, 16])); w = int(r.choice([1, 2, 4, 5, 8, 12, 17]))
        Cin = int(r.choice([8, 16, 32, 64, 128])); Cout = int(r.choice([4, 8, 16, 32, 64]))
        g = torch.Generator().manual_seed(B + 5 * h + 11 * w + Cin + Cout)
        x = torch.randn(B, Cin, h, w, generator=g)
        wt = torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5
        bias = torch.randn(Cout, generator=g)
        dy = torch.randn(B, Cout, 2 * h, 2 * w, generator=g)
        tag = (B, h, w, Cin, Cout)
        xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        ref = F.conv_transpose2d(xr, wr, bias, stride=2)
        ref.backward(dy)
        xd, dyd = nhwc(x, 0), nhwc(dy, 0)
        y = torch.empty(B, 2 * h, 2 * w, Cout, dtype=torch.float32, device="cuda")
        wp = pack(L, 0, L.SX_CONVT_FWD, 0, wt, Cout, Cin, Cout, Cin)
        bd = bias.cuda()
        L.check(L.lib().sx_conv_forward(0, L.SX_CONVT_FWD, B, h, w, Cin, Cout, L.ptr(xd), Cin, h, w, L.ptr(wp), L.ptr(bd),
                                        L.ptr(y), Cout, 2 * h, 2 * w, 0, None, 0, L.stream_ptr()))
        wpb = pack(L, 0, L.SX_CONVT_BWD, 0, wt, Cout, Cin, Cin, Cout)
        dx = torch.empty(B, h, w, Cin, dtype=torch.float32, device="cuda")
        L.check(L.lib().sx_conv_forward(0, L.SX_CONVT_BWD, B, h, w, Cout, Cin, L.ptr(dyd), Cout, 2 * h, 2 * w, L.ptr(wpb), None,
                                        L.ptr(dx), Cin, h, w, 0, None, 0, L.stream_ptr()))
        nb = L.lib().sx_conv_wgrad_workspace_bytes(L.SX_CONVT_FWD, B, h, w, Cin, Cout)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        dw = torch.empty(Cin, Cout, 2, 2, dtype=torch.float32, device="cuda")
        L.check(L.lib().sx_conv_wgrad(0, L.SX_CONVT_FWD, B, h, w, Cin, Cout, Cin, Cout, L.ptr(xd), Cin, L.ptr(dyd), Cout,
                                      2 * h, 2 * w, L.ptr(dw), 0, L.ptr(ws), nb, L.stream_ptr()))
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(y, Cout), ref.detach()) < 1e-4, tag
        assert rel_err(from_nhwc(dx, Cin), xr.grad) < 1e-4, tag
        assert rel_err(dw.cpu(), wr.grad) < 1e-4, tag
