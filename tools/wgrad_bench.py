#!/usr/bin/env python3
"""Per-layer timing of every weight-gradient launch (split-K kernel + slab reduction) of one cfg-A step (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
tdt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream()
C = [32, 64, 128, 256]
HW = [(128, 64), (64, 32), (32, 16), (16, 8)]
CIN0, COUT = 8, 8
if os.environ.get("CFG") == "5":           # BASELINE configs[4]: base 64, 256x128, in 10 (padded 16), out 4 (padded 8)
    C = [64, 128, 256, 512]
    HW = [(256, 128), (128, 64), (64, 32), (32, 16)]
    CIN0 = 16
layers = []          # (mode, h, w, Ca (dY channels), Cb (input channels))
for l in range(4):
    cin = CIN0 if l == 0 else C[l - 1]
    layers += [(0, *HW[l], C[l], cin), (0, *HW[l], C[l], C[l]), (1, *HW[l], C[l], cin)]
for l in (2, 1, 0):
    layers += [(0, *HW[l], C[l], 2 * C[l]), (0, *HW[l], C[l], C[l]), (1, *HW[l], C[l], 2 * C[l]), (2, *HW[l + 1], C[l + 1], C[l])]
layers.append((1, *HW[0], COUT, C[0]))
names = {0: "3x3", 1: "1x1", 2: "cT"}
tot = 0.0
for (mode, h, w, ca, cb) in layers:
    if os.environ.get("SHAPE") and os.environ["SHAPE"] != f"{mode},{h},{w},{ca},{cb}": continue
    hq, wq = (2 * h, 2 * w) if mode == 2 else (h, w)
    P = (torch.rand(B, h, w, ca, device="cuda", generator=g) - 0.5).to(tdt)
    Q = (torch.rand(B, hq, wq, cb, device="cuda", generator=g) - 0.5).to(tdt)
    nb = lib.sx_conv_wgrad_workspace_bytes(mode, B, h, w, ca, cb)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    taps = {0: 9, 1: 1, 2: 4}[mode]
    dw = torch.empty(ca * cb * taps, device="cuda")
    def launch():
        L.check(lib.sx_conv_wgrad(1, mode, B, h, w, ca, cb, ca, cb, L.ptr(P), ca, L.ptr(Q), cb, hq, wq, L.ptr(dw), 0,
                                  L.ptr(ws), nb, L.stream_ptr()))
    for _ in range(3): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): launch()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * B * h * w * ca * cb * taps
    by = (P.numel() + Q.numel()) * 2
    tot += us
    print(f"{names[mode]:4s} {h:4d}x{w:<3d} dY {ca:4d} x in {cb:<4d} {us:8.1f} us  {fl/us/1e6:8.1f} TF/s  {by/us/1e3:8.1f} GB/s  slab {nb/1e6:7.1f} MB")
print(f"total {tot:.1f} us")
