#!/usr/bin/env python3
"""Per-launch timing of every implicit-GEMM conv launch of one cfg-A training step (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unet_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
tdt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream()
tot = 0.0
names = {0: "3x3", 1: "1x1", 2: "cT_f", 3: "cT_b"}
for (mode, h, w, cin, cout, fl) in bench.conv_launches(B):
    if os.environ.get("ONLY3") and mode != 0: continue
    if os.environ.get("SHAPE") and os.environ["SHAPE"] != f"{mode},{h},{w},{cin},{cout}": continue
    hx, wx, hy, wy = h, w, h, w
    if mode == 2: hy, wy = 2 * h, 2 * w
    if mode == 3: hx, wx = 2 * h, 2 * w
    x = (torch.rand(B, hx, wx, cin, device="cuda", generator=g) - 0.5).to(tdt)
    y = torch.empty(B, hy, wy, cout, device="cuda", dtype=tdt)
    nel = lib.sx_packed_weight_elems(mode, 1, cout, cin)
    wp = ((torch.rand(nel, device="cuda", generator=g) - 0.5) * 0.1).to(tdt)
    stats = torch.zeros(B, 8, 2, device="cuda") if (mode == 0 and not os.environ.get("NOSTATS")) else None
    def launch():
        L.check(lib.sx_conv_forward(1, mode, B, h, w, cin, cout, L.ptr(x), cin, hx, wx, L.ptr(wp), None, L.ptr(y), cout, hy, wy, 0,
                                    L.ptr(stats), 8 if stats is not None else 0, L.stream_ptr()))
    for _ in range(3): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): launch()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    by = (x.numel() + y.numel()) * 2
    tot += us
    print(f"{names[mode]:5s} {h:4d}x{w:<3d} {cin:4d}->{cout:<4d} {us:8.1f} us  {fl/us/1e6:8.1f} TF/s  {by/us/1e3:8.1f} GB/s (min traffic {by/1e6:.1f} MB)")
print(f"total {tot:.1f} us")
