#!/usr/bin/env python3
"""Per-launch timing of every conv_ws_kernel launch of one cfg-A training step (GPU box).
env SHAPE="kind,h,w,cin,cout" selects one launch (kind: dual / c3 / c1 / center), NOSTATS=1 drops the GroupNorm statistics."""
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
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) - 0.5).to(tdt)
packed = lambda mode, rows, k: ((torch.rand(lib.sx_packed_weight_elems(mode, 1, rows, k), device="cuda", generator=g) - 0.5) * 0.1).to(tdt)
tot = 0.0
for (kind, h, w, cin, cout, fl) in bench.conv_launches(B):
    if os.environ.get("SHAPE") and os.environ["SHAPE"] != f"{kind},{h},{w},{cin},{cout}": continue
    stats = None if os.environ.get("NOSTATS") else torch.zeros(B, 8, 2, device="cuda")
    if kind == "dual":
        x, y, s2, w3, w1 = rnd(B, h, w, cin), rnd(B, h, w, cout), rnd(B, h, w, cout), packed(0, cout, cin), packed(1, cout, cin)
        by = (x.numel() + 2 * y.numel()) * 2
        def launch():
            L.check(lib.sx_conv3_shortcut_forward(1, B, h, w, cin, cout, L.ptr(x), cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout,
                                                  L.ptr(stats), 8 if stats is not None else 0, L.ptr(s2), cout, L.stream_ptr()))
    elif kind == "center":
        x, y, w3, w1 = rnd(B, h, w, 2 * cin), rnd(B, h, w, cout), packed(0, cout, cin), packed(1, cout, cin)
        by = (x.numel() + y.numel()) * 2
        def launch():
            L.check(lib.sx_conv3_shortcut_dgrad(1, B, h, w, cin, cout, L.ptr(x), 2 * cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout, None, 0, L.stream_ptr()))
    else:
        mode = 0 if kind == "c3" else 1
        x, y, wp = rnd(B, h, w, cin), rnd(B, h, w, cout), packed(mode, cout, cin)
        by = (x.numel() + y.numel()) * 2
        sg = stats if mode == 0 else None
        def launch():
            L.check(lib.sx_conv_forward(1, mode, B, h, w, cin, cout, L.ptr(x), cin, h, w, L.ptr(wp), None, L.ptr(y), cout, h, w, 0,
                                        L.ptr(sg), 8 if sg is not None else 0, L.stream_ptr()))
    for _ in range(3): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): launch()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"{kind:6s} {h:4d}x{w:<3d} {cin:4d}->{cout:<4d} {us:8.1f} us  {fl/us/1e6:8.1f} TF/s  {by/us/1e3:8.1f} GB/s (min traffic {by/1e6:.1f} MB)")
print(f"total {tot:.1f} us")
