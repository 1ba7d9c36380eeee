#!/usr/bin/env python3
"""Diagnostic (library built with -DSX_WS_DBG=1): in-kernel s_memtime stamps of one conv_ws launch.
SHAPE=kind,h,w,cin,cout with kind c3 | c1 | dual | center   (default c3,64,32,64,64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib as L
lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind, h, w, cin, cout = os.environ.get("SHAPE", "c3,64,32,64,64").split(",")
h, w, cin, cout = int(h), int(w), int(cin), int(cout)
tdt = torch.bfloat16
def rnd(*s): return (torch.rand(*s, device="cuda") - 0.5).to(tdt)
def packed(mode, rows, k): return (rnd(lib.sx_packed_weight_elems(mode, 1, rows, k)) * 0.1).to(tdt)
stamps = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
os.environ["SX_WS_STAMPS"] = hex(stamps.data_ptr())
stats = torch.zeros(B, 8, 2, device="cuda")
if kind == "dual":
    x, y, s2 = rnd(B, h, w, cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
    w3, w1 = packed(0, cout, cin), packed(1, cout, cin)
    def launch(): L.check(lib.sx_conv3_shortcut_forward(1, B, h, w, cin, cout, L.ptr(x), cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout, L.ptr(stats), 8, L.ptr(s2), cout, L.stream_ptr()))
elif kind == "center":
    x, y = rnd(B, h, w, 2 * cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
    w3, w1 = packed(0, cout, cin), packed(1, cout, cin)
    def launch(): L.check(lib.sx_conv3_shortcut_dgrad(1, B, h, w, cin, cout, L.ptr(x), 2 * cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout, None, 0, L.stream_ptr()))
else:
    mode = 0 if kind == "c3" else 1
    x, y = rnd(B, h, w, cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
    wp = packed(mode, cout, cin)
    def launch(): L.check(lib.sx_conv_forward(1, mode, B, h, w, cin, cout, L.ptr(x), cin, h, w, L.ptr(wp), None, L.ptr(y), cout, h, w, 0, L.ptr(stats) if mode == 0 else None, 8 if mode == 0 else 0, L.stream_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(5):
    stamps.zero_()
    torch.cuda.synchronize()
    e0.record(); launch(); e1.record()
    torch.cuda.synchronize()
s = stamps.cpu().view(1024, 16).double()
s = s[s[:, 0] > 0]
print(f"{kind} {h}x{w} {cin}->{cout}: {s.shape[0]} blocks, tiles per block {s[:, 6].min():.0f}..{s[:, 6].max():.0f}, event time {e0.elapsed_time(e1)*1e3:.1f} us")
t0 = s[:, 0].min()
for i, nme in enumerate(["entry", "setup done (weights + 3 stages issued)", "first stage landed", "tile loop done", "drained"]):
    v = s[:, i] - t0
    print(f"{nme:40s} min {v.min():8.0f}  median {v.median():8.0f}  max {v.max():8.0f}   (s_memtime ticks, 100 MHz -> x10 ns)")
d = s[:, 1:5] - s[:, 0:4]
for i, nme in enumerate(["setup", "wait first stage", "tile loop", "drain"]):
    print(f"  per-block {nme:18s} median {d[:, i].median():8.0f} max {d[:, i].max():8.0f}")
loop = (s[:, 3] - s[:, 2])
nst = s[:, 10].clamp(min=1)
print(f"  per stage (median block): total {(loop / nst).median():7.0f}  = wait+barrier {(s[:, 7] / nst).median():6.0f} + DMA issue {(s[:, 8] / nst).median():6.0f}"
      f" + LDS reads/MFMA {((loop - s[:, 7] - s[:, 8] - s[:, 9] - s[:, 11]) / nst).median():6.0f}; epilogue per tile {(s[:, 9] / s[:, 6].clamp(min=1)).median():6.0f}"
      + (f"; partial-sum exchange per tile {(s[:, 11] / s[:, 6].clamp(min=1)).median():6.0f}" if float(s[:, 11].max()) > 0 else ""))
