#!/usr/bin/env python3
"""Diagnostic (DBG build of wgrad_ws.hip): in-kernel s_memtime stamps of one wgrad launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib as L
lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode, h, w, ca, cb = [int(v) for v in os.environ.get("SHAPE", "0,128,64,32,32").split(",")]
P = (torch.rand(B, h, w, ca, device="cuda") - 0.5).bfloat16()
Q = (torch.rand(B, h, w, cb, device="cuda") - 0.5).bfloat16()
nb = lib.sx_conv_wgrad_workspace_bytes(mode, B, h, w, ca, cb)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
dw = torch.empty(ca * cb * 9, device="cuda")
stamps = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
os.environ["SX_WG_STAMPS"] = hex(stamps.data_ptr())
for it in range(5):
    L.check(lib.sx_conv_wgrad(1, mode, B, h, w, ca, cb, ca, cb, L.ptr(P), ca, L.ptr(Q), cb, h, w, L.ptr(dw), 0, L.ptr(ws), nb, L.stream_ptr()))
    torch.cuda.synchronize()
s = stamps.cpu().view(512, 8)[:256].double()
t0 = s[:, 0].min()
import numpy as np
names = ["entry", "setup done", "first stage landed", "loop done", "slab stored"]
for i, nme in enumerate(names):
    v = (s[:, i] - t0)
    print(f"{nme:20s} min {v.min():9.0f}  median {v.median():9.0f}  max {v.max():9.0f}   (cycles since first block entry)")
d = s[:, 1:5] - s[:, 0:4]
for i, nme in enumerate(["setup", "wait first stage", "main loop", "slab store"]):
    print(f"  per-block {nme:18s} median {d[:, i].median():9.0f} max {d[:, i].max():9.0f}")
