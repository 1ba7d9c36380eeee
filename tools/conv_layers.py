#!/usr/bin/env python3
"""Per-launch table of the conv_ws_kernel family: every launch of one cfg-A training step (bench.conv_launches), each timed
on its own (20 back-to-back repeats between HIP events -- operands that fit the 256 MB MALL are served from it, so these are
best-case figures; the in-step durations are in profiles/*_sequence.txt).  usage: [CFG=5] conv_layers.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unet_amd import _lib as L

arch = bench.ARCH_B if os.environ.get("CFG") == "5" else bench.ARCH_A       # CFG=5: BASELINE configs[4] (batch 32)
B = int(sys.argv[1]) if len(sys.argv) > 1 else (32 if arch is bench.ARCH_B else 64)
make = bench.conv_launch_maker(B, L.SX_BF16)
st = torch.cuda.current_stream()
tot = 0.0
print(f"{'kind':6s} {'grid':>8s} {'channels':>10s} {'us':>8s} {'TF/s':>8s} {'GB/s':>8s} {'MB':>7s}")
for (kind, h, w, cin, cout, fl) in bench.conv_launches(B, arch):
    launch, tensors = make(kind, h, w, cin, cout)
    by = sum(t.numel() * t.element_size() for t in tensors)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        launch()
    e1.record(st)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"{kind:6s} {h:4d}x{w:<3d} {cin:4d}->{cout:<4d} {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:8.1f} {by / 1e6:7.1f}")
print(f"total {tot:.1f} us over {len(bench.conv_launches(B, arch))} launches")
