#!/usr/bin/env python3
"""BASELINE config 4 shape: batched inverse (cfg A, 128x64), problem-steps/s on one GPU.  usage: inverse_bench.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unet_amd.inverse import batched_inverse
from unet_amd.synth import make_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model = bench.make_model(0).cuda().eval()
model.compute_dtype = torch.bfloat16
bt = make_batch(N, bench.H, bench.W, seed=7)
mask, p_true = bt["mask"].cuda(), bt["params"].cuda()
with torch.no_grad():
    y = torch.cat([model(mask[i:i + 64], params=p_true[i:i + 64]).float() for i in range(0, N, 64)])
p0 = torch.zeros_like(p_true)
for b in (64, 256, 1024):
    if b > N: continue
    batched_inverse(model, mask, y, p0, steps=2, early_stop_patience=0, batch=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = batched_inverse(model, mask, y, p0, steps=steps, early_stop_patience=0, batch=b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"N={N} chunk={b}: {N * steps / dt:9.0f} problem-steps/s  ({1e3 * dt / steps / (N / b):.2f} ms per chunk-step)  "
          f"mean fit {float(r['fit'].mean()):.4e}")
