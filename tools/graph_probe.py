#!/usr/bin/env python3
"""Experiment: one training step captured in a HIP graph and replayed (timing only: step-dependent scalars are baked in)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unet_amd.train import TrainEngine
from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
from unet_amd.losses import edge_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = bench.make_model(0).cuda().train()
bt = make_batch(B, bench.H, bench.W, seed=1234)
x, y, m, p = (bt[k].cuda() for k in ("x", "y", "mask", "params"))
w_map = torch.from_numpy(edge_weights(bt["mask"][0, 0].numpy(), 3.0))
eng = TrainEngine(model, dtype=torch.bfloat16, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, channel_weights=CHANNEL_WEIGHTS_8,
                  w_map=w_map, loss_opts=bench.LOSS)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5):
        eng.train_step(x, y, m, p, lam_grad=0.1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    eng.train_step(x, y, m, p, lam_grad=0.1)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
torch.cuda.synchronize()
print(f"B={B}: graph replay {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step")
