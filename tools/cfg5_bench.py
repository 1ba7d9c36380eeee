#!/usr/bin/env python3
"""BASELINE config 5 (plasma-to-sources: in 10, out 4, base 64, depth 3, no FiLM, 256x128): training step time.
usage: cfg5_bench.py [batch] [steps] [bf16|fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd.models import UNet
from unet_amd.train import TrainEngine
from unet_amd.synth import make_batch
from unet_amd.losses import edge_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt_name = sys.argv[3] if len(sys.argv) > 3 else "bf16"
torch.manual_seed(0)
model = UNet(in_ch=10, out_ch=4, base=64, P=0, depth=3).cuda().train()
bt = make_batch(B, 256, 128, in_ch=10, out_ch=4, P=0, seed=3)
x, y, m = (bt[k].cuda() for k in ("x", "y", "mask"))
w_map = torch.from_numpy(edge_weights(bt["mask"][0, 0].numpy(), 3.0))
opts = dict(lam_w=0.5, lam_grad=0.1, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1", multiscale=0, ms_weight=0.5, grad_ds=2)
eng = TrainEngine(model, dtype=torch.bfloat16 if dt_name == "bf16" else torch.float32, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, w_map=w_map, loss_opts=opts)
for _ in range(5):
    eng.train_step(x, y, m, None, lam_grad=0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.train_step(x, y, m, None, lam_grad=0.1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"cfg5 {dt_name} B={B}: {1e3 * dt:.2f} ms/step, {B / dt:.0f} samples/s, loss {float(eng.log[eng.nlog - 1, 0]):.4f}")
