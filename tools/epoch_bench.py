#!/usr/bin/env python3
"""End-to-end epoch throughput of train_unet's inner loop fed by (a) the reference-style host DataLoader and
(b) the device-resident loader, on a synthetic SOLPS-shaped npz (GPU box).  usage: epoch_bench.py [N] [batch]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unet_amd import data as D
from unet_amd.models import UNet
from unet_amd.train import TrainEngine
from unet_amd.losses import edge_weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W, C = 128, 64, 8
rng = np.random.default_rng(0)
ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
rho = np.sqrt(((ii - H / 2) / (0.47 * H)) ** 2 + ((jj - W / 2) / (0.45 * W)) ** 2)
mask = ((rho >= 0.45) & (rho <= 1.0)).astype(np.float32)
keys = ["Te", "Ti", "ne", "ua", "Sp", "Qe", "Qi", "Sm"]
Y = np.abs(rng.normal(size=(N, C, H, W))).astype(np.float32) * 10
path = os.path.join(tempfile.mkdtemp(), "synth.npz")
np.savez(path, Y=Y, y_keys=np.array(keys), mask=mask, X=rng.normal(size=(N, 5)).astype(np.float32))
for resident in (False, True):
    norms = {k: D.MaskedLogStandardizer(eps=1.0) for k in keys}
    t0 = time.perf_counter()
    tl, vl, norm, P, hw, scaler, pt, pk = D.make_loaders(path, batch_size=B, y_keys=keys, norms_by_name=norms, device="cuda",
                                                         resident=resident, split=1.0 - B / N)
    t_build = time.perf_counter() - t0
    torch.manual_seed(0)
    model = UNet(in_ch=1, out_ch=C, base=32, P=P, film_hidden=128, depth=3).cuda().train()
    w_map = torch.from_numpy(edge_weights(mask, 3.0))
    eng = TrainEngine(model, dtype=torch.bfloat16, lr=3e-4, channel_weights=[1, 1, 1.2, 1.5, 1.2, 1.2, 1.2, 1.2], w_map=w_map,
                      loss_opts=dict(lam_w=0.5, lam_grad=0.1, grad_ds=2))
    for epoch in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        for b in tl:
            x, y, m, p = (b[k].to("cuda", non_blocking=True) for k in ("x", "y", "mask", "params"))
            eng.train_step(x, y, m, p, lam_grad=0.1)
            n += x.shape[0]
        eng.read_epoch()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{'resident' if resident else 'host    '} loader  epoch {epoch}: {n} samples in {dt*1e3:8.1f} ms = {n/dt:9.1f} samples/s"
              f"   (loader build {t_build:.2f} s)", flush=True)
