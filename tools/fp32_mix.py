#!/usr/bin/env python3
"""Which kernels carry the split-bf16 error at BASELINE configs[4]'s full grid?  Worst per-tensor gradient error (smooth loss)
of one training step against the CPU oracle for: everything split / weight gradients exact / convolutions exact / all exact.
Also the step time of each mix at batch 32."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import unet_ref, loss_ref
from oracle.unet_ref import UNetCfg, init_state_dict
from unet_amd.models import UNet
from unet_amd.train import TrainEngine
from unet_amd.synth import make_batch
from unet_amd import _lib as L
torch.set_num_threads(16)
LOSS = dict(lam_w=0.5, lam_grad=0.0, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1", multiscale=0, ms_weight=0.5, grad_ds=2)
cfg = UNetCfg(in_ch=10, out_ch=4, base=64, P=0, depth=3)
sd = init_state_dict(cfg, seed=21)
bt = make_batch(2, 256, 128, in_ch=10, out_ch=4, P=0, seed=6, jitter_masks=True)
w = torch.from_numpy(loss_ref.edge_weights(bt["mask"][0, 0].numpy()))
sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
loss_ref.masked_weighted_loss(unet_ref.unet_forward(sdr, cfg, bt["x"], None), bt["y"], bt["mask"], w=w, **LOSS).backward()
gmax = max(float(v.grad.norm()) for v in sdr.values())
big = make_batch(32, 256, 128, in_ch=10, out_ch=4, P=0, seed=3)
os.environ["SX_FP32_WIDE"] = "2"
lib = L.lib()
for name, conv_exact, wg in (("all split", 0, -1), ("convs split, weight gradients exact", 0, 0), ("convs exact, weight gradients split", 1, 1),
                             ("all exact", 1, -1)):
    lib.sx_set_fp32_exact(conv_exact)
    os.environ["SX_X3_WGRAD"] = str(wg)
    model = UNet(in_ch=10, out_ch=4, base=64, P=0, depth=3)
    model.load_state_dict(sd)
    model = model.cuda().train()
    eng = TrainEngine(model, dtype=torch.float32, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, w_map=w, loss_opts=LOSS)
    eng.train_step(*(bt[k].cuda() for k in ("x", "y", "mask")), None, lam_grad=0.0)
    torch.cuda.synchronize()
    worst = (0.0, "")
    for k, q in model.named_parameters():
        gd, rd = q.grad.detach().cpu().double().reshape(-1), sdr[k].grad.double().reshape(-1)
        worst = max(worst, (float((gd - rd).norm()) / max(float(rd.norm()), 1e-3 * gmax), k))
    x, y, m = (big[k].cuda() for k in ("x", "y", "mask"))
    eng2 = TrainEngine(model, dtype=torch.float32, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, w_map=w, loss_opts=LOSS)
    for _ in range(2): eng2.train_step(x, y, m, None, lam_grad=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): eng2.train_step(x, y, m, None, lam_grad=0.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print(f"{name:40s} worst gradient {worst[0]:.3e} ({worst[1]});  step at batch 32: {1e3 * dt:.2f} ms", flush=True)
    del eng, eng2, model
