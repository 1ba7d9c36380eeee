#!/usr/bin/env python3
"""Stress of the loss tile kernel's last-block total (SX_LOSS_TICKET=1): the same loss 3000 times at the benchmark's shape, other
work in between; every value must agree with the first to summation-order noise.  usage: loss_ticket_stress.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib as L
from unet_amd.losses import LossBuffers, loss_prepare, loss_forward_backward
it = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator().manual_seed(5)
B, C, H, W = 64, 8, 128, 64
pred = torch.randn(B, H, W, 8, generator=g).bfloat16().cuda()
target = torch.randn(B, C, H, W, generator=g).cuda()
mask = (torch.rand(B, 1, H, W, generator=g) > 0.2).float().cuda()
w = (1 + torch.rand(H, W, generator=g)).cuda()
bufs = LossBuffers(B, C, H, W, 0, 2, "cuda")
d = torch.empty_like(pred)
outs = torch.zeros(it, 4, device="cuda")
junk = torch.randn(4096, 4096, device="cuda")
for i in range(it):
    loss_prepare(bufs, B, C, H, W, mask, w, 0, 2, 0.1)
    loss_forward_backward(bufs, L.SX_BF16, B, C, H, W, pred, 8, target, mask, w, None, "huber", 0.05, 0.5, 0.1, "vector", "l1", 0, 0.5, 2,
                          d, 8, 1.0, out=outs[i])
    if i % 3 == 0:
        junk.mul_(1.0001)          # unrelated traffic between the calls
torch.cuda.synchronize()
o = outs.cpu()
ref = o[0]
rel = ((o - ref).abs() / ref.abs().clamp_min(1e-30)).max(0).values
print("values", ref.tolist())
print("worst relative deviation from the first call over", it, "calls:", rel.tolist())
assert float(rel.max()) < 2e-6, "the last-block total disagreed with the first call"
print("ok")
