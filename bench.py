#!/usr/bin/env python3
"""Headline benchmark: conditional U-Net FULL TRAINING STEP, bf16, batch 64 per GPU, 128x64 grid
(BASELINE.json configs[1]; weak scaling at N GPUs = configs[2]).

A step = forward + production loss (huber 0.05 + edge 0.5 + Sobel-L1 0.1 at grad_ds 2, channel weights)
+ backward + gradient all-reduce (N > 1) + clip 1.0 + AdamW(lr 3e-4, wd 1e-4), on a batch already in HBM.
Prints ONE JSON line on rank 0.  `roofline` is measured live for the dominant kernel family (the
implicit-GEMM convolution) by replaying every conv launch of one step through the C ABI between
HIP events; `cpu_baseline` times the CPU oracle (a port of the reference's PyTorch path) on a
bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=24)
    return ap.parse_args()


CFG = dict(in_ch=1, out_ch=8, base=32, depth=3, P=5, film_hidden=128)
H, W = 128, 64
LOSS = dict(lam_w=0.5, lam_grad=0.1, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1",
            multiscale=0, ms_weight=0.5, grad_ds=2)


def make_model(seed=0):
    from unet_amd.models import UNet
    torch.manual_seed(seed)
    m = UNet(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3)
    with torch.no_grad():                      # FiLM output layers are zero at init: make conditioning live
        for n, p in m.named_parameters():
            if "film" in n and ".net.2." in n:
                p.normal_(0.0, 0.05)
    return m


def conv_launches(B):
    """(mode, Hc, Wc, Cin, Cout, flops) of every implicit-GEMM conv launch in one training step of cfg A:
    fprop + dgrad of the 3x3 / 1x1 / transposed convs and the head (wgrad is a different kernel)."""
    C = [32, 64, 128, 256]
    Hs = [(128, 64), (64, 32), (32, 16), (16, 8)]
    out = []

    def add(mode, hw, cin, cout, taps, n=1):
        h, w = hw
        fl = 2.0 * B * h * w * cin * cout * taps
        for _ in range(n):
            out.append((mode, h, w, cin, cout, fl))
    # encoder + bottleneck blocks: conv1, conv2, shortcut fprop; dgrad of conv2, conv1 + shortcut (not for block 0 input)
    cin = 8   # packed input channels (1 real): algorithmic flops count the REAL channel
    for l in range(4):
        real_in = 1 if l == 0 else C[l - 1]
        k_in = 8 if l == 0 else C[l - 1]
        h, w = Hs[l]
        out.append((0, h, w, k_in, C[l], 2.0 * B * h * w * real_in * C[l] * 9))
        add(0, Hs[l], C[l], C[l], 9)
        out.append((1, h, w, k_in, C[l], 2.0 * B * h * w * real_in * C[l]))
        add(0, Hs[l], C[l], C[l], 9)                     # dgrad conv2
        if l > 0:
            add(0, Hs[l], C[l], C[l - 1], 9)             # dgrad conv1
            add(1, Hs[l], C[l], C[l - 1], 1)             # dgrad shortcut
    for l in (2, 1, 0):                                   # decoders
        add(2, Hs[l + 1], C[l + 1], C[l], 4)             # convT fprop
        add(0, Hs[l], 2 * C[l], C[l], 9)
        add(0, Hs[l], C[l], C[l], 9)
        add(1, Hs[l], 2 * C[l], C[l], 1)
        add(0, Hs[l], C[l], C[l], 9)                     # dgrad conv2
        add(0, Hs[l], C[l], 2 * C[l], 9)                 # dgrad conv1
        add(1, Hs[l], C[l], 2 * C[l], 1)                 # dgrad shortcut
        add(3, Hs[l + 1], C[l], C[l + 1], 4)             # convT dgrad
    h, w = Hs[0]
    out.append((1, h, w, 32, 8, 2.0 * B * h * w * 32 * 8))   # head fprop
    out.append((1, h, w, 8, 32, 2.0 * B * h * w * 32 * 8))   # head dgrad
    return out


def measure_conv_roofline(B, sx_dtype, reps=5):
    """Replay every conv launch of a step through sx_conv_forward between HIP events (torch events on the
    stream the kernels are launched on).  Returns (total_flops, total_seconds, n_launches, total_bytes) where
    total_bytes is the algorithmic HBM traffic (input image read once + output written once, packed weights)."""
    from unet_amd import _lib as L
    lib = L.lib()
    tdt = torch.bfloat16 if sx_dtype == L.SX_BF16 else torch.float32
    g = torch.Generator(device="cuda").manual_seed(0)
    total_fl, total_s, n, total_by = 0.0, 0.0, 0, 0.0
    st = torch.cuda.current_stream()
    for (mode, h, w, cin, cout, fl) in conv_launches(B):
        hx, wx, hy, wy = h, w, h, w
        if mode == 2:
            hy, wy = 2 * h, 2 * w
        if mode == 3:
            hx, wx = 2 * h, 2 * w
        x = (torch.rand(B, hx, wx, cin, device="cuda", generator=g) - 0.5).to(tdt)
        y = torch.empty(B, hy, wy, cout, device="cuda", dtype=tdt)
        rows, k = cout, cin
        nel = lib.sx_packed_weight_elems(mode, sx_dtype, rows, k)
        wp = ((torch.rand(nel, device="cuda", generator=g) - 0.5) * 0.1).to(tdt)
        stats = torch.zeros(B, 8, 2, device="cuda") if mode == 0 else None
        def launch():
            L.check(lib.sx_conv_forward(sx_dtype, mode, B, h, w, cin, cout, L.ptr(x), cin, hx, wx, L.ptr(wp), None, L.ptr(y),
                                        cout, hy, wy, 0, L.ptr(stats), 8 if stats is not None else 0, L.stream_ptr()))
        launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            launch()
        e1.record(st)
        e1.synchronize()
        total_s += e0.elapsed_time(e1) * 1e-3 / reps
        total_fl += fl
        total_by += (x.numel() + y.numel() + wp.numel()) * x.element_size()
        n += 1
    return total_fl, total_s, n, total_by


def cpu_baseline(batch, steps):
    """Oracle (CPU port of the reference path) full training step, fp32, on the host cores."""
    from oracle import unet_ref, loss_ref, optim_ref
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
    cfg = UNetCfg(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3)
    sd = {k: v.requires_grad_(True) for k, v in init_state_dict(cfg, seed=0).items()}
    bt = make_batch(batch, H, W, seed=1234)
    w = torch.from_numpy(loss_ref.edge_weights(bt["mask"][0, 0].numpy()))
    cw = torch.tensor(CHANNEL_WEIGHTS_8)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}

    def step(i):
        for p in sd.values():
            p.grad = None
        pred = unet_ref.unet_forward(sd, cfg, bt["x"], bt["params"])
        loss = loss_ref.masked_weighted_loss(pred, bt["y"], bt["mask"], w=w, channel_weights=cw, **LOSS)
        loss.backward()
        names = list(sd)
        gs, _ = optim_ref.clip_grad_norm([sd[k].grad for k in names], 1.0)
        with torch.no_grad():
            for k, g in zip(names, gs):
                pn, m[k], v2[k] = optim_ref.adamw_step(sd[k], g, m[k], v2[k], i + 1, 3e-4, weight_decay=1e-4)
                sd[k].copy_(pn)
    step(0)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 1)
    dt = time.perf_counter() - t0
    return batch * steps / dt, dt


def main():
    a = parse()
    from unet_amd import dist as sxdist
    from unet_amd import _lib as L
    from unet_amd.train import TrainEngine
    from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
    from unet_amd.losses import edge_weights
    rank, world, local = sxdist.init_from_env()
    if world != a.gpus:
        if a.gpus > 1 and world == 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L.lib()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = make_model(0).to(dev).train()
    bt = make_batch(a.batch, H, W, seed=1234 + rank)          # every rank its own shard (weak scaling)
    x, y, m, p = (bt[k].to(dev) for k in ("x", "y", "mask", "params"))
    w_map = torch.from_numpy(edge_weights(bt["mask"][0, 0].numpy(), 3.0))
    eng = TrainEngine(model, dtype=dtype, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, channel_weights=CHANNEL_WEIGHTS_8,
                      w_map=w_map, loss_opts=LOSS)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.train_step(x, y, m, p, lam_grad=LOSS["lam_grad"])
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.train_step(x, y, m, p, lam_grad=LOSS["lam_grad"])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    loss_last = float(eng.log[max(eng.nlog - 1, 0), 0])
    value = world * a.batch * a.steps / dt
    line = {
        "metric": "train samples/s, cond. U-Net 128x64 grid (full training step)", "value": round(value, 2),
        "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: cond. U-Net full training step, batch 64/GPU, 128x64 grid "
                               "(base 32, depth 3, P 5, out 8), huber+edge+Sobel loss, clip 1.0, AdamW",
                   "global_batch": world * a.batch, "grid": [H, W], "parallelism": f"dp{world}"},
        "final_loss": loss_last,
    }
    if rank == 0:
        sx_dtype = L.SX_BF16 if a.dtype == "bf16" else L.SX_F32
        if not a.no_roofline:
            fl, secs, n, nbytes = measure_conv_roofline(a.batch, sx_dtype)
            peak = 2500.0 if a.dtype == "bf16" else 157.3
            ach = fl / secs / 1e12
            line["roofline"] = {"kernel": "conv_ws_kernel + conv_igemm_kernel (all fprop/dgrad launches of one step)",
                                "bound": "mfma",
                                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                "traffic": None, "launches_per_step": n, "flops_per_step": fl,
                                "seconds_per_step": round(secs, 6),
                                "hbm_algorithmic_GBps": round(nbytes / secs / 1e9, 1), "hbm_frac": round(nbytes / secs / 8e12, 4)}
            step_flops = 7.272e9 * a.batch
            line["step_model_tflops"] = round(step_flops * a.steps / dt / 1e12, 2)
        if not a.no_cpu_baseline:
            torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "16")))
            v, secs = cpu_baseline(a.cpu_batch, a.cpu_steps)
            line["cpu_baseline"] = {"value": round(v, 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                                    "kind": "port", "sample": f"{a.cpu_steps} full training steps at batch {a.cpu_batch} "
                                    f"(fp32, same model/grid/loss/optimizer), {secs:.1f} s"}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
