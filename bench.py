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
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-tier leg (what amp=False runs), reported beside the headline")
    ap.add_argument("--no-infer", action="store_true", help="skip the forward-only leg (profiling runs of the training step)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config5 / config4 / epoch legs")
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


ARCH_A = dict(C=[32, 64, 128, 256], Hs=[(128, 64), (64, 32), (32, 16), (16, 8)], cin0=8, cin0_real=1, out_pad=8)
ARCH_B = dict(C=[64, 128, 256, 512], Hs=[(256, 128), (128, 64), (64, 32), (32, 16)], cin0=16, cin0_real=10, out_pad=8)   # BASELINE configs[4]


def conv_launches(B, arch=ARCH_A):
    """Every launch of the dominant kernel (conv_ws_kernel, the weights-stationary implicit-GEMM convolution) in one
    training step of cfg A (or, with ARCH_B, of the plasma-to-sources network of BASELINE configs[4], where the 512-channel
    layers run as two K halves and one fused form falls back to two launches: the closures below call the same C-ABI entry
    points as the step, whatever they launch): (kind, H, W, Cin, Cout, algorithmic flops).
    kind: "dual" = conv1 3x3 + 1x1 shortcut forward (one launch), "c3"/"c1" = plain 3x3 / 1x1 (forward or data gradient),
    "center" = joint data gradient of conv1 + shortcut, "ct" = ConvTranspose2d(2,2) forward (four 1x1 GEMMs + pixel
    shuffle, H x W = its INPUT grid), "ctb" = its data gradient (a 1x1 over 4 taps x Cin gathered from the fine grid).
    Flops count the REAL channels (the 1-channel input is padded to 8).  The weight gradients run in wgrad_ws_kernel."""
    C, Hs = arch["C"], arch["Hs"]
    out = []

    def block(hw, cin, cin_real, c, need_dx):
        h, w = hw
        px = 2.0 * B * h * w
        out.append(("dual", h, w, cin, c, px * cin_real * c * 10))        # conv1 + shortcut forward
        out.append(("c3", h, w, c, c, px * c * c * 9))                     # conv2 forward
        # conv2 data gradient; where the kernel covers it, with GroupNorm-1's backward sums fused (reads Y1 as well)
        out.append(("c3" if c >= 128 else "c3bs", h, w, c, c, px * c * c * 9))
        if need_dx:
            out.append(("center", h, w, c, cin, px * cin_real * c * 10))   # conv1 + shortcut data gradient
    for l in range(4):                                                     # encoders + bottleneck
        block(Hs[l], arch["cin0"] if l == 0 else C[l - 1], arch["cin0_real"] if l == 0 else C[l - 1], C[l], l > 0)
    for l in (2, 1, 0):                                                    # decoders
        h, w = Hs[l + 1]
        out.append(("ct", h, w, C[l + 1], C[l], 2.0 * B * h * w * C[l + 1] * C[l] * 4))
        block(Hs[l], 2 * C[l], 2 * C[l], C[l], True)
        out.append(("ctb", h, w, C[l], C[l + 1], 2.0 * B * h * w * C[l + 1] * C[l] * 4))   # its data gradient (H x W = its OUTPUT grid)
    h, w = Hs[0]
    out.append(("c1", h, w, arch["out_pad"], C[0], 2.0 * B * h * w * C[0] * arch["out_pad"]))   # head data gradient (the head's forward
    return out                                                                                  # has a bias: conv_igemm_kernel)


def conv_launch_maker(B, sx_dtype):
    """make(kind, h, w, cin, cout) -> (launch closure, tensors whose bytes are the algorithmic traffic) for the launches
    conv_launches() lists; shared with tools/conv_layers.py (the per-layer table under profiles/)."""
    from unet_amd import _lib as L
    lib = L.lib()
    tdt = torch.bfloat16 if sx_dtype == L.SX_BF16 else torch.float32
    g = torch.Generator(device="cuda").manual_seed(0)

    def rnd(*shape):
        return (torch.rand(*shape, device="cuda", generator=g) - 0.5).to(tdt)

    def packed(mode, rows, k):
        return ((torch.rand(lib.sx_packed_weight_elems(mode, sx_dtype, rows, k), device="cuda", generator=g) - 0.5) * 0.1).to(tdt)

    def make(kind, h, w, cin, cout):
        """(launch closure, tensors whose bytes are the algorithmic traffic) of one conv_ws launch"""
        if kind == "dual":
            x, y, s2 = rnd(B, h, w, cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
            w3, w1 = packed(0, cout, cin), packed(1, cout, cin)
            stats = torch.zeros(B, 8, 2, device="cuda")
            def launch():
                L.check(lib.sx_conv3_shortcut_forward(sx_dtype, B, h, w, cin, cout, L.ptr(x), cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout,
                                                      L.ptr(stats), 8, L.ptr(s2), cout, L.stream_ptr()))
            return launch, (x, y, s2, w3, w1)
        if kind == "center":
            x, y = rnd(B, h, w, 2 * cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
            w3, w1 = packed(0, cout, cin), packed(1, cout, cin)
            def launch():
                L.check(lib.sx_conv3_shortcut_dgrad(sx_dtype, B, h, w, cin, cout, L.ptr(x), 2 * cin, L.ptr(w3), L.ptr(w1), L.ptr(y), cout,
                                                    None, 0, L.stream_ptr()))
            return launch, (x, y, w3, w1)
        if kind == "ct":
            x, y = rnd(B, h, w, cin), torch.empty(B, 2 * h, 2 * w, 2 * cout, device="cuda", dtype=tdt)   # lower half of a cat buffer
            wp, bias = packed(2, cout, cin), torch.zeros(cout, device="cuda")
            def launch():
                L.check(lib.sx_conv_forward(sx_dtype, 2, B, h, w, cin, cout, L.ptr(x), cin, h, w, L.ptr(wp), L.ptr(bias), L.ptr(y),
                                            2 * cout, 2 * h, 2 * w, 0, None, 0, L.stream_ptr()))
            return launch, (x, y[..., :cout], wp)
        if kind == "ctb":
            x, y = rnd(B, 2 * h, 2 * w, 2 * cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)   # reads the lower half of d(cat)
            wp = packed(3, cout, cin)
            def launch():
                L.check(lib.sx_conv_forward(sx_dtype, 3, B, h, w, cin, cout, L.ptr(x), 2 * cin, 2 * h, 2 * w, L.ptr(wp), None, L.ptr(y),
                                            cout, h, w, 0, None, 0, L.stream_ptr()))
            return launch, (x[..., :cin], y, wp)
        if kind == "c3bs":
            x, y, y1 = rnd(B, h, w, cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt), rnd(B, h, w, cout)
            wp = packed(0, cout, cin)
            stats = torch.stack([torch.zeros(B, 8, device="cuda"), torch.full((B, 8), float(h * w * cout // 8), device="cuda")], -1).contiguous()
            ga, be, sums = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(B, cout, 4, device="cuda")
            def launch():
                L.check(lib.sx_conv_forward_gn_bwd_sums(sx_dtype, 0, B, h, w, cin, cout, L.ptr(x), cin, L.ptr(wp), L.ptr(y), cout, 0,
                                                        L.ptr(y1), cout, None, 0, L.ptr(stats), 8, 1e-5, L.ptr(ga), L.ptr(be), None, None,
                                                        L.ptr(sums), L.stream_ptr()))
            return launch, (x, y, y1, wp)
        mode = 0 if kind == "c3" else 1
        x, y = rnd(B, h, w, cin), torch.empty(B, h, w, cout, device="cuda", dtype=tdt)
        wp = packed(mode, cout, cin)
        stats = torch.zeros(B, 8, 2, device="cuda") if mode == 0 else None
        def launch():
            L.check(lib.sx_conv_forward(sx_dtype, mode, B, h, w, cin, cout, L.ptr(x), cin, h, w, L.ptr(wp), None, L.ptr(y),
                                        cout, h, w, 0, L.ptr(stats), 8 if stats is not None else 0, L.stream_ptr()))
        return launch, (x, y, wp)

    return make


def measure_conv_roofline(B, sx_dtype, reps=5, arch=ARCH_A):
    """Replay every conv_ws_kernel launch of a step, in sequence, through the C ABI between HIP events (torch events on
    the stream the kernels are launched on).  Returns (flops, seconds, launches, bytes): bytes = algorithmic HBM traffic (every input
    tensor read once, every output written once, packed weights once)."""
    make = conv_launch_maker(B, sx_dtype)
    total_fl, total_s, n, total_by = 0.0, 0.0, 0, 0.0
    st = torch.cuda.current_stream()
    launches = []
    for (kind, h, w, cin, cout, fl) in conv_launches(B, arch):
        launch, tensors = make(kind, h, w, cin, cout)
        launches.append(launch)
        total_fl += fl
        total_by += sum(t_.numel() * t_.element_size() for t_ in tensors)
        n += 1
    # the whole sequence back to back, as in a step: every launch finds its operands as cold as the 1.3 GB working set
    # of the sequence leaves them (timing one launch in a loop would let it re-read its own data from the 256 MB MALL)
    for f in launches:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        for f in launches:
            f()
    e1.record(st)
    e1.synchronize()
    total_s = e0.elapsed_time(e1) * 1e-3 / reps
    return total_fl, total_s, n, total_by


def measure_infer(model, x, p, dtype, steps, warmup, world, sync, dev):
    """Forward-only throughput through the public module API (eval mode, packed weights cached)."""
    model.eval()
    model.compute_dtype = dtype
    with torch.no_grad():
        for _ in range(max(warmup, 2)):
            model(x, params=p)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(x, params=p)
        sync()
        dt = time.perf_counter() - t0
    del y
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    model.train()
    return world * x.shape[0] * steps / dt


TRAFFIC_FILE = "profiles/r03_conv_ws_traffic.json"


def conv_traffic(a):
    """HBM bytes per step of the same launches from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, collected in separate
    rocprofv3 --pmc runs of this file and committed under profiles/: tools/pmc_traffic.py).  Only valid for the
    configuration it was collected on; NOT measured by this run (the JSON line says so: `traffic_source`)."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if a.batch != 64 or a.dtype != "bf16" or not os.path.exists(path):
        return None
    with open(path) as f:
        return float(json.load(f)["hbm_bytes_per_step"])


def config5_leg(dev, L, batch=32, steps=8):
    """BASELINE configs[4]: plasma-to-sources U-Net (in 10, out 4, base 64, depth 3, no FiLM) at 256x128, full training step
    (production loss, clip, AdamW) in both tiers, and the convolution family of one step replayed between HIP events.
    scripts/run_source_from_plasma_pipeline.py:199-345."""
    from unet_amd.models import UNet
    from unet_amd.train import TrainEngine
    from unet_amd.synth import make_batch
    from unet_amd.losses import edge_weights
    bt = make_batch(batch, 256, 128, in_ch=10, out_ch=4, P=0, seed=3)
    x, y, m = (bt[k].to(dev) for k in ("x", "y", "mask"))
    w_map = torch.from_numpy(edge_weights(bt["mask"][0, 0].numpy(), 3.0))
    out = {"workload": "BASELINE configs[4]: plasma-to-sources U-Net (in 10, out 4, base 64, depth 3, P 0), 256x128, full training "
                       f"step, batch {batch}", "flops_per_sample_step": 116.85e9}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        torch.manual_seed(0)
        model = UNet(in_ch=10, out_ch=4, base=64, P=0, depth=3).to(dev).train()
        eng = TrainEngine(model, dtype=dt, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, w_map=w_map, loss_opts=LOSS)
        for _ in range(3):
            eng.train_step(x, y, m, None, lam_grad=LOSS["lam_grad"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(x, y, m, None, lam_grad=LOSS["lam_grad"])
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / steps
        out[name] = {"ms_per_step": round(1e3 * d, 3), "samples_per_s": round(batch / d, 1), "steps": steps,
                     "model_TFLOPs": round(116.85e9 * batch / d / 1e12, 1), "final_loss": float(eng.log[eng.nlog - 1, 0])}
        del eng, model
    out["fp32"]["arithmetic"] = ("convolutions on three bf16 planes (six MFMAs per product block, fp32-grade products), weight gradients "
                                 "on two: networks whose widest layer reaches 512 channels leave the two-plane kernels (4.1e-4 worst "
                                 "gradient error measured there against the 1e-4 contract; unet_plan.hip)")
    out["bf16"]["mfma_frac_of_step"] = round(out["bf16"]["model_TFLOPs"] / 2500.0, 4)
    fl, secs, n, nbytes = measure_conv_roofline(batch, L.SX_BF16, reps=3, arch=ARCH_B)
    tfs = fl / secs / 1e12
    # 640 flop per byte: above the machine balance (312), so this family is priced against the dense bf16 MFMA peak
    out["roofline"] = {"kernel": "every convolution entry point of one step (conv_ws_kernel; 512-channel layers as two K halves)",
                       "bound": "mfma", "achieved": round(tfs, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tfs / 2500.0, 4),
                       "traffic": None, "calls_per_step": n, "flops_per_step": fl, "bytes_per_step": nbytes,
                       "seconds_per_step": round(secs, 6)}
    return out


def config4_leg(dev, n=1024, steps=10):
    """BASELINE configs[3] on one GPU: batched inverse, 1024 problems per chunk (the per-GPU share of 8192 over 8 GPUs), forward +
    per-sample fit + data-gradient-only backward + per-sample Adam.  scripts/eval_inverse_cycle_conditional_unet.py:355-416."""
    from unet_amd.inverse import batched_inverse
    from unet_amd.synth import make_batch
    model = make_model(0).to(dev).eval()
    model.compute_dtype = torch.bfloat16
    bt = make_batch(n, H, W, seed=7)
    mask, p_true = bt["mask"].to(dev), bt["params"].to(dev)
    with torch.no_grad():
        y = torch.cat([model(mask[i:i + 64], params=p_true[i:i + 64]).float() for i in range(0, n, 64)])
    p0 = torch.zeros_like(p_true)
    batched_inverse(model, mask, y, p0, steps=2, early_stop_patience=0, batch=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = batched_inverse(model, mask, y, p0, steps=steps, early_stop_patience=0, batch=n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = n * steps / dt
    return {"workload": f"BASELINE configs[3], one GPU's share: batched inverse, {n} problems per chunk, 128x64, bf16, Adam + cosine",
            "problem_steps_per_s": round(rate, 1), "ms_per_chunk_step": round(1e3 * dt / steps, 3), "steps": steps,
            "model_TFLOPs": round(4.846e9 * rate / 1e12, 1), "flops_per_problem_step": 4.846e9, "mean_fit": float(r["fit"].mean())}


def epoch_leg(dev, n_train=1984, batch=64):
    """train_unet (the reference's entry point, solpex/train.py:14-359) fed by the device-resident loader over a synthetic
    SOLPS-shaped npz: does the data path keep up with the step?  Per-epoch time = (5 epochs - 1 epoch) / 4, each run
    including validation on one batch and the best-checkpoint write."""
    import contextlib
    import io
    import tempfile
    import numpy as np
    from unet_amd import data as D
    from unet_amd.train import train_unet
    from unet_amd.synth import CHANNEL_WEIGHTS_8
    n = n_train + batch
    rng = np.random.default_rng(0)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    rho = np.sqrt(((ii - H / 2) / (0.47 * H)) ** 2 + ((jj - W / 2) / (0.45 * W)) ** 2)
    mask = ((rho >= 0.45) & (rho <= 1.0)).astype(np.float32)
    keys = ["Te", "Ti", "ne", "ua", "Sp", "Qe", "Qi", "Sm"]
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "synth.npz")
    np.savez(path, Y=np.abs(rng.normal(size=(n, 8, H, W))).astype(np.float32) * 10, y_keys=np.array(keys), mask=mask,
             X=rng.normal(size=(n, 5)).astype(np.float32))
    norms = {k: D.MaskedLogStandardizer(eps=1.0) for k in keys}
    with contextlib.redirect_stdout(io.StringIO()):
        tl, vl, norm, P, hw, scaler, pt, pk = D.make_loaders(path, batch_size=batch, y_keys=keys, norms_by_name=norms, device=str(dev),
                                                             resident=True, split=1.0 - batch / n)

        def run(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train_unet(tl, vl, norm, 1, dev, out_ch=8, epochs=epochs, base=32, P=P, film_hidden=128, depth=3, lam_grad=0.1, lam_w=0.5,
                       multiscale=0, amp=True, lr_init=3e-4, channel_weights=CHANNEL_WEIGHTS_8, param_scaler=scaler,
                       save_path=os.path.join(tmp, "best.pt"), lam_grad_warmup_start=0, lam_grad_warmup_end=0)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(1)
        t1 = run(1)
        t5 = run(5)
        # the inner loop of an epoch alone (loader + TrainEngine.train_step + the one host read per epoch): the data path's share
        from unet_amd.models import UNet
        from unet_amd.train import TrainEngine
        from unet_amd.losses import edge_weights
        torch.manual_seed(0)
        model = UNet(in_ch=1, out_ch=8, base=32, P=P, film_hidden=128, depth=3).to(dev).train()
        eng = TrainEngine(model, dtype=torch.bfloat16, lr=3e-4, channel_weights=CHANNEL_WEIGHTS_8,
                          w_map=torch.from_numpy(edge_weights(mask, 3.0)), loss_opts=dict(lam_w=0.5, lam_grad=0.1, grad_ds=2))
        loop = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_seen = 0
            for b in tl:
                eng.train_step(b["x"], b["y"], b["mask"], b["params"], lam_grad=0.1)
                n_seen += b["x"].shape[0]
            eng.read_epoch()
            torch.cuda.synchronize()
            loop.append(n_seen / (time.perf_counter() - t0))
    per = (t5 - t1) / 4
    return {"workload": f"train_unet(amp=True) + ResidentLoader, {n_train} training samples per epoch, batch {batch}, validation on "
                        f"{batch}, best-checkpoint writes", "samples_per_s": round(n_train / per, 1), "ms_per_epoch": round(1e3 * per, 2),
            "first_epoch_run_s": round(t1, 3),
            "loader_plus_step_samples_per_s": round(max(loop[1:]), 1),
            "note": "loader_plus_step = the epoch's training loop alone (ResidentLoader batches -> train_step -> one host read); the "
                    "rest of ms_per_epoch is what the reference also does once per epoch: validation, history, torch.save of the best "
                    "checkpoint (model + optimizer state)"}


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


def config1_gpu(dev, reps=50):
    """BASELINE configs[0] on the GPU: forward + masked MSE (base="mse", lam_grad=0, lam_w=0), batch 8, fp32 tier."""
    from unet_amd.losses import masked_weighted_loss
    from unet_amd.synth import make_batch
    model = make_model(0).to(dev).eval()
    model.compute_dtype = torch.float32
    bt = make_batch(8, H, W, seed=77)
    x, y, m, p = (bt[k].to(dev) for k in ("x", "y", "mask", "params"))
    with torch.no_grad():
        for _ in range(5):
            loss = masked_weighted_loss(model(x, params=p), y, m, base="mse", lam_grad=0.0, lam_w=0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = masked_weighted_loss(model(x, params=p), y, m, base="mse", lam_grad=0.0, lam_w=0.0)
        torch.cuda.synchronize()
    return 8 * reps / (time.perf_counter() - t0), float(loss)


def config1_cpu(reps=20):
    """The same workload through the CPU oracle (port of the reference's PyTorch path), fp32, on the host cores."""
    from oracle import unet_ref, loss_ref
    from oracle.unet_ref import UNetCfg
    from unet_amd.synth import make_batch
    cfg = UNetCfg(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3)
    sd = {k: v.detach().cpu().clone() for k, v in make_model(0).state_dict().items()}
    bt = make_batch(8, H, W, seed=77)
    with torch.no_grad():
        loss = loss_ref.masked_weighted_loss(unet_ref.unet_forward(sd, cfg, bt["x"], bt["params"]), bt["y"], bt["mask"], base="mse",
                                             lam_grad=0.0, lam_w=0.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = loss_ref.masked_weighted_loss(unet_ref.unet_forward(sd, cfg, bt["x"], bt["params"]), bt["y"], bt["mask"],
                                                 base="mse", lam_grad=0.0, lam_w=0.0)
    return 8 * reps / (time.perf_counter() - t0), float(loss)


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
    local = local % max(torch.cuda.device_count(), 1)      # rehearsals: several gloo ranks on one card (SX_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    # SX_DP_FORCE=1 on one GPU: a ONE-rank RCCL group, so that the whole data-parallel protocol of the step (asynchronous
    # denominator all-reduce, three gradient buckets, split encoder backward) runs against real RCCL kernels and its cost on
    # one card goes on record (`dp` in the JSON line; profiles/r03_dp_force.json)
    forced = world == 1 and os.environ.get("SX_DP_FORCE", "0") not in ("", "0")
    if forced and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1)
    dp_on = world > 1 or forced
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
        if dp_on:
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
    lt = eng.log[max(eng.nlog - 1, 0), 0:1].clone()
    sxdist.all_reduce_sum(lt)            # under DP a rank logs its SHARE of the batch-global loss
    loss_last = float(lt)
    value = world * a.batch * a.steps / dt
    infer = None if a.no_infer else measure_infer(model, x, p, dtype, a.steps, a.warmup, world, sync, dev)
    dp_diag = None
    if dp_on:
        # diagnostic for the scaling runs: the same step with every collective switched off (each rank on its own from here
        # on -- after the timed region, nothing reads these weights again).  The difference is the communication the
        # three-bucket overlap did NOT hide.
        eng.dp = False
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            eng.train_step(x, y, m, p, lam_grad=LOSS["lam_grad"])
        sync()
        dt0 = time.perf_counter() - t1
        tt = torch.tensor([dt0], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt0 = float(tt)
        eng.dp = True
        dp_diag = {"ms_per_step_without_collectives": round(1e3 * dt0 / a.steps, 4),
                   "exposed_communication_ms": round(1e3 * (dt - dt0) / a.steps, 4),
                   "buckets_bytes": [4 * eng.n_dec, 4 * (eng.n_bot - eng.n_dec), 4 * (eng.flat_g.numel() - eng.n_bot)],
                   "ranks": world, "backend": torch.distributed.get_backend(),
                   "what": "the step with its collectives (timed as `ms_per_step`) against the same step with every collective "
                           "switched off afterwards" + ("; ONE-rank RCCL group (SX_DP_FORCE=1)" if forced else "")}
    line = {
        "metric": "train samples/s, cond. U-Net 128x64 grid (full training step)", "value": round(value, 2),
        "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: cond. U-Net full training step, batch 64/GPU, 128x64 grid "
                               "(base 32, depth 3, P 5, out 8), huber+edge+Sobel loss, clip 1.0, AdamW",
                   "global_batch": world * a.batch, "grid": [H, W], "parallelism": f"dp{world}"},
        "final_loss": loss_last,
        # the metric's "infer" half: UNet.forward (solpex/models.py:205-208) under no_grad, same batch / dtype, all ranks
    }
    if dp_diag is not None:
        line["dp"] = dp_diag
    if infer is not None:
        line["infer"] = {"value": round(infer, 2), "unit": "samples/s", "batch_per_gpu": a.batch,
                         "what": "UNet.forward(x, params), eval mode, no_grad, inputs resident in HBM"}
    if rank == 0:
        sx_dtype = L.SX_BF16 if a.dtype == "bf16" else L.SX_F32
        if not a.no_roofline:
            fl, secs, n, nbytes = measure_conv_roofline(a.batch, sx_dtype)
            mfma_peak = 2500.0 if a.dtype == "bf16" else 157.3
            tfs = fl / secs / 1e12
            gbs = nbytes / secs / 1e9
            # aggregate arithmetic intensity fl/nbytes (~180 flop/B) is below the machine balance (2.5 PF / 8 TB/s = 312):
            # the family is bounded by HBM, so that is the roofline quoted; the MFMA view rides along
            line["roofline"] = {"kernel": "conv_ws_kernel (every launch of one step: 3x3 / 1x1 / transposed forward, 3x3 / 1x1 / transposed data gradients)",
                                "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                "frac": round(gbs / 8000.0, 4), "traffic": conv_traffic(a),
                                "traffic_source": TRAFFIC_FILE + " (PMC passes of an earlier run of this workload, not this run)",
                                "launches_per_step": n,
                                "bytes_per_step": nbytes, "flops_per_step": fl, "seconds_per_step": round(secs, 6),
                                "mfma_TFLOPs": round(tfs, 2), "mfma_frac": round(tfs / mfma_peak, 4)}
            step_flops = 7.272e9 * a.batch
            line["step_model_tflops"] = round(step_flops * a.steps / dt / 1e12, 2)
        if not a.no_fp32 and world == 1 and a.dtype == "bf16":
            # the fp32 tier (amp=False: what the reference's own scripts pass) on the same workload, beside the headline
            m32 = make_model(0).to(dev).train()
            e32 = TrainEngine(m32, dtype=torch.float32, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, channel_weights=CHANNEL_WEIGHTS_8,
                              w_map=w_map, loss_opts=LOSS)
            for _ in range(5):
                e32.train_step(x, y, m, p, lam_grad=LOSS["lam_grad"])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n32 = 20
            for _ in range(n32):
                e32.train_step(x, y, m, p, lam_grad=LOSS["lam_grad"])
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t1) / n32
            exact = L.lib().sx_set_fp32_exact(-1)
            line["fp32_tier"] = {"ms_per_step": round(1e3 * d32, 4), "samples_per_s": round(a.batch / d32, 1), "steps": n32,
                                 "arithmetic": "exact fp32 MFMA" if exact else "split-bf16: bf16 hi + lo planes, 3 bf16 MFMAs per product block, fp32 accumulation",
                                 "final_loss": float(e32.log[max(e32.nlog - 1, 0), 0])}
            del e32, m32
        if not a.no_extra and world == 1 and a.dtype == "bf16":
            # the other BASELINE configurations and the end-to-end epoch, a few steps each, beside the headline
            del eng
            torch.cuda.empty_cache()
            line["config5"] = config5_leg(dev, L)
            line["config4"] = config4_leg(dev)
            line["epoch"] = epoch_leg(dev)
        if not a.no_cpu_baseline and world == 1:      # the CPU leg is a single-GPU-run item (rank 0, N = 1 only)
            # threads: the job's CPU share (OMP_NUM_THREADS is set to it on the GPU box, where os.cpu_count() reports the whole
            # machine); "cores" = the threads the oracle actually ran on, host_cpu_count = what the host reports
            threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1))
            torch.set_num_threads(threads)
            v, secs = cpu_baseline(a.cpu_batch, a.cpu_steps)
            line["cpu_baseline"] = {"value": round(v, 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                                    "host_cpu_count": os.cpu_count(), "kind": "port",
                                    "sample": f"{a.cpu_steps} full training steps at batch {a.cpu_batch} "
                                    f"(fp32, same model/grid/loss/optimizer), {secs:.1f} s"}
            # BASELINE configs[0] (the reference's own CPU-runnable case) beside it: forward + masked MSE, batch 8, fp32
            g_v, g_loss = config1_gpu(dev)
            c_v, c_loss = config1_cpu()
            line["config1"] = {"workload": "BASELINE configs[0]: forward + masked MSE, batch 8, 128x64, fp32",
                               "gpu_samples_per_s": round(g_v, 1), "cpu_samples_per_s": round(c_v, 1), "cpu_threads": threads,
                               "loss_gpu": g_loss, "loss_cpu": c_loss}
        print(json.dumps(line), flush=True)
    if dp_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
