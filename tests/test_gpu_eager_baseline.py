"""Same-GPU comparison: the oracle (a plain-PyTorch restatement of the reference path, i.e. what the reference itself
executes: ATen / MIOpen / rocBLAS eager kernels) against the HIP path, full production training step at BASELINE
configs[1] (batch 64, 128x64).  The oracle is the checker's yardstick here, never the product path.  Writes the two
step times to gpurun_out/eager_baseline.json (read into DESIGN.md).  Both sides start from the SAME weights and see the
same batch, and what the test asserts is parity: the losses of the first steps agree (fp32 <= 1e-3 relative over four
optimizer steps, bf16 <= 2e-2 against autocast).  The timings are recorded, not asserted (a parity suite must not flake on
a busy box)."""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSS = dict(lam_w=0.5, lam_grad=0.1, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1",
            multiscale=0, ms_weight=0.5, grad_ds=2)


def _eager_step_ms(dtype, B=64, H=128, W=64, steps=6):
    from oracle import unet_ref, loss_ref, optim_ref
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
    dev = torch.device("cuda")
    cfg = UNetCfg(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in init_state_dict(cfg, seed=0).items()}
    bt = {k: v.to(dev) for k, v in make_batch(B, H, W, seed=1234).items()}
    w = torch.from_numpy(loss_ref.edge_weights(bt["mask"][0, 0].cpu().numpy())).to(dev)
    cw = torch.tensor(CHANNEL_WEIGHTS_8, device=dev)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    names = list(sd)

    losses = []

    def step(i):
        for p in sd.values():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            pred = unet_ref.unet_forward(sd, cfg, bt["x"], bt["params"])
        loss = loss_ref.masked_weighted_loss(pred.float(), bt["y"], bt["mask"], w=w, channel_weights=cw, **LOSS)
        losses.append(loss.detach())
        loss.backward()
        gs, _ = optim_ref.clip_grad_norm([sd[k].grad for k in names], 1.0)
        with torch.no_grad():
            for k, g in zip(names, gs):
                pn, m[k], v2[k] = optim_ref.adamw_step(sd[k], g, m[k], v2[k], i + 1, 3e-4, weight_decay=1e-4)
                sd[k].copy_(pn)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(3 + i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, [float(v) for v in losses]


def _hip_step_ms(dtype, B=64, H=128, W=64, steps=20):
    from unet_amd.models import UNet
    from unet_amd.train import TrainEngine
    from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
    from unet_amd.losses import edge_weights
    from oracle.unet_ref import UNetCfg, init_state_dict
    model = UNet(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3)
    model.load_state_dict(init_state_dict(UNetCfg(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3), seed=0))
    model = model.cuda().train()
    bt = make_batch(B, H, W, seed=1234)
    x, y, mk, p = (bt[k].cuda() for k in ("x", "y", "mask", "params"))
    w_map = torch.from_numpy(edge_weights(bt["mask"][0, 0].numpy(), 3.0))
    eng = TrainEngine(model, dtype=dtype, lr=3e-4, weight_decay=1e-4, clip_grad=1.0, channel_weights=CHANNEL_WEIGHTS_8,
                      w_map=w_map, loss_opts=LOSS)
    losses = []
    for _ in range(5):
        losses.append(eng.train_step(x, y, mk, p, lam_grad=LOSS["lam_grad"])[0:1].clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(x, y, mk, p, lam_grad=LOSS["lam_grad"])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, [float(v) for v in losses]


def test_hip_step_matches_eager_torch_losses_on_the_same_gpu():
    res, losses = {}, {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        e_ms, e_loss = _eager_step_ms(dt)
        h_ms, h_loss = _hip_step_ms(dt)
        res[name] = {"eager_torch_ms": round(e_ms, 3), "hip_ms": round(h_ms, 3)}
        losses[name] = (e_loss, h_loss)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "eager_baseline.json"), "w") as f:
        json.dump({"workload": "full training step, batch 64, 128x64, cfg A, production loss, clip + AdamW", **res}, f)
    print(res, losses)
    for name, tol in (("fp32", 1e-3), ("bf16", 2e-2)):
        e_loss, h_loss = losses[name]
        for i in range(4):                      # steps 0..3: the same trajectory, not just the same first loss
            assert abs(h_loss[i] - e_loss[i]) < tol * abs(e_loss[i]), (name, i, h_loss[:4], e_loss[:4])


def test_latent_cycle_trainer_step_vs_eager_torch_on_the_same_gpu():
    """train_cycle_consistent's step (two 256-wide MLPs, batch 64, z_dim 256, P 5): HIP kernels vs the oracle in eager
    PyTorch on the same GPU.  Both are launch-latency bound; recorded in gpurun_out/eager_baseline_latent.json."""
    from oracle import latent_ref
    from unet_amd import latent as HL
    rng = __import__("numpy").random.default_rng(0)
    N, zd, Pd, bs = 64 * 40, 256, 5, 64
    Z = rng.normal(size=(N, zd)).astype("float32")
    P = rng.normal(size=(N, Pd)).astype("float32")
    torch.manual_seed(0)
    p2z, z2p = HL.ParamToZ(Pd, zd), HL.ZToParam(zd, Pd)
    sd_a = {k: v.detach().clone().cuda().requires_grad_(True) for k, v in p2z.state_dict().items()}
    sd_b = {k: v.detach().clone().cuda().requires_grad_(True) for k, v in z2p.state_dict().items()}
    opt = latent_ref._Opt([sd_a, sd_b], 1e-3, 1e-4)
    Zd, Pdv = torch.from_numpy(Z).cuda(), torch.from_numpy(P).cuda()

    e_rows = []

    def eager_epoch(record=False):
        for i in range(0, N, bs):
            z, p = Zd[i:i + bs], Pdv[i:i + bs]
            t0_ = latent_ref.smooth_l1(latent_ref.mlp_forward(sd_b, z), p)
            t1_ = latent_ref.smooth_l1(latent_ref.mlp_forward(sd_a, p), z)
            t2_ = latent_ref.smooth_l1(latent_ref.mlp_forward(sd_b, latent_ref.mlp_forward(sd_a, p)), p)
            t3_ = latent_ref.smooth_l1(latent_ref.mlp_forward(sd_a, latent_ref.mlp_forward(sd_b, z)), z)
            if record:
                e_rows.append(torch.stack([t0_, t1_, t2_, t3_]).detach())
            (t0_ + t1_ + 0.1 * (t2_ + t3_)).backward()
            opt.step()
    eager_epoch(record=True)                # the untimed first epoch: its loss trajectory is what the HIP side must reproduce
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager_epoch()
    torch.cuda.synchronize()
    eager_us = 1e6 * (time.perf_counter() - t0) / (N // bs)

    p2z, z2p = p2z.cuda(), z2p.cuda()
    eng = HL._Trainer([p2z, z2p], 1e-3, 1e-4, torch.device("cuda"))

    def hip_epoch():
        for i in range(0, N, bs):
            HL._cycle_losses(eng, p2z, z2p, Zd[i:i + bs], Pdv[i:i + bs], 0.1, True)
            eng.step()
        return eng.read()[0]
    h_rows = hip_epoch()
    # same weights, same batches, same optimizer: the four loss terms of every step of the first epoch (40 AdamW steps)
    e_np = torch.stack(e_rows).double().cpu().numpy()
    assert h_rows.shape == e_np.shape
    assert abs(h_rows[0] - e_np[0]).max() < 1e-5 * abs(e_np[0]).max()          # step 0: same weights, forward only
    assert abs(h_rows - e_np).max() < 2e-3 * abs(e_np).max(), (h_rows[-1], e_np[-1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip_epoch()
    torch.cuda.synchronize()
    hip_us = 1e6 * (time.perf_counter() - t0) / (N // bs)
    res = {"workload": "train_cycle_consistent step, batch 64, z_dim 256, P 5, hidden (256,256)",
           "eager_torch_us_per_step": round(eager_us, 1), "hip_us_per_step": round(hip_us, 1)}
    with open(os.path.join(ROOT, "gpurun_out", "eager_baseline_latent.json"), "w") as f:
        json.dump(res, f)
    print(res)
