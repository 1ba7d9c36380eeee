#!/usr/bin/env python3
"""Generate golden vectors from the LIVE reference (build container only).

Imports `/root/reference/solpex` (with an empty `h5py` stub: the reference only
needs h5py for geometry file I/O, off the hot path), pushes seeded inputs and
seeded weights through it and stores inputs + expected outputs as small .npz
fixtures under tests/golden/.  Nothing of the reference's source travels: only
numbers.  Run:  python tools/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types
import itertools
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")
import solpex as ref                      # noqa: E402  (the reference)
from solpex import data as ref_data       # noqa: E402
from solpex import losses as ref_losses   # noqa: E402

from oracle.unet_ref import UNetCfg, init_state_dict   # noqa: E402
from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)
torch.manual_seed(0)


def ref_model(cfg: UNetCfg, sd):
    m = ref.UNet(in_ch=cfg.in_ch, out_ch=cfg.out_ch, base=cfg.base, groups_gn=cfg.groups_gn,
                 dropout=cfg.dropout, P=cfg.P, film_hidden=cfg.film_hidden, z_dim=cfg.z_dim, depth=cfg.depth)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m


def sd_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def cfg_dict(cfg):
    return {f"cfg_{k}": np.asarray(getattr(cfg, k)) for k in
            ("in_ch", "out_ch", "base", "groups_gn", "P", "film_hidden", "z_dim", "depth")}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, torch_version=np.asarray(torch.__version__), **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def unet_case(name, cfg, B, H, W, seed, loss_kw, store_weights=True, jitter=True, empty_last=True,
              summary_only=False, use_cw=True):
    sd = init_state_dict(cfg, seed=seed)
    model = ref_model(cfg, sd)
    model.train()
    bt = make_batch(B, H, W, in_ch=cfg.in_ch, out_ch=cfg.out_ch, P=cfg.P, seed=1234 + seed,
                    jitter_masks=jitter, empty_last=empty_last)
    x, y, mask = bt["x"], bt["y"], bt["mask"]
    params = bt["params"].clone().requires_grad_(True) if cfg.P > 0 else None
    w_map = torch.from_numpy(ref_losses.edge_weights(mask[0, 0].numpy(), 3.0))[None, None]
    cw = None
    if use_cw:
        cw = torch.tensor((CHANNEL_WEIGHTS_8 * 2)[:cfg.out_ch], dtype=torch.float32)
    pred, b = model(x, params=params, return_bottleneck=True)
    loss = ref_losses.masked_weighted_loss(pred, y, mask, w=w_map, channel_weights=cw, **loss_kw)
    loss.backward()
    arrs = dict(cfg_dict(cfg))
    arrs.update(B=B, H=H, W=W, seed=seed, x=x, y=y, mask=mask, w_map=w_map,
                sd_checksum=sd_checksum(sd), loss=loss.detach())
    if cw is not None:
        arrs["cw"] = cw
    if params is not None:
        arrs["params"] = params.detach()
        arrs["dparams"] = params.grad
    for k, v in loss_kw.items():
        arrs[f"losskw_{k}"] = np.asarray(v)
    named = dict(model.named_parameters())
    if summary_only:
        arrs["pred"] = pred.detach()
        arrs["b_norm"] = b.detach().norm()
        arrs["grad_names"] = np.asarray(list(named.keys()))
        arrs["grad_norms"] = np.asarray([float(p.grad.norm()) for p in named.values()])
        arrs["grad_sums"] = np.asarray([float(p.grad.double().sum()) for p in named.values()])
        # a few full small gradients
        for k in ("out.weight", "out.bias", "encoders.0.net.0.weight", "film_bot.net.2.bias", "up_convs.2.bias"):
            if k in named:
                arrs["grad::" + k] = named[k].grad
    else:
        arrs["pred"] = pred.detach()
        arrs["bott"] = b.detach()
        for k, p in named.items():
            if store_weights:
                arrs["w::" + k] = p.detach()
            arrs["grad::" + k] = p.grad
    save(name, **arrs)


def latent_case():
    cfg = UNetCfg(in_ch=1, out_ch=2, base=8, P=5, film_hidden=16, z_dim=4, depth=3)
    sd = init_state_dict(cfg, seed=3)
    model = ref_model(cfg, sd).eval()
    bt = make_batch(2, 16, 16, in_ch=1, out_ch=2, P=5, seed=77, jitter_masks=True)
    x, params = bt["x"], bt["params"]
    with torch.no_grad():
        y, b = model(x, params=params, return_bottleneck=True)
        b2, skips = model.encode(x, params=params)
        z = model.project_z(b)
        z0 = ref.bottleneck_to_z(b)
        bb = model.unproject_z(z, b.shape)
        y2 = model.decode_from_bottleneck(x, bb, params=params)
        y_nop = model(x)            # params=None path: FiLM skipped entirely
    arrs = dict(cfg_dict(cfg))
    arrs.update(x=x, params=params, y=y, bott=b, skip0=skips[0], skip2=skips[2], z=z, z0=z0,
                bb=bb.contiguous(), y2=y2, y_nop=y_nop, seed=3, sd_checksum=sd_checksum(sd))
    save("unet_latent.npz", **arrs)


def latent_mlp_case():
    """solpex/latent.py: the two MLPs, one forward + gradient check each, and a few epochs of every trainer."""
    from solpex import latent as ref_latent
    rng = np.random.default_rng(5)
    Pd, zd, hid = 5, 16, (32, 32)
    Ntr, Nva = 72, 24
    Wm = rng.normal(size=(Pd, zd)).astype(np.float32) * 0.6
    P_tr = rng.normal(size=(Ntr, Pd)).astype(np.float32); P_va = rng.normal(size=(Nva, Pd)).astype(np.float32)
    Z_tr = np.tanh(P_tr @ Wm).astype(np.float32) + 0.05 * rng.normal(size=(Ntr, zd)).astype(np.float32)
    Z_va = np.tanh(P_va @ Wm).astype(np.float32) + 0.05 * rng.normal(size=(Nva, zd)).astype(np.float32)
    arrs = dict(P_tr=P_tr, P_va=P_va, Z_tr=Z_tr, Z_va=Z_va, hidden=np.asarray(hid), epochs=3, batch_size=16, lam_cycle=0.1)

    def put(prefix, sd):
        for k, v in sd.items():
            arrs[f"{prefix}::{k}"] = v.detach().clone()

    def clone(sd):
        return {k: v.detach().clone() for k, v in sd.items()}
    torch.manual_seed(21)
    p2z = ref_latent.ParamToZ(Pd, zd, hidden=hid)
    z2p = ref_latent.ZToParam(zd, Pd, hidden=hid)
    z2p_ln = ref_latent.ZToParam(zd, Pd, hidden=hid, use_layernorm=True)
    with torch.no_grad():                      # LayerNorm affine away from the identity
        for m in z2p_ln.net:
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.2 * torch.randn_like(m.weight)); m.bias.add_(0.1 * torch.randn_like(m.bias))
    init = dict(p2z=clone(p2z.state_dict()), z2p=clone(z2p.state_dict()), z2p_ln=clone(z2p_ln.state_dict()))
    for k, sd in init.items():
        put(f"init_{k}", sd)
    # forward + gradients of a SmoothL1 loss (both MLP variants)
    for name, m, a, b in (("p2z", p2z, P_va, Z_va), ("z2p_ln", z2p_ln, Z_va, P_va)):
        x = torch.from_numpy(a).clone().requires_grad_(True)
        out = m(x)
        loss = torch.nn.SmoothL1Loss(beta=0.05)(out, torch.from_numpy(b))
        loss.backward()
        arrs[f"fwd_{name}::out"] = out
        arrs[f"fwd_{name}::loss"] = loss
        arrs[f"fwd_{name}::dx"] = x.grad
        for k, p in m.named_parameters():
            arrs[f"fwd_{name}::grad::{k}"] = p.grad.clone()
            p.grad = None
    kw = dict(device="cpu", lr=2e-3, epochs=3, batch_size=16, wd=1e-4)
    torch.manual_seed(31)
    m, h = ref_latent.train_param2z(P_train=P_tr, Z_train=Z_tr, P_val=P_va, Z_val=Z_va, model=p2z, **kw)
    put("final_param2z", m.state_dict()); arrs["hist_param2z_tr"] = h["tr"]; arrs["hist_param2z_va"] = h["va"]
    torch.manual_seed(32)
    m, h = ref_latent.train_z2param(Z_train=Z_tr, P_train=P_tr, Z_val=Z_va, P_val=P_va, model=z2p_ln, **kw)
    put("final_z2param_ln", m.state_dict()); arrs["hist_z2param_ln_tr"] = h["tr"]; arrs["hist_z2param_ln_va"] = h["va"]
    p2z2 = ref_latent.ParamToZ(Pd, zd, hidden=hid); p2z2.load_state_dict(init["p2z"])
    z2p2 = ref_latent.ZToParam(zd, Pd, hidden=hid); z2p2.load_state_dict(init["z2p"])
    torch.manual_seed(33)
    a, b, h = ref_latent.train_cycle_consistent(Z_train=Z_tr, P_train=P_tr, Z_val=Z_va, P_val=P_va, p2z_model=p2z2,
                                                z2p_model=z2p2, lam_cycle=0.1, **kw)
    put("final_cycle_p2z", a.state_dict()); put("final_cycle_z2p", b.state_dict())
    for k, v in h.items():
        arrs[f"hist_cycle_{k}"] = v
    save("latent_mlp.npz", **arrs)


def loss_cases():
    g = torch.Generator().manual_seed(11)
    B, C, H, W = 2, 3, 12, 10
    pred = torch.randn(B, C, H, W, generator=g) * 0.3
    target = pred + torch.randn(B, C, H, W, generator=g) * 0.06     # residuals straddle beta = 0.05
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
    mask[1, :, :, :3] = 0.0
    w = torch.rand(1, 1, H, W, generator=g)
    cw = torch.tensor([1.0, 1.5, 0.7])
    combos = []
    for base, gmode, gbase, gds, ms in itertools.product(
            ("huber", "l1", "mse"), ("vector", "magnitude"), ("l1", "huber", "mse"), (1, 2), (0, 1, 2)):
        combos.append((base, gmode, gbase, gds, ms, True, True, 0.5, 0.1))
    combos.append(("mse", "vector", "l1", 1, 0, False, False, 0.0, 0.0))    # plain masked MSE
    combos.append(("huber", "vector", "l1", 2, 0, True, False, 0.5, 0.1))
    combos.append(("huber", "vector", "l1", 4, 2, False, True, 1.0, 0.2))
    combos.append(("huber", "vector", "l1", 2, 1, True, True, 0.5, 0.0))    # train_unet before the warm-up
    losses, grads, desc = [], [], []
    for (base, gmode, gbase, gds, ms, use_w, use_cw, lam_w, lam_grad) in combos:
        p = pred.clone().requires_grad_(True)
        L = ref_losses.masked_weighted_loss(p, target, mask, w=w if use_w else None, lam_grad=lam_grad, lam_w=lam_w,
                                            base=base, huber_beta=0.05, grad_mode=gmode, grad_base=gbase,
                                            multiscale=ms, ms_weight=0.5, grad_ds=gds,
                                            channel_weights=cw if use_cw else None)
        L.backward()
        losses.append(float(L))
        grads.append(p.grad.numpy().copy())
        desc.append(f"{base}|{gmode}|{gbase}|{gds}|{ms}|{int(use_w)}|{int(use_cw)}|{lam_w}|{lam_grad}")
    # all-zero mask: denominators clamp to 1e-8
    z = ref_losses.masked_weighted_loss(pred, target, torch.zeros_like(mask), w=w, lam_grad=0.1, lam_w=0.5,
                                        grad_ds=2, channel_weights=cw)
    save("loss_cases.npz", pred=pred, target=target, mask=mask, w=w, cw=cw, desc=np.asarray(desc),
         losses=np.asarray(losses, dtype=np.float64), grads=np.stack(grads), loss_zero_mask=z,
         mae_norm=ref_losses.mae_norm(pred, target, mask))


def norm_cases():
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 6, 5
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.25).float()
    keys = ["Te", "ne", "Sp", "lin"]
    n_log = ref_data.MaskedLogStandardizer(eps=1.0)
    n_log.mu, n_log.sigma = torch.tensor(2.5), torch.tensor(1.3)
    n_log2 = ref_data.MaskedLogStandardizer(eps=1e-3)
    n_log2.mu, n_log2.sigma = torch.tensor(-1.0), torch.tensor(0.7)
    n_sym = ref_data.MaskedSymLogStandardizer(scale=10.0)
    n_sym.mu, n_sym.sigma = torch.tensor(0.2), torch.tensor(1.7)
    n_lin = ref_data.MaskedLinearStandardizer()
    n_lin.mu, n_lin.sigma = torch.tensor(3.0), torch.tensor(2.0)
    norm = ref_data.MultiChannelNormalizer(keys, dict(Te=n_log, ne=n_log2, Sp=n_sym, lin=n_lin))
    y_phys = torch.stack([torch.rand(B, H, W, generator=g) * 100, torch.rand(B, H, W, generator=g) * 5,
                          torch.randn(B, H, W, generator=g) * 200, torch.randn(B, H, W, generator=g) * 4], dim=1)
    y_phys[0, 0, 0, 0] = float("nan")
    y_phys[0, 1, 1, 1] = float("inf")
    y_phys[1, 2, 2, 2] = float("-inf")
    y_phys[1, 0, 3, 3] = -4.0
    z = norm.transform(y_phys, mask)
    zz = torch.randn(B, 4, H, W, generator=g) * 1.5
    zz[0, 2, 0, 0] = 80.0       # symexp overflow -> inf -> 0
    zz[1, 0, 1, 1] = 200.0      # exp overflow -> inf -> 0
    inv = norm.inverse(zz, mask)
    pred = zz + 0.1 * torch.randn(B, 4, H, W, generator=g)
    a, s, px = ref_losses.batch_error_sums_ev(pred, zz, mask, norm)
    em = np.zeros((20, 14), dtype=np.float32)
    em[3:17, 2:12] = 1
    em[8:11, 5:8] = 0
    ew = ref_losses.edge_weights(em, 3.0)
    ew0 = ref_losses.edge_weights(np.zeros((4, 4), dtype=np.float32), 3.0)
    save("norm_cases.npz", mask=mask, y_phys=y_phys, z=z, zz=zz, inv=inv, pred=pred,
         abs_sum=a, sq_sum=s, px=px, kinds=np.asarray([0, 0, 2, 1]),
         mus=np.asarray([2.5, -1.0, 0.2, 3.0], dtype=np.float32),
         sigmas=np.asarray([1.3, 0.7, 1.7, 2.0], dtype=np.float32),
         epss=np.asarray([1.0, 1e-3, 1e-12, 1e-12]), scales=np.asarray([1.0, 1.0, 10.0, 1.0]),
         edge_mask=em, edge_w=ew, edge_w_empty=ew0)


def adamw_case():
    g = torch.Generator().manual_seed(9)
    shapes = [(4, 3, 3, 3), (7,), (5, 6)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    opt = torch.optim.AdamW(ps, lr=3e-3, weight_decay=1e-4)
    arrs = {"n_tensors": len(ps), "lr": 3e-3, "wd": 1e-4, "clip": 1.0}
    for i, p in enumerate(ps):
        arrs[f"p0_{i}"] = p.detach().clone()
    for step in range(3):
        gs = [torch.randn(s, generator=g) * (2.0 if step == 0 else 0.05) for s in shapes]
        for p, gg in zip(ps, gs):
            p.grad = gg.clone()
        tot = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
        arrs[f"gnorm_{step}"] = tot
        for i, (p, gg) in enumerate(zip(ps, gs)):
            arrs[f"g{step}_{i}"] = gg
            arrs[f"p{step + 1}_{i}"] = p.detach().clone()
    save("adamw_clip.npz", **arrs)


class _ListDS(torch.utils.data.Dataset):
    def __init__(self, bt):
        self.bt = bt

    def __len__(self):
        return self.bt["x"].shape[0]

    def __getitem__(self, i):
        return {k: v[i] for k, v in self.bt.items()}


def train_case():
    """Two epochs of the reference train_unet on a tiny synthetic loader (fp32, CPU)."""
    cfg = UNetCfg(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2)
    sd = init_state_dict(cfg, seed=4)
    tr = make_batch(6, 16, 8, seed=100, jitter_masks=True)
    va = make_batch(4, 16, 8, seed=101, jitter_masks=True)
    tl = torch.utils.data.DataLoader(_ListDS(tr), batch_size=2, shuffle=False)
    vl = torch.utils.data.DataLoader(_ListDS(va), batch_size=2, shuffle=False)
    keys = [f"c{i}" for i in range(8)]
    norms = {}
    for i, k in enumerate(keys):
        if i % 3 == 0:
            n = ref_data.MaskedLogStandardizer(eps=1.0); n.mu, n.sigma = torch.tensor(1.0 + 0.1 * i), torch.tensor(0.8)
        elif i % 3 == 1:
            n = ref_data.MaskedSymLogStandardizer(scale=5.0); n.mu, n.sigma = torch.tensor(0.1 * i), torch.tensor(1.2)
        else:
            n = ref_data.MaskedLinearStandardizer(); n.mu, n.sigma = torch.tensor(0.5), torch.tensor(2.0)
        norms[k] = n
    norm = ref_data.MultiChannelNormalizer(keys, norms)

    def mk(in_ch, out_ch):
        m = ref.UNet(in_ch=in_ch, out_ch=out_ch, base=8, P=5, film_hidden=16, depth=2)
        m.load_state_dict(sd)
        return m
    path = "/tmp/_golden_train.pt"
    model, hist = ref.train_unet(tl, vl, norm, 1, "cpu", out_ch=8, lam_grad=0.1, lam_w=0.5, epochs=2, base=8,
                                 model_cls=mk, P=5, film_hidden=16, depth=2, save_path=path, multiscale=1,
                                 lr_init=2e-3, amp=False, lam_grad_warmup_start=0, lam_grad_warmup_end=1,
                                 channel_weights=CHANNEL_WEIGHTS_8, log_every=1)
    fin = model.state_dict()
    arrs = {f"hist_{k}": np.asarray(v, dtype=np.float64) for k, v in hist.items()}
    arrs.update(cfg_dict(cfg))
    arrs["final_checksum"] = sd_checksum(fin)
    arrs["final::out.weight"] = fin["out.weight"]
    arrs["final::encoders.0.net.0.weight"] = fin["encoders.0.net.0.weight"]
    arrs["final::film_bot.net.2.bias"] = fin["film_bot.net.2.bias"]
    ck = torch.load(path, weights_only=False)
    arrs["ckpt_keys"] = np.asarray(sorted(ck.keys()))
    arrs["ckpt_epoch"] = ck["epoch"]
    arrs["norm_kinds"] = np.asarray([i % 3 for i in range(8)])
    # the same run with the physical-space L1 term switched on (train.py:200-205, lam_ev > 0)
    model_ev, hist_ev = ref.train_unet(tl, vl, norm, 1, "cpu", out_ch=8, lam_grad=0.1, lam_w=0.5, lam_ev=0.05, epochs=2, base=8,
                                       model_cls=mk, P=5, film_hidden=16, depth=2, save_path=path, multiscale=1,
                                       lr_init=2e-3, amp=False, lam_grad_warmup_start=0, lam_grad_warmup_end=1,
                                       channel_weights=CHANNEL_WEIGHTS_8, log_every=1)
    for k, v in hist_ev.items():
        arrs[f"ev_hist_{k}"] = np.asarray(v, dtype=np.float64)
    fin_ev = model_ev.state_dict()
    arrs["ev_final::out.weight"] = fin_ev["out.weight"]
    arrs["ev_final::encoders.0.net.0.weight"] = fin_ev["encoders.0.net.0.weight"]
    # the term alone, value and gradient, on a given prediction (the expression of train.py:202-204 on the live normalizer)
    g = torch.Generator().manual_seed(12)
    pe = (0.8 * torch.randn(3, 8, 16, 8, generator=g)).requires_grad_(True)
    ye = 0.8 * torch.randn(3, 8, 16, 8, generator=g)
    me = make_batch(3, 16, 8, seed=102, jitter_masks=True, empty_last=True)["mask"]
    loss_ev = ((norm.inverse(pe.float(), me.float()) - norm.inverse(ye.float(), me.float())).abs() * me.float()).sum() / me.sum().clamp_min(1e-8)
    loss_ev.backward()
    arrs.update(ev_pred=pe.detach(), ev_target=ye, ev_mask=me, ev_loss=loss_ev.detach(), ev_grad=pe.grad)
    save("train_history.npz", **arrs)


def inverse_case():
    """The per-case inverse loop of scripts/eval_inverse_cycle_conditional_unet.py (:332-416, Adam path) on the live
    reference: the script's own module-level functions (weighted_masked_mse, masked_mae, build_x_from_mask) are imported;
    the loop body, which lives inside its main(), is driven here statement by statement in the same order."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_inverse_script",
                                                  "/root/reference/scripts/eval_inverse_cycle_conditional_unet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = UNetCfg(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2)
    sd = init_state_dict(cfg, seed=11)
    model = ref_model(cfg, sd).eval()
    N, H, W, steps, lr, reg, clip = 3, 16, 16, 12, 5e-2, 1e-4, 4.0
    bt = make_batch(N, H, W, in_ch=1, out_ch=8, P=5, seed=4242, jitter_masks=True, empty_last=False)
    mask = bt["mask"]                                   # (N,1,H,W)
    p_true = bt["params"]
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        y_true_norm = model(mod.build_x_from_mask(mask), params=p_true) + 0.05 * torch.randn(N, 8, H, W, generator=g)
    field_idx = [0, 1, 2, 4, 5, 7]
    cw_sel = torch.tensor([1.0, 1.0, 0.5, 2.0, 0.25, 1.0])
    p0 = 0.3 * torch.randn(N, 5, generator=g)
    arrs = dict(cfg_dict(cfg))
    arrs.update(mask=mask, y_true_norm=y_true_norm, p0=p0, field_idx=np.asarray(field_idx), cw_sel=cw_sel, steps=steps, lr=lr,
                reg=reg, clip=clip, seed=11, sd_checksum=sd_checksum(sd), tol=1e-6)
    for patience in (0, 4):
        hist = np.full((steps, N), np.nan, dtype=np.float64)
        p_fin, fit_fin, mae_fin, nrun = [], [], [], []
        for n in range(N):
            m_t = mask[n:n + 1]
            y_sel = y_true_norm[n:n + 1][:, field_idx]
            p_var = torch.nn.Parameter(p0[n:n + 1].clone())
            opt = torch.optim.Adam([p_var], lr=lr)
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=1e-5)
            es_best, es_wait, ran = float("inf"), 0, 0
            for step in range(steps):
                x = mod.build_x_from_mask(m_t)
                y_pred = model(x, params=p_var)
                loss_fit = mod.weighted_masked_mse(y_pred[:, field_idx], y_sel, m_t, cw_sel)
                loss = loss_fit + reg * (p_var ** 2).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                sched.step()
                loss_item = float(loss.item())
                with torch.no_grad():
                    p_var.clamp_(-clip, clip)
                hist[step, n] = loss_item
                ran += 1
                if patience > 0:
                    rel_improv = (es_best - loss_item) / max(abs(es_best), 1e-12)
                    if rel_improv > 1e-6:
                        es_best, es_wait = loss_item, 0
                    else:
                        es_wait += 1
                        if es_wait >= patience:
                            break
            with torch.no_grad():
                y_pred = model(mod.build_x_from_mask(m_t), params=p_var)
                fit_fin.append(float(mod.weighted_masked_mse(y_pred[:, field_idx], y_sel, m_t, cw_sel)))
                mae_fin.append(float(mod.masked_mae(y_pred[:, field_idx], y_sel, m_t)))
            p_fin.append(p_var.detach().numpy()[0].copy())
            nrun.append(ran)
        arrs.update({f"hist_p{patience}": hist, f"p_final_p{patience}": np.stack(p_fin), f"fit_p{patience}": np.asarray(fit_fin),
                     f"mae_p{patience}": np.asarray(mae_fin), f"steps_run_p{patience}": np.asarray(nrun)})
    for k, v in sd.items():
        arrs["w::" + k] = v
    save("inverse_case.npz", **arrs)


def inverse_lbfgs_case():
    """The script's DEFAULT optimiser path (:336-342,356-373: torch.optim.LBFGS, lr 1, max_iter 20, strong Wolfe) on the
    live reference, same imported script functions and set-up as inverse_case; the fixture of inverse_case.npz supplies
    the weights, masks, targets and starting points (this file only adds the L-BFGS trajectories)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_inverse_script",
                                                  "/root/reference/scripts/eval_inverse_cycle_conditional_unet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g0 = np.load(os.path.join(OUT, "inverse_case.npz"))
    cfg = UNetCfg(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2)
    sd = {k[3:]: torch.from_numpy(g0[k]).clone() for k in g0.files if k.startswith("w::")}
    model = ref_model(cfg, sd).eval()
    mask, y_true_norm, p0 = (torch.from_numpy(g0[k]) for k in ("mask", "y_true_norm", "p0"))
    field_idx = [int(i) for i in g0["field_idx"]]
    cw_sel = torch.from_numpy(g0["cw_sel"])
    N, steps, reg, clip = 2, 4, float(g0["reg"]), float(g0["clip"])
    hist = np.zeros((steps, N)); hfit = np.zeros((steps, N)); nev = np.zeros((steps, N))
    p_fin, fit_fin = [], []
    for n in range(N):
        m_t = mask[n:n + 1]
        y_sel = y_true_norm[n:n + 1][:, field_idx]
        p_var = torch.nn.Parameter(p0[n:n + 1].clone())
        opt = torch.optim.LBFGS([p_var], lr=1.0, max_iter=20, line_search_fn="strong_wolfe")
        for step in range(steps):
            _loss_val, _fit_val, _n = [None], [None], [0]

            def closure():
                opt.zero_grad(set_to_none=True)
                x = mod.build_x_from_mask(m_t)
                y_pred_norm = model(x, params=p_var)
                loss_fit = mod.weighted_masked_mse(y_pred_norm[:, field_idx], y_sel, m_t, cw_sel)
                loss = loss_fit + reg * (p_var ** 2).mean()
                loss.backward()
                _loss_val[0] = float(loss.item()); _fit_val[0] = float(loss_fit.item()); _n[0] += 1
                return loss
            opt.step(closure)
            with torch.no_grad():
                p_var.clamp_(-clip, clip)
            hist[step, n], hfit[step, n], nev[step, n] = _loss_val[0], _fit_val[0], _n[0]
        with torch.no_grad():
            y_pred = model(mod.build_x_from_mask(m_t), params=p_var)
            fit_fin.append(float(mod.weighted_masked_mse(y_pred[:, field_idx], y_sel, m_t, cw_sel)))
        p_fin.append(p_var.detach().numpy()[0].copy())
    save("inverse_lbfgs.npz", n_cases=N, steps=steps, hist=hist, hist_fit=hfit, n_evals=nev, p_final=np.stack(p_fin),
         fit_final=np.asarray(fit_fin))


def loaders_case():
    """solpex/data.py:467-572 make_loaders on a synthetic npz (multi-channel Y with non-finite entries, per-sample masks,
    `throughput_ratio` parameter transform, Log / SymLog / Linear normalizers): fitted statistics, parameter scaler, the
    train / val split and the first batches.  The npz arrays travel inside the fixture."""
    import tempfile
    rng = np.random.default_rng(11)
    N, H, W = 41, 16, 8
    mask = np.zeros((N, H, W), np.float32)
    for i in range(N):
        mask[i, 2 + i % 3:H - 2, 1:W - 1 - i % 2] = 1
    Y = rng.normal(size=(N, 3, H, W)).astype(np.float32) * 10
    Y[:, 0] = np.abs(Y[:, 0]) + 0.5
    Y[1, 0, 3, 3] = np.nan
    Y[2, 1, 4, 2] = np.inf
    Y[3, 2, 5, 3] = -np.inf
    X = np.exp(rng.normal(size=(N, 5))).astype(np.float32)
    X[:, 0] *= 1e21
    X[:, 2] *= 3e20
    y_keys_all = np.array(["Te", "S", "ne"])
    param_keys = np.array(["Gamma_D2", "Ptot_W", "n_core", "dna", "hci"])
    out = dict(Y=Y, mask=mask, X=X, y_keys_all=y_keys_all, param_keys_in=param_keys)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "d.npz")
        np.savez(path, Y=Y, y_keys=y_keys_all, mask=mask, X=X, param_keys=param_keys)
        norms = dict(Te=ref_data.MaskedLogStandardizer(eps=1.0), S=ref_data.MaskedSymLogStandardizer(scale=5.0),
                     ne=ref_data.MaskedLinearStandardizer())
        tl, vl, norm, P, hw, (pmu, pstd), pt, pk = ref_data.make_loaders(
            path, inputs_mode="params", batch_size=8, split=0.85, seed=42, device="cpu", y_keys=["Te", "S", "ne"],
            norms_by_name=norms, param_transform="throughput_ratio")
        out.update(P=P, H=hw[0], W=hw[1], param_mu=pmu, param_std=pstd, param_transform=np.asarray(pt),
                   param_keys_out=np.array([str(k) for k in pk]), idx_tr=np.asarray(tl.dataset.indices),
                   idx_va=np.asarray(vl.dataset.indices), n_train_batches=len(tl), n_val_batches=len(vl))
        for k in ("Te", "S", "ne"):
            out[f"mu_{k}"] = np.float64(norm.norms[k].mu)
            out[f"sigma_{k}"] = np.float64(norm.norms[k].sigma)
        torch.manual_seed(123)
        bt = next(iter(tl))
        bv = next(iter(vl))
        for tag, b in (("tr", bt), ("va", bv)):
            for k in ("x", "y", "mask", "params", "idx"):
                out[f"{tag}_{k}"] = b[k]
        # every validation batch's idx (order of an un-shuffled loader) and the last, short, training batch size
        out["va_idx_all"] = np.concatenate([np.asarray(b["idx"]) for b in vl])
        # legacy single-output path: y_keys=None -> MaskedLogStandardizer(eps=1) on y_key="Te", no parameter transform
        tl1, vl1, norm1, P1, hw1, (pmu1, pstd1), pt1, pk1 = ref_data.make_loaders(path, inputs_mode="params", batch_size=4,
                                                                                split=0.7, seed=7, device="cpu", y_key="Te")
        out.update(leg_mu=np.float64(norm1.mu), leg_sigma=np.float64(norm1.sigma), leg_param_mu=pmu1, leg_param_std=pstd1,
                   leg_idx_tr=np.asarray(tl1.dataset.indices), leg_y0=vl1.dataset[0]["y"], leg_param_keys=np.array([str(k) for k in pk1]))
        # autoencoder mode: x = normalised y, no parameter scaler
        norms = dict(Te=ref_data.MaskedLogStandardizer(eps=1.0), S=ref_data.MaskedSymLogStandardizer(scale=5.0))
        tl2, vl2, norm2, P2, hw2, (pmu2, pstd2), pt2, pk2 = ref_data.make_loaders(
            path, inputs_mode="autoencoder", batch_size=8, y_keys=["Te", "S"], norms_by_name=norms, device="cpu")
        assert pmu2 is None
        b2 = next(iter(vl2))
        out.update(ae_x=b2["x"], ae_params=b2["params"], ae_mu_S=np.float64(norm2.norms["S"].mu))
    save("loaders_case.npz", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    prod = dict(lam_grad=0.1, lam_w=0.5, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1",
                multiscale=0, ms_weight=0.5, grad_ds=2)
    unet_case("unet_small.npz", UNetCfg(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2),
              B=3, H=16, W=8, seed=0, loss_kw=prod)
    unet_case("unet_odd.npz", UNetCfg(in_ch=3, out_ch=4, base=8, P=0, depth=2),
              B=2, H=13, W=7, seed=1, loss_kw=dict(lam_grad=0.0, lam_w=0.0, base="mse"), empty_last=False)
    unet_case("unet_d3_ms.npz", UNetCfg(in_ch=1, out_ch=8, base=16, P=5, film_hidden=32, depth=3),
              B=2, H=32, W=16, seed=2, loss_kw=dict(prod, multiscale=1), store_weights=False, empty_last=False)
    unet_case("unet_cfgA_summary.npz", UNetCfg(in_ch=1, out_ch=8, base=32, P=5, film_hidden=128, depth=3),
              B=2, H=128, W=64, seed=0, loss_kw=prod, store_weights=False, summary_only=True, empty_last=False)
    latent_case()
    latent_mlp_case()
    loss_cases()
    norm_cases()
    adamw_case()
    train_case()
    inverse_case()
    inverse_lbfgs_case()
    loaders_case()


if __name__ == "__main__":
    if len(sys.argv) > 1:               # regenerate only the named cases, e.g.  make_golden.py inverse_case
        os.makedirs(OUT, exist_ok=True)
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
