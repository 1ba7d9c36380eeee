"""How far bf16 itself sits from fp32 on tiny problems: this package's bf16 tier and eager PyTorch under bf16 autocast (the oracle's
functional model on the GPU), both against the fp32 CPU oracle, on cases of tests/test_gpu_unet.py::test_random_architectures_against_oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_unet import _random_arch, build
from oracle import unet_ref, loss_ref
from oracle.unet_ref import UNetCfg, init_state_dict
from unet_amd.synth import make_batch
from unet_amd.losses import masked_weighted_loss
for seed in (9, 10, 3):
    a = _random_arch(seed); print(a)
    cfg = UNetCfg(in_ch=a["in_ch"], out_ch=a["out_ch"], base=a["base"], P=a["P"], film_hidden=32, depth=a["depth"])
    sd = init_state_dict(cfg, seed=50 + seed)
    bt = make_batch(a["B"], a["H"], a["W"], in_ch=a["in_ch"], out_ch=a["out_ch"], P=a["P"], seed=60 + seed, jitter_masks=True)
    kw = dict(base="huber", huber_beta=0.05, lam_grad=0.0, lam_w=0.0)
    def ref_grads(device, autocast):
        sdr = {k: v.clone().to(device).requires_grad_(True) for k, v in sd.items()}
        par = bt["params"].to(device) if a["P"] > 0 else None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            o = unet_ref.unet_forward(sdr, cfg, bt["x"].to(device), par)
        l = loss_ref.masked_weighted_loss(o.float(), bt["y"].to(device), bt["mask"].to(device), **kw)
        l.backward()
        return {k: v.grad.detach().cpu() for k, v in sdr.items() if v.grad is not None}, float(l)
    g32, l32 = ref_grads("cpu", False)
    gbf, lbf = ref_grads("cuda", True)
    model = build(cfg, sd).train(); model.compute_dtype = torch.bfloat16
    par = bt["params"].cuda() if a["P"] > 0 else None
    pred = model(bt["x"].cuda(), params=par)
    loss = masked_weighted_loss(pred, bt["y"].cuda(), bt["mask"].cuda(), **kw); loss.backward()
    k = "encoders.0.net.0.weight"
    ours = dict(model.named_parameters())[k].grad.cpu()
    e_ours = float((ours.double() - g32[k].double()).norm() / g32[k].double().norm())
    e_eager = float((gbf[k].double() - g32[k].double()).norm() / g32[k].double().norm())
    print(f"  loss fp32 {l32:.6f} eager-bf16 {lbf:.6f} ours {float(loss):.6f};  {k}: ours {e_ours:.3e}  eager autocast {e_eager:.3e}")
    worst_o = max((float((dict(model.named_parameters())[kk].grad.cpu().double() - g32[kk].double()).norm() / max(float(g32[kk].double().norm()), 1e-30)), kk) for kk in g32)
    worst_e = max((float((gbf[kk].double() - g32[kk].double()).norm() / max(float(g32[kk].double().norm()), 1e-30)), kk) for kk in g32)
    print("  worst ours", worst_o, " worst eager", worst_e)
