"""GPU: batched inverse driver (unet_amd/inverse.py, C ABI sx_inverse_*) against the live-reference per-case runs of
tests/golden/inverse_case.npz and against size-independent properties at the BASELINE config-4 shape.

Tolerances: fp32 tier.  A 12-step Adam trajectory amplifies the ~1e-6 forward differences (step-1 Adam moves every
parameter by ~lr*sign(g)), so parameters are held to 2e-3 absolute (lr = 5e-2, |p| ~ 0.3) and losses to 1e-3 relative.
"""
import numpy as np
import pytest
import torch

from helpers import cfg_from_golden, sd_from_golden, t

pytestmark = pytest.mark.gpu


def _model(g):
    from unet_amd.models import UNet
    cfg = cfg_from_golden(g)
    m = UNet(in_ch=cfg.in_ch, out_ch=cfg.out_ch, base=cfg.base, groups_gn=cfg.groups_gn, P=cfg.P,
             film_hidden=cfg.film_hidden, z_dim=cfg.z_dim, depth=cfg.depth)
    m.load_state_dict(sd_from_golden(g, cfg), strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("patience", [0, 4])
def test_batched_inverse_matches_reference_cases(golden, patience):
    from unet_amd.inverse import batched_inverse
    g = golden("inverse_case.npz")
    model = _model(g)
    res = batched_inverse(model, t(g["mask"]).cuda(), t(g["y_true_norm"]).cuda(), t(g["p0"]).cuda(),
                          field_idx=[int(i) for i in g["field_idx"]], channel_weights=t(g["cw_sel"]), steps=int(g["steps"]),
                          lr=float(g["lr"]), reg=float(g["reg"]), clip_scaled=float(g["clip"]), early_stop_patience=patience,
                          early_stop_tol=float(g["tol"]), return_history=True)
    torch.cuda.synchronize()
    assert [int(v) for v in res["steps_run"].cpu()] == [int(v) for v in g[f"steps_run_p{patience}"]]
    hist, ref = res["loss_history"].cpu().numpy(), g[f"hist_p{patience}"]
    ran = ~np.isnan(ref)
    assert np.array_equal(~np.isnan(hist), ran)
    assert np.allclose(hist[ran], ref[ran], rtol=1e-3, atol=1e-6)
    assert np.allclose(res["p_scaled"].cpu().numpy(), g[f"p_final_p{patience}"], rtol=0, atol=2e-3)
    assert np.allclose(res["fit"].cpu().numpy(), g[f"fit_p{patience}"], rtol=2e-3)
    assert np.allclose(res["cycle_mae_norm"].cpu().numpy(), g[f"mae_p{patience}"], rtol=2e-3)


def test_weighted_masked_mse_and_first_step_gradient_against_oracle(golden):
    """One forward + data-only backward: per-sample fit and d fit / d p against autograd on the CPU oracle."""
    from oracle import unet_ref, inverse_ref
    from unet_amd.inverse import _Problem, _weights, _mask3
    from unet_amd import _lib as L
    g = golden("inverse_case.npz")
    cfg = cfg_from_golden(g)
    sd = sd_from_golden(g, cfg)
    model = _model(g)
    mask, yt, p0 = t(g["mask"]), t(g["y_true_norm"]), t(g["p0"])
    fidx = [int(i) for i in g["field_idx"]]
    w, sel = _weights(model, fidx, t(g["cw_sel"]), "cuda")
    prob = _Problem(model, mask.shape[0], mask.shape[2], mask.shape[3], L.SX_F32)
    m3 = _mask3(mask.cuda())
    p = p0.cuda().contiguous()
    prob.forward_fit(m3.unsqueeze(1), p, yt.cuda().contiguous(), m3, w, sel, True)
    grad = prob.backward_params(p).clone()
    torch.cuda.synchronize()
    for n in range(mask.shape[0]):
        pv = p0[n:n + 1].clone().requires_grad_(True)
        y = unet_ref.unet_forward(sd, cfg, mask[n:n + 1], pv)
        fit = inverse_ref.weighted_masked_mse(y[:, fidx], yt[n:n + 1][:, fidx], mask[n:n + 1], t(g["cw_sel"]))
        fit.backward()
        assert abs(float(prob.fit[n]) - float(fit)) < 1e-4 * abs(float(fit))
        assert float((grad[n].cpu() - pv.grad[0]).norm()) < 2e-4 * float(pv.grad.norm())


def test_config4_shape_properties_bf16():
    """BASELINE config 4 shape (cfg A, 128x64, bf16) at 128 cases: (1) cases are independent - solving them in chunks of
    32 gives the same parameters as one batch; (2) the fit never gets worse than at p0 and improves for most cases;
    (3) parameters stay inside the clamp; (4) a case whose mask is empty keeps fit 0 and only feels the regulariser."""
    from unet_amd.models import UNet
    from unet_amd.inverse import batched_inverse
    from unet_amd.synth import make_batch
    torch.manual_seed(0)
    model = UNet(in_ch=1, out_ch=8, base=32, P=5, depth=3).cuda().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".net.2." in n and "film" in n:
                p.normal_(0, 0.05)
    model.compute_dtype = torch.bfloat16
    N = 128
    bt = make_batch(N, 128, 64, seed=77, jitter_masks=True)
    mask, p_true = bt["mask"].cuda(), bt["params"].cuda()
    mask[5] = 0.0
    with torch.no_grad():
        y_true = model(mask, params=p_true).float()
    p0 = torch.zeros_like(p_true)
    kw = dict(steps=6, lr=5e-2, early_stop_patience=0)
    r0 = batched_inverse(model, mask, y_true, p0, steps=0, early_stop_patience=0)
    full = batched_inverse(model, mask, y_true, p0, batch=128, **kw)
    part = batched_inverse(model, mask, y_true, p0, batch=32, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(full["p_scaled"]).all() and float(full["p_scaled"].abs().max()) <= 4.0
    assert float((full["p_scaled"] - part["p_scaled"]).abs().max()) < 2e-2
    better = (full["fit"] < r0["fit"]).float().mean()
    assert float(better) > 0.8
    assert float(full["fit"][5]) == 0.0 and float(full["p_scaled"][5].abs().max()) < 1e-2
    assert [int(v) for v in full["steps_run"].cpu()] == [6] * N


def test_lbfgs_path_matches_reference_trajectories(golden):
    """optimizer="lbfgs" (the reference script's default): host L-BFGS + strong Wolfe around the device closure.  fp32 tier;
    the line search amplifies rounding, so the recovered parameters get 2e-2 and the losses 1e-3."""
    from unet_amd.inverse import batched_inverse
    g, gl = golden("inverse_case.npz"), golden("inverse_lbfgs.npz")
    model = _model(g)
    N = int(gl["n_cases"])
    mask, yt, p0 = (t(g[k])[:N].cuda() for k in ("mask", "y_true_norm", "p0"))
    res = batched_inverse(model, mask, yt, p0, field_idx=[int(i) for i in g["field_idx"]], channel_weights=t(g["cw_sel"]),
                          steps=int(gl["steps"]), optimizer="lbfgs", reg=float(g["reg"]), clip_scaled=float(g["clip"]),
                          early_stop_patience=0, dtype=torch.float32, return_history=True)
    h = res["loss_history"].cpu().numpy()
    assert np.allclose(h, gl["hist"], rtol=1e-3), (h, gl["hist"])
    assert np.allclose(res["fit"].cpu().numpy(), gl["fit_final"], rtol=1e-3)
    assert np.abs(res["p_scaled"].cpu().numpy() - gl["p_final"]).max() < 2e-2
    assert res["steps_run"].tolist() == [int(gl["steps"])] * N
    with pytest.raises(ValueError):
        batched_inverse(model, mask, yt, p0, optimizer="sgd")
