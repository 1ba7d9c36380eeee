"""CPU: the oracle (oracle/) reproduces every golden vector captured from the live reference."""
import numpy as np
import pytest
import torch

import oracle
from oracle import unet_ref, loss_ref, norm_ref, optim_ref
from helpers import cfg_from_golden, sd_from_golden, loss_kw_from_golden, t, rel_err, max_err

TOL = 2e-6


def _run_unet_case(g, summary=False):
    cfg = cfg_from_golden(g)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_from_golden(g, cfg).items()}
    x, y, mask = t(g["x"]), t(g["y"]), t(g["mask"])
    params = t(g["params"]).requires_grad_(True) if cfg.P > 0 else None
    cw = t(g["cw"]) if "cw" in g.files else None
    pred, b = unet_ref.unet_forward(sd, cfg, x, params, return_bottleneck=True)
    loss = loss_ref.masked_weighted_loss(pred, y, mask, w=t(g["w_map"]), channel_weights=cw, **loss_kw_from_golden(g))
    loss.backward()
    assert max_err(pred.detach(), g["pred"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    if params is not None:
        assert max_err(params.grad, g["dparams"]) < 2e-5
    return sd, b


@pytest.mark.parametrize("name", ["unet_small.npz", "unet_odd.npz", "unet_d3_ms.npz"])
def test_unet_forward_backward(golden, name):
    g = golden(name)
    sd, b = _run_unet_case(g)
    assert max_err(b.detach(), g["bott"]) < TOL
    for k, v in sd.items():
        assert max_err(v.grad, g["grad::" + k]) < 2e-5, k


def test_unet_cfgA_summary(golden):
    g = golden("unet_cfgA_summary.npz")
    sd, b = _run_unet_case(g, summary=True)
    assert abs(float(b.detach().norm()) - float(g["b_norm"])) < 1e-5 * float(g["b_norm"])
    names = [str(s) for s in g["grad_names"]]
    for n, gn in zip(names, g["grad_norms"]):
        assert abs(float(sd[n].grad.norm()) - gn) <= 2e-4 * gn + 1e-9, n
    for k in g.files:
        if k.startswith("grad::"):
            assert max_err(sd[k[6:]].grad, g[k]) < 1e-4, k


def test_latent_helpers(golden):
    g = golden("unet_latent.npz")
    cfg = cfg_from_golden(g)
    sd = sd_from_golden(g, cfg)
    x, params = t(g["x"]), t(g["params"])
    with torch.no_grad():
        y, b = unet_ref.unet_forward(sd, cfg, x, params, return_bottleneck=True)
        b2, skips = unet_ref.unet_encode(sd, cfg, x, params)
        z = unet_ref.project_z(sd, cfg, b)
        bb = unet_ref.unproject_z(sd, cfg, z, b.shape)
        y2 = unet_ref.unet_decode(sd, cfg, bb, skips, params)
        y_nop = unet_ref.unet_forward(sd, cfg, x, None)
    assert max_err(y, g["y"]) < TOL and max_err(b, g["bott"]) < TOL
    assert max_err(skips[0], g["skip0"]) < TOL and max_err(skips[2], g["skip2"]) < TOL
    assert max_err(unet_ref.bottleneck_to_z(b), g["z0"]) < TOL
    assert max_err(z, g["z"]) < TOL and max_err(bb, g["bb"]) < TOL
    assert max_err(y2, g["y2"]) < TOL and max_err(y_nop, g["y_nop"]) < TOL


def test_loss_cases(golden):
    g = golden("loss_cases.npz")
    pred, target, mask, w, cw = (t(g[k]) for k in ("pred", "target", "mask", "w", "cw"))
    for i, d in enumerate(g["desc"]):
        base, gmode, gbase, gds, ms, use_w, use_cw, lam_w, lam_grad = str(d).split("|")
        p = pred.clone().requires_grad_(True)
        L = loss_ref.masked_weighted_loss(p, target, mask, w=w if int(use_w) else None, lam_grad=float(lam_grad),
                                          lam_w=float(lam_w), base=base, huber_beta=0.05, grad_mode=gmode,
                                          grad_base=gbase, multiscale=int(ms), ms_weight=0.5, grad_ds=int(gds),
                                          channel_weights=cw if int(use_cw) else None)
        L.backward()
        assert abs(float(L) - g["losses"][i]) < TOL * abs(g["losses"][i]), d
        assert max_err(p.grad, g["grads"][i]) < 1e-5, d
    z = loss_ref.masked_weighted_loss(pred, target, torch.zeros_like(mask), w=w, lam_grad=0.1, lam_w=0.5, grad_ds=2,
                                      channel_weights=cw)
    assert float(z) == float(g["loss_zero_mask"])
    assert abs(float(loss_ref.mae_norm(pred, target, mask)) - float(g["mae_norm"])) < 1e-6
    with pytest.raises(ValueError):
        loss_ref.masked_weighted_loss(pred, target, mask, base="nope")
    with pytest.raises(ValueError):
        loss_ref.masked_weighted_loss(pred, target, mask, grad_base="nope")


def test_normalizers_and_metrics(golden):
    g = golden("norm_cases.npz")
    specs = [norm_ref.NormSpec(int(k), float(m), float(s), float(e), float(sc)) for k, m, s, e, sc in
             zip(g["kinds"], g["mus"], g["sigmas"], g["epss"], g["scales"])]
    mask = t(g["mask"])
    z = norm_ref.norm_transform(specs, t(g["y_phys"]), mask)
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=2e-6, atol=1e-6)
    inv = norm_ref.norm_inverse(specs, t(g["zz"]), mask)
    np.testing.assert_allclose(inv.numpy(), g["inv"], rtol=2e-6, atol=1e-6)
    a, s, px = loss_ref.batch_error_sums_ev(t(g["pred"]), t(g["zz"]), mask, lambda q, m: norm_ref.norm_inverse(specs, q, m))
    assert abs(float(a) - float(g["abs_sum"])) < 1e-5 * float(g["abs_sum"])
    assert abs(float(s) - float(g["sq_sum"])) < 1e-5 * float(g["sq_sum"])
    assert float(px) == float(g["px"])
    np.testing.assert_allclose(loss_ref.edge_weights(g["edge_mask"], 3.0), g["edge_w"], rtol=1e-6, atol=1e-7)
    assert not loss_ref.edge_weights(np.zeros((4, 4), np.float32)).any()


def test_adamw_clip(golden):
    g = golden("adamw_clip.npz")
    n = int(g["n_tensors"])
    ps = [t(g[f"p0_{i}"]) for i in range(n)]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for step in range(3):
        gs = [t(g[f"g{step}_{i}"]) for i in range(n)]
        gs, tot = optim_ref.clip_grad_norm(gs, float(g["clip"]))
        assert abs(float(tot) - float(g[f"gnorm_{step}"])) < 1e-6 * float(tot)
        for i in range(n):
            ps[i], ms[i], vs[i] = optim_ref.adamw_step(ps[i], gs[i], ms[i], vs[i], step + 1, float(g["lr"]),
                                                       weight_decay=float(g["wd"]))
            assert max_err(ps[i], g[f"p{step + 1}_{i}"]) < 1e-6


def test_inverse_loop_matches_reference(golden):
    """Per-case inverse loop (Adam + cosine, clamp, the reference's early-stop arithmetic) against the live-reference run."""
    from oracle import inverse_ref
    g = golden("inverse_case.npz")
    cfg = cfg_from_golden(g)
    sd = sd_from_golden(g, cfg)
    mask, yt, p0 = t(g["mask"]), t(g["y_true_norm"]), t(g["p0"])
    fidx = [int(i) for i in g["field_idx"]]
    for patience in (0, 4):
        for n in range(mask.shape[0]):
            p, fit, mae, losses = inverse_ref.inverse_case(sd, cfg, mask[n:n + 1], yt[n:n + 1], p0[n:n + 1], fidx, t(g["cw_sel"]),
                                                           steps=int(g["steps"]), lr=float(g["lr"]), reg=float(g["reg"]),
                                                           clip=float(g["clip"]), patience=patience, tol=float(g["tol"]))
            assert len(losses) == int(g[f"steps_run_p{patience}"][n])
            ref_l = g[f"hist_p{patience}"][:len(losses), n]
            assert np.allclose(losses, ref_l, rtol=2e-5, atol=1e-7)
            assert np.allclose(p.numpy(), g[f"p_final_p{patience}"][n], rtol=0, atol=2e-5)
            assert abs(fit - float(g[f"fit_p{patience}"][n])) < 2e-5 * abs(float(g[f"fit_p{patience}"][n]))
            assert abs(mae - float(g[f"mae_p{patience}"][n])) < 2e-5 * abs(float(g[f"mae_p{patience}"][n]))
    # the reference's early stop fires after exactly `patience` steps (rel_improv is NaN while best is inf)
    assert [int(v) for v in g["steps_run_p4"]] == [4, 4, 4]


def _sd(g, prefix):
    return {k[len(prefix) + 2:]: t(g[k]) for k in g.files if k.startswith(prefix + "::")}


def test_latent_mlps_and_trainers_match_reference(golden):
    """solpex/latent.py MLPs (forward, gradients) and the three trainers' histories / final weights vs the live reference."""
    from oracle import latent_ref
    g = golden("latent_mlp.npz")
    P_tr, P_va, Z_tr, Z_va = (g[k] for k in ("P_tr", "P_va", "Z_tr", "Z_va"))
    for name, a, b in (("p2z", P_va, Z_va), ("z2p_ln", Z_va, P_va)):
        sd = {k: v.requires_grad_(True) for k, v in _sd(g, f"init_{name}").items()}
        x = t(a).requires_grad_(True)
        out = latent_ref.mlp_forward(sd, x)
        loss = latent_ref.smooth_l1(out, t(b))
        loss.backward()
        assert max_err(out, g[f"fwd_{name}::out"]) < 2e-6 and abs(float(loss) - float(g[f"fwd_{name}::loss"])) < 1e-6
        assert max_err(x.grad, g[f"fwd_{name}::dx"]) < 1e-6
        for k, v in sd.items():
            assert max_err(v.grad, g[f"fwd_{name}::grad::{k}"]) < 1e-6, k
    kw = dict(lr=2e-3, epochs=int(g["epochs"]), batch_size=int(g["batch_size"]), wd=1e-4)
    torch.manual_seed(31)
    fin, h = latent_ref.train_single(_sd(g, "init_p2z"), P_tr, Z_tr, P_va, Z_va, **kw)
    assert np.allclose(h["tr"], g["hist_param2z_tr"], rtol=2e-5) and np.allclose(h["va"], g["hist_param2z_va"], rtol=2e-5)
    for k, v in fin.items():
        assert max_err(v, g[f"final_param2z::{k}"]) < 2e-5, k
    torch.manual_seed(32)
    fin, h = latent_ref.train_single(_sd(g, "init_z2p_ln"), Z_tr, P_tr, Z_va, P_va, **kw)
    assert np.allclose(h["tr"], g["hist_z2param_ln_tr"], rtol=2e-5) and np.allclose(h["va"], g["hist_z2param_ln_va"], rtol=2e-5)
    for k, v in fin.items():
        assert max_err(v, g[f"final_z2param_ln::{k}"]) < 2e-5, k
    torch.manual_seed(33)
    fa, fb, h = latent_ref.train_cycle(_sd(g, "init_p2z"), _sd(g, "init_z2p"), Z_tr, P_tr, Z_va, P_va,
                                       lam_cycle=float(g["lam_cycle"]), **kw)
    for k in h:
        assert np.allclose(h[k], g[f"hist_cycle_{k}"], rtol=2e-5), k
    for k, v in fa.items():
        assert max_err(v, g[f"final_cycle_p2z::{k}"]) < 2e-5, k
    for k, v in fb.items():
        assert max_err(v, g[f"final_cycle_z2p::{k}"]) < 2e-5, k


def test_inverse_lbfgs_matches_reference(golden):
    """The script's default L-BFGS path (strong Wolfe) against the live-reference trajectories: same closure counts,
    same last-closure losses, same recovered parameters."""
    from oracle import inverse_ref
    g, gl = golden("inverse_case.npz"), golden("inverse_lbfgs.npz")
    cfg = cfg_from_golden(g)
    sd = sd_from_golden(g, cfg)
    mask, yt, p0 = t(g["mask"]), t(g["y_true_norm"]), t(g["p0"])
    fidx = [int(i) for i in g["field_idx"]]
    for n in range(int(gl["n_cases"])):
        p, fit, losses, counts = inverse_ref.inverse_case_lbfgs(sd, cfg, mask[n:n + 1], yt[n:n + 1], p0[n:n + 1], fidx, t(g["cw_sel"]),
                                                                steps=int(gl["steps"]), reg=float(g["reg"]), clip=float(g["clip"]))
        assert counts == [int(v) for v in gl["n_evals"][:, n]]
        assert np.allclose(losses, gl["hist"][:, n], rtol=1e-5)
        assert abs(fit - float(gl["fit_final"][n])) < 1e-5 * abs(fit)
        assert max_err(p, gl["p_final"][n]) < 1e-3
