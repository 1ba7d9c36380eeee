"""Latent MLPs and trainers on the HIP path against the golden vectors captured from the live reference
(tests/golden/latent_mlp.npz) and, for LayerNorm + Dropout, against the oracle with the same masks.  fp32: <= 1e-4."""
import numpy as np
import pytest
import torch

from helpers import t, rel_err, max_err

pytestmark = pytest.mark.gpu


def _sd(g, prefix):
    return {k[len(prefix) + 2:]: t(g[k]) for k in g.files if k.startswith(prefix + "::")}


def _models(g):
    from unet_amd.latent import ParamToZ, ZToParam
    hid = tuple(int(v) for v in g["hidden"])
    Pd, zd = g["P_tr"].shape[1], g["Z_tr"].shape[1]
    p2z = ParamToZ(Pd, zd, hidden=hid); p2z.load_state_dict(_sd(g, "init_p2z"))
    z2p = ZToParam(zd, Pd, hidden=hid); z2p.load_state_dict(_sd(g, "init_z2p"))
    z2p_ln = ZToParam(zd, Pd, hidden=hid, use_layernorm=True); z2p_ln.load_state_dict(_sd(g, "init_z2p_ln"))
    return p2z, z2p, z2p_ln


def test_state_dict_keys_match_reference(golden):
    g = golden("latent_mlp.npz")
    p2z, z2p, z2p_ln = _models(g)
    assert sorted(p2z.state_dict()) == sorted(_sd(g, "init_p2z"))
    assert sorted(z2p_ln.state_dict()) == sorted(_sd(g, "init_z2p_ln"))


def test_mlp_forward_and_gradients_through_autograd(golden):
    g = golden("latent_mlp.npz")
    p2z, _, z2p_ln = _models(g)
    for name, m, a, b in (("p2z", p2z, g["P_va"], g["Z_va"]), ("z2p_ln", z2p_ln, g["Z_va"], g["P_va"])):
        m = m.cuda()
        x = t(a).cuda().requires_grad_(True)
        out = m(x)
        assert rel_err(out.detach().cpu(), t(g[f"fwd_{name}::out"])) < 1e-5
        # the SmoothL1 kernel supplies the loss value and d loss / d out; autograd carries it through the HIP backward
        from unet_amd import _lib as L
        val = torch.zeros(1, device="cuda")
        dout = torch.empty_like(out)
        L.check(L.lib().sx_smooth_l1(out.numel(), L.ptr(out.detach()), L.ptr(t(b).cuda()), 0.05, L.ptr(val), L.ptr(dout), 1.0, 0,
                                     L.stream_ptr()))
        out.backward(dout)
        assert abs(float(val) - float(g[f"fwd_{name}::loss"])) < 1e-5 * abs(float(g[f"fwd_{name}::loss"]))
        assert rel_err(x.grad.cpu(), t(g[f"fwd_{name}::dx"])) < 1e-4
        for k, p in m.named_parameters():
            assert rel_err(p.grad.cpu(), t(g[f"fwd_{name}::grad::{k}"])) < 1e-4, k
        with torch.no_grad():                       # inference path (no autograd) gives the same numbers
            assert torch.equal(m(x.detach()), out.detach())
    with pytest.raises(RuntimeError):
        p2z.cpu()(t(g["P_va"]))


def test_cycle_losses_with_dropout_against_oracle_with_the_same_masks():
    """train_cycle_consistent's step with dropout > 0 (solpex/latent.py:243-251): the reference evaluates p2z(p) and z2p(z)
    again, with fresh masks, inside the cycle terms.  Six forward runs, each through its own masks: the four loss values
    and every parameter gradient against the oracle fed the masks the HIP path drew."""
    from oracle import latent_ref
    from unet_amd import latent as HL
    torch.manual_seed(11)
    p2z = HL.ParamToZ(5, 24, hidden=(32, 16), dropout=0.2).cuda().train()
    z2p = HL.ZToParam(24, 5, hidden=(40, 24), dropout=0.25, use_layernorm=True).cuda().train()
    z, p = torch.randn(21, 24), torch.randn(21, 5)
    lam = 0.1
    eng = HL._Trainer([p2z, z2p], 1e-3, 1e-4, torch.device("cuda"))
    ctxs = HL._cycle_losses(eng, p2z, z2p, z.cuda(), p.cuda(), lam, True)
    torch.cuda.synchronize()
    mk = {k: [b["mask"].cpu() for b in c.blocks[:-1]] for k, c in ctxs.items()}
    assert ctxs["inv_in"] is not ctxs["inv"] and not torch.equal(mk["inv_in"][0], mk["inv"][0])     # fresh masks
    sa = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in p2z.state_dict().items()}
    sb = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in z2p.state_dict().items()}
    f = latent_ref.mlp_forward
    t_fwd = latent_ref.smooth_l1(f(sb, z, mk["fwd"]), p)
    t_inv = latent_ref.smooth_l1(f(sa, p, mk["inv"]), z)
    t_pzp = latent_ref.smooth_l1(f(sb, f(sa, p, mk["inv_in"]), mk["pzp"]), p)
    t_zpz = latent_ref.smooth_l1(f(sa, f(sb, z, mk["fwd_in"]), mk["zpz"]), z)
    (t_fwd + t_inv + lam * (t_pzp + t_zpz)).backward()
    row = eng.log[0].cpu()
    for got, ref in zip(row, (t_fwd, t_inv, t_pzp, t_zpz)):
        assert abs(float(got) - float(ref)) < 1e-5 * max(abs(float(ref)), 1e-3)
    for m, sd in ((p2z, sa), (z2p, sb)):
        for k, q in m.named_parameters():
            assert rel_err(eng.grads[q].cpu(), sd[k].grad) < 1e-4, k
    # eval mode: no masks, two runs per MLP as before
    p2z.eval(); z2p.eval()
    c2 = HL._cycle_losses(eng, p2z, z2p, z.cuda(), p.cuda(), lam, False)
    assert c2["inv_in"] is c2["inv"] and c2["fwd_in"] is c2["fwd"]


def test_layernorm_dropout_mlp_against_oracle_with_the_same_masks():
    from oracle import latent_ref
    from unet_amd.latent import ZToParam
    torch.manual_seed(3)
    m = ZToParam(24, 5, hidden=(40, 24), dropout=0.25, use_layernorm=True).cuda().train()
    x = torch.randn(19, 24)
    tgt = torch.randn(19, 5)
    ctx = m.hip_forward(x.cuda(), train=True)
    masks = [b["mask"].cpu() for b in ctx.blocks[:-1]]
    assert all(mk is not None and 0.5 < float((mk > 0).float().mean()) < 0.95 for mk in masks)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    out = latent_ref.mlp_forward(sd, xr, drop_masks=masks)
    loss = latent_ref.smooth_l1(out, tgt)
    loss.backward()
    assert rel_err(ctx.out.cpu(), out.detach()) < 1e-5
    grads = {p: torch.zeros_like(p) for p in m.parameters()}
    o2 = out.detach().clone().requires_grad_(True)
    latent_ref.smooth_l1(o2, tgt).backward()
    dx = m.hip_backward(ctx, o2.grad.cuda(), grads, need_dx=True)
    assert rel_err(dx.cpu(), xr.grad) < 1e-4
    for k, p in m.named_parameters():
        assert rel_err(grads[p].cpu(), sd[k].grad) < 1e-4, k


def test_trainers_reproduce_reference_histories(golden, tmp_path):
    from unet_amd import latent as HL
    g = golden("latent_mlp.npz")
    P_tr, P_va, Z_tr, Z_va = (g[k] for k in ("P_tr", "P_va", "Z_tr", "Z_va"))
    kw = dict(device="cuda", lr=2e-3, epochs=int(g["epochs"]), batch_size=int(g["batch_size"]), wd=1e-4)
    p2z, z2p, z2p_ln = _models(g)
    torch.manual_seed(31)
    path = str(tmp_path / "p2z.pt")
    m, h = HL.train_param2z(P_train=P_tr, Z_train=Z_tr, P_val=P_va, Z_val=Z_va, model=p2z, save_path=path, **kw)
    assert np.allclose(h["tr"], g["hist_param2z_tr"], rtol=1e-4) and np.allclose(h["va"], g["hist_param2z_va"], rtol=1e-4)
    for k, v in m.state_dict().items():
        assert max_err(v.cpu(), g[f"final_param2z::{k}"]) < 1e-4, k
    ck = torch.load(path, weights_only=True)
    assert sorted(ck) == ["best", "model"] and abs(ck["best"] - min(h["va"])) < 1e-12
    torch.manual_seed(32)
    path = str(tmp_path / "z2p.pt")
    m, h = HL.train_z2param(Z_train=Z_tr, P_train=P_tr, Z_val=Z_va, P_val=P_va, model=z2p_ln, save_path=path, **kw)
    assert np.allclose(h["tr"], g["hist_z2param_ln_tr"], rtol=1e-4) and np.allclose(h["va"], g["hist_z2param_ln_va"], rtol=1e-4)
    for k, v in m.state_dict().items():
        assert max_err(v.cpu(), g[f"final_z2param_ln::{k}"]) < 1e-4, k
    assert sorted(torch.load(path, weights_only=True)) == ["P", "best", "model", "z_dim"]
    p2z, z2p, _ = _models(g)
    torch.manual_seed(33)
    path = str(tmp_path / "cyc.pt")
    a, b, h = HL.train_cycle_consistent(Z_train=Z_tr, P_train=P_tr, Z_val=Z_va, P_val=P_va, p2z_model=p2z, z2p_model=z2p,
                                        lam_cycle=float(g["lam_cycle"]), save_path=path, **kw)
    for k in h:
        assert np.allclose(h[k], g[f"hist_cycle_{k}"], rtol=1e-4), k
    for k, v in a.state_dict().items():
        assert max_err(v.cpu(), g[f"final_cycle_p2z::{k}"]) < 1e-4, k
    for k, v in b.state_dict().items():
        assert max_err(v.cpu(), g[f"final_cycle_z2p::{k}"]) < 1e-4, k
    assert sorted(torch.load(path, weights_only=True)) == ["P", "best", "p2z_model", "z2p_model", "z_dim"]
    with pytest.raises(RuntimeError):
        HL.train_param2z(P_train=P_tr, Z_train=Z_tr, P_val=P_va, Z_val=Z_va, model=p2z, device="cpu")


def test_extract_z_dataset_matches_module_calls(tmp_path):
    from unet_amd import data as D
    from unet_amd.models import UNet, bottleneck_to_z
    from unet_amd.latent import extract_z_dataset
    rng = np.random.default_rng(2)
    N, H, W = 11, 16, 8
    path = str(tmp_path / "d.npz")
    np.savez(path, Te=np.abs(rng.normal(size=(N, H, W))).astype(np.float32), mask=np.ones((H, W), np.float32),
             X=rng.normal(size=(N, 5)).astype(np.float32))
    ds = D.SOLPSDataset(path, inputs="params")
    torch.manual_seed(0)
    model = UNet(in_ch=1, out_ch=1, base=8, P=5, film_hidden=16, z_dim=4, depth=2).cuda().eval()
    Z, P, idx = extract_z_dataset(ae_model=model, dataset=ds, device="cuda", batch_size=4)
    assert Z.shape == (N, 4) and P.shape == (N, 5) and idx.tolist() == list(range(N))
    assert np.allclose(P, ds.params)
    with torch.no_grad():
        item = ds[5]
        _, b = model(item["x"][None].cuda(), params=item["params"][None].cuda(), return_bottleneck=True)
        z5 = model.z_proj(bottleneck_to_z(b)).cpu().numpy()[0]
    assert np.allclose(Z[5], z5, rtol=1e-4, atol=1e-5)
    Z2, _, _ = extract_z_dataset(ae_model=model, dataset=ds, device="cuda", batch_size=4, max_batches=2)
    assert Z2.shape[0] == 8


def test_eval_param2z_one_and_sample_from_resident_loader(tmp_path):
    """The latent pipeline's evaluation helpers (solpex/utils.py:11-15,50-97; scripts/run_latent_pipeline.py:323-350) on the
    HIP path: a sample picked from a device-resident loader, the autoencoder's latent against the params -> z MLP's."""
    from oracle import unet_ref
    from oracle.unet_ref import UNetCfg
    from unet_amd import data as D
    from unet_amd.models import UNet
    from unet_amd.latent import ParamToZ
    from unet_amd.utils import sample_from_loader, eval_param2z_one, nearest_neighbor_in_Z, pick_device
    rng = np.random.default_rng(3)
    N, H, W = 12, 16, 8
    path = str(tmp_path / "d.npz")
    np.savez(path, Te=np.abs(rng.normal(size=(N, H, W))).astype(np.float32), mask=np.ones((H, W), np.float32),
             X=rng.normal(size=(N, 5)).astype(np.float32))
    dev = pick_device()
    assert dev.type == "cuda"
    tl, vl, norm, P, hw, (mu, std), pt, pk = D.make_loaders(path, batch_size=4, device=dev, resident=True)
    assert isinstance(tl, D.ResidentLoader)
    gidx, s = sample_from_loader(tl, k=2)
    assert gidx == tl.dataset.indices[2] and int(s["idx"]) == gidx and s["x"].shape == (1, H, W)
    torch.manual_seed(0)
    ae = UNet(in_ch=1, out_ch=1, base=8, P=5, film_hidden=16, depth=2).to(dev)
    mlp = ParamToZ(5, 32, hidden=(16, 16)).to(dev)
    z_mu, z_std = np.linspace(-0.1, 0.1, 32).astype(np.float32), np.linspace(0.5, 1.5, 32).astype(np.float32)
    out = eval_param2z_one(ae=ae, mlp=mlp, sample=s, device=dev, label="", z_mu=z_mu, z_std=z_std, p_mu=mu, p_std=std)
    assert set(out) == {"z_true", "z_pred", "cos", "z_true_norm", "z_pred_norm"}
    assert out["z_true"].shape == (1, 32) and out["z_pred"].shape == (1, 32) and not out["z_true"].is_cuda
    # oracle: the same two evaluations in plain PyTorch on the CPU
    cfg = UNetCfg(in_ch=1, out_ch=1, base=8, P=5, film_hidden=16, depth=2)
    sd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    _, b = unet_ref.unet_forward(sd, cfg, s["x"][None], s["params"][None], return_bottleneck=True)
    z_true = b.mean(dim=(2, 3))
    msd = {k: v.detach().cpu() for k, v in mlp.state_dict().items()}
    p = (s["params"].numpy() - mu) / std
    h = torch.from_numpy(p.astype(np.float32))[None]
    lin = [k[:-len(".weight")] for k in msd if k.endswith(".weight")]
    for i, name in enumerate(lin):
        h = h @ msd[name + ".weight"].T + msd[name + ".bias"]
        if i + 1 < len(lin):
            h = torch.nn.functional.silu(h)
    z_pred = h * torch.from_numpy(z_std)[None] + torch.from_numpy(z_mu)[None]
    assert rel_err(out["z_true"], z_true) < 1e-4 and rel_err(out["z_pred"], z_pred) < 1e-4
    cos = float(torch.nn.functional.cosine_similarity(z_true, z_pred).item())
    assert abs(out["cos"] - cos) < 1e-4
    j, d2 = nearest_neighbor_in_Z(out["z_pred"][0].numpy(), np.stack([z_true[0].numpy(), out["z_pred"][0].numpy()]))
    assert j == 1 and d2 == 0.0
