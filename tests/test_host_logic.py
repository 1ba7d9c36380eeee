"""CPU: host-side logic of the mirror package that needs no GPU: schedules, checkpoint (de)serialisation,
parameter transforms, dataset/loader plumbing, state_dict compatibility, loud failure without CUDA."""
import numpy as np
import pytest
import torch


def test_state_dict_names_match_oracle_shapes_and_reference_order():
    from oracle.unet_ref import UNetCfg, param_shapes
    from unet_amd.models import UNet
    from unet_amd._engine import param_names
    for cfg in (UNetCfg(1, 8, 32, P=5, depth=3), UNetCfg(10, 4, 16, P=0, depth=2), UNetCfg(1, 2, 8, P=5, z_dim=4, depth=4)):
        m = UNet(in_ch=cfg.in_ch, out_ch=cfg.out_ch, base=cfg.base, P=cfg.P, z_dim=cfg.z_dim, depth=cfg.depth)
        sd = m.state_dict()
        want = param_shapes(cfg)
        assert list(sd.keys()) == list(want.keys())
        assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
        names = param_names(cfg.in_ch, cfg.out_ch, cfg.base, cfg.depth, cfg.P)
        assert names == [k for k in want if not k.startswith("z_")]
    assert sum(p.numel() for p in UNet(1, 8, 32, P=5).parameters()) == 2199720


def test_no_cpu_fallback():
    from unet_amd.models import UNet
    from unet_amd.losses import masked_weighted_loss, mae_norm
    from unet_amd.train import train_unet
    m = UNet(1, 8, 8, P=5, depth=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 16, 8), params=torch.zeros(1, 5))
    with pytest.raises(RuntimeError):
        masked_weighted_loss(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4), torch.ones(1, 1, 4, 4))
    with pytest.raises(RuntimeError):
        mae_norm(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4), torch.ones(1, 1, 4, 4))
    with pytest.raises(RuntimeError):
        train_unet(None, None, None, 1, "cpu")
    with pytest.raises(ValueError):
        masked_weighted_loss(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4), torch.ones(1, 1, 4, 4), base="nope")


def test_lam_grad_schedule_matches_reference_ramp():
    from unet_amd.train import lam_grad_schedule
    assert lam_grad_schedule(0, 0.1, 20, 60) == 0.0 and lam_grad_schedule(19, 0.1, 20, 60) == 0.0
    assert lam_grad_schedule(20, 0.1, 20, 60) == 0.0 and abs(lam_grad_schedule(40, 0.1, 20, 60) - 0.05) < 1e-12
    assert lam_grad_schedule(60, 0.1, 20, 60) == 0.1 and lam_grad_schedule(5, 0.0, 0, 1) == 0.0
    assert lam_grad_schedule(1, 0.1, 0, 1) == 0.1


def test_norm_checkpoint_round_trip_and_errors():
    from unet_amd import data as D
    from unet_amd.train import _norm_to_ckpt
    from unet_amd.predict import _norm_from_ckpt, scale_params
    a = D.MaskedLogStandardizer(eps=1e-3); a.mu, a.sigma = torch.tensor(1.0), torch.tensor(2.0)
    b = D.MaskedSymLogStandardizer(scale=7.0); b.mu, b.sigma = torch.tensor(0.5), torch.tensor(1.5)
    c = D.MaskedLinearStandardizer(); c.mu, c.sigma = torch.tensor(0.1), torch.tensor(3.0)
    n = D.MultiChannelNormalizer(["Te", "S", "l"], dict(Te=a, S=b, l=c))
    pk = _norm_to_ckpt(n)
    assert pk["kind"] == "MultiChannelNormalizer" and pk["norms"]["S"]["scale"] == 7.0 and pk["norms"]["Te"]["eps"] == 1e-3
    n2 = _norm_from_ckpt(pk)
    y = torch.rand(2, 3, 4, 5) * 10
    m = torch.ones(2, 1, 4, 5)
    assert torch.allclose(n2.transform(y, m), n.transform(y, m))
    assert torch.allclose(n2.inverse(n2.transform(y, m), m), y, rtol=1e-4, atol=1e-4)
    leg = _norm_from_ckpt((torch.tensor(1.0), torch.tensor(2.0), 1.0))
    assert isinstance(leg, D.MaskedLogStandardizer)
    with pytest.raises(ValueError):
        _norm_from_ckpt({"kind": "Nope"})
    with pytest.raises(ValueError):
        _norm_from_ckpt(3)
    assert np.allclose(scale_params([1, 2], [1, 1], [2, 2]), [0, 0.5]) and scale_params([1, 2], None, None).dtype == np.float32


def test_param_transform_round_trip():
    from unet_amd.data import apply_param_transform, invert_param_transform, fit_param_scaler
    p = np.array([[1e21, 3e6, 2e20, 1.0, 2.0], [5e20, 2e6, 5e20, 0.5, 1.0]], dtype=np.float32)
    keys = ["Gamma_D2", "Ptot_W", "n_core", "dna", "hci"]
    q, k2 = apply_param_transform(p, keys, "throughput_ratio")
    assert k2[0] == "log10_thr" and k2[2] == "ratio_nc" and abs(q[0, 0] - np.log10(1.2e21)) < 1e-4
    r, k3 = invert_param_transform(q, k2, "throughput_ratio")
    assert k3[0] == "Gamma_D2" and k3[2] == "Gamma_core" and np.allclose(r, p, rtol=1e-4)
    with pytest.raises(ValueError):
        apply_param_transform(p, keys, "bogus")
    with pytest.raises(KeyError):
        apply_param_transform(p, ["a", "b", "c", "d", "e"], "throughput_ratio")
    mu, std = fit_param_scaler(np.array([[1.0, 2.0], [1.0, 4.0]]))
    assert std[0] == 1.0 and abs(std[1] - 1.0) < 1e-6


def test_dataset_and_loaders(tmp_path):
    from unet_amd import data as D
    rng = np.random.default_rng(0)
    N, Cc, H, W = 10, 3, 8, 8
    mask = np.zeros((N, H, W), np.float32); mask[:, 2:7, 1:7] = 1
    Y = np.abs(rng.normal(size=(N, Cc, H, W))).astype(np.float32) * 10
    path = str(tmp_path / "d.npz")
    np.savez(path, Y=Y, y_keys=np.array(["Te", "ne", "S"]), mask=mask, X=rng.normal(size=(N, 5)).astype(np.float32),
             param_keys=np.array(["Gamma_D2", "Ptot_W", "n_core", "dna", "hci"]))
    norms = dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0))
    tl, vl, norm, P, hw, (mu, std), pt, pk = D.make_loaders(path, batch_size=4, y_keys=["Te", "S"], norms_by_name=norms, device="cpu")
    assert P == 5 and hw == (8, 8) and mu.shape == (5,) and len(pk) == 5
    b = next(iter(tl))
    assert b["x"].shape[1:] == (1, 8, 8) and b["y"].shape[1:] == (2, 8, 8) and b["mask"].shape[1:] == (1, 8, 8) and b["params"].shape[1] == 5
    assert float(norm.norms["Te"].sigma) > 0
    with pytest.raises(KeyError):
        D.SOLPSDataset(path, y_keys=["nope"])
    with pytest.raises(ValueError):
        D.make_loaders(path, y_keys=["Te"], device="cpu")


def test_resident_loader_batch_order_mirrors_the_host_loader(tmp_path):
    """ResidentLoader takes its batch order from torch's own DataLoader over positions: for one torch seed it is the
    order of the reference's DataLoader(Subset(...), shuffle=True) (same RNG draws), checked here without a GPU."""
    import torch
    from torch.utils.data import DataLoader
    from unet_amd import data as D
    rng = np.random.default_rng(1)
    N, H, W = 23, 8, 8
    path = str(tmp_path / "d.npz")
    np.savez(path, Te=np.abs(rng.normal(size=(N, H, W))).astype(np.float32), mask=np.ones((H, W), np.float32),
             X=rng.normal(size=(N, 5)).astype(np.float32))
    tl, vl, *_ = D.make_loaders(path, batch_size=4, device="cpu")          # host loaders
    assert isinstance(tl, DataLoader)
    torch.manual_seed(77)
    host = [b["idx"].tolist() for b in tl] + [b["idx"].tolist() for b in tl]      # two epochs
    torch.manual_seed(77)
    pos = DataLoader(D._Positions(len(tl.dataset)), batch_size=4, shuffle=True)
    ind = np.asarray(tl.dataset.indices)
    mirror = [ind[b.numpy()].tolist() for b in pos] + [ind[b.numpy()].tolist() for b in pos]
    assert host == mirror
    with pytest.raises(RuntimeError):
        D.ResidentLoader(tl.dataset, 4, True, "cpu")


def test_utils_helpers_have_the_reference_return_values(tmp_path):
    """solpex/utils.py:11-48 as the pipeline scripts call them: `idx, sample = sample_from_loader(loader, k)`,
    `pick_device()` -> torch.device, parameter scaling, nearest latent."""
    import torch
    from unet_amd import data as D
    from unet_amd import utils as U
    import unet_amd
    import solpex
    rng = np.random.default_rng(2)
    N, H, W = 12, 8, 8
    path = str(tmp_path / "d.npz")
    np.savez(path, Te=np.abs(rng.normal(size=(N, H, W))).astype(np.float32), mask=np.ones((H, W), np.float32),
             X=rng.normal(size=(N, 5)).astype(np.float32))
    tl, vl, *_ = D.make_loaders(path, batch_size=4, device="cpu")
    for loader in (tl, vl):
        gidx, s = U.sample_from_loader(loader, k=1)
        assert gidx == loader.dataset.indices[1]
        ref = loader.dataset.dataset[gidx]
        assert set(s.keys()) == set(ref.keys()) and int(s["idx"]) == gidx
        assert torch.equal(s["x"], ref["x"]) and torch.equal(s["y"], ref["y"])
    dev = U.pick_device()
    assert isinstance(dev, torch.device) and dev.type in ("cuda", "mps", "cpu")
    p = U.scale_params_for_inference([1.0, 2.0, 3.0], [1.0, 1.0, 1.0], [2.0, 2.0, 2.0])
    assert p.dtype == np.float32 and np.allclose(p, [0.0, 0.5, 1.0])
    assert np.allclose(U.scale_params_for_inference([1, 2], None, None), [1.0, 2.0])
    Z = np.array([[0.0, 0.0], [1.0, 1.0], [3.0, 0.0]])
    j, d2 = U.nearest_neighbor_in_Z(np.array([0.9, 1.2]), Z)
    assert j == 1 and abs(d2 - 0.05) < 1e-12
    for name in ("pick_device", "sample_from_loader", "eval_param2z_one", "nearest_neighbor_in_Z"):   # solpex/__init__.py:11
        assert getattr(solpex, name) is getattr(unet_amd, name)


def test_fp32_mode_switch_needs_no_gpu():
    """unet_amd.fp32_exact / sx_set_fp32_exact: query, set, restore (host-side state of the library; no kernel runs)."""
    import unet_amd
    was = unet_amd.fp32_exact()
    assert unet_amd.fp32_exact(True) == was
    assert unet_amd.fp32_exact() is True
    assert unet_amd.fp32_exact(False) is True
    assert unet_amd.fp32_exact() is False
    unet_amd.fp32_exact(was)
