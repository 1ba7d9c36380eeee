"""The data path against the live reference's make_loaders (tests/golden/loaders_case.npz, written by
tools/make_golden.py: solpex/data.py:467-572 on a synthetic npz): fitted normalizer statistics, parameter transform and
scaler, train / val split, batch contents.  CPU tests cover the host loaders and the data-parallel sharding logic; the GPU
tests cover the device-resident loader and its per-rank shares."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_npz(g, tmp_path):
    path = str(tmp_path / "d.npz")
    np.savez(path, Y=g["Y"], y_keys=g["y_keys_all"], mask=g["mask"], X=g["X"], param_keys=g["param_keys_in"])
    return path


def _loaders(D, path, device, resident, **kw):
    norms = dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0),
                 ne=D.MaskedLinearStandardizer())
    return D.make_loaders(path, inputs_mode="params", batch_size=8, split=0.85, seed=42, device=device,
                          y_keys=["Te", "S", "ne"], norms_by_name=norms, param_transform="throughput_ratio",
                          resident=resident, **kw)


def _check_common(g, out):
    tl, vl, norm, P, hw, (pmu, pstd), pt, pk = out
    assert P == int(g["P"]) and tuple(hw) == (int(g["H"]), int(g["W"])) and pt == str(g["param_transform"])
    assert [str(k) for k in pk] == [str(k) for k in g["param_keys_out"]]
    for k in ("Te", "S", "ne"):
        assert abs(float(norm.norms[k].mu) - float(g[f"mu_{k}"])) < 1e-6 * max(1.0, abs(float(g[f"mu_{k}"]))), k
        assert abs(float(norm.norms[k].sigma) - float(g[f"sigma_{k}"])) < 1e-6 * float(g[f"sigma_{k}"]), k
    assert np.allclose(pmu, g["param_mu"], rtol=1e-6) and np.allclose(pstd, g["param_std"], rtol=1e-6)
    assert len(tl) == int(g["n_train_batches"]) and len(vl) == int(g["n_val_batches"])
    return tl, vl


def _check_batch(g, tag, b, tol):
    assert [int(i) for i in b["idx"]] == [int(i) for i in g[f"{tag}_idx"]]
    for k in ("x", "y", "mask", "params"):
        ref = torch.from_numpy(g[f"{tag}_{k}"])
        assert tuple(b[k].shape) == tuple(ref.shape), (tag, k)
        assert torch.allclose(b[k].cpu(), ref, rtol=tol, atol=tol), (tag, k, float((b[k].cpu() - ref).abs().max()))


def test_host_loaders_match_reference_fixture(golden, tmp_path):
    from unet_amd import data as D
    g = golden("loaders_case.npz")
    path = _write_npz(g, tmp_path)
    tl, vl = _check_common(g, _loaders(D, path, "cpu", False))
    assert list(tl.dataset.indices) == list(g["idx_tr"]) and list(vl.dataset.indices) == list(g["idx_va"])
    torch.manual_seed(123)
    _check_batch(g, "tr", next(iter(tl)), 1e-6)
    _check_batch(g, "va", next(iter(vl)), 1e-6)
    assert np.concatenate([np.asarray(b["idx"]) for b in vl]).tolist() == g["va_idx_all"].tolist()
    # legacy single-output path and autoencoder mode
    tl1, vl1, norm1, P1, hw1, (pmu1, pstd1), pt1, pk1 = D.make_loaders(path, inputs_mode="params", batch_size=4, split=0.7,
                                                                      seed=7, device="cpu", y_key="Te", resident=False)
    assert abs(float(norm1.mu) - float(g["leg_mu"])) < 1e-6 and abs(float(norm1.sigma) - float(g["leg_sigma"])) < 1e-6
    assert np.allclose(pmu1, g["leg_param_mu"], rtol=1e-6) and np.allclose(pstd1, g["leg_param_std"], rtol=1e-6)
    assert list(tl1.dataset.indices) == list(g["leg_idx_tr"]) and [str(k) for k in pk1] == [str(k) for k in g["leg_param_keys"]]
    assert torch.allclose(vl1.dataset[0]["y"], torch.from_numpy(g["leg_y0"]), rtol=1e-6, atol=1e-6)
    norms = dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0))
    tl2, vl2, norm2, P2, hw2, (pmu2, pstd2), pt2, pk2 = D.make_loaders(path, inputs_mode="autoencoder", batch_size=8,
                                                                      y_keys=["Te", "S"], norms_by_name=norms, device="cpu",
                                                                      resident=False)
    b2 = next(iter(vl2))
    assert pmu2 is None and abs(float(norm2.norms["S"].mu) - float(g["ae_mu_S"])) < 1e-6
    assert torch.allclose(b2["x"], torch.from_numpy(g["ae_x"]), rtol=1e-6, atol=1e-6)
    assert torch.allclose(b2["params"], torch.from_numpy(g["ae_params"]), rtol=1e-6, atol=1e-6)


def test_rank_share_partitions_every_global_batch():
    from unet_amd.data import rank_share
    for n, bs, world in [(34, 8, 2), (7, 8, 2), (100, 16, 8), (33, 4, 4), (5, 4, 3)]:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(n))
        seen, nb = [[] for _ in range(world)], [0] * world
        for lo in range(0, n, bs * world):
            gb = order[lo:lo + bs * world]
            shares = [rank_share(gb, r, world, n) for r in range(world)]
            assert len({int(s.numel()) for s in shares}) == 1            # same rows on every rank (same step count)
            cat = torch.cat(shares)
            assert cat[:gb.numel()].tolist() == gb.tolist() and (cat[gb.numel():] == n).all()   # global batch + padding
            for r in range(world):
                seen[r] += [int(v) for v in shares[r] if int(v) != n]
                nb[r] += 1
        flat = sum(seen, [])
        assert sorted(flat) == list(range(n)) and len(set(nb)) == 1       # disjoint, cover the data, equal batch counts


def _ds_worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from unet_amd import dist as D
    from unet_amd import data as DD
    D.init_from_env(backend="gloo")
    norms = dict(Te=DD.MaskedLogStandardizer(eps=1.0), S=DD.MaskedSymLogStandardizer(scale=5.0), ne=DD.MaskedLinearStandardizer())
    tl, vl, *_ = DD.make_loaders(path, batch_size=4, device="cpu", y_keys=["Te", "S", "ne"], norms_by_name=norms, resident=False)
    tl.sampler.set_epoch(0)
    D.check_equal_across_ranks(len(tl), "train batches")
    err = ""
    try:
        D.check_equal_across_ranks(len(tl) + rank, "a deliberately unequal count")
    except RuntimeError as e:
        err = str(e)
    q.put((rank, len(tl), len(vl), [int(i) for b in tl for i in b["idx"]], [int(i) for b in vl for i in b["idx"]], err,
           int(D.broadcast0(torch.tensor([rank + 5]))[0])))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_host_loaders_split_the_dataset(golden, tmp_path):
    """gloo, world 2: make_loaders picks up rank / world from the process group; the two ranks' index sets are disjoint,
    cover the split (DistributedSampler pads the odd one out by wrapping) and have equal batch counts; unequal counts raise
    on every rank instead of hanging."""
    g = golden("loaders_case.npz")
    path = _write_npz(g, tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 200
    procs = [ctx.Process(target=_ds_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nt0, nv0, tr0, va0, e0, b0), (_, nt1, nv1, tr1, va1, e1, b1) = res
    assert nt0 == nt1 and nv0 == nv1 and len(tr0) == len(tr1)
    assert set(tr0) | set(tr1) == set(int(i) for i in g["idx_tr"]) and len(set(tr0) & set(tr1)) <= 1
    assert set(va0) | set(va1) == set(int(i) for i in g["idx_va"])
    assert "differs between ranks" in e0 and "differs between ranks" in e1
    assert b0 == 5 and b1 == 5


@pytest.mark.gpu
def test_resident_loaders_match_reference_fixture(golden, tmp_path):
    from unet_amd import data as D
    g = golden("loaders_case.npz")
    path = _write_npz(g, tmp_path)
    tl, vl = _check_common(g, _loaders(D, path, "cuda", True))
    assert isinstance(tl, D.ResidentLoader)
    torch.manual_seed(123)
    _check_batch(g, "tr", next(iter(tl)), 1e-5)         # normalised on the device: fp32 rounding of exp / log
    _check_batch(g, "va", next(iter(vl)), 1e-5)


@pytest.mark.gpu
def test_resident_loader_rank_shares_rebuild_the_global_batches(golden, tmp_path):
    """DP-2 at per-rank batch 4: as many steps as the single-process loader at batch 8, the two ranks' rows of a step are
    disjoint and together cover the split exactly once per epoch; the short last batch is padded with an empty-mask sample
    so that both ranks run the same number of steps (rank_share, tested above, is what cuts a global batch)."""
    from unet_amd import data as D
    g = golden("loaders_case.npz")
    path = _write_npz(g, tmp_path)
    one, _v, *_ = _loaders(D, path, "cuda", True)
    n = len(g["idx_tr"])
    ranks = [D.make_loaders(path, inputs_mode="params", batch_size=4, split=0.85, seed=42, device="cuda",
                            y_keys=["Te", "S", "ne"],
                            norms_by_name=dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0),
                                               ne=D.MaskedLinearStandardizer()),
                            param_transform="throughput_ratio", resident=True, rank=r, world=2)[0] for r in range(2)]
    assert len(ranks[0]) == len(ranks[1]) == len(one)
    # every rank seeds its generator from the (identically seeded) global RNG; with a process group rank 0 broadcasts
    parts = []
    for ld in ranks:
        torch.manual_seed(77)
        parts.append(list(ld))
    seen = []
    for i in range(len(one)):
        a, b = parts[0][i], parts[1][i]
        assert a["x"].shape[0] == b["x"].shape[0]
        idx = torch.cat([a["idx"], b["idx"]])
        real = idx >= 0
        seen += [int(v) for v in idx[real]]
        for k in ("x", "y", "mask", "params"):
            cat = torch.cat([a[k], b[k]])
            assert float(cat[~real.to(cat.device)].abs().max() if (~real).any() else 0.0) == 0.0, k   # padding rows are empty
    assert sorted(seen) == sorted(int(i) for i in g["idx_tr"]) and len(seen) == n
