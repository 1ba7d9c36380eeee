"""CPU, gloo, world_size 2: the data-parallel protocol of unet_amd.dist (global loss denominators, SUM of
gradients, sharding helpers).  The compute stand-in is the oracle (the HIP kernels need a GPU); what is under
test is the host-side protocol the TrainEngine follows: DP-2 must equal the un-split batch."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_terms(pred, y, mask, w, cw):
    """Raw numerators / denominators of base+edge loss (mse), as sx_loss_prepare / _forward_backward split them."""
    scw = cw.sum()
    den = torch.stack([mask.sum() , (mask * w).sum()])
    px = (pred - y) ** 2 * mask * cw.view(1, -1, 1, 1)
    return den, px, scw


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from unet_amd import dist as D
    from oracle import unet_ref
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.synth import make_batch
    r, w_, _ = D.init_from_env(backend="gloo")
    assert (r, w_) == (rank, world) and D.is_active() and D.rank() == rank
    cfg = UNetCfg(in_ch=1, out_ch=4, base=8, P=5, film_hidden=8, depth=2)
    sd = {k: v.requires_grad_(True) for k, v in init_state_dict(cfg, seed=0).items()}
    bt = make_batch(4, 16, 8, out_ch=4, seed=5, jitter_masks=True, empty_last=True)     # per-sample masks differ
    b0, b1 = D.shard_range(4)
    sl = slice(b0, b1)
    wmap = torch.rand(16, 8, generator=torch.Generator().manual_seed(1))
    cw = torch.tensor([1.0, 1.5, 0.7, 1.2])
    pred = unet_ref.unet_forward(sd, cfg, bt["x"][sl], bt["params"][sl])
    den_local, px, scw = _local_terms(pred, bt["y"][sl], bt["mask"][sl], wmap, cw)

    def terms(den):          # step 3 of the protocol: numerators over GLOBAL denominators
        return px.sum() / (den[0] * scw).clamp_min(1e-8) + 0.5 * (px * wmap).sum() / (den[1] * scw).clamp_min(1e-8)
    loss_share = D.global_loss_protocol(den_local.detach(), terms)
    loss_share.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in sd.values()])
    h = D.all_reduce_sum(flat, async_op=True)            # step 4: SUM, no division
    h.wait()
    tot = loss_share.detach().clone()
    D.all_reduce_sum(tot)
    rows = D.gather_rows(bt["params"][sl] * 2.0, 4)
    q.put((rank, float(tot), flat.clone(), rows.clone()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp2_equals_unsplit_batch():
    sys.path.insert(0, ROOT)
    from oracle import unet_ref, loss_ref
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.synth import make_batch
    from unet_amd import dist as D
    assert D.shard_range(5, 0, 2) == (0, 3) and D.shard_range(5, 1, 2) == (3, 5) and D.world_size() == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = UNetCfg(in_ch=1, out_ch=4, base=8, P=5, film_hidden=8, depth=2)
    sd = {k: v.requires_grad_(True) for k, v in init_state_dict(cfg, seed=0).items()}
    bt = make_batch(4, 16, 8, out_ch=4, seed=5, jitter_masks=True, empty_last=True)
    wmap = torch.rand(16, 8, generator=torch.Generator().manual_seed(1))
    cw = torch.tensor([1.0, 1.5, 0.7, 1.2])
    pred = unet_ref.unet_forward(sd, cfg, bt["x"], bt["params"])
    full = loss_ref.masked_weighted_loss(pred, bt["y"], bt["mask"], w=wmap, lam_grad=0.0, lam_w=0.5, base="mse", channel_weights=cw)
    full.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in sd.values()])
    for rank, tot, g, rows in res:
        assert abs(tot - float(full)) < 1e-5 * abs(float(full))
        assert float((g - flat).norm() / flat.norm()) < 1e-5
        assert torch.allclose(rows, bt["params"] * 2.0)
    # the naive "mean of per-rank means" is NOT equivalent when masks differ -- guard the premise
    halves = []
    for sl in (slice(0, 2), slice(2, 4)):
        pr = unet_ref.unet_forward(sd, cfg, bt["x"][sl], bt["params"][sl])
        halves.append(float(loss_ref.masked_weighted_loss(pr, bt["y"][sl], bt["mask"][sl], w=wmap, lam_grad=0.0, lam_w=0.5,
                                                          base="mse", channel_weights=cw)))
    assert abs(0.5 * sum(halves) - float(full)) > 1e-4 * abs(float(full))


def _sweep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from unet_amd import dist as D
    from unet_amd import sweep
    D.init_from_env(backend="gloo")
    seen = []

    def run(trial):
        # inside a trial the process is a single-rank job: nothing would all-reduce
        assert D.world_size() == 1 and D.rank() == 0 and not D.is_active()
        if trial["tag"] == "bad":
            raise RuntimeError("boom")
        seen.append(trial["tag"])
        return {"tag": trial["tag"], "val": trial["x"] ** 2, "rank": rank}
    trials = [{"tag": f"t{i}", "x": i} for i in range(5)] + [{"tag": "bad", "x": 0}]
    res = sweep.run_trials(trials, run, costs=[5, 1, 1, 1, 1, 1])
    assert D.world_size() == world and D.is_active()          # local mode ended
    q.put((rank, res, seen))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sweep_trials_are_partitioned_and_gathered():
    sys.path.insert(0, ROOT)
    from unet_amd import sweep
    assert sweep.assign(5, 2) == [[0, 2, 4], [1, 3]]
    assert sweep.assign(4, 2, costs=[10, 1, 1, 1]) == [[0], [1, 2, 3]]
    with pytest.raises(ValueError):
        sweep.assign(3, 2, costs=[1.0])
    assert sweep.run_trials([1, 2, 3], lambda v: v * 10) == [10, 20, 30]        # single process
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(2)], key=lambda v: v[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, res0, seen0), (r1, res1, seen1) = out
    assert res0 == res1                                                          # every rank holds the full table
    assert [r.get("val") for r in res0[:5]] == [0, 1, 4, 9, 16] and res0[5] == {"error": "boom"}
    assert sorted(seen0 + seen1) == [f"t{i}" for i in range(5)] and set(seen0).isdisjoint(seen1)
    assert seen0 == ["t0"]                                                       # the costly trial alone on rank 0 (LPT)
