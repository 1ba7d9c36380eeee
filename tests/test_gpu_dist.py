"""Two ranks on ONE GPU (gloo moves the CUDA tensors; the box has a single card, RCCL needs one device per rank): the real
TrainEngine data-parallel step -- global loss denominators, the decoder-bucket all-reduce launched between the two backward
halves, the late bucket (FiLM generators, decoder GroupNorm affines and biases finished by the encoder half under
SX_BWD_DEFER_FILM) -- must leave every rank with the parameters a single process gets from the un-split batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTS = dict(lam_w=0.5, lam_grad=0.1, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1", multiscale=1,
            ms_weight=0.5, grad_ds=2)
STEPS = 3


def _engine(dtype):
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.models import UNet
    from unet_amd.train import TrainEngine
    from unet_amd.losses import edge_weights
    from unet_amd.synth import make_batch
    cfg = UNetCfg(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2)
    sd = init_state_dict(cfg, seed=6)
    bt = {k: v.cuda() for k, v in make_batch(8, 16, 16, seed=3, jitter_masks=True, empty_last=True).items()}
    w = torch.from_numpy(edge_weights(bt["mask"][0, 0].cpu().numpy()))
    m = UNet(in_ch=1, out_ch=8, base=8, P=5, film_hidden=16, depth=2)
    m.load_state_dict(sd)
    m = m.cuda().train()
    eng = TrainEngine(m, dtype=dtype, lr=1e-3, weight_decay=1e-4, clip_grad=0.5, w_map=w, loss_opts=OPTS)
    return eng, bt


def _run(eng, bt, sl):
    losses, grads = [], None
    for i in range(STEPS):
        out = eng.train_step(bt["x"][sl], bt["y"][sl], bt["mask"][sl], bt["params"][sl], lam_grad=0.1)
        losses.append(out[0:1].clone())
        if i == 0:      # the arena still holds the (all-reduced) gradient the optimizer just consumed
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().cpu().clone() for n, p in eng.model.named_parameters()}
    torch.cuda.synchronize()
    names = [n for n, _ in eng.model.named_parameters()]
    return torch.cat(losses).cpu(), {n: p.detach().cpu().clone() for n, p in eng.model.named_parameters()}, names, grads


def _worker(rank, world, port, dtype_name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    _fh = open(os.path.join(ROOT, "gpurun_out", f"dist_worker{rank}.txt"), "w")
    faulthandler.dump_traceback_later(45, file=_fh)
    from unet_amd import dist as D
    D.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    eng, bt = _engine(getattr(torch, dtype_name))
    assert eng.world == world
    b0, b1 = D.shard_range(bt["x"].shape[0])
    losses, params, _, grads = _run(eng, bt, slice(b0, b1))
    tot = losses.cuda()
    D.all_reduce_sum(tot)                   # per-rank shares of the global loss
    # numpy: pickled by value (torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, tot.cpu().numpy(), {k: v.numpy() for k, v in params.items()}, {k: v.numpy() for k, v in grads.items()}))
    faulthandler.cancel_dump_traceback_later()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("dtype_name,tol", [("float32", 2e-4), ("bfloat16", 3e-2)])
def test_two_rank_engine_steps_equal_single_process(dtype_name, tol):
    sys.path.insert(0, ROOT)
    from helpers import rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    try:
        res = sorted([q.get(timeout=75) for _ in range(2)], key=lambda t: t[0])
    except queue.Empty:
        # a stall is a FAILURE: the workers' faulthandler stacks (dumped after 45 s) say which call each rank sat in
        for p in procs:
            p.terminate()
        for p in procs:
            p.join(timeout=20)
        stacks = ""
        for r in range(2):
            try:
                stacks += f"\n--- rank {r}\n" + open(os.path.join(ROOT, "gpurun_out", f"dist_worker{r}.txt")).read()[-3000:]
            except OSError:
                pass
        pytest.fail("the two ranks sharing one GPU did not finish in 75 s" + stacks)
    res = [(r, torch.from_numpy(t), {k: torch.from_numpy(v) for k, v in pr.items()}, {k: torch.from_numpy(v) for k, v in gr.items()})
           for r, t, pr, gr in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eng, bt = _engine(getattr(torch, dtype_name))
    losses, params, names, grads = _run(eng, bt, slice(0, bt["x"].shape[0]))
    for rank, tot, pr, gr in res:
        assert float((tot - losses).abs().max()) < tol * float(losses.abs().max()), (rank, tot, losses)
        for n in names:
            # the gradient is the sharp check (AdamW's first steps move a weight by ~lr whatever the gradient's size)
            assert rel_err(gr[n], grads[n]) < tol, (rank, n, "grad")
            assert rel_err(pr[n], params[n]) < tol, (rank, n)
    # both ranks hold the same parameters bit for bit (same reduced gradients, same optimizer arithmetic)
    for n in names:
        assert torch.equal(res[0][2][n], res[1][2][n]), n


# ---------------------------------------------------------------------------------------------------------------------
# The REAL RCCL path on one card: a one-rank "nccl" group with SX_DP_FORCE=1.  A single-rank all-reduce is a valid RCCL
# collective, so TrainEngine.train_step runs its data-parallel protocol with genuine ProcessGroupNCCL Work handles on RCCL's
# own stream: denominators reduced asynchronously before the forward pass, the three gradient buckets reduced while the
# rest of the backward still writes the arena, wait() ordering before clip + AdamW.  The result must equal the plain step.
def _nccl_worker(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SX_DP_FORCE="1")
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fh = open(os.path.join(ROOT, "gpurun_out", "dist_nccl_worker.txt"), "w")
    faulthandler.dump_traceback_later(60, file=fh)
    import torch.distributed as dist
    from unet_amd import dist as D
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    out = {}
    for dtype_name in ("float32", "bfloat16"):
        assert D.is_active() and D.world_size() == 1
        eng, bt = _engine(getattr(torch, dtype_name))
        assert eng.dp and eng.n_dec < eng.n_bot < eng.flat_g.numel()
        losses, params, names, grads = _run(eng, bt, slice(0, bt["x"].shape[0]))
        with D.local_mode():
            assert not D.is_active()
            eng0, _ = _engine(getattr(torch, dtype_name))
            assert not eng0.dp
            losses0, params0, _, grads0 = _run(eng0, bt, slice(0, bt["x"].shape[0]))
        out[dtype_name] = (losses.numpy(), losses0.numpy(), {k: v.numpy() for k, v in params.items()},
                           {k: v.numpy() for k, v in params0.items()}, {k: v.numpy() for k, v in grads.items()},
                           {k: v.numpy() for k, v in grads0.items()})
    # benchmark-sized step through the same protocol (the stall of the gloo rehearsal appeared at these sizes only)
    from unet_amd.models import UNet
    from unet_amd.train import TrainEngine
    from unet_amd.synth import make_batch
    torch.manual_seed(0)
    model = UNet(in_ch=1, out_ch=8, base=32, P=5, depth=3).cuda().train()
    bt = {k: v.cuda() for k, v in make_batch(64, 128, 64, seed=1234).items()}
    eng = TrainEngine(model, dtype=torch.bfloat16, lr=3e-4, clip_grad=1.0, loss_opts=dict(OPTS, multiscale=0))
    for _ in range(4):
        o = eng.train_step(bt["x"], bt["y"], bt["mask"], bt["params"], lam_grad=0.1)
    torch.cuda.synchronize()
    out["big_loss"] = float(o[0])
    q.put(out)
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_async_dp_protocol():
    import queue
    sys.path.insert(0, ROOT)
    from helpers import rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    p = ctx.Process(target=_nccl_worker, args=(port, q))
    p.start()
    try:
        out = q.get(timeout=100)
    except queue.Empty:
        p.terminate()
        p.join(timeout=20)
        try:
            stack = open(os.path.join(ROOT, "gpurun_out", "dist_nccl_worker.txt")).read()[-4000:]
        except OSError:
            stack = ""
        pytest.fail("the one-rank RCCL worker did not finish in 100 s (exit code %s)\n%s" % (p.exitcode, stack))
    p.join(timeout=60)
    assert p.exitcode == 0
    assert np.isfinite(out["big_loss"])
    for dtype_name, tol in (("float32", 2e-4), ("bfloat16", 3e-2)):
        losses, losses0, params, params0, grads, grads0 = out[dtype_name]
        # identical arithmetic (one rank: the all-reduce is the identity); only the fp32 atomic order of the GroupNorm
        # statistics differs between two runs, which Adam's first steps (update ~ lr * sign(g)) amplify: the first loss and
        # the first gradients agree to rounding, the three-step trajectory to the tolerances of the two-rank test above
        assert abs(losses[0] - losses0[0]) < (1e-5 if dtype_name == "float32" else 1e-3) * abs(losses0[0]), (losses, losses0)
        assert np.allclose(losses, losses0, rtol=tol, atol=0), (losses, losses0)
        for n in params:
            assert rel_err(grads[n], grads0[n]) < (1e-5 if dtype_name == "float32" else 2e-2), (n, "grad")
            assert rel_err(params[n], params0[n]) < tol, n


# ---------------------------------------------------------------------------------------------------------------------
# Bucket membership: the engine all-reduces flat_g[:n_dec] right after the decoder half's backward and
# flat_g[n_dec:n_bot] after the bottleneck part of the encoder half, while the rest of the backward keeps writing the arena.
# A parameter filed in the wrong bucket would be a silent gradient race on real multi-GPU runs only -- so: poison the arena,
# run the halves with the engine's flags, and check which slots each call wrote.
@pytest.mark.parametrize("depth", [2, 3])
@pytest.mark.parametrize("film", [True, False])
def test_gradient_buckets_are_written_by_the_backward_half_that_owns_them(depth, film):
    import numpy as np
    from oracle.unet_ref import UNetCfg, init_state_dict
    from unet_amd.models import UNet
    from unet_amd.train import TrainEngine
    from unet_amd.synth import make_batch
    from unet_amd import _lib as L
    P = 5 if film else 0
    cfg = UNetCfg(in_ch=1, out_ch=8, base=8, P=P, film_hidden=16, depth=depth)
    m = UNet(in_ch=1, out_ch=8, base=8, P=P, film_hidden=16, depth=depth)
    m.load_state_dict(init_state_dict(cfg, seed=2))
    eng = TrainEngine(m.cuda().train(), dtype=torch.float32, lr=1e-3, clip_grad=1.0, loss_opts=dict(OPTS, multiscale=0))
    bt = {k: v.cuda() for k, v in make_batch(4, 32, 16, P=P, seed=8, jitter_masks=True).items()}
    x, fp = eng._inputs(bt["x"], bt["params"] if film else None)
    sess, bufs = eng._session(4, 32, 16, fp is not None)
    begun = eng._loss_begin(sess, bufs, bt["mask"], 0.1)
    sess.pack_weights()
    sess.encode(x, fp)
    sess.decode(fp, None)
    eng._loss_end(sess, bufs, bt["y"], begun, 0.1, True, 1.0)
    flags = L.SX_BWD_WEIGHT_GRADS | L.SX_BWD_DEFER_FILM | L.SX_BWD_NO_DPARAMS

    def slot_mask(lo, hi):
        """elements of [lo, hi) that belong to a parameter (slots are padded to multiples of four)"""
        mk = torch.zeros(eng.flat_g.numel(), dtype=torch.bool)
        for n, (o, k) in eng.slots.items():
            mk[o:o + k] = True
        mk[:lo] = False
        mk[hi:] = False
        return mk
    n_all = eng.flat_g.numel()
    assert 0 < eng.n_dec < eng.n_bot < n_all
    eng.flat_g.fill_(float("nan"))
    sess.decode_backward(fp, None, flags)
    torch.cuda.synchronize()
    g1 = eng.flat_g.cpu().clone()
    assert torch.isfinite(g1[slot_mask(0, eng.n_dec)]).all(), "decoder bucket incomplete after sx_unet_decode_backward"
    assert torch.isnan(g1[eng.n_dec:]).all(), "sx_unet_decode_backward wrote outside its bucket"
    sess.encode_backward(fp, flags | L.SX_BWD_ONLY_BOTTLENECK)
    torch.cuda.synchronize()
    g2 = eng.flat_g.cpu().clone()
    assert torch.isfinite(g2[slot_mask(eng.n_dec, eng.n_bot)]).all(), "bottleneck bucket incomplete after ONLY_BOTTLENECK"
    assert torch.isnan(g2[eng.n_bot:]).all(), "ONLY_BOTTLENECK wrote into the last bucket"
    assert torch.equal(g2[:eng.n_dec].nan_to_num(7.0), g1[:eng.n_dec].nan_to_num(7.0)), "ONLY_BOTTLENECK touched the decoder bucket"
    sess.encode_backward(fp, flags | L.SX_BWD_AFTER_BOTTLENECK)
    torch.cuda.synchronize()
    g3 = eng.flat_g.cpu().clone()
    assert torch.isfinite(g3[slot_mask(0, n_all)]).all()
    assert torch.equal(g3[:eng.n_bot].nan_to_num(7.0), g2[:eng.n_bot].nan_to_num(7.0)), "AFTER_BOTTLENECK touched an earlier bucket"
