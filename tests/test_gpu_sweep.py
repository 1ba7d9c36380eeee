"""GPU: trial-level parallelism for sweeps (reference: scripts/run_ablation_sweep.py:257-271 runs its
experiments one after the other).  Two ranks share the one card under gloo; `unet_amd.sweep.run_trials` hands each rank a
share of the trial list, every trial is a REAL `train_unet` run on the HIP path as a single-rank job (`dist.local_mode()`:
no gradient all-reduce although a process group is up, its own checkpoint), and the rows gathered on both ranks must equal
the rows of the same trials run one after the other in a single process -- i.e. sharding the sweep changes nothing but
who runs what.  A trial that raises yields an error row and does not end the sweep."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRIALS = [dict(name="base8_lr2e-3", base=8, lr=2e-3, amp=False),
          dict(name="base16_lr1e-3", base=16, lr=1e-3, amp=False),
          dict(name="base8_bf16", base=8, lr=2e-3, amp=True),
          dict(name="bad_base", base=12, lr=1e-3, amp=False),            # GroupNorm(8, 12): the reference's own ValueError
          dict(name="base8_ev", base=8, lr=1e-3, amp=False, lam_ev=0.05)]
COSTS = [1, 4, 1, 1, 1]


def _run_trial(tr, tmpdir):
    """One sweep experiment: two epochs of train_unet on a small synthetic split; returns a picklable result row."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_train import _ListDS, _norm
    from unet_amd.train import train_unet
    from unet_amd.synth import make_batch, CHANNEL_WEIGHTS_8
    torch.manual_seed(1234)                                   # the reference seeds every experiment the same way
    trn = make_batch(6, 16, 8, seed=100, jitter_masks=True)
    val = make_batch(4, 16, 8, seed=101, jitter_masks=True)
    tl = torch.utils.data.DataLoader(_ListDS(trn), batch_size=2, shuffle=False)
    vl = torch.utils.data.DataLoader(_ListDS(val), batch_size=2, shuffle=False)
    path = os.path.join(tmpdir, tr["name"] + ".pt")
    model, hist = train_unet(tl, vl, _norm(), 1, "cuda", out_ch=8, lam_grad=0.1, lam_w=0.5, epochs=2, base=tr["base"], P=5,
                             film_hidden=16, depth=2, save_path=path, multiscale=1, lr_init=tr["lr"], amp=tr["amp"],
                             lam_ev=tr.get("lam_ev", 0.0), channel_weights=CHANNEL_WEIGHTS_8,
                             param_scaler=(np.zeros(5, np.float32), np.ones(5, np.float32)))
    return {"name": tr["name"], "tr_mse": [float(v) for v in hist["tr_mse"]], "va_mse": [float(v) for v in hist["va_mse"]],
            "saved": os.path.exists(path), "n_params": sum(p.numel() for p in model.parameters())}


def _worker(rank, world, port, tmpdir, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fh = open(os.path.join(ROOT, "gpurun_out", f"sweep_worker{rank}.txt"), "w")
    faulthandler.dump_traceback_later(90, file=fh)
    from unet_amd import dist as D, sweep
    D.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    ran = []
    rows = sweep.run_trials(TRIALS, lambda tr: _run_trial(tr, os.path.join(tmpdir, f"rank{rank}")), costs=COSTS,
                            on_result=lambda i, r: ran.append(i))
    q.put((rank, rows, ran))
    faulthandler.cancel_dump_traceback_later()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sweep_trials_sharded_over_two_ranks_equal_the_serial_sweep(tmp_path):
    sys.path.insert(0, ROOT)
    from unet_amd import sweep
    for r in range(2):
        os.makedirs(tmp_path / f"rank{r}")
    os.makedirs(tmp_path / "serial")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 100
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    try:
        res = sorted([q.get(timeout=150) for _ in range(2)], key=lambda t: t[0])
    except queue.Empty:
        for p in procs:
            p.terminate()
        stacks = ""
        for r in range(2):
            try:
                stacks += f"\n--- rank {r}\n" + open(os.path.join(ROOT, "gpurun_out", f"sweep_worker{r}.txt")).read()[-3000:]
            except OSError:
                pass
        pytest.fail("the sharded sweep did not finish in 150 s" + stacks)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # who ran what: the expensive trial alone on one rank, everything else on the other (longest first, least loaded)
    shares = sweep.assign(len(TRIALS), 2, COSTS)
    assert [sorted(res[r][2]) for r in range(2)] == [sorted(s) for s in shares]
    assert sorted(shares[0] + shares[1]) == list(range(len(TRIALS)))
    # every rank holds every row, identical
    assert res[0][1] == res[1][1]
    rows = res[0][1]
    # the same experiments one after the other in this process (no process group at all)
    serial = sweep.run_trials(TRIALS, lambda tr: _run_trial(tr, str(tmp_path / "serial")))
    for tr, a, b in zip(TRIALS, rows, serial):
        if tr["name"] == "bad_base":
            assert "error" in a and "error" in b and a["error"] == b["error"] and "divisible" in a["error"]
            continue
        assert a["saved"] and b["saved"] and a["n_params"] == b["n_params"]
        tol = 3e-2 if tr["amp"] else 2e-4           # fp32 atomics' summation order; bf16 rounding of the same
        np.testing.assert_allclose(a["tr_mse"], b["tr_mse"], rtol=tol, err_msg=tr["name"])
        np.testing.assert_allclose(a["va_mse"], b["va_mse"], rtol=tol, err_msg=tr["name"])
        assert np.isfinite(a["tr_mse"]).all() and a["tr_mse"][1] < a["tr_mse"][0] * 1.5
    # different experiments really are different runs
    assert rows[0]["tr_mse"] != rows[1]["tr_mse"] and rows[0]["n_params"] != rows[1]["n_params"]
