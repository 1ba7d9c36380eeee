"""Trial-level parallelism for hyper-parameter / ablation sweeps.

Reference: scripts/run_ablation_sweep.py:257-271 runs its experiments one after the other on one device; each is a small
model (base 16-64, batch 4-16) that cannot fill an MI355X, let alone eight.  The trials are independent, so the natural
partition is one trial per GPU at a time: every rank of a `torch.distributed.run` job takes a share of the trial list,
trains each of its trials as a single-rank job (`dist.local_mode()`: no gradient all-reduce, its own checkpoint file) and
the result rows are gathered on every rank at the end -- no data-path collective, like the batched inverse.
"""
from __future__ import annotations

import torch.distributed as tdist

from . import dist as sxdist


def assign(n_trials, world, costs=None):
    """Trial indices per rank.  Without costs: round robin.  With costs (e.g. parameter count x epochs): longest
    processing time first onto the least loaded rank, so one large trial does not serialise the sweep."""
    shares = [[] for _ in range(world)]
    if costs is None:
        for i in range(n_trials):
            shares[i % world].append(i)
        return shares
    if len(costs) != n_trials:
        raise ValueError(f"costs has {len(costs)} entries for {n_trials} trials")
    load = [0.0] * world
    for i in sorted(range(n_trials), key=lambda k: (-float(costs[k]), k)):
        r = min(range(world), key=lambda k: (load[k], k))
        shares[r].append(i)
        load[r] += float(costs[i])
    return shares


def run_trials(trials, run_fn, costs=None, on_result=None):
    """results[i] = run_fn(trials[i]) for every trial, each executed by exactly one rank.

    A trial that raises yields {"error": str(exception)} (the reference's sweep records failures and goes on,
    run_ablation_sweep.py:264-268).  `on_result(i, result)` is called on the executing rank after each trial (incremental
    CSV writes).  Every rank returns the complete list, in trial order.  Results must be picklable."""
    real_world = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
    real_rank = tdist.get_rank() if real_world > 1 else 0
    mine = assign(len(trials), real_world, costs)[real_rank]
    local = {}
    for i in mine:
        with sxdist.local_mode():
            try:
                local[i] = run_fn(trials[i])
            except Exception as e:                      # noqa: BLE001  (a failed trial must not end the sweep)
                local[i] = {"error": str(e)}
        if on_result is not None:
            on_result(i, local[i])
    if real_world > 1:
        parts = [None] * real_world
        tdist.all_gather_object(parts, local)
        merged = {}
        for part in parts:
            merged.update(part)
    else:
        merged = local
    return [merged[i] for i in range(len(trials))]
