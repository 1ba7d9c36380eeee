"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The reference is single-device; what DP must preserve is
its batch-GLOBAL loss denominators (solpex/losses.py:77-79):

    1. every rank computes its local raw denominators  sum(m), sum(m*w), sum(m_g)       (sx_loss_prepare)
    2. all-reduce(SUM) those <= 9 floats                                                (this module)
    3. every rank forms numerators / gradients with the GLOBAL denominators            (sx_loss_forward_backward)
    4. all-reduce(SUM) the flat gradient arena (no division by world size), then clip + AdamW, identical on
       every rank.
With identical masks on every sample this reduces to ordinary gradient averaging.
"""
from __future__ import annotations

import contextlib
import os
import torch
import torch.distributed as dist


_LOCAL = 0          # depth of local_mode(): inside it this process behaves as a single-rank job


@contextlib.contextmanager
def local_mode():
    """Run the enclosed code as if there were one rank: train_unet / TrainEngine / batched_inverse then neither split nor
    all-reduce anything.  For trial-level parallelism (unet_amd.sweep), where every rank trains its OWN model."""
    global _LOCAL
    _LOCAL += 1
    try:
        yield
    finally:
        _LOCAL -= 1


def _group_ready():
    return dist.is_available() and dist.is_initialized()


def _forced():
    # SX_DP_FORCE=1: run the data-parallel protocol (denominator all-reduce, bucketed asynchronous gradient all-reduces, split
    # encoder backward) even in a group of ONE rank.  A single-rank all-reduce is a valid RCCL collective, so a one-GPU box
    # exercises the real ProcessGroupNCCL path: Work handles, RCCL's own stream, wait() ordering (tests/test_gpu_dist.py).
    return os.environ.get("SX_DP_FORCE", "0") not in ("", "0")


def is_active():
    return not _LOCAL and _group_ready() and (dist.get_world_size() > 1 or _forced())


def world_size():
    return dist.get_world_size() if (_group_ready() and not _LOCAL) else 1


def rank():
    return dist.get_rank() if (_group_ready() and not _LOCAL) else 0


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    r = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(r)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        # SX_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a one-card box; RCCL wants a device per rank)
        backend = os.environ.get("SX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=r, world_size=world)
    return r, world, local


class _Done:
    def wait(self):
        return True


def all_reduce_sum(t, async_op=False):
    """In-place SUM all-reduce of a tensor (no-op when there is one rank)."""
    if not is_active():
        return _Done() if async_op else t
    if t.is_cuda and dist.get_backend() == "gloo":
        # gloo is the stand-in transport of the tests / one-card rehearsals.  Its asynchronous CUDA path (worker threads staging
        # through pinned memory while the caller keeps launching kernels) stalled with two ranks sharing one GPU (both ranks
        # stuck in wait(), reproducibly at the benchmark's sizes); stage through the host here, synchronously.
        torch.cuda.current_stream().synchronize()
        c = t.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t.copy_(c)
        return _Done() if async_op else t
    h = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
    return h if async_op else t


def broadcast0(t, device=None):
    """Rank 0's value of a small host tensor on every rank (returned on the host)."""
    if not is_active():
        return t
    if dist.get_backend() == "nccl":
        d = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(d, src=0)
        return d.cpu()
    c = t.clone()
    dist.broadcast(c, src=0)
    return c


def check_equal_across_ranks(value, what):
    """Raise on every rank when an integer that must agree (e.g. batches per epoch) differs between ranks: unequal loop
    lengths would otherwise hang in the first all-reduce one rank never reaches."""
    if not is_active():
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    lo = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo) != int(hi):
        raise RuntimeError(f"data parallel: {what} differs between ranks ({int(lo)}..{int(hi)}); shard the data with "
                           "make_loaders(rank=, world=) / ResidentLoader.shard() so that every rank runs the same number of steps")


def shard_range(n, r=None, w=None):
    """Contiguous [begin, end) share of n independent units for rank r of w (embarrassingly parallel paths)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per, rem = divmod(n, w)
    b = r * per + min(r, rem)
    return b, b + per + (1 if r < rem else 0)


def gather_rows(local, n_total):
    """All-gather row blocks of unequal length (shard_range layout) into an (n_total, ...) tensor on every rank."""
    if not is_active():
        return local
    w = world_size()
    sizes = [shard_range(n_total, r, w) for r in range(w)]
    mx = max(e - b for b, e in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:e - b] for o, (b, e) in zip(outs, sizes)], dim=0)


def global_loss_protocol(local_den, local_terms_fn):
    """The 4-step protocol above in host terms (used by the CPU gloo tests with the oracle as the
    compute): all-reduce the raw denominators, then let the caller build loss / gradients from them."""
    den = local_den.clone()
    all_reduce_sum(den)
    return local_terms_fn(den)
