"""MI355X-native mirror of `solpex.train` (reference: solpex/train.py).

`train_unet` keeps the reference signature, history keys, checkpoint dictionary and scheduling logic
(lam_grad ramp, ReduceLROnPlateau, best-val checkpointing, early stop).  The per-batch work runs
through `TrainEngine`: one persistent HIP session per batch shape, forward + loss + backward +
clip + AdamW as a handful of C-ABI calls, no autograd graph, no per-step host synchronisation
(the reference's three `.item()` calls per step become device-side accumulators read once per epoch).
Under `torch.distributed` (one process per GPU, RCCL) the engine all-reduces the loss denominators
(they are batch-global in the reference, solpex/losses.py:77-79) and the flat gradient arena in two
buckets, the first overlapped with the encoder's backward.
"""
from __future__ import annotations

import os
import numpy as np
import torch

from . import _lib as L
from ._engine import UNetSession
from .losses import edge_weights, LossBuffers, loss_prepare, loss_forward_backward, _prep_cw
from . import dist as sxdist


def lam_grad_schedule(epoch, lam_grad, warmup_start, warmup_end):
    """Linear ramp of the gradient-loss weight (reference: solpex/train.py:155-166)."""
    if lam_grad == 0.0 or epoch < int(warmup_start):
        return 0.0
    if epoch >= int(warmup_end):
        return lam_grad
    span = max(int(warmup_end) - int(warmup_start), 1)
    frac = min(max(float(epoch - int(warmup_start)) / float(span), 0.0), 1.0)
    return float(lam_grad) * frac


def flatten_parameters(model, order=None):
    """Move every parameter of `model` into one flat fp32 arena (16-byte aligned slots) and make
    `.grad` a view of a second arena.  Returns (flat_p, flat_g, slots) with slots[name] = (offset, numel)."""
    named = list(model.named_parameters())
    if order is not None:
        rank = {n: i for i, n in enumerate(order)}
        named.sort(key=lambda kv: rank.get(kv[0], len(rank)))
    dev = named[0][1].device
    slots, off = {}, 0
    for n, p in named:
        slots[n] = (off, p.numel())
        off += (p.numel() + 3) // 4 * 4
    flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
    flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for n, p in named:
            o, k = slots[n]
            flat_p[o:o + k].copy_(p.detach().reshape(-1).float())
            p.data = flat_p[o:o + k].view(p.shape)
            p.grad = flat_g[o:o + k].view(p.shape)
    return flat_p, flat_g, slots


class TrainEngine:
    """Fused training / evaluation steps of a `unet_amd.models.UNet` on one GPU (or one DP rank)."""

    def __init__(self, model, *, dtype=torch.float32, lr=1e-3, weight_decay=1e-4, clip_grad=1.0, betas=(0.9, 0.999),
                 eps=1e-8, channel_weights=None, w_map=None, loss_opts=None, lam_ev=0.0, norm=None):
        from .models import UNet
        if not isinstance(model, UNet):
            raise TypeError("TrainEngine drives unet_amd.models.UNet instances")
        self.model = model
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("TrainEngine: the model must live on a CUDA (ROCm) device; this build has no CPU path")
        self.sx_dtype = L.SX_BF16 if dtype == torch.bfloat16 else L.SX_F32
        self.lr, self.wd, self.clip, self.betas, self.eps = float(lr), float(weight_decay), clip_grad, betas, float(eps)
        self.cfg = model._cfg
        names = model._names
        # arena order: decoder half first (its bucket is all-reduced while the encoder's backward runs).  The decoder's
        # FiLM generators and GroupNorm affine parameters belong to the second bucket: their gradients come out of
        # sx_unet_encode_backward
        # (SX_BWD_DEFER_FILM: one FiLM backward launch pair per step)
        def late(n):       # decoder-half parameters whose gradients SX_BWD_DEFER_FILM finishes in the encoder half
            return (n.startswith("film_dec.") or (n.startswith("decoders.") and (".net.1." in n or ".net.4." in n))
                    or n == "out.bias" or (n.startswith("up_convs.") and n.endswith(".bias")))
        # third bucket boundary: the bottleneck's convolution weights (40 % of the parameters) are complete after the first
        # part of the encoder half (SX_BWD_ONLY_BOTTLENECK); their all-reduce overlaps the encoder levels' backward
        def botw(n):
            return n in ("bot.net.0.weight", "bot.net.3.weight", "bot.shortcut.weight")
        const = model._const_tensors()           # table entries that are no parameters (identity shortcut): not in the arena
        dec = [names[i] for i in model._dec_idx if not late(names[i])]
        bot = [names[i] for i in model._enc_idx if botw(names[i])]
        enc = ([names[i] for i in model._dec_idx if late(names[i])] +
               [names[i] for i in model._enc_idx if not botw(names[i]) and names[i] not in const])
        # arena tail: every FiLM generator (gradients exist only when the step is conditioned), then parameters the kernels
        # never produce gradients for (z_proj / z_unproj).  The reference's optimizer skips parameters whose grad is None:
        # the clip norm and AdamW run over the arena prefix that got gradients this step (no decay of the rest)
        film = [n for n in enc if n.startswith("film_")]
        enc = [n for n in enc if not n.startswith("film_")] + film
        self.flat_p, self.flat_g, self.slots = flatten_parameters(model, order=dec + bot + enc)
        model._wgen += 1
        self.n_dec = self.slots[(bot + enc)[0]][0] if (bot + enc) else self.flat_p.numel()
        self.n_bot = self.slots[enc[0]][0] if enc else self.flat_p.numel()
        owned = set(dec + bot + enc)
        rest = [n for n, _ in model.named_parameters() if n not in owned]
        self.n_kernel = self.slots[rest[0]][0] if rest else self.flat_p.numel()       # end of the kernel-owned parameters
        self.n_film = self.slots[film[0]][0] if film else self.n_kernel               # start of the FiLM generators
        self._film_pending = False
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.step_count = 0
        self.film_steps = 0        # optimizer steps that included the FiLM generators (torch's AdamW counts steps per parameter)
        self._ss = torch.zeros(1, dtype=torch.float32, device=self.device)
        named = dict(model.named_parameters())
        self._const_grads = {n: torch.zeros_like(t) for n, t in const.items()}      # scratch the kernels may write
        self.params = [named[n] if n in named else const[n] for n in names]
        self.grads = [named[n].grad if n in named else self._const_grads[n] for n in names]
        self.cw = _prep_cw(channel_weights, self.cfg["out_ch"], self.device)
        self.w_map = None if w_map is None else w_map.to(self.device, torch.float32).reshape(w_map.shape[-2], w_map.shape[-1]).contiguous()
        self.loss_opts = dict(lam_grad=0.2, lam_w=1.0, base="huber", huber_beta=0.05, grad_mode="vector", grad_base="l1",
                              multiscale=0, ms_weight=0.5, grad_ds=1)
        if loss_opts:
            self.loss_opts.update(loss_opts)
        # optional physical-space L1 term of the training loss (reference: solpex/train.py:200-205)
        self.lam_ev = float(lam_ev or 0.0)
        self._head_bias_ready = False
        self._ev_table = None
        if self.lam_ev > 0:
            if norm is None:
                raise ValueError("TrainEngine: lam_ev > 0 needs the output normalizer (`norm`)")
            from .data import norm_table
            self._ev_table = norm_table(norm, self.cfg["out_ch"])
        self._sessions = {}
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.sumsq_parts = torch.zeros(256, dtype=torch.float32, device=self.device)     # SX_SUMSQ_PARTS: this engine's norm partials
        self.accum = 0
        # device-side per-batch log rows [loss, sum|p-t|*m, px (global under DP), px (local)], read once per epoch
        self.log = torch.zeros(4096, 4, dtype=torch.float32, device=self.device)
        self.nlog = 0
        self.world = sxdist.world_size()
        self.dp = sxdist.is_active()          # world > 1 (or SX_DP_FORCE with a one-rank group)

    # ------------------------------------------------------------------ sessions
    def _session(self, B, H, W, use_film):
        key = (B, H, W, bool(use_film))
        ent = self._sessions.get(key)
        if ent is None:
            any_sess = next(iter(self._sessions.values()))[0] if self._sessions else None
            sess = UNetSession(self.cfg, B, H, W, self.sx_dtype, use_film, self.device,
                               wpack=None if any_sess is None else any_sess.wpack, film_shared=True)
            sess.set_params(self.params)
            sess.set_grads(self.grads)
            o = self.loss_opts
            bufs = LossBuffers(B, self.cfg["out_ch"], H, W, 2, max(int(o["grad_ds"]), 1), self.device)
            ent = (sess, bufs)
            self._sessions[key] = ent
        return ent

    def _inputs(self, x, params):
        if not x.is_cuda:
            raise RuntimeError("TrainEngine: batches must be CUDA (ROCm) tensors")
        x = x if x.dtype == torch.float32 else x.float()
        fp = None
        if params is not None and self.cfg["P"] > 0:
            fp = params.detach().float().contiguous()
        return x, fp

    def _drop_table(self, sess):
        m = self.model
        if not (m.training and m.dropout and m.dropout > 0):
            return None, None
        keep = 1.0 - float(m.dropout)
        masks = []
        for j in range(m.depth):
            c = self.cfg["base"] * (2 ** (m.depth - 1 - j))
            masks.append(torch.bernoulli(torch.full((sess.B, c), keep, device=self.device, dtype=torch.float32)) / keep)
        return masks, L.ptr_table(masks)

    def _loss_begin(self, sess, bufs, mask, lam_grad):
        """Phase 1 of the loss: the denominators depend on the masks only, so they are formed -- and under DP all-reduced,
        asynchronously -- BEFORE the network's forward pass; the collective's latency hides behind it."""
        o = self.loss_opts
        B, H, W, Cc = sess.B, sess.H, sess.W, self.cfg["out_ch"]
        m = mask if (mask.dtype == torch.float32 and mask.is_contiguous()) else mask.float().contiguous()
        loss_prepare(bufs, B, Cc, H, W, m, self.w_map, o["multiscale"], o["grad_ds"], float(lam_grad))
        px_local, handle = None, None
        if self.dp:
            # (single scale: phase 1 keeps this rank's own pixel count in the loss workspace and phase 2 reports it as out[3];
            # the multi-scale path has no such slot: copy den[0] aside before the all-reduce)
            px_local = bufs.den[0].clone() if int(o["multiscale"]) != 0 else None
            handle = sxdist.all_reduce_sum(bufs.den, async_op=True)        # batch-global denominators
        return m, px_local, handle

    def _loss_end(self, sess, bufs, y, begun, lam_grad, want_grad, grad_scale):
        o = dict(self.loss_opts)
        o["lam_grad"] = float(lam_grad)
        m, px_local, handle = begun
        B, H, W, Cc = sess.B, sess.H, sess.W, self.cfg["out_ch"]
        tgt = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous()
        if handle is not None:
            handle.wait()
        pred = sess.view(L.SX_T_PRED)
        dpred = sess.view(L.SX_T_DPRED) if want_grad else None
        ld = pred.stride(3)
        if self.nlog >= self.log.shape[0]:
            self.log = torch.cat([self.log, torch.zeros_like(self.log)])
        row = self.log[self.nlog]                       # [loss, sum|p-t|*m, sum m (global), sum m (local)]
        # the head's bias gradient rides on the loss's gradient kernel when this step's encode cleared its slots (FiLM on) and
        # nothing adds to d(pred) afterwards (lam_ev)
        bias_sums = sess.head_bias_sums() if (want_grad and sess.use_film and not self.lam_ev > 0) else None
        self._head_bias_ready = loss_forward_backward(
            bufs, self.sx_dtype, B, Cc, H, W, pred, ld, tgt, m, self.w_map, self.cw, o["base"],
            o["huber_beta"], o["lam_w"], o["lam_grad"], o["grad_mode"], o["grad_base"], o["multiscale"],
            o["ms_weight"], o["grad_ds"], dpred, ld, grad_scale, out=row, bias_sums=bias_sums)
        if self.lam_ev > 0 and want_grad:
            # training only (the reference's validation loss has no lam_ev term, train.py:262-276); den[0] = sum(mask), global under DP
            L.check(L.lib().sx_loss_ev_l1(self.sx_dtype, B, Cc, H, W, L.ptr(pred), ld, L.ptr(tgt), L.ptr(m), *self._ev_table.c_args(),
                                          L.ptr(bufs.den), self.lam_ev, L.ptr(row), L.ptr(dpred), ld, float(grad_scale),
                                          L.stream_ptr()), "sx_loss_ev_l1")
        if px_local is not None:
            row[3:4].copy_(px_local.reshape(1))
        self.nlog += 1
        return row[:3]

    # ------------------------------------------------------------------ steps
    def train_step(self, x, y, mask, params=None, *, lam_grad=None, grad_accum_steps=1, lr=None, step_now=True):
        """Forward + loss + backward (+ optimizer step when the accumulation window closes).
        Returns the device tensor [loss, sum|p-t|*m, sum m] of this batch (no host sync)."""
        x, fp = self._inputs(x, params)
        B, _, H, W = x.shape
        sess, bufs = self._session(B, H, W, fp is not None)
        lg = self.loss_opts["lam_grad"] if lam_grad is None else lam_grad
        begun = self._loss_begin(sess, bufs, mask, lg)
        masks, drop_table = self._drop_table(sess)
        sess.pack_encode(x, fp)          # the weights changed in the last optimizer step: repack, in the encode head's launch
        sess.decode(fp, drop_table)
        out = self._loss_end(sess, bufs, y, begun, lg, True, 1.0 / max(int(grad_accum_steps), 1))
        # (sessions are film_shared: with FiLM on, this step's encode cleared the backward sums with the statistics)
        flags = (L.SX_BWD_WEIGHT_GRADS | L.SX_BWD_DEFER_FILM | L.SX_BWD_NO_DPARAMS | L.SX_BWD_SUMS_CLEARED |
                 (L.SX_BWD_ACCUMULATE if self.accum > 0 else 0))
        sess.decode_backward(fp, drop_table, flags | (L.SX_BWD_HEAD_BIAS_READY if self._head_bias_ready else 0))
        h_dec = None
        closing = step_now and (self.accum + 1 >= max(int(grad_accum_steps), 1))
        if self.dp and closing:
            h_dec = sxdist.all_reduce_sum(self.flat_g[:self.n_dec], async_op=True)
        h_bot = None
        if self.dp and closing and self.n_bot > self.n_dec and not os.environ.get("SX_DP_NO_SPLIT"):
            sess.encode_backward(fp, flags | L.SX_BWD_ONLY_BOTTLENECK)
            h_bot = sxdist.all_reduce_sum(self.flat_g[self.n_dec:self.n_bot], async_op=True)
            sess.encode_backward(fp, flags | L.SX_BWD_AFTER_BOTTLENECK)
        else:
            sess.encode_backward(fp, flags)
        self.accum += 1
        if closing:
            if self.dp:
                h_enc = sxdist.all_reduce_sum(self.flat_g[(self.n_bot if h_bot is not None else self.n_dec):], async_op=True)
                # one communicator stream executes the collectives in issue order: the last bucket's completion implies the
                # earlier ones' (one stream hand-over instead of three)
                h_enc.wait()
                del h_dec, h_bot
            self.optimizer_step(lr, n=self.n_kernel if (fp is not None or self._film_pending) else self.n_film)
            self._film_pending = False
        elif fp is not None:
            self._film_pending = True            # an accumulation window with at least one conditioned step has FiLM gradients
        return out

    def optimizer_step(self, lr=None, n=None):
        """clip_grad_norm_ + AdamW over the flat arena (reference: solpex/train.py:226-233); `n`: arena prefix that holds
        gradients (default: every kernel-owned parameter)."""
        lib = L.lib()
        self.step_count += 1
        n = self.n_kernel if n is None else int(n)
        use_clip = bool(self.clip and self.clip > 0)
        lr_now = float(self.lr if lr is None else lr)
        with_film = n > self.n_film
        if with_film:
            self.film_steps += 1
        if with_film and self.film_steps != self.step_count:
            # The generators joined late (unconditioned steps came first): torch.optim.AdamW keeps a step count per parameter,
            # so their bias correction runs on their own count.  One norm over everything, two AdamW launches.
            gsq = None
            if use_clip:
                self._ss.zero_()
                L.check(lib.sx_grad_sumsq(L.ptr(self.flat_g), n, L.ptr(self._ss), L.ptr(self.sumsq_parts), L.stream_ptr()), "sx_grad_sumsq")
                gsq = L.ptr(self._ss)
            for lo, hi, stp in ((0, self.n_film, self.step_count), (self.n_film, n, self.film_steps)):
                L.check(lib.sx_adamw_step(L.ptr(self.flat_p[lo:hi]), L.ptr(self.flat_g[lo:hi]), L.ptr(self.flat_m[lo:hi]),
                                          L.ptr(self.flat_v[lo:hi]), hi - lo, lr_now, self.betas[0], self.betas[1], self.eps, self.wd,
                                          stp, float(self.clip) if use_clip else 0.0, gsq, 1.0,
                                          L.ptr(self.gnorm) if use_clip else None, None, L.stream_ptr()), "sx_adamw_step")
        elif use_clip:
            # deterministic block partials of sum g^2 (in this engine's own scratch); the AdamW blocks add them up themselves
            # (no one-block launch between)
            nparts = lib.sx_grad_sumsq_partial(L.ptr(self.flat_g), n, L.ptr(self.sumsq_parts), L.stream_ptr())
            L.check(min(nparts, 0), "sx_grad_sumsq_partial")
            L.check(lib.sx_adamw_step_fused(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.flat_m), L.ptr(self.flat_v), n, lr_now,
                                            self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, float(self.clip),
                                            L.ptr(self.sumsq_parts), nparts, 1.0, L.ptr(self.gnorm), L.stream_ptr()),
                    "sx_adamw_step_fused")
        else:
            L.check(lib.sx_adamw_step(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.flat_m), L.ptr(self.flat_v), n, lr_now,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 0.0, None, 1.0, None, None,
                                      L.stream_ptr()), "sx_adamw_step")
        self.accum = 0
        self.model._wgen += 1          # the kernel wrote the parameters through raw pointers: packed-weight caches are stale

    def flush(self, lr=None):
        """Step with whatever gradients are pending (epoch ended mid-accumulation, train.py:235-242)."""
        if self.accum > 0:
            if self.dp:
                sxdist.all_reduce_sum(self.flat_g)
            self.optimizer_step(lr, n=self.n_kernel if self._film_pending else self.n_film)
            self._film_pending = False

    @torch.no_grad()
    def eval_step(self, x, y, mask, params=None, *, lam_grad=None, norm=None, err_acc=None):
        """Forward + loss (no gradient).  If `norm` is given, adds batch_error_sums_ev into err_acc[0:3]."""
        x, fp = self._inputs(x, params)
        B, _, H, W = x.shape
        sess, bufs = self._session(B, H, W, fp is not None)
        lg = self.loss_opts["lam_grad"] if lam_grad is None else lam_grad
        begun = self._loss_begin(sess, bufs, mask, lg)
        sess.pack_weights()
        sess.encode(x, fp)
        sess.decode(fp, None)
        out = self._loss_end(sess, bufs, y, begun, lg, False, 1.0)
        if norm is not None and err_acc is not None:
            from .data import norm_table
            pred = sess.view(L.SX_T_PRED)
            tab = norm_table(norm, self.cfg["out_ch"])
            sb, sc, sh, sw = pred.stride()
            tgt = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous()
            m = mask if (mask.dtype == torch.float32 and mask.is_contiguous()) else mask.float().contiguous()
            L.check(L.lib().sx_masked_err_sums(self.sx_dtype, 1, B, self.cfg["out_ch"], H, W, L.ptr(pred), sb, sc, sh, sw,
                                               L.ptr(tgt), L.ptr(m), *tab.c_args(), L.ptr(err_acc), L.stream_ptr()),
                    "sx_masked_err_sums")
        return out

    def read_epoch(self):
        """One host sync per epoch: returns (sum loss*px, sum |p-t|*m, sum px, n_batches) accumulated in
        float64 on the host from the per-batch device log (as the reference does with `.item()` per batch),
        SUM-reduced over DP ranks; resets the log."""
        rows = self.log[:self.nlog].double().cpu()
        n = self.nlog
        self.nlog = 0
        vals = torch.tensor([float((rows[:, 0] * rows[:, 2]).sum()), float(rows[:, 1].sum()), float(rows[:, 3].sum())],
                            dtype=torch.float64)
        if self.dp:
            # rows[:,0] is this rank's share of the global loss and rows[:,2] the GLOBAL px: SUM-reducible
            v = vals.to(self.device)
            sxdist.all_reduce_sum(v)
            vals = v.cpu()
        return float(vals[0]), float(vals[1]), float(vals[2]), n

    # ------------------------------------------------------------------ optimizer state <-> torch.optim.AdamW
    def export_opt_state(self, opt):
        """Fill a torch.optim.AdamW's state with views of the engine's moments (checkpoint format parity)."""
        for n, p in self.model.named_parameters():
            o, k = self.slots[n]
            film = self.n_film <= o < self.n_kernel
            steps = self.film_steps if film else self.step_count
            if o >= self.n_kernel or steps == 0:      # never stepped (no gradient): the reference's optimizer holds no state for it
                continue
            opt.state[p] = {"step": torch.tensor(float(steps)),
                            "exp_avg": self.flat_m[o:o + k].view(p.shape), "exp_avg_sq": self.flat_v[o:o + k].view(p.shape)}

    def import_opt_state(self, opt):
        for n, p in self.model.named_parameters():
            st = opt.state.get(p)
            if not st:
                continue
            o, k = self.slots[n]
            self.flat_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            if self.n_film <= o < self.n_kernel:
                self.film_steps = int(float(st["step"]))
            else:
                self.step_count = int(float(st["step"]))


def train_unet(
    train_loader, val_loader, norm, in_ch, device, *,
    out_ch: int | None = None,
    inputs_mode="params",
    lam_grad=0.2, lam_w=1.0, lam_ev=0.0,
    epochs=10, base=32, param_scaler=None, model_cls=None,
    P: int = 0, film_hidden: int = 128, z_dim: int = 0, depth: int = 3, dropout: float = 0.0,
    save_path="unet_best.pt",
    return_history: bool = True,
    history_path=None,
    multiscale=1,
    grad_base="l1",
    ms_weight=0.5,
    resume_path: str | None = None,
    strict: bool = True,
    lr_init: float = 1e-3, lr_factor: float = 0.5, lr_patience: int = 5, lr_min: float = 1e-5,
    lr_threshold: float = 1e-4,
    amp: bool = True,
    grad_accum_steps: int = 1,
    non_blocking: bool = True,
    clip_grad: float = 1.0,
    channels_last: bool = True,
    cudnn_benchmark: bool = True,
    log_every: int = 1,
    early_stop_patience: int | None = None,
    early_stop_min_delta: float = 0.0,
    weight_decay: float = 1e-4,
    lam_grad_warmup_start: int = 20,
    lam_grad_warmup_end: int = 60,
    channel_weights=None,
    param_transform=None,
    param_keys=None,
):
    """Same arguments and return value as the reference (solpex/train.py:14-359).

    `amp=True` on a GPU selects the bf16 tier (bf16 activations, fp32 accumulation / master weights),
    the reference's autocast semantics; `channels_last` and `cudnn_benchmark` are accepted and have
    no effect (the kernels are NHWC throughout).  `lam_ev > 0` adds the physical-space masked L1 term (sx_loss_ev_l1).
    """
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("train_unet (MI355X build): device must be a CUDA/ROCm device; there is no CPU path")
    if out_ch is None:
        out_ch = 1
    if model_cls is None:
        from .models import UNet
        model_cls = lambda in_ch, out_ch: UNet(in_ch=in_ch, out_ch=out_ch, base=base, P=P, film_hidden=film_hidden,
                                               z_dim=z_dim, depth=depth, dropout=dropout)
    model = model_cls(in_ch, out_ch).to(device)

    cw_list = None
    if channel_weights is not None:
        cw_arr = np.asarray(channel_weights, dtype=np.float32).reshape(-1)
        if cw_arr.size != int(out_ch):
            raise ValueError(f"channel_weights length={cw_arr.size} but out_ch={out_ch}")
        cw_list = cw_arr

    # torch objects kept for their host-side logic and checkpoint format; the arithmetic of the step is ours
    opt = torch.optim.AdamW(model.parameters(), lr=lr_init, weight_decay=weight_decay)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=lr_factor, patience=lr_patience,
                                                           threshold=lr_threshold, threshold_mode="rel", cooldown=0,
                                                           min_lr=lr_min, eps=1e-8)
    use_amp = bool(amp)
    # bf16 needs no loss scaling: nothing here scales.  The object only provides the checkpoint's "scaler" entry in the form
    # the reference's resume path loads (train.py:117-118 calls scaler.load_state_dict when amp is on: an enabled scaler's dict)
    scaler = torch.amp.GradScaler("cuda", enabled=use_amp)

    start_epoch, best_val = 0, float("inf")
    ckpt_resume = None
    if resume_path is not None and os.path.exists(resume_path):
        ckpt_resume = torch.load(resume_path, map_location=device, weights_only=False)
        model.load_state_dict(ckpt_resume["model"], strict=strict)
        if "sched" in ckpt_resume:
            scheduler.load_state_dict(ckpt_resume["sched"])
        start_epoch = int(ckpt_resume.get("epoch", 0)) + 1
        best_val = float(ckpt_resume.get("best_val", best_val))
        print(f"[resume] loaded {resume_path} @ epoch={start_epoch} best_val={best_val:.6g}")

    m0 = train_loader.dataset[0]["mask"]
    w_map = torch.from_numpy(edge_weights(m0.numpy()[0], sigma_px=3.0))

    engine = TrainEngine(model, dtype=torch.bfloat16 if use_amp else torch.float32, lr=lr_init, weight_decay=weight_decay,
                         clip_grad=clip_grad, channel_weights=cw_list, w_map=w_map, lam_ev=lam_ev, norm=norm,
                         loss_opts=dict(lam_w=lam_w, base="huber", huber_beta=0.05, grad_mode="vector", grad_base=grad_base,
                                        multiscale=multiscale, ms_weight=ms_weight, grad_ds=2))
    if ckpt_resume is not None and "opt" in ckpt_resume:
        opt.load_state_dict(ckpt_resume["opt"])
        engine.import_opt_state(opt)

    if engine.dp:
        # every rank must run the same number of steps (each step all-reduces); resident loaders built before the process
        # group existed are re-sharded here, anything else is only checked
        for ld in (train_loader, val_loader):
            if hasattr(ld, "shard") and getattr(ld, "world", 1) != sxdist.world_size():
                ld.shard(sxdist.rank(), sxdist.world_size())
        sxdist.check_equal_across_ranks(len(train_loader), "the number of training batches per epoch")
        sxdist.check_equal_across_ranks(len(val_loader), "the number of validation batches per epoch")
    history = {"tr_mse": [], "tr_maeN": [], "va_mse": [], "va_maeN": [], "va_mae_eV": [], "va_rmse_eV": [], "lr": []}
    rank0 = sxdist.rank() == 0
    no_improve = 0
    err_acc = torch.zeros(3, dtype=torch.float32, device=device)
    for epoch in range(start_epoch, start_epoch + epochs):
        model.train()
        lam_grad_epoch = lam_grad_schedule(epoch, lam_grad, lam_grad_warmup_start, lam_grad_warmup_end)
        lr_now = opt.param_groups[0]["lr"]
        engine.accum = 0
        for b in train_loader:
            x = b["x"].to(device, non_blocking=non_blocking)
            y = b["y"].to(device, non_blocking=non_blocking)
            m = b["mask"].to(device, non_blocking=non_blocking)
            params = b["params"].to(device, non_blocking=non_blocking) if P > 0 else None
            engine.train_step(x, y, m, params, lam_grad=lam_grad_epoch, grad_accum_steps=grad_accum_steps, lr=lr_now)
        engine.flush(lr_now)
        s_loss, s_abs, s_px, _ = engine.read_epoch()
        tr_loss, tr_maeN = s_loss / max(s_px, 1.0), s_abs / max(s_px, 1.0)

        model.eval()
        err_acc.zero_()
        for b in val_loader:
            x = b["x"].to(device, non_blocking=non_blocking)
            y = b["y"].to(device, non_blocking=non_blocking)
            m = b["mask"].to(device, non_blocking=non_blocking)
            params = b["params"].to(device, non_blocking=non_blocking) if P > 0 else None
            engine.eval_step(x, y, m, params, lam_grad=lam_grad_epoch, norm=norm, err_acc=err_acc)
        v_loss, v_abs, v_px, _ = engine.read_epoch()
        if engine.dp:
            sxdist.all_reduce_sum(err_acc)
        ev_abs, ev_sq, _ = err_acc.tolist()
        va_loss, va_maeN = v_loss / max(v_px, 1.0), v_abs / max(v_px, 1.0)
        va_MAE_eV, va_RMSE_eV = ev_abs / max(v_px, 1.0), (ev_sq / max(v_px, 1.0)) ** 0.5

        history["tr_mse"].append(tr_loss)
        history["tr_maeN"].append(tr_maeN)
        history["va_mse"].append(va_loss)
        history["va_maeN"].append(va_maeN)
        history["va_mae_eV"].append(va_MAE_eV)
        history["va_rmse_eV"].append(va_RMSE_eV)
        history["lr"].append(opt.param_groups[0]["lr"])

        lr_before = opt.param_groups[0]["lr"]
        show = rank0 and (epoch % max(int(log_every), 1)) == 0
        if show:
            print(f"Epoch {epoch:03d} | train {tr_loss:.4f} | val {va_loss:.4f} | lr {lr_before:.2e} | lam_grad_eff {lam_grad_epoch:.3g}")
        scheduler.step(va_loss)
        lr_after = opt.param_groups[0]["lr"]
        if lr_after < lr_before and show:
            print(f"  ↳ ReduceLROnPlateau: lr {lr_before:.2e} → {lr_after:.2e}")

        if (best_val - va_loss) > float(early_stop_min_delta):
            best_val, no_improve = va_loss, 0
            if rank0:
                mu, std = param_scaler if param_scaler is not None else (None, None)
                engine.export_opt_state(opt)
                ckpt = {
                    "epoch": epoch, "model": {k: v.detach().clone() for k, v in model.state_dict().items()},
                    "opt": opt.state_dict(), "sched": scheduler.state_dict(), "scaler": scaler.state_dict(),
                    "best_val": float(best_val), "norm": _norm_to_ckpt(norm),
                    "in_ch": in_ch, "out_ch": out_ch, "base": base, "dropout": getattr(model, "dropout", 0.0),
                    "depth": getattr(model, "depth", 3), "P": P, "film_hidden": film_hidden, "z_dim": z_dim,
                    "inputs_mode": inputs_mode, "param_mu": mu, "param_std": std, "param_scaler": (mu, std),
                    "channel_weights": None if cw_list is None else [float(v) for v in cw_list],
                    "param_transform": param_transform, "param_keys": param_keys, "history": history,
                }
                torch.save(ckpt, save_path)
        else:
            no_improve += 1
        if early_stop_patience is not None and no_improve >= int(early_stop_patience):
            if rank0:
                print(f"[early-stop] no val improvement for {no_improve} epochs; stopping at epoch {epoch:03d}")
            break

    if history_path is not None and rank0:
        np.savez(history_path, **history)
    return (model, history) if return_history else model


def _norm_to_ckpt(norm):
    """Serialise a normalizer for the checkpoint (reference: solpex/train.py:361-376)."""
    if hasattr(norm, "y_keys") and hasattr(norm, "norms"):
        return {"kind": type(norm).__name__, "y_keys": list(norm.y_keys),
                "norms": {k: _norm_to_ckpt(n) for k, n in norm.norms.items()}}
    if all(hasattr(norm, a) for a in ("mu", "sigma", "eps")):
        pack = {"kind": type(norm).__name__, "mu": norm.mu, "sigma": norm.sigma, "eps": float(norm.eps)}
        if hasattr(norm, "scale"):
            pack["scale"] = float(norm.scale)
        return pack
    return {"kind": type(norm).__name__}
