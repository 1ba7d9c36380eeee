"""MI355X-native mirror of `solpex.latent` (reference: solpex/latent.py): the latent MLPs `ParamToZ` / `ZToParam`,
their three trainers and `extract_z_dataset`.

Same constructor signatures, `state_dict` keys (`net.{i}.weight/bias`), trainer arguments, printed lines, checkpoint
dicts and return values as the reference.  The arithmetic (Linear / SiLU / LayerNorm / Dropout forward and backward,
SmoothL1, clip + AdamW) runs in the HIP kernels of `csrc/mlp.hip` and `csrc/optim.hip` through the C ABI; torch is the
allocator, the RNG (batch order, dropout masks) and the container of the parameters.  No CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, TensorDataset

from . import _lib as L
from .models import bottleneck_to_z

SMOOTH_L1_BETA = 0.05          # latent.py:74,148,229


def _build(d_in, d_out, hidden, dropout, use_layernorm):
    layers, d = [], d_in
    for h in hidden:
        layers += [nn.Linear(d, h), nn.SiLU()]
        if use_layernorm:
            layers += [nn.LayerNorm(h)]
        if dropout > 0:
            layers += [nn.Dropout(dropout)]
        d = h
    layers += [nn.Linear(d, d_out)]
    return nn.Sequential(*layers)


class _Ctx:
    """Activations one forward pass saved for its backward."""
    __slots__ = ("x", "blocks", "out")


class _MLP(nn.Module):
    """Linear -> SiLU [-> LayerNorm] [-> Dropout] blocks + a final Linear, evaluated by the HIP kernels."""

    def _specs(self):
        specs, cur = [], None
        for m in self.net:
            if isinstance(m, nn.Linear):
                if cur is not None:
                    specs.append(cur)
                cur = dict(lin=m, act=False, ln=None, p=0.0)
            elif isinstance(m, nn.SiLU):
                cur["act"] = True
            elif isinstance(m, nn.LayerNorm):
                cur["ln"] = m
            elif isinstance(m, nn.Dropout):
                cur["p"] = float(m.p)
        specs.append(cur)
        return specs

    # ------------------------------------------------------------------ kernels
    def hip_forward(self, x, train=False):
        """x (B, d_in) fp32 CUDA -> _Ctx (ctx.out = (B, d_out))."""
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} (HIP): inputs must be CUDA (ROCm) tensors; this build has no CPU path")
        if x.dim() != 2:
            raise ValueError(f"{type(self).__name__} (HIP): expected a (B, features) tensor, got {tuple(x.shape)}")
        lib, st = L.lib(), L.stream_ptr()
        x = x.detach()
        x = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        ctx = _Ctx()
        ctx.x, ctx.blocks = x, []
        B, h = x.shape[0], x
        for s in self._specs():
            lin = s["lin"]
            N, K = lin.weight.shape
            if h.shape[1] != K:
                raise ValueError(f"{type(self).__name__} (HIP): expected {K} input features, got {h.shape[1]}")
            y = torch.empty(B, N, dtype=torch.float32, device=x.device)
            pre = torch.empty_like(y) if s["act"] else None
            L.check(lib.sx_gemm_f32(B, N, K, L.ptr(h), K, 1, L.ptr(lin.weight), K, 1, L.ptr(y), N, L.ptr(lin.bias),
                                    1 if s["act"] else 0, L.ptr(pre), N, None, 0, None, 0, None, 0, st), "sx_gemm_f32")
            blk = dict(inp=h, pre=pre, act=y, stats=None, mask=None)
            h = y
            if s["ln"] is not None:
                ln = s["ln"]
                stats = torch.empty(B, 2, dtype=torch.float32, device=x.device)
                o = torch.empty_like(h)
                L.check(lib.sx_layernorm_forward(B, N, L.ptr(h), L.ptr(ln.weight), L.ptr(ln.bias), float(ln.eps), L.ptr(o),
                                                 L.ptr(stats), st), "sx_layernorm_forward")
                blk["stats"], h = stats, o
            if train and s["p"] > 0:
                keep = 1.0 - s["p"]
                mask = torch.bernoulli(torch.full((B, N), keep, device=x.device, dtype=torch.float32)) / keep
                o = torch.empty_like(h)
                L.check(lib.sx_mul(B * N, L.ptr(h), L.ptr(mask), L.ptr(o), st), "sx_mul")
                blk["mask"], h = mask, o
            ctx.blocks.append(blk)
        ctx.out = h
        return ctx

    def hip_backward(self, ctx, dout, grads=None, need_dx=False):
        """Back-propagate dout (B, d_out).  `grads`: dict parameter -> fp32 buffer the gradient is ACCUMULATED into
        (None = input gradient only).  Returns dx (B, d_in) if need_dx."""
        lib, st = L.lib(), L.stream_ptr()
        specs = self._specs()
        B = ctx.x.shape[0]
        d = dout if (dout.dtype == torch.float32 and dout.is_contiguous()) else dout.float().contiguous()
        # d is the gradient of block i's Linear OUTPUT (pre-activation for hidden blocks) on entry to iteration i
        for i in range(len(specs) - 1, -1, -1):
            s, blk = specs[i], ctx.blocks[i]
            lin = s["lin"]
            N, K = lin.weight.shape
            if grads is not None:
                gw, gb = grads[lin.weight], grads[lin.bias]
                # dW[n][k] += sum_b d[b][n] x[b][k]; db[n] += sum_b d[b][n] (row sums of A = d^T)
                L.check(lib.sx_gemm_f32(N, K, B, L.ptr(d), 1, N, L.ptr(blk["inp"]), 1, K, L.ptr(gw), K, None, 0, None, 0,
                                        None, 0, None, 0, L.ptr(gb), 1, st), "sx_gemm_f32")
            if i == 0 and not need_dx:
                return None
            # dx[b][k] = sum_n d[b][n] W[n][k], then walk back through the previous block's Dropout / LayerNorm / SiLU
            prev = ctx.blocks[i - 1] if i > 0 else None
            ps = specs[i - 1] if i > 0 else None
            dx = torch.empty(B, K, dtype=torch.float32, device=d.device)
            mask = prev["mask"] if prev is not None else None
            fuse_act = prev is not None and ps["act"] and ps["ln"] is None
            L.check(lib.sx_gemm_f32(B, K, N, L.ptr(d), N, 1, L.ptr(lin.weight), 1, K, L.ptr(dx), K, None, 0, None, 0,
                                    L.ptr(mask), K, L.ptr(prev["pre"]) if fuse_act else None, K, None, 0, st), "sx_gemm_f32")
            if prev is not None and ps["ln"] is not None:
                ln = ps["ln"]
                d2 = torch.empty_like(dx)
                gg = grads[ln.weight] if grads is not None else None
                gbt = grads[ln.bias] if grads is not None else None
                L.check(lib.sx_layernorm_backward(B, K, L.ptr(prev["act"]), L.ptr(ln.weight), L.ptr(prev["stats"]), L.ptr(dx),
                                                  L.ptr(d2), L.ptr(gg), L.ptr(gbt), L.ptr(prev["pre"]) if ps["act"] else None, st),
                        "sx_layernorm_backward")
                dx = d2
            d = dx
        return d

    # ------------------------------------------------------------------ nn.Module API
    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        needs = torch.is_grad_enabled() and (x2.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs:
            out = _MLPFn.apply(self, x2, *self.parameters())
        else:
            out = self.hip_forward(x2, train=self.training).out
        return out.reshape(*lead, out.shape[-1])


class _MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        c = mod.hip_forward(x, train=mod.training)
        ctx.mod, ctx.c = mod, c
        return c.out.view(c.out.shape)

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        params = list(mod.parameters())
        want_w = any(ctx.needs_input_grad[2:])
        grads = {p: torch.zeros_like(p, dtype=torch.float32) for p in params} if want_w else None
        dx = mod.hip_backward(ctx.c, dout, grads, need_dx=ctx.needs_input_grad[1])
        outs = [grads[p] if (want_w and ctx.needs_input_grad[2 + i]) else None for i, p in enumerate(params)]
        return (None, dx, *outs)


class ParamToZ(_MLP):
    """params -> z (reference: solpex/latent.py:14-30)."""

    def __init__(self, P, latent_dim, hidden=(256, 256), dropout=0.0, use_layernorm=False):
        super().__init__()
        self.net = _build(P, latent_dim, hidden, dropout, use_layernorm)


class ZToParam(_MLP):
    """Amortized inverse z -> params (reference: solpex/latent.py:31-48)."""

    def __init__(self, z_dim, P, hidden=(256, 256), dropout=0.0, use_layernorm=False):
        super().__init__()
        self.net = _build(z_dim, P, hidden, dropout, use_layernorm)


# ---------------------------------------------------------------------------------------------- training engine
class _Trainer:
    """Flat fp32 arenas (parameters, gradients, Adam moments) over one or two MLPs; clip 1.0 + AdamW in two launches
    (sx_grad_sumsq + sx_adamw_step), SmoothL1 values written to a device log and read once per epoch."""

    def __init__(self, models, lr, wd, device):
        self.models, self.lr, self.wd, self.device = models, float(lr), float(wd), device
        params = [p for m in models for p in m.parameters()]
        off, slots = 0, []
        for p in params:
            slots.append((off, p.numel()))
            off += (p.numel() + 3) // 4 * 4
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.grads = {}
        with torch.no_grad():
            for p, (o, k) in zip(params, slots):
                self.flat_p[o:o + k].copy_(p.detach().reshape(-1).float())
                p.data = self.flat_p[o:o + k].view(p.shape)
                self.grads[p] = self.flat_g[o:o + k].view(p.shape)
        self.step_count = 0
        self._ss = torch.zeros(2, dtype=torch.float32, device=device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=device)
        self._parts = torch.zeros(256, dtype=torch.float32, device=device)          # SX_SUMSQ_PARTS: norm partials of this trainer
        self.log = torch.zeros(1024, 4, dtype=torch.float32, device=device)
        self.nlog = 0
        self.sizes = []

    def new_row(self, bs):
        if self.nlog >= self.log.shape[0]:
            self.log = torch.cat([self.log, torch.zeros_like(self.log)])
        self.sizes.append(int(bs))
        self.nlog += 1
        return self.log[self.nlog - 1]

    def loss(self, pred, tgt, out, dpred=None, weight=1.0, accumulate=False):
        tgt = tgt if (tgt.dtype == torch.float32 and tgt.is_contiguous()) else tgt.float().contiguous()
        L.check(L.lib().sx_smooth_l1(pred.numel(), L.ptr(pred), L.ptr(tgt), SMOOTH_L1_BETA, L.ptr(out), L.ptr(dpred),
                                     float(weight), 1 if accumulate else 0, L.stream_ptr()), "sx_smooth_l1")

    def step(self):
        lib = L.lib()
        self.step_count += 1
        n = self.flat_p.numel()
        cur, nxt = self._ss[self.step_count & 1:], self._ss[(self.step_count + 1) & 1:]
        L.check(lib.sx_grad_sumsq(L.ptr(self.flat_g), n, L.ptr(cur), L.ptr(self._parts), L.stream_ptr()), "sx_grad_sumsq")
        L.check(lib.sx_adamw_step(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.flat_m), L.ptr(self.flat_v), n, self.lr,
                                  0.9, 0.999, 1e-8, self.wd, self.step_count, 1.0, L.ptr(cur), 1.0, L.ptr(self.gnorm),
                                  L.ptr(nxt), L.stream_ptr()), "sx_adamw_step")

    def read(self):
        """(rows (n,4) float64, batch sizes) of the epoch so far; resets the log (one host sync)."""
        rows = self.log[:self.nlog].double().cpu().numpy()
        sizes = np.asarray(self.sizes, dtype=np.float64)
        self.log[:self.nlog].zero_()
        self.nlog, self.sizes = 0, []
        return rows, sizes


def _loaders(a_tr, b_tr, a_va, b_va, batch_size, num_workers):
    tr = DataLoader(TensorDataset(torch.from_numpy(a_tr), torch.from_numpy(b_tr)), batch_size=batch_size, shuffle=True,
                    num_workers=num_workers)
    va = DataLoader(TensorDataset(torch.from_numpy(a_va), torch.from_numpy(b_va)), batch_size=batch_size, shuffle=False,
                    num_workers=num_workers)
    return tr, va


def _check_device(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("solpex.latent (MI355X build): device must be a CUDA/ROCm device; there is no CPU path")
    return device


def _train_single(tag, model, a_tr, b_tr, a_va, b_va, device, lr, epochs, batch_size, wd, num_workers, save_path, extra):
    """Shared body of train_param2z / train_z2param (latent.py:51-199): model maps a -> b."""
    device = _check_device(device)
    model = model.to(device)
    tr, va = _loaders(a_tr, b_tr, a_va, b_va, batch_size, num_workers)
    eng = _Trainer([model], lr, wd, device)
    best, hist = float("inf"), {"tr": [], "va": []}
    for ep in range(epochs):
        model.train()
        for a, b in tr:
            a, b = a.to(device), b.to(device)
            ctx = model.hip_forward(a, train=True)
            dpred = torch.empty_like(ctx.out)
            eng.loss(ctx.out, b, eng.new_row(a.shape[0]), dpred)
            eng.flat_g.zero_()
            model.hip_backward(ctx, dpred, eng.grads)
            eng.step()
        rows, sizes = eng.read()
        tr_loss = float((rows[:, 0] * sizes).sum()) / max(float(sizes.sum()), 1.0)
        model.eval()
        for a, b in va:
            a, b = a.to(device), b.to(device)
            eng.loss(model.hip_forward(a).out, b, eng.new_row(a.shape[0]))
        rows, sizes = eng.read()
        va_loss = float((rows[:, 0] * sizes).sum()) / max(float(sizes.sum()), 1.0)
        hist["tr"].append(tr_loss)
        hist["va"].append(va_loss)
        if (ep % 10) == 0 or ep == epochs - 1:
            print(f"[{tag}] ep {ep:04d} tr {tr_loss:.5f} va {va_loss:.5f}")
        if va_loss < best:
            best = va_loss
            if save_path:
                torch.save({"model": {k: v.detach().clone() for k, v in model.state_dict().items()}, "best": best, **extra},
                           save_path)
    return model, hist


def train_param2z(*, P_train, Z_train, P_val, Z_val, model, device, lr=1e-3, epochs=200, batch_size=256, wd=1e-4,
                  num_workers=0, save_path=None):
    """Reference: solpex/latent.py:51-121."""
    return _train_single("param2z", model, P_train, Z_train, P_val, Z_val, device, lr, epochs, batch_size, wd, num_workers,
                         save_path, {})


def train_z2param(*, Z_train, P_train, Z_val, P_val, model, device, lr=1e-3, epochs=200, batch_size=256, wd=1e-4,
                  num_workers=0, save_path=None):
    """Train ZToParam inverse MLP: latent z -> scaled params (reference: solpex/latent.py:124-199)."""
    return _train_single("z2param", model, Z_train, P_train, Z_val, P_val, device, lr, epochs, batch_size, wd, num_workers,
                         save_path, {"z_dim": Z_train.shape[1], "P": P_train.shape[1]})


def _cycle_losses(eng, p2z, z2p, z, p, lam, train):
    """The four SmoothL1 terms of latent.py:243-251 into one log row [fwd, inv, cyc_pzp, cyc_zpz]; with `train`, the
    gradients of fwd + inv + lam * (cyc_pzp + cyc_zpz) accumulate into the arena.  Without dropout the reference's second
    evaluations of p2z(p) and z2p(z) inside the cycle terms equal the first ones, so each MLP runs twice, not three times.
    With dropout > 0 (training) they do not: the reference draws fresh masks for every call, so the inner p2z(p) / z2p(z) of
    the cycle terms are evaluated on their own, three runs per MLP, each back-propagated through its own masks.
    Returns the forward contexts (tests read the masks that were drawn)."""
    resample = train and (_has_dropout(p2z) or _has_dropout(z2p))
    row = eng.new_row(z.shape[0])
    c_fwd = z2p.hip_forward(z, train)                  # z2p(z)           vs p
    c_inv = p2z.hip_forward(p, train)                  # p2z(p)           vs z
    c_inv_in = p2z.hip_forward(p, train) if resample else c_inv       # the p2z(p) inside z2p(p2z(p))
    c_fwd_in = z2p.hip_forward(z, train) if resample else c_fwd       # the z2p(z) inside p2z(z2p(z))
    c_pzp = z2p.hip_forward(c_inv_in.out, train)       # z2p(p2z(p))      vs p
    c_zpz = p2z.hip_forward(c_fwd_in.out, train)       # p2z(z2p(z))      vs z
    ctxs = dict(fwd=c_fwd, inv=c_inv, inv_in=c_inv_in, fwd_in=c_fwd_in, pzp=c_pzp, zpz=c_zpz)
    if not train:
        eng.loss(c_fwd.out, p, row[0:1])
        eng.loss(c_inv.out, z, row[1:2])
        eng.loss(c_pzp.out, p, row[2:3])
        eng.loss(c_zpz.out, z, row[3:4])
        return ctxs
    g_pzp, g_zpz = torch.empty_like(c_pzp.out), torch.empty_like(c_zpz.out)
    eng.loss(c_pzp.out, p, row[2:3], g_pzp, weight=lam)
    eng.loss(c_zpz.out, z, row[3:4], g_zpz, weight=lam)
    eng.flat_g.zero_()
    g_inv = z2p.hip_backward(c_pzp, g_pzp, eng.grads, need_dx=True)      # gradient reaching p2z(p) through the cycle
    g_fwd = p2z.hip_backward(c_zpz, g_zpz, eng.grads, need_dx=True)      # gradient reaching z2p(z) through the cycle
    if resample:
        p2z.hip_backward(c_inv_in, g_inv, eng.grads)                     # the cycle terms' own inner evaluations
        z2p.hip_backward(c_fwd_in, g_fwd, eng.grads)
        g_fwd, g_inv = torch.empty_like(c_fwd.out), torch.empty_like(c_inv.out)
        eng.loss(c_fwd.out, p, row[0:1], g_fwd)                          # the direct terms on theirs
        eng.loss(c_inv.out, z, row[1:2], g_inv)
    else:
        eng.loss(c_fwd.out, p, row[0:1], g_fwd, accumulate=True)         # + the direct terms
        eng.loss(c_inv.out, z, row[1:2], g_inv, accumulate=True)
    z2p.hip_backward(c_fwd, g_fwd, eng.grads)
    p2z.hip_backward(c_inv, g_inv, eng.grads)
    return ctxs


def _has_dropout(m):
    return any(isinstance(l, nn.Dropout) and l.p > 0 for l in m.net)


def train_cycle_consistent(*, Z_train, P_train, Z_val, P_val, p2z_model, z2p_model, device, lr=1e-3, epochs=400,
                           batch_size=64, wd=1e-4, lam_cycle=0.1, num_workers=0, save_path=None):
    """Joint training of ParamToZ and ZToParam with cycle consistency losses (reference: solpex/latent.py:202-310)."""
    device = _check_device(device)
    p2z_model, z2p_model = p2z_model.to(device), z2p_model.to(device)
    tr, va = _loaders(Z_train, P_train, Z_val, P_val, batch_size, num_workers)
    eng = _Trainer([p2z_model, z2p_model], lr, wd, device)        # one AdamW / one clip over both (latent.py:227-228,255)
    best = float("inf")
    hist = {"tr": [], "va": [], "tr_fwd": [], "tr_inv": [], "tr_cyc_pzp": [], "tr_cyc_zpz": []}
    lam = float(lam_cycle)
    for ep in range(epochs):
        p2z_model.train()
        z2p_model.train()
        for z, p in tr:
            z, p = z.to(device), p.to(device)
            _cycle_losses(eng, p2z_model, z2p_model, z, p, lam, True)
            eng.step()
        rows, sizes = eng.read()
        n = max(float(sizes.sum()), 1.0)
        # the reference adds the four fp32 scalars on the device before .item(): reproduce that rounding
        tot = (rows[:, 0].astype(np.float32) + rows[:, 1].astype(np.float32)
               + np.float32(lam) * (rows[:, 2].astype(np.float32) + rows[:, 3].astype(np.float32))).astype(np.float64)
        tr_loss = float((tot * sizes).sum()) / n
        hist["tr"].append(tr_loss)
        for j, k in enumerate(("tr_fwd", "tr_inv", "tr_cyc_pzp", "tr_cyc_zpz")):
            hist[k].append(float((rows[:, j] * sizes).sum()) / n)
        p2z_model.eval()
        z2p_model.eval()
        for z, p in va:
            z, p = z.to(device), p.to(device)
            _cycle_losses(eng, p2z_model, z2p_model, z, p, lam, False)
        rows, sizes = eng.read()
        tot = (rows[:, 0].astype(np.float32) + rows[:, 1].astype(np.float32)
               + np.float32(lam) * (rows[:, 2].astype(np.float32) + rows[:, 3].astype(np.float32))).astype(np.float64)
        va_loss = float((tot * sizes).sum()) / max(float(sizes.sum()), 1.0)
        hist["va"].append(va_loss)
        if (ep % 10) == 0 or ep == epochs - 1:
            print(f"[cycle] ep {ep:04d} tr {tr_loss:.5f} va {va_loss:.5f} "
                  f"fwd {hist['tr_fwd'][-1]:.5f} inv {hist['tr_inv'][-1]:.5f} "
                  f"cyc_pzp {hist['tr_cyc_pzp'][-1]:.5f} cyc_zpz {hist['tr_cyc_zpz'][-1]:.5f}")
        if va_loss < best:
            best = va_loss
            if save_path:
                torch.save({"z2p_model": {k: v.detach().clone() for k, v in z2p_model.state_dict().items()},
                            "p2z_model": {k: v.detach().clone() for k, v in p2z_model.state_dict().items()},
                            "best": best, "z_dim": Z_train.shape[1], "P": P_train.shape[1]}, save_path)
    return p2z_model, z2p_model, hist


@torch.no_grad()
def extract_z_dataset(*, ae_model, dataset, device, batch_size=64, num_workers=0, max_batches=None):
    """Returns Z (N, z_dim) float32, P (N, P) float32, idxs (N,) int64 (reference: solpex/latent.py:313-372).
    `dataset` may also be a `ResidentLoader` (its batches are then used as they are, already on the device)."""
    from .data import ResidentLoader
    device = torch.device(device)
    ae_model.eval()
    if isinstance(dataset, ResidentLoader):
        loader, dataset = dataset, dataset.dataset
    else:
        loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers)
    def batch_params(batch):
        """(B, P) float32 numpy: from the batch itself, else looked up in the dataset by the batch's indices"""
        if "params" in batch:
            return batch["params"].detach().to("cpu", torch.float32).numpy()
        if "idx" not in batch or not hasattr(dataset, "params"):
            raise KeyError("Need params for param→z: provide batch['params'] or batch['idx'] with dataset.params.")
        rows = batch["idx"].detach().cpu().numpy() if torch.is_tensor(batch["idx"]) else batch["idx"]
        return np.asarray(dataset.params[np.asarray(rows, dtype=int)], dtype=np.float32)

    z_parts, p_parts = [], []
    for bi, batch in enumerate(loader):
        if max_batches is not None and bi >= max(int(max_batches), 1):      # (the reference always takes the first batch)
            break
        x = batch["x"].to(device)
        cond = batch["params"].to(device) if "params" in batch else None
        _, bott = ae_model(x, params=cond, return_bottleneck=True)
        z_t = bottleneck_to_z(bott)
        if getattr(ae_model, "z_proj", None) is not None:
            z_t = ae_model.z_proj(z_t)
        z_parts.append(z_t.detach().to("cpu", torch.float32).numpy())
        p_parts.append(batch_params(batch))
    Z = np.ascontiguousarray(np.concatenate(z_parts, axis=0), dtype=np.float32)
    P = np.ascontiguousarray(np.concatenate(p_parts, axis=0), dtype=np.float32)
    return Z, P, np.arange(Z.shape[0], dtype=np.int64)        # running positions, as the reference numbers them
