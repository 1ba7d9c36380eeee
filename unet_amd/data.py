"""Mirror of `solpex.data` (reference: solpex/data.py).

Two kinds of code live here:
  * the output normalizers' pointwise maps (`transform` / `inverse`), which sit inside the validation
    loop and inference (solpex/train.py:284, solpex/predict.py:134).  On CUDA tensors they run the fused
    HIP kernel `sx_norm_apply` (one pass for all channels instead of slice/map/cat per channel);
  * host-side dataset preparation (npz loading, fitting, parameter transforms, DataLoaders), which the
    reference runs once on the CPU and which is deliberately plain Python/numpy here too (SURVEY 2.1 #5:
    out of scope as kernels, kept for drop-in).  CPU tensors handed to a normalizer by that pipeline
    (`SOLPSDataset.__getitem__`) are mapped with the same formulas in torch on the host.
"""
from __future__ import annotations

import ctypes as C
import numpy as np
import torch
from torch.utils.data import Dataset, DataLoader, Subset

from . import _lib as L


def normalize_coords(R2d, Z2d, mask=None):
    """Reference: solpex/data.py:11-22: affine map of (R,Z) to [-1,1] over the masked region."""
    sel = (mask > 0.5) if mask is not None else np.ones_like(R2d, dtype=bool)
    lo_r, hi_r = R2d[sel].min(), R2d[sel].max()
    lo_z, hi_z = Z2d[sel].min(), Z2d[sel].max()
    Rn = 2.0 * (R2d - lo_r) / (hi_r - lo_r) - 1.0
    Zn = 2.0 * (Z2d - lo_z) / (hi_z - lo_z) - 1.0
    return Rn.astype(np.float32), Zn.astype(np.float32)


def load_geometry_h5(fname):
    """Reference: solpex/data.py:24-29.  h5py is imported lazily: it is only needed for geometry files."""
    import h5py
    with h5py.File(fname, "r") as f:
        return np.array(f["R2D"]), np.array(f["Z2D"])


# ------------------------------------------------------------------------------------------ normalizers
class NormTable:
    """Per-channel (kind, mu, sigma, eps, scale) as host arrays for the C ABI."""

    def __init__(self, kinds, mus, sigmas, epss, scales):
        n = len(kinds)
        self.n = n
        self.kinds = (C.c_int * n)(*[int(k) for k in kinds])
        self.mu = (C.c_float * n)(*[float(v) for v in mus])
        self.sigma = (C.c_float * n)(*[float(v) for v in sigmas])
        self.eps = (C.c_float * n)(*[float(v) for v in epss])
        self.scale = (C.c_float * n)(*[float(v) for v in scales])

    def c_args(self):
        return (self.kinds, self.mu, self.sigma, self.eps, self.scale)


def _spec(norm):
    if isinstance(norm, MaskedLogStandardizer):
        return (L.SX_NORM_LOG, float(norm.mu), float(norm.sigma), float(norm.eps), 1.0)
    if isinstance(norm, MaskedSymLogStandardizer):
        return (L.SX_NORM_SYMLOG, float(norm.mu), float(norm.sigma), float(norm.eps), float(norm.scale))
    if isinstance(norm, MaskedLinearStandardizer):
        return (L.SX_NORM_LINEAR, float(norm.mu), float(norm.sigma), float(norm.eps), 1.0)
    raise TypeError(f"normalizer {type(norm).__name__} has no device kernel; use the solpex normalizer classes")


def norm_table(norm, n_channels):
    """NormTable for a single normalizer (applied to every channel) or a MultiChannelNormalizer."""
    if isinstance(norm, MultiChannelNormalizer):
        specs = [_spec(norm.norms[k]) for k in norm.y_keys]
        if len(specs) != n_channels:
            raise ValueError(f"normalizer has {len(specs)} channels but the tensor has {n_channels}")
    else:
        specs = [_spec(norm)] * n_channels
    return NormTable(*zip(*specs))


def _device_apply(norm, t, mask, inverse):
    """Run sx_norm_apply on a CUDA (B,C,H,W) tensor."""
    if t.dim() != 4:
        raise ValueError(f"normalizer: expected (B,C,H,W), got {tuple(t.shape)}")
    B, Cc, H, W = t.shape
    tab = norm_table(norm, Cc)
    src = t.detach()
    if src.dtype not in (torch.float32, torch.bfloat16):
        src = src.float()
    dt = L.SX_BF16 if src.dtype == torch.bfloat16 else L.SX_F32
    m = None
    if mask is not None:
        m = mask.to(device=t.device, dtype=torch.float32)
        if m.numel() != B * H * W:
            m = m.expand(B, 1, H, W)
        m = m.reshape(B, 1, H, W).contiguous()
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=t.device)
    sb, sc, sh, sw = src.stride()
    L.check(L.lib().sx_norm_apply(dt, 1 if inverse else 0, B, Cc, H, W, L.ptr(src), sb, sc, sh, sw, L.ptr(m),
                                  *tab.c_args(), L.ptr(out), L.stream_ptr()), "sx_norm_apply")
    return out


def _finite(t):
    return torch.where(torch.isfinite(t), t, torch.zeros_like(t))


def _masked_moments(loader, channel_index, key, mask_key, pre):
    """Masked mean / variance of pre(y[:, channel]) accumulated over a loader (host side, float64 sums)."""
    s1 = s2 = n = 0.0
    for batch in loader:
        y = torch.as_tensor(batch[key])
        m = torch.as_tensor(batch[mask_key])
        if y.dim() == 3:
            y = y.unsqueeze(1)
        elif y.dim() != 4:
            raise ValueError(f"Unexpected y shape {tuple(y.shape)}")
        if m.dim() == 3:
            m = m.unsqueeze(1)
        elif m.dim() != 4:
            raise ValueError(f"Unexpected mask shape {tuple(m.shape)}")
        yc = y[:, channel_index:channel_index + 1]
        vals = pre(yc)[m > 0.5]
        if vals.numel() == 0:
            continue
        s1 += vals.sum().item()
        s2 += (vals * vals).sum().item()
        n += float(vals.numel())
    mu = s1 / max(n, 1.0)
    var = s2 / max(n, 1.0) - mu * mu
    return mu, max(var, 0.0) ** 0.5


class MaskedLogStandardizer:
    """z = (log(clamp(y,0)+eps) - mu) / (sigma + 1e-8), masked (reference: solpex/data.py:32-110)."""

    def __init__(self, eps=1.0, sigma_floor=1e-6):
        self.mu = None
        self.sigma = None
        self.eps = float(eps)
        self.sigma_floor = float(sigma_floor)

    def fit(self, loader, channel_index: int = 0, key: str = "y", mask_key: str = "mask"):
        def pre(y):
            y = torch.nan_to_num(y, nan=0.0, posinf=0.0, neginf=0.0)
            return _finite(torch.log(torch.clamp(y, min=0) + self.eps))
        mu, sd = _masked_moments(loader, channel_index, key, mask_key, pre)
        self.mu = torch.tensor(mu, dtype=torch.float32)
        self.sigma = torch.tensor(max(sd, self.sigma_floor), dtype=torch.float32)

    def transform(self, y, mask):
        if y.is_cuda:
            return _device_apply(self, y, mask, inverse=False)
        y = torch.nan_to_num(y, nan=0.0, posinf=0.0, neginf=0.0)
        z = (torch.log(torch.clamp(y, min=0) + self.eps) - self.mu) / (self.sigma + 1e-8)
        return _finite(z) * mask

    def inverse(self, y_norm, mask=None):
        if y_norm.is_cuda:
            return _device_apply(self, y_norm, mask, inverse=True)
        y = _finite(torch.exp(y_norm * (self.sigma + 1e-8) + self.mu) - self.eps)
        return y if mask is None else y * mask


class MaskedLinearStandardizer:
    """z = (y - mu) / sigma (reference: solpex/data.py:112-143; the mask is not applied)."""

    def __init__(self, eps=1e-12):
        self.eps = eps
        self.mu = None
        self.sigma = None

    def fit(self, loader, key="y", mask_key="mask", channel_index=0):
        mu, sd = _masked_moments(loader, channel_index, key, mask_key, lambda y: y)
        self.mu = torch.tensor(mu, dtype=torch.float32)
        self.sigma = torch.tensor(sd + self.eps, dtype=torch.float32)

    def transform(self, y, m):
        if y.is_cuda:
            return _device_apply(self, y, m, inverse=False)
        return (y - self.mu) / self.sigma

    def inverse(self, z, m):
        if z.is_cuda:
            return _device_apply(self, z, m, inverse=True)
        return z * self.sigma + self.mu


class MaskedSymLogStandardizer:
    """z = (sign(y) log1p(|y|/s) - mu) / sigma, masked (reference: solpex/data.py:146-199)."""

    def __init__(self, scale, eps=1e-12):
        self.scale = float(scale)
        self.eps = eps
        self.mu = None
        self.sigma = None

    def _symlog(self, y):
        return torch.sign(y) * torch.log1p(torch.abs(y) / self.scale)

    def _symexp(self, z):
        return torch.sign(z) * self.scale * torch.expm1(torch.abs(z))

    def fit(self, loader, key="y", mask_key="mask", channel_index=0):
        mu, sd = _masked_moments(loader, channel_index, key, mask_key, lambda y: self._symlog(torch.nan_to_num(y)))
        self.mu = torch.tensor(mu, dtype=torch.float32)
        self.sigma = torch.tensor(sd + self.eps, dtype=torch.float32)

    def transform(self, y, m):
        if y.is_cuda:
            return _device_apply(self, y, m, inverse=False)
        z = (self._symlog(torch.nan_to_num(y)) - self.mu) / self.sigma
        return _finite(z) * m

    def inverse(self, zhat, m=None):
        if zhat.is_cuda:
            return _device_apply(self, zhat, m, inverse=True)
        y = _finite(self._symexp(zhat * self.sigma + self.mu))
        return y if m is None else y * m


class MultiChannelNormalizer:
    """One normalizer per output channel (reference: solpex/data.py:202-223)."""

    def __init__(self, y_keys, norms_by_name):
        self.y_keys = [str(k) for k in y_keys]
        self.norms = {k: norms_by_name[k] for k in self.y_keys}

    def fit(self, loader):
        for ci, name in enumerate(self.y_keys):
            self.norms[name].fit(loader, channel_index=ci)

    def transform(self, y, m):
        if y.is_cuda:
            return _device_apply(self, y, m, inverse=False)
        return torch.cat([self.norms[k].transform(y[:, i:i + 1], m) for i, k in enumerate(self.y_keys)], dim=1)

    def inverse(self, z, m):
        if z.is_cuda:
            return _device_apply(self, z, m, inverse=True)
        return torch.cat([self.norms[k].inverse(z[:, i:i + 1], m) for i, k in enumerate(self.y_keys)], dim=1)


# ------------------------------------------------------------------------------------------ parameter transforms
PARAM_TRANSFORMS = {"none", "throughput_ratio"}


def _col(keys, *names):
    keys = [str(k) for k in keys]
    for n in names:
        if n in keys:
            return keys.index(n)
    return None


def apply_param_transform(params, param_keys, transform):
    """(Gamma_D2, Gamma_core|n_core) -> (log10 throughput, core ratio); reference: solpex/data.py:405-428."""
    if transform is None or transform == "none":
        return params.copy(), list(param_keys)
    if transform != "throughput_ratio":
        raise ValueError(f"Unknown param_transform={transform!r}. Choose from {PARAM_TRANSFORMS}")
    ig, ic = _col(param_keys, "Gamma_D2"), _col(param_keys, "Gamma_core", "n_core")
    if ig is None or ic is None:
        raise KeyError(f"throughput_ratio requires Gamma_D2 and Gamma_core (or n_core) in param_keys={list(param_keys)}")
    out = params.astype(np.float64).copy()
    puff, core = out[:, ig].copy(), out[:, ic].copy()
    out[:, ig] = np.log10(puff + core)
    out[:, ic] = core / (puff + core)
    keys = list(param_keys)
    keys[ig], keys[ic] = "log10_thr", "ratio_nc"
    return out.astype(np.float32), keys


def invert_param_transform(params, param_keys, transform):
    """Inverse of apply_param_transform; reference: solpex/data.py:430-451."""
    if transform is None or transform == "none":
        return params.copy(), list(param_keys)
    if transform != "throughput_ratio":
        raise ValueError(f"Unknown param_transform={transform!r}")
    it, ir = _col(param_keys, "log10_thr"), _col(param_keys, "ratio_nc")
    if it is None or ir is None:
        raise KeyError(f"invert_param_transform requires log10_thr and ratio_nc in param_keys={list(param_keys)}")
    out = params.astype(np.float64).copy()
    thr = 10.0 ** out[:, it]
    ratio = np.clip(out[:, ir], 1e-12, 1.0 - 1e-12)
    out[:, it] = thr * (1.0 - ratio)
    out[:, ir] = thr * ratio
    keys = list(param_keys)
    keys[it], keys[ir] = "Gamma_D2", "Gamma_core"
    return out.astype(np.float32), keys


def fit_param_scaler(raw_params_train):
    """Reference: solpex/data.py:454-460."""
    x = raw_params_train.astype(np.float64)
    mu, std = x.mean(axis=0), x.std(axis=0)
    std = np.where(~np.isfinite(std) | (std < 1e-12), 1.0, std)
    mu = np.where(~np.isfinite(mu), 0.0, mu)
    return mu.astype(np.float32), std.astype(np.float32)


def apply_param_scaler(dataset, mu, std):
    if dataset.params.size and mu is not None:
        dataset.params = (dataset.params.astype(np.float32) - mu) / std


# ------------------------------------------------------------------------------------------ dataset / loaders
class SOLPSDataset(Dataset):
    """npz-backed dataset (reference: solpex/data.py:226-391).  Items: x (Cin,H,W), y (Cout,H,W),
    mask (1,H,W), params (P,), idx."""

    def __init__(self, path, geom_h5=None, normalizer=None, inputs="autoencoder", y_key="Te", y_keys=None,
                 param_transform=None):
        self.normalizer = normalizer
        self.inputs = str(inputs)
        self.y_key = str(y_key)
        self.y_keys_req = None if y_keys is None else [str(k) for k in y_keys]
        d = np.load(path, allow_pickle=True)
        self._has_Y = "Y" in d.files
        if self._has_Y:
            self.Y = d["Y"].astype(np.float32)
            self.y_keys_all = np.array(d["y_keys"]).astype(str)
            self.N, self.C_all, self.H, self.W = self.Y.shape
        else:
            self.Te = d["Te"].astype(np.float32)
            self.y_keys_all = np.array(["Te"], dtype=str)
            self.N, self.H, self.W = self.Te.shape
            self.C_all = 1

        def find(name):
            hits = np.where(self.y_keys_all == name)[0]
            if hits.size == 0:
                raise KeyError(f"y_key={name!r} not found in y_keys={list(self.y_keys_all)}")
            return int(hits[0])

        if self.y_keys_req is None:
            if not self._has_Y and self.y_key != "Te":
                raise KeyError("Legacy dataset has only 'Te'.")
            self.ks = [find(self.y_key)] if self._has_Y else [0]
            self.y_keys = np.array([self.y_key if self._has_Y else "Te"], dtype=str)
        else:
            if not self._has_Y:
                raise KeyError("Requested y_keys but dataset has no 'Y' array (only legacy Te).")
            self.ks = [find(k) for k in self.y_keys_req]
            self.y_keys = np.array(self.y_keys_req, dtype=str)
        self.C_out = len(self.ks)

        mask = d["mask"] if "mask" in d.files else None
        if mask is None or (isinstance(mask, np.ndarray) and mask.shape == ()):
            mask = np.ones((self.N, self.H, self.W), dtype=np.float32)
        else:
            mask = np.array(mask)
            if mask.ndim == 2 and mask.shape == (self.H, self.W):
                mask = np.repeat(mask[None], self.N, axis=0)
            if mask.shape != (self.N, self.H, self.W):
                raise ValueError(f"mask has wrong shape {mask.shape}, expected {(self.N, self.H, self.W)}")
            mask = mask.astype(np.float32)
        self.mask = mask

        self.Rn = self.Zn = None
        if geom_h5 is not None:
            R2d, Z2d = load_geometry_h5(geom_h5)
            if R2d.shape != (self.H, self.W) or Z2d.shape != (self.H, self.W):
                raise ValueError(f"Geometry shape mismatch: R2d{R2d.shape}, Z2d{Z2d.shape}, expected {(self.H, self.W)}")
            self.Rn, self.Zn = normalize_coords(R2d, Z2d, mask=(self.mask[0] > 0.5))

        params = d["params"] if "params" in d.files else (d["X"] if "X" in d.files else None)
        if params is None:
            params = np.zeros((self.N, 0), dtype=np.float32)
        else:
            params = np.array(params).astype(np.float32)
            if params.ndim != 2 or params.shape[0] != self.N:
                raise ValueError(f"params has shape {params.shape}, expected (N,P) with N={self.N}")
        keys = list(d["param_keys"]) if "param_keys" in d.files else [f"p{i}" for i in range(params.shape[1])]
        self.param_transform = param_transform
        if param_transform and param_transform != "none" and params.shape[1] > 0:
            params, keys = apply_param_transform(params, keys, param_transform)
        self.param_keys = keys
        self.params = params

    def __len__(self):
        return self.N

    def __getitem__(self, idx):
        raw = self.Y[idx, self.ks] if self._has_Y else self.Te[idx][None]
        raw = torch.nan_to_num(torch.from_numpy(np.ascontiguousarray(raw)).float(), nan=0.0, posinf=0.0, neginf=0.0)
        mask = (torch.from_numpy(self.mask[idx]).unsqueeze(0).float() > 0.5).float()
        if self.normalizer is None:
            y = raw * mask
        else:
            y = self.normalizer.transform(raw.unsqueeze(0), mask.unsqueeze(0)).squeeze(0)
        params = torch.from_numpy(self.params[idx]).float()
        if self.inputs == "autoencoder":
            return {"x": y.clone(), "y": y, "mask": mask, "params": params, "idx": int(idx)}
        chans = [mask]
        if self.Rn is not None:
            chans += [torch.from_numpy(self.Rn).unsqueeze(0).float(), torch.from_numpy(self.Zn).unsqueeze(0).float()]
        return {"x": torch.cat(chans, dim=0), "y": y, "mask": mask, "params": params, "idx": int(idx)}


class _Positions(Dataset):
    """Positions 0..n-1: lets torch's own DataLoader / RandomSampler decide the batch order, so a ResidentLoader
    consumes the global RNG exactly as the reference's DataLoader(shuffle=True) does (same permutations for a seed)."""

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return int(i)


def rank_share(global_batch, rank, world, pad):
    """Rank `rank`'s rows of one GLOBAL batch of positions (a LongTensor): the batch is cut into `world` equal consecutive
    shares; a short last batch is padded with `pad` (the position of an all-zero-mask sample, which adds nothing to any
    masked sum), so every rank sees the same number of rows and of batches.  Concatenating the shares of ranks 0..world-1
    gives the global batch back (plus padding): DP-N at per-rank batch b sees, step for step, the global batches a run at
    batch N*b would see FROM THE SAME PERMUTATION.  (The world == 1 loader draws its permutation from torch's global RNG like
    the reference's DataLoader; world > 1 loaders draw from a generator seeded by rank 0 -- equal batch contents need the
    serial run to be given that generator, otherwise the two runs are equivalent only up to the order of the epoch.)"""
    g = int(global_batch.numel())
    per = (g + world - 1) // world
    mine = global_batch[rank * per:(rank + 1) * per]
    if mine.numel() < per:
        mine = torch.cat([mine, torch.full((per - mine.numel(),), int(pad), dtype=global_batch.dtype)])
    return mine


class ResidentLoader:
    """Device-resident replacement of `DataLoader(Subset(SOLPSDataset, indices), batch_size, shuffle)`.

    The reference re-normalises every sample on the host every epoch (solpex/data.py:359-366) and collates item by
    item; at the HIP path's step time that host loop is the bottleneck.  Here the subset is uploaded ONCE, normalised
    ONCE on the device (sx_norm_apply), and a batch is three row gathers (sx_gather_rows) by a device index vector.
    Yields the same dict (`x`, `y`, `mask`, `params`, `idx`) with the same values and the same batch order as the
    host loader for a given torch seed; tensors are already on `device` (`idx` stays a host LongTensor).
    """

    _warned_unsynced = False

    def __init__(self, subset, batch_size, shuffle, device, drop_last=False, chunk=256, rank=None, world=None):
        """Under data parallelism (`rank` / `world`, default: the process group's) `batch_size` is the PER-RANK batch: the
        batch order is drawn for global batches of world * batch_size from a generator all ranks seed alike, and each rank
        gathers its share (`rank_share`)."""
        from . import dist as sxdist
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("ResidentLoader needs a CUDA (ROCm) device; use the host DataLoader otherwise")
        if isinstance(subset, Subset):
            ds, indices = subset.dataset, np.asarray(subset.indices, dtype=np.int64)
        else:
            ds, indices = subset, np.arange(len(subset), dtype=np.int64)
        if not isinstance(ds, SOLPSDataset):
            raise TypeError("ResidentLoader wraps a SOLPSDataset (or a Subset of one)")
        self.dataset, self.batch_size, self.device = subset, int(batch_size), device
        self._shuffle, self._drop_last = bool(shuffle), bool(drop_last)
        self._indices = torch.cat([torch.from_numpy(indices), torch.tensor([-1], dtype=torch.int64)])   # [n] = padding row
        n = len(indices)
        self.n, self.C, self.H, self.W = n, ds.C_out, ds.H, ds.W
        self.shard(sxdist.rank() if rank is None else rank, sxdist.world_size() if world is None else world)
        self.autoencoder = ds.inputs == "autoencoder"
        mask = (ds.mask[indices] > 0.5).astype(np.float32)[:, None]                     # (n,1,H,W), data.py:369-370
        # one extra all-zero row at position n: the padding sample of a short last global batch under data parallelism
        self.mask = torch.zeros(n + 1, 1, self.H, self.W, dtype=torch.float32, device=device)
        self.mask[:n] = torch.from_numpy(mask).to(device)
        self.y = torch.zeros(n + 1, self.C, self.H, self.W, dtype=torch.float32, device=device)
        for i0 in range(0, n, chunk):
            sel = indices[i0:i0 + chunk]
            raw = ds.Y[sel][:, ds.ks] if ds._has_Y else ds.Te[sel][:, None]
            raw = np.nan_to_num(np.ascontiguousarray(raw, dtype=np.float32), nan=0.0, posinf=0.0, neginf=0.0)
            m = self.mask[i0:min(i0 + chunk, n)]
            if ds.normalizer is None:
                self.y[i0:i0 + len(sel)] = torch.from_numpy(raw * mask[i0:i0 + chunk]).to(device)
            else:
                self.y[i0:i0 + len(sel)] = ds.normalizer.transform(torch.from_numpy(raw).to(device), m)
        prm = np.ascontiguousarray(ds.params[indices], dtype=np.float32)
        self.params = torch.from_numpy(np.concatenate([prm, np.zeros((1, prm.shape[1]), np.float32)])).to(device)
        self.P = self.params.shape[1]
        self.geo = None
        if not self.autoencoder and ds.Rn is not None:
            self.geo = torch.from_numpy(np.stack([ds.Rn, ds.Zn]).astype(np.float32)).to(device).contiguous()
        self.in_ch = self.C if self.autoencoder else (3 if self.geo is not None else 1)

    def shard(self, rank, world):
        """(Re)define this loader as rank `rank` of `world` data-parallel ranks."""
        self.rank, self.world = int(rank), max(int(world), 1)
        if not 0 <= self.rank < self.world:
            raise ValueError(f"ResidentLoader: rank {rank} outside 0..{world - 1}")
        # world == 1: torch's global RNG decides the order, exactly as the reference's DataLoader(shuffle=True) does
        self._gen = torch.Generator() if self.world > 1 else None
        self._order = DataLoader(_Positions(self.n), batch_size=self.batch_size * self.world, shuffle=self._shuffle,
                                 drop_last=self._drop_last, generator=self._gen)
        return self

    def __len__(self):
        return len(self._order)

    def _sync_seed(self):
        """All ranks must draw the same permutation: rank 0's next value of the global RNG seeds every rank's generator."""
        from . import dist as sxdist
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        if sxdist.is_active() and sxdist.world_size() == self.world:
            seed = sxdist.broadcast0(seed, self.device)
        elif not ResidentLoader._warned_unsynced:
            # explicit rank= / world= without a matching process group: nothing to agree on a seed through.  Correct only if
            # every rank's global RNG is in the same state (same torch.manual_seed, same draws so far)
            import warnings
            ResidentLoader._warned_unsynced = True
            warnings.warn(f"ResidentLoader: world={self.world} but no process group of that size is active: the epoch permutation "
                          "is seeded from this process's global RNG alone -- ranks whose RNG states differ will duplicate or drop "
                          "samples.  Seed every rank identically, or initialise torch.distributed first.")
        self._gen.manual_seed(int(seed))

    def _gather(self, src, n_src, row, pos, nb, dst, dst_stride):
        L.check(L.lib().sx_gather_rows(L.ptr(src), row, n_src, L.ptr(pos), nb, row, L.ptr(dst), dst_stride, L.stream_ptr()),
                "sx_gather_rows")

    def __iter__(self):
        HW = self.H * self.W
        if self.world > 1:
            self._sync_seed()
        for pos in self._order:                      # host LongTensor of positions inside the subset
            if int(pos.min()) < 0 or int(pos.max()) >= self.n:
                raise IndexError("ResidentLoader: sampler produced an index outside the subset")
            if self.world > 1:
                pos = rank_share(pos, self.rank, self.world, self.n)
            nb = int(pos.numel())
            pos_d = pos.to(self.device, non_blocking=True)
            y = torch.empty(nb, self.C, self.H, self.W, dtype=torch.float32, device=self.device)
            mask = torch.empty(nb, 1, self.H, self.W, dtype=torch.float32, device=self.device)
            self._gather(self.y, self.n + 1, self.C * HW, pos_d, nb, y, self.C * HW)
            self._gather(self.mask, self.n + 1, HW, pos_d, nb, mask, HW)
            params = torch.empty(nb, self.P, dtype=torch.float32, device=self.device)
            if self.P > 0:
                self._gather(self.params, self.n + 1, self.P, pos_d, nb, params, self.P)
            if self.autoencoder:
                x = y.clone()
            elif self.geo is None:
                x = mask.clone()
            else:
                x = torch.empty(nb, 3, self.H, self.W, dtype=torch.float32, device=self.device)
                self._gather(self.mask, self.n + 1, HW, pos_d, nb, x, 3 * HW)
                zero = torch.zeros(nb, dtype=torch.int64, device=self.device)
                self._gather(self.geo, 1, 2 * HW, zero, nb, x.view(nb, -1)[:, HW:], 3 * HW)
            yield {"x": x, "y": y, "mask": mask, "params": params, "idx": self._indices[pos]}


def make_loaders(npz_path, inputs_mode="params", batch_size=16, split=0.85, seed=42,
                 device="cuda" if torch.cuda.is_available() else "cpu", y_key="Te", y_keys=None, norm_fit_batch=16,
                 geom_h5=None, norms_by_name=None, num_workers=0, param_transform=None, resident=None, rank=None,
                 world=None):
    """Reference: solpex/data.py:467-572.  Returns the same 8-tuple.  `resident` (extra): True = the two loaders are
    `ResidentLoader`s (dataset held and normalised in HBM, batches gathered on the device); None = automatically when
    `device` is a GPU and one is present; False = the reference's host DataLoaders.
    `rank` / `world` (extra; default: the torch.distributed group's): under data parallelism `batch_size` is the per-rank
    batch and every rank iterates its own share of each global batch -- same number of batches on every rank (resident
    loaders pad a short last batch with an empty-mask sample; host loaders use torch's DistributedSampler, which pads by
    wrapping around: call `loader.sampler.set_epoch(epoch)` for a fresh shuffle per epoch)."""
    from . import dist as sxdist
    rank = sxdist.rank() if rank is None else int(rank)
    world = sxdist.world_size() if world is None else int(world)
    geom = geom_h5 if inputs_mode != "autoencoder" else None
    kw = dict(geom_h5=geom, inputs=inputs_mode, y_key=y_key, y_keys=y_keys, param_transform=param_transform)
    raw_all = SOLPSDataset(npz_path, normalizer=None, **kw)
    order = np.arange(len(raw_all))
    np.random.default_rng(seed).shuffle(order)
    cut = int(split * len(raw_all))
    idx_tr, idx_va = order[:cut], order[cut:]
    fit_loader = DataLoader(Subset(raw_all, idx_tr), batch_size=norm_fit_batch, shuffle=False, num_workers=num_workers)
    if y_keys is None:
        norm = MaskedLogStandardizer(eps=1.0)
    else:
        if norms_by_name is None:
            raise ValueError("y_keys provided but norms_by_name is None.")
        norm = MultiChannelNormalizer(y_keys=y_keys, norms_by_name=norms_by_name)
    norm.fit(fit_loader)
    ds_tr = SOLPSDataset(npz_path, normalizer=norm, **kw)
    ds_va = SOLPSDataset(npz_path, normalizer=norm, **kw)
    param_mu = param_std = None
    if inputs_mode != "autoencoder":
        param_mu, param_std = fit_param_scaler(raw_all.params[idx_tr])
        apply_param_scaler(ds_tr, param_mu, param_std)
        apply_param_scaler(ds_va, param_mu, param_std)
    train_set, val_set = Subset(ds_tr, idx_tr), Subset(ds_va, idx_va)
    pin = torch.device(device).type == "cuda"
    extra = dict(num_workers=num_workers, pin_memory=pin, persistent_workers=(num_workers > 0),
                 prefetch_factor=(4 if num_workers > 0 else None))
    if resident is None:
        resident = pin and torch.cuda.is_available()
    if resident:
        train_loader = ResidentLoader(train_set, batch_size, True, device, rank=rank, world=world)
        val_loader = ResidentLoader(val_set, batch_size, False, device, rank=rank, world=world)
    elif world > 1:
        from torch.utils.data.distributed import DistributedSampler
        s_tr = DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True, seed=seed)
        s_va = DistributedSampler(val_set, num_replicas=world, rank=rank, shuffle=False)
        train_loader = DataLoader(train_set, batch_size=batch_size, sampler=s_tr, **extra)
        val_loader = DataLoader(val_set, batch_size=batch_size, sampler=s_va, **extra)
    else:
        train_loader = DataLoader(train_set, batch_size=batch_size, shuffle=True, **extra)
        val_loader = DataLoader(val_set, batch_size=batch_size, shuffle=False, **extra)
    P = train_set.dataset.params.shape[1] if hasattr(train_set.dataset, "params") else 0
    H, W = train_set[0]["mask"].shape[-2:]
    return train_loader, val_loader, norm, P, (H, W), (param_mu, param_std), param_transform, raw_all.param_keys
