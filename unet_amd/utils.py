"""Host helpers with the reference's names, arguments and return values (solpex/utils.py:11-97): the pipeline scripts
call them around the hot path (`pick_device()`, `idx, sample = sample_from_loader(loader)`, `eval_param2z_one(...)`).
Nothing here is arithmetic of the path; the model calls inside `eval_param2z_one` run on the HIP kernels."""
from __future__ import annotations

import time

import numpy as np
import torch


def sample_from_loader(loader, k=0):
    """(global index, sample dict) of the k-th entry of the split behind `loader` (reference: solpex/utils.py:11-15).

    Both loader kinds `make_loaders` returns keep the split as `loader.dataset` = `torch.utils.data.Subset` of the
    `SOLPSDataset` (host `DataLoader` and device-resident `ResidentLoader` alike), so the sample is the dataset's own
    item: normalised x / y / mask / params on the host, exactly what the reference hands to its plotting code."""
    split = loader.dataset
    gidx = split.indices[k]
    return gidx, split.dataset[gidx]


def pick_device():
    """`torch.device` to run on (reference: solpex/utils.py:18-23).  PyTorch-ROCm reports an MI355X as "cuda"."""
    if torch.cuda.is_available():
        return torch.device("cuda")
    mps = getattr(torch.backends, "mps", None)
    if mps is not None and mps.is_available():
        return torch.device("mps")
    return torch.device("cpu")


def scale_params_for_inference(params, param_mu, param_std):
    """Raw control parameters -> the scaled float32 vector the network was trained on (reference: solpex/utils.py:25-29);
    without a scaler the values pass through."""
    out = np.asarray(params, dtype=np.float32)
    if param_mu is None or param_std is None:
        return out
    return (out - np.asarray(param_mu, dtype=np.float32)) / np.asarray(param_std, dtype=np.float32)


def save_geometry_h5(path, r2d, z2d, case_name=None, units="m", level=4):
    """Write the cell-centre coordinates as float32 datasets R2D / Z2D (gzip + shuffle, chunks of at most 512 x 512) with
    the reference's attributes (solpex/utils.py:31-43); `load_geometry_h5` reads them back.  h5py is imported lazily."""
    import h5py
    fields = {"R2D": np.ascontiguousarray(r2d, dtype=np.float32), "Z2D": np.ascontiguousarray(z2d, dtype=np.float32)}
    rows, cols = fields["R2D"].shape
    with h5py.File(path, "w") as f:
        for name, arr in fields.items():
            ds = f.create_dataset(name, data=arr, chunks=(min(rows, 512), min(cols, 512)), compression="gzip",
                                  compression_opts=int(level), shuffle=True)
            ds.attrs["units"] = units
            ds.attrs["grid"] = "cell centers"
        f.attrs["created"] = time.strftime("%Y-%m-%d %H:%M:%S")
        if case_name is not None:
            f.attrs["case_name"] = str(case_name)


def nearest_neighbor_in_Z(z_pred, Z_ref):
    """(row index, squared distance) of the latent in Z_ref (N, z_dim) closest to z_pred (z_dim,)
    (reference: solpex/utils.py:45-48)."""
    d2 = ((np.asarray(Z_ref) - np.asarray(z_pred)[None, :]) ** 2).sum(axis=1)
    j = int(np.argmin(d2))
    return j, float(d2[j])


@torch.no_grad()
def eval_param2z_one(*, ae, mlp, sample, device, label="", z_mu=None, z_std=None, p_mu=None, p_std=None):
    """Latent of one sample from the autoencoder against the latent the params -> z MLP predicts
    (reference: solpex/utils.py:50-97).  Returns {"z_true", "z_pred" (CPU tensors), "cos", "z_true_norm", "z_pred_norm"}.

    As in the reference: the MLP sees the parameters scaled by (p_mu, p_std) -- the arguments, else the sample's own
    "p_mu"/"p_std" entries, else unscaled -- while the autoencoder's FiLM input is the sample's `params` as stored; if
    (z_mu, z_std) are given the MLP output is un-normalised before the comparison."""
    from .models import bottleneck_to_z
    ae.eval()
    mlp.eval()
    x = sample["x"].unsqueeze(0).to(device)
    p_in = sample["params"].detach().cpu().numpy().astype(np.float32, copy=False)
    if (p_mu is None or p_std is None) and "p_mu" in sample and "p_std" in sample:
        p_mu, p_std = sample["p_mu"], sample["p_std"]
    p_in = scale_params_for_inference(p_in, p_mu, p_std)
    film_in = sample["params"].unsqueeze(0).to(device) if "params" in sample else None
    _, bott = ae(x, params=film_in, return_bottleneck=True)
    z_true = bottleneck_to_z(bott)
    z_pred = mlp(torch.from_numpy(np.ascontiguousarray(p_in)).unsqueeze(0).to(device))
    if z_mu is not None and z_std is not None:
        z_pred = z_pred * torch.as_tensor(z_std, device=device).view(1, -1) + torch.as_tensor(z_mu, device=device).view(1, -1)
    tiny = 1e-12
    ut = z_true / (z_true.norm(dim=1, keepdim=True) + tiny)
    up = z_pred / (z_pred.norm(dim=1, keepdim=True) + tiny)
    res = {"z_true": z_true.detach().cpu(), "z_pred": z_pred.detach().cpu(), "cos": float((ut * up).sum(dim=1).item()),
           "z_true_norm": float(z_true.norm().item()), "z_pred_norm": float(z_pred.norm().item())}
    if label:
        print(f"[{label}] cos={res['cos']:.4f} |z_true|={res['z_true_norm']:.4g} |z_pred|={res['z_pred_norm']:.4g}")
    return res


def fp32_exact(on=None):
    """fp32 arithmetic of the convolution kernels (amp=False): False = split-bf16 (each operand as bf16 hi + lo, three bf16 MFMAs
    per product block, ~4e-6 per convolution against fp64; the default), True = exact fp32 MFMA (1.7x slower per training step).
    Returns the previous setting; `on=None` only queries.  The HIP build's counterpart of `torch.backends.cudnn.allow_tf32`
    (no equivalent in the reference's own code); the environment variable SX_FP32_EXACT=1 sets it at start-up."""
    from . import _lib as L
    return bool(L.lib().sx_set_fp32_exact(-1 if on is None else int(bool(on))))
