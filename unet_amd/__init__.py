"""unet_amd: MI355X-native (gfx950 HIP) implementation of the SOLPEx conditional U-Net hot path.

Import surface mirrors the part of `solpex/__init__.py` (reference :6-13) that belongs to the path:
models, losses, train, predict, data, latent.  The arithmetic lives in unet_amd/csrc (libsolpex_hip.so).
"""
from .data import (MaskedLogStandardizer, MaskedLinearStandardizer, MaskedSymLogStandardizer, MultiChannelNormalizer,
                   SOLPSDataset, make_loaders, normalize_coords, load_geometry_h5, apply_param_transform,
                   invert_param_transform)
from .models import UNet, ParamToZ, bottleneck_to_z, z_to_bottleneck
from .losses import edge_weights, masked_weighted_loss, mae_norm, batch_error_sums_ev
from .train import train_unet, TrainEngine
from .predict import load_checkpoint, predict_fields, predict_te, predict_fields_batch, scale_params
from .utils import (pick_device, sample_from_loader, eval_param2z_one, nearest_neighbor_in_Z, scale_params_for_inference,
                    save_geometry_h5, fp32_exact)
from .latent import (extract_z_dataset, ParamToZ, ZToParam, train_param2z, train_z2param,   # noqa: F811  (as the reference,
                     train_cycle_consistent)                                              # latent.ParamToZ wins)
from .data import ResidentLoader

__all__ = [
    "MaskedLogStandardizer", "MaskedLinearStandardizer", "MaskedSymLogStandardizer", "MultiChannelNormalizer",
    "SOLPSDataset", "make_loaders", "normalize_coords", "load_geometry_h5", "apply_param_transform",
    "invert_param_transform", "UNet", "ParamToZ", "bottleneck_to_z", "z_to_bottleneck", "edge_weights",
    "masked_weighted_loss", "mae_norm", "batch_error_sums_ev", "train_unet", "TrainEngine", "load_checkpoint",
    "predict_fields", "predict_te", "predict_fields_batch", "scale_params", "pick_device", "sample_from_loader", "eval_param2z_one",
    "nearest_neighbor_in_Z", "scale_params_for_inference", "save_geometry_h5", "extract_z_dataset", "ZToParam", "train_param2z", "train_z2param",
    "train_cycle_consistent", "ResidentLoader", "fp32_exact",
]
