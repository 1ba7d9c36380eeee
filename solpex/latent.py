"""Alias of unet_amd.latent (reference: solpex/latent.py)."""
from unet_amd.latent import *  # noqa: F401,F403
from unet_amd.latent import (ParamToZ, ZToParam, train_param2z, train_z2param, train_cycle_consistent,  # noqa: F401
                             extract_z_dataset)
