"""Drop-in alias: `solpex.data` -> `unet_amd.data` (the MI355X-native implementation)."""
from unet_amd.data import *  # noqa: F401,F403
from unet_amd import data as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
