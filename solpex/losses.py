"""Drop-in alias: `solpex.losses` -> `unet_amd.losses` (the MI355X-native implementation)."""
from unet_amd.losses import *  # noqa: F401,F403
from unet_amd import losses as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
