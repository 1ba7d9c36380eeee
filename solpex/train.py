"""Drop-in alias: `solpex.train` -> `unet_amd.train` (the MI355X-native implementation)."""
from unet_amd.train import *  # noqa: F401,F403
from unet_amd import train as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
