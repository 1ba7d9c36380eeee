"""Drop-in alias: `solpex.predict` -> `unet_amd.predict` (the MI355X-native implementation)."""
from unet_amd.predict import *  # noqa: F401,F403
from unet_amd import predict as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
