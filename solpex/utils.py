"""Drop-in alias: `solpex.utils` -> `unet_amd.utils` (the MI355X-native implementation)."""
from unet_amd.utils import *  # noqa: F401,F403
from unet_amd import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
