"""Drop-in alias: `solpex.models` -> `unet_amd.models` (the MI355X-native implementation)."""
from unet_amd.models import *  # noqa: F401,F403
from unet_amd import models as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
