"""Drop-in `solpex` package name for the MI355X build: the reference's scripts do `import solpex` /
`from solpex.models import UNet` (pyproject.toml:6,18-20); everything resolves to `unet_amd`."""
from unet_amd import *  # noqa: F401,F403
from unet_amd import __all__  # noqa: F401
