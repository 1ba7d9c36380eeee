"""CPU oracle for the SOLPEx conditional U-Net hot path.

TEST INFRASTRUCTURE ONLY.  This package is a fresh, plain-PyTorch (CPU, fp32)
restatement of the reference algorithm.  It may be imported only by `tests/`,
by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py`,
and there only as the checker / the timed CPU baseline -- never by the
product package `unet_amd` (which must fail loudly without its HIP library).

Parity pin: every function here is checked against golden vectors captured
from the live reference (`/root/reference/solpex`, imported in the build
container by `tools/make_golden.py`; vectors committed under `tests/golden/`).
The reference itself ships no tests or golden vectors for this path
(SURVEY.md section 4), so those captured vectors are the only pin.
"""
from .unet_ref import UNetCfg, init_state_dict, unet_forward, unet_encode, unet_decode, film_generator  # noqa: F401
from .loss_ref import masked_weighted_loss, masked_mean, mae_norm, batch_error_sums_ev, edge_weights, sobel_xy  # noqa: F401
from .norm_ref import norm_transform, norm_inverse, NormSpec  # noqa: F401
from .optim_ref import clip_grad_norm, adamw_step  # noqa: F401
