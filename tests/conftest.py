import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(params=["split", "exact", "x6"])
def fp32_mode(request):
    """fp32 arithmetic of the convolution kernels for one test: "split" = the default (bf16 hi + lo planes, three bf16 MFMAs
    per product block, unet_amd/csrc/conv_x3.hip), "exact" = fp32 MFMA, "x6" = three planes / six MFMAs (sx_set_fp32_planes).
    Restored afterwards."""
    from unet_amd import _lib
    lib = _lib.lib()
    was = lib.sx_set_fp32_planes({"exact": 0, "split": 2, "x6": 3}[request.param])
    yield request.param
    lib.sx_set_fp32_planes(was)
