"""CPU: the C-ABI library loads and exports every symbol include/solpex_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "solpex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(sx_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from unet_amd import _lib
    decl = _declared()
    assert len(decl) >= 30
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(h, name), f"{name} declared in include/solpex_hip.h but not exported"
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, n in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} args, binding {len(_lib.SIGNATURES[name][1])}"
    assert _lib.lib().sx_version() == 100


def test_host_side_queries_need_no_gpu():
    from unet_amd import _lib as L
    lib = L.lib()
    d = L.UNetDesc(in_ch=1, out_ch=8, base=32, depth=3, groups_gn=8, P=5, film_hidden=128, B=64, H=128, W=64,
                   dtype=L.SX_BF16, use_film=1)
    assert lib.sx_unet_num_params(ctypes.byref(d)) == 85
    assert lib.sx_unet_workspace_bytes(ctypes.byref(d)) > 0
    assert lib.sx_unet_wpack_bytes(ctypes.byref(d)) > 0
    bad = L.UNetDesc(in_ch=1, out_ch=8, base=12, depth=3, groups_gn=8, P=5, film_hidden=128, B=1, H=16, W=16,
                     dtype=L.SX_F32, use_film=1)
    assert lib.sx_unet_workspace_bytes(ctypes.byref(bad)) < 0
    with pytest.raises(ValueError):
        L.check(lib.sx_unet_workspace_bytes(ctypes.byref(bad)))
    assert b"base" in lib.sx_last_error()
