"""CPU: the C-ABI shared library loads and exports every symbol include/matcouply_hip.h declares; without a GPU the
product fails loudly (no compute calls are made here)."""
import ctypes
import os
import re

import pytest

from matcouply_amd import _engine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "matcouply_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcl_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert len(declared) >= 28
    assert sorted(_engine.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_engine.LIB_PATH):
        pytest.fail("libmatcouply_hip.so has not been built: run __graft_entry__.build()")
    lib = ctypes.CDLL(_engine.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert _engine.load_library().mcl_version() >= 100


def test_struct_layouts_match_header():
    # mcl_penalty_desc: 2 x int32, 2 x double, 3 pointers + (ABI 400) the matrix pointer and its int64 row count;
    # mcl_options: 4 doubles + 4 int32
    assert ctypes.sizeof(_engine.PenaltyDesc) == 8 + 16 + 24 + 16
    assert ctypes.sizeof(_engine.Options) == 40 + 16  # 5 doubles (ABI 400: + inner_tol) + 4 int32
    assert _engine.DIAG_LEN == 8 + 3 * _engine.MCL_MAX_REGS * 2


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = _engine.load_library()
    h = ctypes.c_void_p()
    rc = lib.mcl_create(ctypes.byref(h), 0, None)
    assert rc != 0 and not h.value
    assert b"no HIP device" in lib.mcl_last_error(None)
