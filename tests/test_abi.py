"""The C-ABI library loads on a CPU-only host and exports exactly what include/tgcn.h declares.
No compute entry point is called here (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgcn_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ["tgcn_plan_create", "tgcn_plan_destroy", "tgcn_spmm", "tgcn_colsum",
                 "tgcn_last_error", "tgcn_abi_version"]:
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from pytextgcn_amd import _lib, build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/tgcn.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    loaded = _lib.load()
    assert loaded.tgcn_abi_version() == _lib.ABI_VERSION
    assert loaded.tgcn_last_error() is not None


def test_argument_errors_do_not_need_a_gpu():
    from pytextgcn_amd import _lib
    lib = _lib.load()
    out = ctypes.c_void_p()
    st = lib.tgcn_plan_create(0, 0, None, 1, None, 1, None, 1, 1, 0, 0, 0, None, ctypes.byref(out))
    assert st == _lib.E_INVALID and b"n_nodes" in lib.tgcn_last_error()
    with pytest.raises(ValueError):
        _lib.check(st)
    assert lib.tgcn_spmm_workspace_bytes(None, 0, 8) == 0
    # the accumulate form (ABI 6) shares tgcn_spmm's argument checks: a NULL plan is refused before anything is enqueued
    assert lib.tgcn_spmm_acc(None, 0, None, 8, None, 0, 0, 8, None, 8, None, 0, None) == _lib.E_INVALID
    assert b"NULL" in lib.tgcn_last_error()
    assert lib.tgcn_colsum_workspace_bytes(1000, 200) > 0
    assert lib.tgcn_plan_destroy(None) == 0
    q = ctypes.c_int64()
    assert lib.tgcn_plan_query(None, 0, ctypes.byref(q)) == _lib.E_INVALID
