"""ctypes binding of libtgcn.so -- one Python line per entry point of include/tgcn.h.

There is deliberately no fallback: if the HIP library is missing or fails to load, every
operator of this package raises.  (The CPU oracle under oracle/ is test infrastructure and is
never imported from here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

from .build import LIB_PATH

ABI_VERSION = 6

OK, E_INVALID, E_RANGE, E_HIP, E_NOMEM, E_WORKSPACE = 0, -1, -2, -3, -4, -5
NORM_OFF, NORM_ACCURATE, NORM_REFERENCE = 0, 1, 2          # `normalize` of tgcn_plan_create
DEGREE_ACCURATE, DEGREE_REFERENCE = 0, 1                   # `degree_sum` of tgcn_gcn_norm
DEGREE_SUMS = {"accurate": DEGREE_ACCURATE, "reference": DEGREE_REFERENCE}

(Q_N_NODES, Q_N_ROWS, Q_NNZ, Q_NNZ_T, Q_SYMMETRIC, Q_ITEMS, Q_ITEMS_T, Q_LONG_ROWS, Q_LONG_ROWS_T,
 Q_SEGMENTS, Q_SEGMENTS_T, Q_DEVICE_BYTES, Q_ROW_BEGIN, Q_HAS_TRANSPOSE, Q_N_ROWS_T, Q_HOT_ROWS,
 Q_HOT_ROWS_T) = range(17)

# name -> (restype, argtypes); tests/test_abi.py checks this table against include/tgcn.h
SIGNATURES = {
    "tgcn_abi_version": (c_int, []),
    "tgcn_last_error": (c_char_p, []),
    "tgcn_plan_create": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                 c_int, c_int, c_int64, c_int64, c_int, c_void_p,
                                 POINTER(c_void_p)]),
    "tgcn_plan_create_coo": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int, c_void_p, POINTER(c_void_p)]),
    "tgcn_gcn_norm": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p,
                              c_void_p, c_int, c_void_p]),
    "tgcn_plan_destroy": (c_int, [c_void_p]),
    "tgcn_plan_query": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "tgcn_plan_export": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tgcn_spmm_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "tgcn_spmm": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64,
                          c_void_p, c_size_t, c_void_p]),
    "tgcn_spmm_split": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int,
                                c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "tgcn_spmm_acc": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int,
                              c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "tgcn_spmm_adam": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int64, c_double, c_double, c_double, c_double, c_double, c_int64, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "tgcn_spmm_adam_split": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double, c_double, c_int64,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_rows_gather": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "tgcn_rows_scatter": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "tgcn_rows_reduce_ranked": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p, c_int64,
                                        c_int64, c_int64, c_int64, c_void_p]),
    "tgcn_colsum_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "tgcn_colsum": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t,
                            c_void_p]),
    "tgcn_masked_ce_workspace_bytes": (c_size_t, []),
    "tgcn_masked_ce": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_float,
                               c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "tgcn_masked_ce_pred": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_masked_ce_grad_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "tgcn_masked_ce_grad": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_scale_by_device_scalar": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "tgcn_gemm_nn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_void_p]),
    "tgcn_gemm_nt": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_void_p]),
    "tgcn_gemm_tn_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "tgcn_gemm_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_void_p, c_size_t, c_void_p]),
    "tgcn_gemm_nn_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_double, c_void_p, c_void_p]),
    "tgcn_gemm_nt_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_double, c_void_p, c_void_p]),
    "tgcn_gemm_tn_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_double, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_dropout_mask_words": (c_size_t, [c_int, c_int]),
    "tgcn_gemm_nn_dropout_mask": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                          c_double, c_void_p, c_void_p, c_int64, c_void_p]),
    "tgcn_gemm_tn_dropout_mask": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                          c_double, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "tgcn_gemm_nt_colsum_mask": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                         c_double, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_set_gemm_split": (c_int, [c_int]),
    "tgcn_set_dropout_row_keys": (c_int, [c_int64, c_int64, c_int64]),
    "tgcn_gemm_nt_colsum_workspace_bytes": (c_size_t, [c_int]),
    "tgcn_gemm_nt_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                    c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tgcn_wwedges_create": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p,
                                    POINTER(c_void_p)]),
    "tgcn_wwedges_query": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "tgcn_wwedges_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tgcn_wwedges_destroy": (c_int, [c_void_p]),
    "tgcn_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                               c_double, c_double, c_double, c_double, c_int64, c_void_p]),
    "tgcn_adam_step_capturable": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                                          c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                          c_void_p]),
}

_lib = None


class TgcnError(RuntimeError):
    """A libtgcn.so call returned a HIP / allocation / workspace error status."""


def load() -> ctypes.CDLL:
    """Load libtgcn.so (built by `python -m pytextgcn_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TGCN_LIB_PATH", LIB_PATH)      # A/B runs against another build of the library
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -m pytextgcn_amd.build`); pytextgcn_amd has no CPU fallback")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    from .build import build_report
    note = build_report().get("sources", {}).get("dense.hip", {}).get("pipe_kernel_isa_check", "")
    if "skipped" in note:
        import warnings
        warnings.warn("pytextgcn_amd: " + note)
    got = lib.tgcn_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libtgcn.so ABI version {got}, this package expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int) -> None:
    """Map a status code onto the reference's error behaviour (Python exceptions)."""
    if status == OK:
        return
    msg = load().tgcn_last_error().decode("utf-8", "replace")
    if status == E_INVALID:
        raise ValueError(msg)
    if status == E_RANGE:
        raise IndexError(msg)
    if status == E_NOMEM:
        raise MemoryError(msg)
    raise TgcnError(f"libtgcn status {status}: {msg}")
