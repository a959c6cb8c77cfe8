"""ctypes front-end of oracle/csr_spmm.c -- TEST INFRASTRUCTURE (see that file's header).

Turns the oracle's normalised edge list (gcn_oracle.gcn_norm, PyG-1.6.3 order) into CSR by a
STABLE sort on the target index, so the per-row summation order equals the reference's
scatter_add order on CPU, and runs the C row loop on it.  PARITY UNPINNED by the reference
(no golden exists for this path, oracle/gcn_oracle.py header); pinned in tests/test_oracle.py
against gcn_oracle.propagate.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np
import torch

from . import gcn_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_csr.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "csr_spmm.c")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        for name in ("oracle_csr_spmm_f32", "oracle_csr_spmm_f64acc"):
            fn = getattr(_lib, name)
            fn.argtypes = [i64, p, p, p, p, i64, i32, p, p, i64]
            fn.restype = None
        _lib.oracle_colsum_f64acc.argtypes = [i64, i32, p, i64, p]
        _lib.oracle_colsum_f64acc.restype = None
    return _lib


def coo_to_csr(target: torch.Tensor, source: torch.Tensor, val: torch.Tensor, n_rows: int
               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Stable sort by target -> (rowptr int64 [n_rows+1], col int32 [nnz], val f32 [nnz])."""
    order = torch.argsort(target, stable=True)
    counts = torch.bincount(target, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, source[order].to(torch.int32).contiguous(), val[order].float().contiguous()


def normalized_csr(edge_index, edge_weight, num_nodes: int, add_self_loops: bool = True,
                   transpose: bool = False):
    """CSR of M (M[target, source] = w_hat), or of M^T when `transpose`."""
    tgt, src, w = gcn_oracle.normalized_coo(edge_index, edge_weight, num_nodes, add_self_loops)
    if transpose:
        tgt, src = src, tgt
    return coo_to_csr(tgt, src, w, num_nodes)


def csr_spmm(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, x: torch.Tensor,
             bias: Optional[torch.Tensor] = None, acc64: bool = False) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.stride(1) == 1 and not x.is_cuda
    n_rows = rowptr.numel() - 1
    F = x.size(1)
    y = torch.empty(n_rows, F, dtype=torch.float32)
    fn = lib().oracle_csr_spmm_f64acc if acc64 else lib().oracle_csr_spmm_f32
    b = bias.contiguous().float() if bias is not None else None
    fn(n_rows, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), x.data_ptr(), x.stride(0), F,
       b.data_ptr() if b is not None else None, y.data_ptr(), y.stride(0))
    return y


def colsum(g: torch.Tensor) -> torch.Tensor:
    assert g.dtype == torch.float32 and g.stride(1) == 1 and not g.is_cuda
    out = torch.empty(g.size(1), dtype=torch.float32)
    lib().oracle_colsum_f64acc(g.size(0), g.size(1), g.data_ptr(), g.stride(0), out.data_ptr())
    return out
