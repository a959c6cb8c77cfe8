"""Word-word PMI edges: drop-in for the reference's Cython module
`textgcn.lib.clib.graphbuilder` (exported through textgcn/lib/__init__.py:1-4), running on the GPU
through libtgcn.so (`tgcn_wwedges_*`, pytextgcn_amd/csrc/graphbuilder.hip).

Same call signatures and return conventions as graphbuilder.pyx:23-68 and :263-275: host numpy
arrays in, host numpy arrays out (`coo` int32 [n_edges, 2], `weights` float32 [n_edges]), so that
`Text2GraphTransformer` (text2graph.py:156-160) and the reference's tests (test_cfunc.py:81-111)
read the same.  `n_jobs` is accepted and unused, as in the reference (graphbuilder.pyx:36).

Two counters behind the same results (csrc/graphbuilder.hip): the dense packed triangle while it takes <= 16 GiB
(V <= ~92 000) and a sorted list of distinct pairs beyond (the reference's own array is O(V^2), graphbuilder.pyx:44,134,
and its index wraps beyond V = 65 535, :250).  `counter="dense" | "sparse"` pins one (None: by size).
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np
import torch

from . import _lib
from .plan import _stream_ptr


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pytextgcn_amd.graphbuilder runs on an AMD GPU through libtgcn.so "
                           "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _run(X, n_vocab: int, n_documents: int, seq_len: int, window_size: int, counter=None):
    import os
    if counter not in (None, "dense", "sparse"):
        raise ValueError('counter must be None, "dense" or "sparse"')
    lib = _lib.load()
    dev = _device()
    Xt = torch.as_tensor(np.ascontiguousarray(X, dtype=np.int32) if not torch.is_tensor(X) else X)
    if Xt.dtype != torch.int32:
        Xt = Xt.to(torch.int32)
    if tuple(Xt.shape) != (n_documents, seq_len):
        raise ValueError(f"X has shape {tuple(Xt.shape)}, expected ({n_documents}, {seq_len})")
    if Xt.numel() and (int(Xt.max()) >= n_vocab or int(Xt.min()) < -1):
        raise IndexError("tokens must lie in [0, n_vocab) or be the padding value -1")
    Xd = Xt.to(dev).contiguous()
    h = ctypes.c_void_p()
    saved = os.environ.get("TGCN_WW_COUNTER")
    if counter is not None:
        os.environ["TGCN_WW_COUNTER"] = counter          # read by tgcn_wwedges_create
    try:
        _lib.check(lib.tgcn_wwedges_create(Xd.data_ptr() if Xd.numel() else None, n_documents, seq_len,
                                           n_vocab, window_size, dev.index, _stream_ptr(dev),
                                           ctypes.byref(h)))
    finally:
        if counter is not None:
            if saved is None:
                os.environ.pop("TGCN_WW_COUNTER", None)
            else:
                os.environ["TGCN_WW_COUNTER"] = saved
    return lib, h, dev


def _query(lib, h, what: int) -> int:
    out = ctypes.c_int64()
    _lib.check(lib.tgcn_wwedges_query(h, what, ctypes.byref(out)))
    return int(out.value)


def compute_word_word_edges(X, n_vocab: int, n_documents: int, seq_len: int, window_size: int = 20,
                            n_jobs: int = 1, verbose: int = 0, counter=None) -> Tuple[np.ndarray, np.ndarray]:
    lib, h, dev = _run(X, n_vocab, n_documents, seq_len, window_size, counter)
    try:
        n = _query(lib, h, 0)
        coo = np.empty((n, 2), dtype=np.int32)
        w = np.empty(n, dtype=np.float32)
        _lib.check(lib.tgcn_wwedges_export(h, coo.ctypes.data if n else None,
                                           w.ctypes.data if n else None, None, _stream_ptr(dev)))
        if verbose > 1:
            print(f"Number of word-word-edges: {n}" + (f" ({_query(lib, h, 4)} distinct pairs, sorted-list counter)"
                                                        if _query(lib, h, 3) else ""))
        return coo, w
    finally:
        lib.tgcn_wwedges_destroy(h)


def sliding_window_tester(X, n_vocab: int, n_documents: int, seq_len: int, window_size: int = 20,
                          n_jobs: int = 1, counter=None) -> np.ndarray:
    """Packed upper triangle (incl. diagonal) of the co-occurrence counts, uint32."""
    lib, h, dev = _run(X, n_vocab, n_documents, seq_len, window_size, counter)
    try:
        c = np.empty(_query(lib, h, 2), dtype=np.uint32)
        _lib.check(lib.tgcn_wwedges_export(h, None, None, c.ctypes.data, _stream_ptr(dev)))
        return c
    finally:
        lib.tgcn_wwedges_destroy(h)


def n_windows(X, n_vocab: int, n_documents: int, seq_len: int, window_size: int = 20) -> int:
    lib, h, _ = _run(X, n_vocab, n_documents, seq_len, window_size)
    try:
        return _query(lib, h, 1)
    finally:
        lib.tgcn_wwedges_destroy(h)


def counter_stats(X, n_vocab: int, n_documents: int, seq_len: int, window_size: int = 20, counter=None) -> dict:
    """Which counter a call with these arguments runs and what it holds: {"sparse", "n_pairs" (distinct i <= j pairs with a
    count; -1 for the dense triangle), "n_edges", "n_windows"}."""
    lib, h, _ = _run(X, n_vocab, n_documents, seq_len, window_size, counter)
    try:
        return {"sparse": bool(_query(lib, h, 3)), "n_pairs": _query(lib, h, 4), "n_edges": _query(lib, h, 0),
                "n_windows": _query(lib, h, 1)}
    finally:
        lib.tgcn_wwedges_destroy(h)


def _sym_diag_idx(row: int, col: int, n: int) -> int:
    """Packed upper triangle incl. the diagonal, row-major -- the layout of `sliding_window_tester`'s
    result (graphbuilder.pyx:214-227)."""
    if row < col:
        row, col = col, row
    return col * n + row - (col + 1) * col // 2


def test_sym_matrix() -> int:
    """The reference's self-check of the packed index (graphbuilder.pyx:277-296), on all 10 slots
    (the reference compares only the first 6)."""
    mat = [0.0] * 10                                   # 4 x 4
    mat[_sym_diag_idx(1, 1, 4)] = 10
    mat[_sym_diag_idx(1, 2, 4)] = 20
    mat[_sym_diag_idx(2, 0, 4)] = 30
    mat[_sym_diag_idx(3, 3, 4)] = 100
    mat[_sym_diag_idx(2, 3, 4)] = 120
    return int(mat == [0, 0, 30, 0, 10, 20, 0, 0, 120, 100])


test_sym_matrix.__test__ = False                      # a library function, not a pytest case
