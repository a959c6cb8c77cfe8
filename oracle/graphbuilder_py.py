"""ctypes front-end of oracle/graphbuilder_oracle.c and loader of oracle/_ref (the reference's own
Cython module compiled in place) -- TEST INFRASTRUCTURE, see the C file's header."""
from __future__ import annotations

import ctypes
import importlib.util
import glob
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle_graphbuilder.so")
_lib = None


def build(force: bool = False) -> None:
    src = os.path.join(_HERE, "graphbuilder_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", _HERE, "_build/liboracle_graphbuilder.so"], check=True)


def build_ref() -> bool:
    """Compile the reference .pyx into oracle/_ref/ when /root/reference is present."""
    if glob.glob(os.path.join(_HERE, "_ref", "graphbuilder*.so")):
        return True
    if not os.path.exists("/root/reference/textgcn/lib/clib/graphbuilder.pyx"):
        return False
    r = subprocess.run(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return r.returncode == 0


def load_ref():
    """The compiled reference module (compute_word_word_edges, sliding_window_tester, ...) or None."""
    found = glob.glob(os.path.join(_HERE, "_ref", "graphbuilder*.so"))
    if not found:
        return None
    spec = importlib.util.spec_from_file_location("graphbuilder", found[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        _lib.oracle_sym_size_diag.argtypes, _lib.oracle_sym_size_diag.restype = [i64], i64
        _lib.oracle_sliding_window.argtypes = [p, p, i64, i64, i64, i64]
        _lib.oracle_sliding_window.restype = i64
        _lib.oracle_count_edges.argtypes, _lib.oracle_count_edges.restype = [p, i64, i64], i64
        _lib.oracle_emit_edges.argtypes, _lib.oracle_emit_edges.restype = [p, i64, i64, p, p], None
    return _lib


def sliding_window(X: np.ndarray, n_vocab: int, window: int) -> Tuple[np.ndarray, int]:
    X = np.ascontiguousarray(X, dtype=np.int32)
    c = np.zeros(lib().oracle_sym_size_diag(n_vocab), dtype=np.uint32)
    nw = lib().oracle_sliding_window(X.ctypes.data, c.ctypes.data, window, n_vocab, X.shape[0], X.shape[1])
    return c, int(nw)


def compute_word_word_edges(X: np.ndarray, n_vocab: int, window: int) -> Tuple[np.ndarray, np.ndarray]:
    """(coo int32 [n_edges, 2], weights float32 [n_edges]) as graphbuilder.pyx:23-68 returns them."""
    c, nw = sliding_window(X, n_vocab, window)
    n = lib().oracle_count_edges(c.ctypes.data, n_vocab, nw)
    coo = np.zeros((n, 2), dtype=np.int32)
    w = np.zeros(n, dtype=np.float32)
    lib().oracle_emit_edges(c.ctypes.data, n_vocab, nw, coo.ctypes.data, w.ctypes.data)
    return coo, w
