"""Golden vectors for the graph builder, produced by the REFERENCE's own Cython module compiled in
place as oracle/_ref (oracle/Makefile target `ref`; needs /root/reference, build container only).

Fixtures are data only: token matrices and the reference's outputs (packed c_ij, COO edges, PMI
weights).  Case 0 is the reference's own test input and golden (textgcn/test/test_cfunc.py:83-99).

Vocabulary sizes are restricted to values for which the reference is memory-safe: its
`SymMat_NoDiag_idx` (graphbuilder.pyx:229-241) leaves slot 0 unused and addresses one element past
the end of `edge_field` (V = 3: idx(1,2) = 3 with SymMatSize_NoDiag = 3), a 4-byte heap overflow that
corrupts the allocator whenever 2*V*(V-1) bytes exactly fills a malloc chunk.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import graphbuilder_py as G  # noqa: E402


def ref_safe_vocab(V: int) -> bool:
    R = 2 * V * (V - 1)                       # bytes of the reference's edge_field
    return (R % 16 != 0) if R <= 512 else (R % 16 != 8)


def main():
    assert G.build_ref(), "oracle/_ref could not be built (needs /root/reference and cython)"
    ref = G.load_ref()
    rng = np.random.default_rng(44)
    cases = [(np.array([[0, 1, 2, 3, 4, -1, -1, -1], [5, 3, 4, 1, 2, 0, 5, 1]], dtype=np.int32), 6, 3)]
    shapes = [(11, 9, 12, 5), (30, 40, 25, 20), (50, 25, 60, 7), (6, 3, 5, 5), (23, 64, 33, 1), (14, 10, 8, 8)]
    for V, D, L, win in shapes:
        assert ref_safe_vocab(V), V
        X = rng.integers(0, V, size=(D, L)).astype(np.int32)
        # zipf-ish skew so that PMI takes both signs, ragged lengths incl. empty and full documents
        X = np.minimum(X, rng.integers(0, V, size=(D, L))).astype(np.int32)
        lens = rng.integers(0, L + 1, size=D)
        lens[0], lens[-1] = L, 0
        for d in range(D):
            X[d, lens[d]:] = -1
        cases.append((X, V, win))
    out = {"n_cases": len(cases)}
    for i, (X, V, win) in enumerate(cases):
        D, L = X.shape
        cij = np.asarray(ref.sliding_window_tester(X, V, D, L, win)).copy()
        coo, w = ref.compute_word_word_edges(X, V, D, L, win)
        out[f"X{i}"], out[f"V{i}"], out[f"win{i}"] = X, V, win
        out[f"cij{i}"], out[f"coo{i}"], out[f"w{i}"] = cij, np.asarray(coo).copy(), np.asarray(w).copy()
    np.savez(os.path.join(HERE, "graphbuilder_ref.npz"), **out)
    print("wrote graphbuilder_ref.npz with", len(cases), "cases; edges per case:",
          [out[f"coo{i}"].shape[0] for i in range(len(cases))])


if __name__ == "__main__":
    main()
