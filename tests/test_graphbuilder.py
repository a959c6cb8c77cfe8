"""Graph builder.  CPU part: the C restatement (oracle/graphbuilder_oracle.c)
against the reference's own golden (textgcn/test/test_cfunc.py:83-99) and against vectors produced
by the reference module itself (tests/golden/graphbuilder_ref.npz).  GPU part: the HIP builder,
through the C ABI, bit-exact against the oracle and the same vectors."""
import os

import numpy as np
import pytest

from oracle import graphbuilder_py as G

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "graphbuilder_ref.npz"))
REF_X = np.array([[0, 1, 2, 3, 4, -1, -1, -1], [5, 3, 4, 1, 2, 0, 5, 1]], dtype=np.int32)
REF_CIJ = np.array([4, 3, 3, 0, 0, 2, 6, 4, 2, 2, 1, 6, 2, 2, 1, 4, 3, 1, 4, 1, 3], dtype=np.uint32)


def random_tokens(rng, V, D, L):
    X = np.minimum(rng.integers(0, V, size=(D, L)), rng.integers(0, V, size=(D, L))).astype(np.int32)
    lens = rng.integers(0, L + 1, size=D)
    for d in range(D):
        X[d, lens[d]:] = -1
    return X


def test_oracle_reproduces_the_reference_golden_vector():
    c, nw = G.sliding_window(REF_X, 6, 3)                       # test_cfunc.py:97
    np.testing.assert_equal(c, REF_CIJ)
    assert nw == 9
    coo, w = G.compute_word_word_edges(REF_X, 6, 3)             # SURVEY.md section 4 capture
    assert coo.T.tolist() == [[0, 1, 0, 2, 0, 5, 3, 4], [1, 0, 2, 0, 5, 0, 4, 3]]
    exp = np.log(np.array([1.125, 1.125, 1.125, 1.125, 1.5, 1.5, 1.6875, 1.6875])).astype(np.float32)
    np.testing.assert_allclose(w, exp, rtol=1e-6)


def test_oracle_equals_vectors_from_the_reference_module():
    for i in range(int(GOLD["n_cases"])):
        X, V, win = GOLD[f"X{i}"], int(GOLD[f"V{i}"]), int(GOLD[f"win{i}"])
        c, _ = G.sliding_window(X, V, win)
        np.testing.assert_equal(c, GOLD[f"cij{i}"])
        coo, w = G.compute_word_word_edges(X, V, win)
        np.testing.assert_equal(coo, GOLD[f"coo{i}"])
        np.testing.assert_equal(w, GOLD[f"w{i}"])               # bit-exact float32


def test_oracle_against_live_reference_module_when_built():
    ref = G.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    c = np.asarray(ref.sliding_window_tester(REF_X, 6, 2, 8, 3))
    np.testing.assert_equal(c, REF_CIJ)
    coo, w = ref.compute_word_word_edges(REF_X, 6, 2, 8, 3)
    a, b = G.compute_word_word_edges(REF_X, 6, 3)
    np.testing.assert_equal(np.asarray(coo), a)
    np.testing.assert_equal(np.asarray(w), b)


@pytest.mark.gpu
def test_gpu_builder_reference_golden(cuda):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, n_windows, sliding_window_tester
    np.testing.assert_equal(sliding_window_tester(REF_X, 6, 2, 8, window_size=3), REF_CIJ)   # test_cfunc.py:97-99
    assert n_windows(REF_X, 6, 2, 8, 3) == 9
    coo, w = compute_word_word_edges(REF_X, 6, 2, 8, 3)
    assert coo.dtype == np.int32 and w.dtype == np.float32
    assert coo.T.tolist() == [[0, 1, 0, 2, 0, 5, 3, 4], [1, 0, 2, 0, 5, 0, 4, 3]]
    for i in range(int(GOLD["n_cases"])):
        X, V, win = GOLD[f"X{i}"], int(GOLD[f"V{i}"]), int(GOLD[f"win{i}"])
        np.testing.assert_equal(sliding_window_tester(X, V, X.shape[0], X.shape[1], win), GOLD[f"cij{i}"])
        coo, w = compute_word_word_edges(X, V, X.shape[0], X.shape[1], win)
        np.testing.assert_equal(coo, GOLD[f"coo{i}"])
        np.testing.assert_equal(w, GOLD[f"w{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("V,D,L,win", [(2, 1, 1, 1), (7, 5, 4, 9), (40, 30, 17, 17), (300, 500, 64, 20),
                                       (1000, 2000, 120, 20), (5000, 3000, 50, 5)])
def test_gpu_builder_equals_oracle_bit_exact(cuda, V, D, L, win):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    X = random_tokens(np.random.default_rng(V + D), V, D, L)
    c_ref, _ = G.sliding_window(X, V, win)
    np.testing.assert_equal(sliding_window_tester(X, V, D, L, win), c_ref)
    coo_ref, w_ref = G.compute_word_word_edges(X, V, win)
    coo, w = compute_word_word_edges(X, V, D, L, win)
    np.testing.assert_equal(coo, coo_ref)
    np.testing.assert_equal(w, w_ref)
    assert coo.shape[0] % 2 == 0
    if coo.shape[0]:
        assert (coo[0::2] == coo[1::2, ::-1]).all() and (coo[:, 0] != coo[:, 1]).all()


@pytest.mark.gpu
def test_gpu_builder_with_zipf_tokens_hot_pairs_counted_in_lds(cuda):
    """A Zipf-distributed corpus large enough for the hot-pair path (k_pair_counts_hot: the pairs among the 128
    most frequent words are counted in LDS and flushed once per workgroup): counts, edges and weights still
    equal the CPU restatement bit for bit, including the diagonal entries every token adds to."""
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    rng = np.random.default_rng(7)
    V, D, L, win = 3000, 1500, 80, 20
    p = 1.0 / np.arange(1, V + 1) ** 1.05
    X = rng.permutation(V)[rng.choice(V, size=(D, L), p=p / p.sum())].astype(np.int32)   # hot words at scattered ids
    lens = rng.integers(L // 2, L + 1, size=D)
    for d in range(D):
        X[d, lens[d]:] = -1
    assert D * L >= 1 << 16 and V > 128                         # the conditions of the hot-pair path
    c_ref, _ = G.sliding_window(X, V, win)
    np.testing.assert_equal(sliding_window_tester(X, V, D, L, win), c_ref)
    coo_ref, w_ref = G.compute_word_word_edges(X, V, win)
    coo, w = compute_word_word_edges(X, V, D, L, win)
    np.testing.assert_equal(coo, coo_ref)
    np.testing.assert_equal(w, w_ref)


@pytest.mark.gpu
def test_gpu_builder_edge_cases_and_errors(cuda):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    X = np.full((4, 6), -1, dtype=np.int32)                     # only padding: windows but no counts
    coo, w = compute_word_word_edges(X, 5, 4, 6, 3)
    assert coo.shape == (0, 2) and w.shape == (0,)
    assert sliding_window_tester(X, 5, 4, 6, 3).sum() == 0
    with pytest.raises(IndexError):
        compute_word_word_edges(np.array([[0, 7]], dtype=np.int32), 5, 1, 2, 2)
    with pytest.raises(ValueError):
        compute_word_word_edges(np.zeros((2, 3), dtype=np.int32), 5, 3, 3, 2)
