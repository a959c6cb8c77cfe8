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


# both counters of the HIP builder: the dense packed triangle (small vocabularies) and the sorted list of distinct pairs
# (everything larger; SURVEY.md 8(f) #2 "lift the O(V^2) memory"); "sparse_chunked": the same with 5 000 records per chunk
# of documents, i.e. many sort / sum / merge rounds on these inputs
COUNTERS = ["dense", "sparse", "sparse_chunked"]


def _pin(monkeypatch, counter):
    monkeypatch.delenv("TGCN_WW_CHUNK_PAIRS", raising=False)
    if counter == "sparse_chunked":
        monkeypatch.setenv("TGCN_WW_CHUNK_PAIRS", "5000")
    return "dense" if counter == "dense" else "sparse"


@pytest.mark.gpu
@pytest.mark.parametrize("counter", COUNTERS)
def test_gpu_builder_reference_golden(cuda, monkeypatch, counter):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, counter_stats, n_windows, sliding_window_tester
    kind = _pin(monkeypatch, counter)
    np.testing.assert_equal(sliding_window_tester(REF_X, 6, 2, 8, window_size=3, counter=kind), REF_CIJ)   # test_cfunc.py:97-99
    assert n_windows(REF_X, 6, 2, 8, 3) == 9
    coo, w = compute_word_word_edges(REF_X, 6, 2, 8, 3, counter=kind)
    assert coo.dtype == np.int32 and w.dtype == np.float32
    assert coo.T.tolist() == [[0, 1, 0, 2, 0, 5, 3, 4], [1, 0, 2, 0, 5, 0, 4, 3]]
    st = counter_stats(REF_X, 6, 2, 8, 3, counter=kind)
    assert st["sparse"] == (kind == "sparse") and st["n_edges"] == 8 and st["n_windows"] == 9
    assert st["n_pairs"] == (int((REF_CIJ != 0).sum()) if kind == "sparse" else -1)
    for i in range(int(GOLD["n_cases"])):
        X, V, win = GOLD[f"X{i}"], int(GOLD[f"V{i}"]), int(GOLD[f"win{i}"])
        np.testing.assert_equal(sliding_window_tester(X, V, X.shape[0], X.shape[1], win, counter=kind), GOLD[f"cij{i}"])
        coo, w = compute_word_word_edges(X, V, X.shape[0], X.shape[1], win, counter=kind)
        np.testing.assert_equal(coo, GOLD[f"coo{i}"])
        np.testing.assert_equal(w, GOLD[f"w{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("counter", COUNTERS)
@pytest.mark.parametrize("V,D,L,win", [(2, 1, 1, 1), (7, 5, 4, 9), (40, 30, 17, 17), (300, 500, 64, 20),
                                       (1000, 2000, 120, 20), (5000, 3000, 50, 5)])
def test_gpu_builder_equals_oracle_bit_exact(cuda, monkeypatch, V, D, L, win, counter):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    kind = _pin(monkeypatch, counter)
    X = random_tokens(np.random.default_rng(V + D), V, D, L)
    c_ref, _ = G.sliding_window(X, V, win)
    np.testing.assert_equal(sliding_window_tester(X, V, D, L, win, counter=kind), c_ref)
    coo_ref, w_ref = G.compute_word_word_edges(X, V, win)
    coo, w = compute_word_word_edges(X, V, D, L, win, counter=kind)
    np.testing.assert_equal(coo, coo_ref)
    np.testing.assert_equal(w, w_ref)
    assert coo.shape[0] % 2 == 0
    if coo.shape[0]:
        assert (coo[0::2] == coo[1::2, ::-1]).all() and (coo[:, 0] != coo[:, 1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("counter", COUNTERS)
def test_gpu_builder_with_zipf_tokens_hot_pairs_counted_in_lds(cuda, monkeypatch, counter):
    """A Zipf-distributed corpus large enough for the hot-pair path (k_pair_counts_hot: the pairs among the 128
    most frequent words are counted in LDS and flushed once per workgroup): counts, edges and weights still
    equal the CPU restatement bit for bit, including the diagonal entries every token adds to.  With the sparse counter
    the hot pairs reach the sorted list as one more chunk of records."""
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    kind = _pin(monkeypatch, counter)
    rng = np.random.default_rng(7)
    V, D, L, win = 3000, 1500, 80, 20
    p = 1.0 / np.arange(1, V + 1) ** 1.05
    X = rng.permutation(V)[rng.choice(V, size=(D, L), p=p / p.sum())].astype(np.int32)   # hot words at scattered ids
    lens = rng.integers(L // 2, L + 1, size=D)
    for d in range(D):
        X[d, lens[d]:] = -1
    assert D * L >= 1 << 16 and V > 128                         # the conditions of the hot-pair path
    c_ref, _ = G.sliding_window(X, V, win)
    np.testing.assert_equal(sliding_window_tester(X, V, D, L, win, counter=kind), c_ref)
    coo_ref, w_ref = G.compute_word_word_edges(X, V, win)
    coo, w = compute_word_word_edges(X, V, D, L, win, counter=kind)
    np.testing.assert_equal(coo, coo_ref)
    np.testing.assert_equal(w, w_ref)


@pytest.mark.gpu
def test_gpu_builder_at_a_vocabulary_whose_triangle_no_device_holds(cuda, monkeypatch):
    """V = 400 000: the dense triangle of the reference (graphbuilder.pyx:44,134) and of the dense counter would take
    320 GB -- more than the device has -- and the reference's own uint32 index wraps beyond V = 65 535 (:250).  The sorted
    pair list needs O(chunk + distinct pairs).  What is checked (the CPU oracle restates the reference and is O(V^2) too):
      * the call completes, picks the sparse counter BY ITSELF, and its edges are well formed and in the reference's
        order (upper triangle row-major, (i,j),(j,i) interleaved);
      * on the documents that only use the first 30 000 words (a third of the corpus by construction) the same call at
        V = 400 000 gives, bit for bit, what the DENSE counter gives for them at V = 30 000 -- and that is checked against
        the CPU oracle on a slice;
      * the weights of sampled edges are the reference's float sequence evaluated from counts taken by brute force."""
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, counter_stats
    monkeypatch.delenv("TGCN_WW_COUNTER", raising=False)
    monkeypatch.setenv("TGCN_WW_CHUNK_PAIRS", str(1 << 24))          # several chunks at this size
    rng = np.random.default_rng(400)
    V, Vs, D, L, win = 400_000, 30_000, 150_000, 40, 10
    p = 1.0 / np.arange(1, V + 1) ** 1.0
    ids = rng.choice(V, size=(D, L), p=p / p.sum()).astype(np.int32)
    small = np.arange(D) % 3 == 0                                     # every third document: words below Vs only
    ps = p[:Vs] / p[:Vs].sum()
    ids[small] = rng.choice(Vs, size=(int(small.sum()), L), p=ps).astype(np.int32)
    lens = rng.integers(L // 2, L + 1, size=D)
    ids[np.arange(L)[None, :] >= lens[:, None]] = -1
    st = counter_stats(ids, V, D, L, win)
    assert st["sparse"] and st["n_pairs"] > 10_000_000 and st["n_edges"] > 0, st
    coo, w = compute_word_word_edges(ids, V, D, L, win)
    assert coo.shape == (st["n_edges"], 2) and w.shape == (st["n_edges"],) and coo.shape[0] % 2 == 0
    a, b = coo[0::2], coo[1::2]
    assert (a == b[:, ::-1]).all() and (a[:, 0] < a[:, 1]).all() and (w[0::2] == w[1::2]).all() and (w > 1e-10).all()
    key = a[:, 0].astype(np.int64) * V + a[:, 1]
    assert (np.diff(key) > 0).all()                                   # row-major upper triangle, no pair twice
    # the sub-corpus: same documents, V = 400 000 (sparse) against V = 30 000 (dense)
    sub = ids[small]
    big_c, big_w = compute_word_word_edges(sub, V, sub.shape[0], L, win)
    den_c, den_w = compute_word_word_edges(sub, Vs, sub.shape[0], L, win, counter="dense")
    np.testing.assert_equal(big_c, den_c)
    np.testing.assert_equal(big_w, den_w)
    slice_ = sub[:3000]
    o_c, o_w = G.compute_word_word_edges(slice_, Vs, win)
    s_c, s_w = compute_word_word_edges(slice_, V, slice_.shape[0], L, win)
    np.testing.assert_equal(s_c, o_c)
    np.testing.assert_equal(s_w, o_w)
    # sampled edges of the full run: counts by brute force over the corpus -> the reference's float sequence
    # (graphbuilder.pyx:147-162)
    last = np.array([max(0, min(L - win, int(n) - win)) for n in lens])        # last window start per document
    def count(i, j):
        c = 0
        for d in np.nonzero(((ids == i).any(1)) & ((ids == j).any(1)))[0]:
            x = ids[d]
            for s0 in range(0, last[d] + 1):
                win_ = x[s0:s0 + win]
                for k in range(len(win_)):
                    if win_[k] == -1:
                        break
                    for l in range(k, len(win_)):
                        if win_[l] == -1:
                            break
                        if {int(win_[k]), int(win_[l])} == {i, j}:
                            c += 1
        return c
    nw = np.float32(st["n_windows"])
    rare = np.nonzero(a[:, 0] > 20_000)[0]                            # (pairs of rare words: few documents to walk)
    assert rare.size > 100
    for e in rng.choice(rare, size=4, replace=False):
        i, j = int(a[e, 0]), int(a[e, 1])
        pi, pj, pij = (np.float32(count(i, i)) / nw, np.float32(count(j, j)) / nw, np.float32(count(i, j)) / nw)
        want = np.float32(np.log(np.float64(np.float32(pij / np.float32(pi * pj)))))
        assert w[2 * e] == want, (i, j, w[2 * e], want)


@pytest.mark.gpu
@pytest.mark.parametrize("counter", ["dense", "sparse"])
def test_gpu_builder_edge_cases_and_errors(cuda, counter):
    from pytextgcn_amd.graphbuilder import compute_word_word_edges, sliding_window_tester
    X = np.full((4, 6), -1, dtype=np.int32)                     # only padding: windows but no counts
    coo, w = compute_word_word_edges(X, 5, 4, 6, 3, counter=counter)
    assert coo.shape == (0, 2) and w.shape == (0,)
    assert sliding_window_tester(X, 5, 4, 6, 3, counter=counter).sum() == 0
    coo, w = compute_word_word_edges(np.zeros((0, 6), dtype=np.int32), 5, 0, 6, 3, counter=counter)   # no documents at all
    assert coo.shape == (0, 2) and w.shape == (0,)
    one = np.full((3, 4), 2, dtype=np.int32)                    # one word only: diagonal counts, no pair, no edge
    coo, w = compute_word_word_edges(one, 5, 3, 4, 9, counter=counter)                                 # window > seq_len
    assert coo.shape == (0, 2)
    c = sliding_window_tester(one, 5, 3, 4, 9, counter=counter)
    assert int(c.sum()) == int(c[G._sym_diag_idx(2, 2, 5)] if hasattr(G, "_sym_diag_idx") else c.max()) == 3 * 10
    with pytest.raises(ValueError):
        compute_word_word_edges(X, 5, 4, 6, 3, counter="hashed")
    with pytest.raises(IndexError):
        compute_word_word_edges(np.array([[0, 7]], dtype=np.int32), 5, 1, 2, 2)
    with pytest.raises(ValueError):
        compute_word_word_edges(np.zeros((2, 3), dtype=np.int32), 5, 3, 3, 2)
