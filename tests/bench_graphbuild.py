#!/usr/bin/env python3
"""Word-word edge construction: the GPU builder (libtgcn.so) next to the CPU restatement of the
reference's Cython loops (oracle/graphbuilder_oracle.c) and, when oracle/_ref was built, the
reference module itself.  The reference's `benchmark_graph.py` has no timer; this is its counterpart."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this script lives in tests/: it uses the CPU oracle)
sys.path.insert(0, ROOT)
from oracle import graphbuilder_py as G  # noqa: E402
from pytextgcn_amd import graphbuilder, synth, text2graph  # noqa: E402
from sklearn.feature_extraction.text import CountVectorizer  # noqa: E402

for n_docs, vocab in [(1000, 1500), (20000, 6000), (100000, 20000)]:
    docs, _ = synth.synthetic_corpus(n_docs, vocab, seed=44, min_len=40, max_len=200)
    cv = CountVectorizer(min_df=5).fit(docs)
    X, L = text2graph._encode_input(docs, 1, cv.vocabulary_, 0, n_docs, None)
    V = len(cv.vocabulary_)
    graphbuilder.compute_word_word_edges(X[:10], V, 10, L, 20)           # warm-up (library load)
    t0 = time.perf_counter()
    coo, w = graphbuilder.compute_word_word_edges(X, V, n_docs, L, 20, counter="dense")
    t_gpu = time.perf_counter() - t0
    graphbuilder.compute_word_word_edges(X[:10], V, 10, L, 20, counter="sparse")
    t0 = time.perf_counter()
    coo_s, w_s = graphbuilder.compute_word_word_edges(X, V, n_docs, L, 20, counter="sparse")
    t_sparse = time.perf_counter() - t0
    assert np.array_equal(coo, coo_s) and np.array_equal(w, w_s)
    st = graphbuilder.counter_stats(X, V, n_docs, L, 20, counter="sparse")
    line = (f"docs={n_docs} V={V} L={L} tokens={int((X >= 0).sum())} edges={coo.shape[0]}  GPU dense triangle {t_gpu*1e3:8.1f} ms"
            f" ({V * (V + 1) // 2 * 4 / 1e9:.2f} GB)   sorted pair list {t_sparse*1e3:8.1f} ms ({st['n_pairs']} distinct pairs, "
            f"{st['n_pairs'] * 12 / 1e9:.2f} GB)")
    if n_docs <= 20000:
        t0 = time.perf_counter()
        coo2, w2 = G.compute_word_word_edges(X, V, 20)
        t_cpu = time.perf_counter() - t0
        assert np.array_equal(coo, coo2) and np.array_equal(w, w2)
        line += f"   CPU restatement (1 core) {t_cpu*1e3:9.1f} ms  ({t_cpu / t_gpu:.0f}x)"
        ref = G.load_ref()
        if ref is not None and V < 65535:
            from tests.golden.make_graphbuilder_golden import ref_safe_vocab
            if ref_safe_vocab(V):
                t0 = time.perf_counter()
                ref.compute_word_word_edges(X, V, n_docs, L, 20)
                line += f"   reference module {1e3*(time.perf_counter() - t0):9.1f} ms"
    print(line, flush=True)

# beyond the dense triangle: Zipf tokens over a vocabulary of 400 000 (the triangle would take 320 GB; the reference's index
# wraps beyond 65 535, graphbuilder.pyx:250) -- the sorted-pair-list counter alone (times include the host <-> device copies)
rng = np.random.default_rng(400)
for V, D, L, win in [(200_000, 500_000, 40, 10), (400_000, 1_000_000, 40, 10), (400_000, 1_000_000, 64, 20)]:
    p = 1.0 / np.arange(1, V + 1)
    X = rng.choice(V, size=(D, L), p=p / p.sum()).astype(np.int32)
    lens = rng.integers(L // 2, L + 1, size=D)
    X[np.arange(L)[None, :] >= lens[:, None]] = -1
    t0 = time.perf_counter()
    st = graphbuilder.counter_stats(X, V, D, L, win)
    t = time.perf_counter() - t0
    print(f"docs={D} V={V} L={L} window={win} tokens={int((X >= 0).sum())} edges={st['n_edges']}  sorted pair list "
          f"{t*1e3:8.1f} ms ({st['n_pairs']} distinct pairs, {st['n_pairs'] * 12 / 1e9:.2f} GB; a dense triangle would take "
          f"{V * (V + 1) // 2 * 4 / 1e9:.0f} GB)", flush=True)
