#!/usr/bin/env python3
"""Runs only the GPU word-word edge builder (tgcn_wwedges_*) on a synthetic corpus, for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_graphbuild.py [n_docs vocab]
Prints the wall time of each of three calls (host arrays in and out, as the reference's call site
text2graph.py:156-160 has them)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import graphbuilder, synth, text2graph  # noqa: E402
from sklearn.feature_extraction.text import CountVectorizer  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
counter = sys.argv[3] if len(sys.argv) > 3 else None          # "dense" / "sparse": pin the counter (default: by size)
docs, _ = synth.synthetic_corpus(n_docs, vocab, seed=44, min_len=40, max_len=200)
cv = CountVectorizer(min_df=5).fit(docs)
X, L = text2graph._encode_input(docs, 1, cv.vocabulary_, 0, n_docs, None)
V = len(cv.vocabulary_)
graphbuilder.compute_word_word_edges(X[:10], V, 10, L, 20)
for _ in range(3):
    t0 = time.perf_counter()
    coo, w = graphbuilder.compute_word_word_edges(X, V, n_docs, L, 20, counter=counter)
    print(f"docs={n_docs} V={V} L={L} tokens={int((X >= 0).sum())} edges={coo.shape[0]}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
