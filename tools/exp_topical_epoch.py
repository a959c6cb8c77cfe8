#!/usr/bin/env python3
"""The whole epoch (pytextgcn_amd.train.FlatLoop, bench.py's `epoch_ms_flat_loop`) on a topical corpus at the benchmark shape:
the file sorted by topic, shuffled, and the shuffled one after `pytextgcn_amd.reorder_documents` (plain / words=True)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytextgcn_amd import reorder_documents, synth  # noqa: E402

dev = torch.device("cuda:0")
N, E, F, C = bench.CONFIGS["c4"]
kw = dict(seed=44, device=dev, n_classes=C, n_topics=C)
variants = {"sorted by topic": lambda: synth.word_doc_graph(N, E, doc_order="by_topic", **kw),
            "shuffled": lambda: synth.word_doc_graph(N, E, doc_order="shuffled", **kw),
            "shuffled + reorder_documents": lambda: reorder_documents(synth.word_doc_graph(N, E, doc_order="shuffled", **kw))[0],
            "shuffled + reorder_documents(words=True)":
                lambda: reorder_documents(synth.word_doc_graph(N, E, doc_order="shuffled", **kw), words=True)[0],
            "the benchmark generator (no topics)": lambda: synth.word_doc_graph(N, E, seed=44, device=dev, n_classes=C)}
for name, make in variants.items():
    g = make()
    ms = bench.flat_loop_epoch_ms(g, F, C)
    print(f"{name:45s} epoch {ms:7.2f} ms", flush=True)
    del g
    torch.cuda.empty_cache()
