#!/usr/bin/env python3
"""Experiment: does clustering the document rows by a shared rare word help the SpMM (temporal
locality of the tail-word gathers)?  Relabels the documents of config c4 and times the SpMM."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.sweep_spmm import time_spmm  # noqa: E402

N, E, F = 2_000_000, 50_000_000, 200
dev = "cuda:0"
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
V = g.n_vocab
ei = g.edge_index
x = torch.randn(N, F, device=dev)
plan = GraphPlan(ei, g.edge_attr, N)
print(json.dumps({"order": "as generated", "ms": time_spmm(plan, x)[0]}))
plan.close()
deg = torch.bincount(ei[1], minlength=N)
# doc -> word edges: source doc (>= V), target word (< V)
m = (ei[0] >= V) & (ei[1] < V)
doc, word = ei[0][m] - V, ei[1][m]
for name, keyfn in [("rarest word", lambda: torch.full((N - V,), 1 << 40, device=dev, dtype=torch.int64).scatter_reduce_(0, doc, deg[word] * V + word, "amin")),
                    ("most frequent word", lambda: torch.zeros(N - V, device=dev, dtype=torch.int64).scatter_reduce_(0, doc, deg[word] * V + word, "amax"))]:
    key = keyfn()
    order = torch.argsort(key, stable=True)              # new position -> old doc
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(N - V, device=dev)
    relabel = torch.cat([torch.arange(V, device=dev), new_id + V])
    ei2 = relabel[ei]
    plan = GraphPlan(ei2, g.edge_attr, N)
    print(json.dumps({"order": f"documents sorted by their {name}", "ms": time_spmm(plan, x)[0]}))
    plan.close()
perm = torch.randperm(N - V, device=dev)
relabel = torch.cat([torch.arange(V, device=dev), perm + V])
plan = GraphPlan(relabel[ei], g.edge_attr, N)
print(json.dumps({"order": "documents shuffled", "ms": time_spmm(plan, x)[0]}))
