#!/usr/bin/env python3
"""SpMM on BASELINE.json config c5 (8 M nodes / 200 M edges, power-law degrees, F = 256) on one GPU:
a graph with no word/document structure and a 8.2 GB operand."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.sweep_spmm import time_spmm  # noqa: E402

N, E, F = 8_000_000, 200_000_000, 256
g = synth.power_law_graph(N, E, seed=44, device="cuda:0")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
del g
x = torch.randn(N, F, device="cuda:0")
med, best = time_spmm(plan, x, reps=6)
b = plan.algorithmic_bytes(F)
print(json.dumps({"config": "c5", "ms": round(med, 2), "alg_GB": round(b / 1e9, 1), "alg_TBps": round(b / med / 1e9, 2),
                  "edges_per_s_fwd_only": round(E / med * 1e3), **plan.stats()}))
