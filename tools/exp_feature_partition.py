#!/usr/bin/env python3
"""What a FEATURE-dimension partition of the SpMM pair would cost per rank (run on the GPU box): the whole c4 operator at
widths F / W.  Rank r of W would own the columns [r F / W, (r + 1) F / W) of the operand and of the result -- no exchange in
the SpMM at all, every rank runs the WHOLE operator at a narrow width.  north_star prescribes the row partition; this
measures the alternative's compute so that the choice is on record."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.sweep_spmm import graph, time_spmm  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

g, N, E, F0 = graph("c4")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
base = None
for W, F in ((1, 200), (2, 100), (4, 50), (4, 52), (8, 25), (8, 28), (8, 32)):
    x = torch.randn(N, F, device="cuda:0")
    fwd, _ = time_spmm(plan, x)
    bwd, _ = time_spmm(plan, x, transpose=True)
    pair = fwd + bwd
    base = base or pair
    print(json.dumps({"ranks": W, "width_per_rank": F, "fwd_ms": round(fwd, 3), "bwd_ms": round(bwd, 3),
                      "pair_ms": round(pair, 3), "speedup_vs_one_gpu_if_nothing_else_costs": round(base / pair, 2)}), flush=True)
    del x
