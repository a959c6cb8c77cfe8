#!/usr/bin/env python3
"""Where is the SpMM kernel's own ceiling?  Same row structure as config c4, but the columns are
folded into a small set so that every gathered row is served by L2 (or by the Infinity Cache)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.sweep_spmm import time_spmm  # noqa: E402

N, E = 2_000_000, 50_000_000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200      # python tools/ceiling_spmm.py [F]
g = synth.word_doc_graph(N, E, seed=44, device="cuda:0", features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
rp, col, val = plan.export_csr()
row = torch.repeat_interleave(torch.arange(N, device="cuda:0"), (rp[1:] - rp[:-1]).long())
x = torch.randn(N, F, device="cuda:0")
print(json.dumps({"case": "c4 as is", "ms": time_spmm(plan, x)[0]}))
for fold in (1024, 4096, 32768, 200_000):
    p2 = GraphPlan.from_coo(row, col.long() % fold, val, N, N)
    med, best = time_spmm(p2, x)
    print(json.dumps({"case": f"columns folded mod {fold} ({fold*F*4/1e6:.1f} MB of X live)", "ms": round(med, 3),
                      "alg_TBps": round(p2.algorithmic_bytes(F) / med / 1e9, 2)}))
    p2.close()
