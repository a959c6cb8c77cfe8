#!/usr/bin/env python3
"""How fast is k_spmm_sweep when every gather hits L2?  c4's row structure with the columns folded onto 8192
operand rows (6.5 MB at F = 200): run under `rocprofv3 --kernel-trace --stats` and read the kernel's time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.sweep_spmm import time_spmm  # noqa: E402

N, E = 2_000_000, 50_000_000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fold = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
g = synth.word_doc_graph(N, E, seed=44, device="cuda:0", features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
rp, col, val = plan.export_csr()
row = torch.repeat_interleave(torch.arange(N, device="cuda:0"), (rp[1:] - rp[:-1]).long())
del g
p2 = GraphPlan.from_coo(row, col.long() % fold, val, N, N)
x = torch.randn(N, F, device="cuda:0")
print({"case": f"c4 as is, F={F}", "ms": time_spmm(plan, x)[0], **plan.stats()})
print({"case": f"columns folded mod {fold}", "ms": time_spmm(p2, x)[0], **p2.stats()})
