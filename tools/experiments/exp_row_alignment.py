#!/usr/bin/env python3
"""Experiment: does it pay to pad the operand rows of the c4 SpMM to whole 128-byte lines?  800-byte rows start at
every multiple of 32 bytes within a line and touch 7 or 8 lines (7.25 on average); a 896-byte stride makes it 7."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
N, E, F = 2_000_000, 50_000_000, 200
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
for ld in (200, 224, 256):
    buf = torch.randn(N, ld, device=dev)
    x = buf[:, :F]
    y = torch.empty(N, F, device=dev)
    for _ in range(5):
        plan.spmm(x, None, out=y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        plan.spmm(x, None, out=y)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
    print(json.dumps({"operand_row_stride_floats": ld, "ms_median": round(ts[10], 4), "ms_min": round(ts[0], 4)}), flush=True)
    del buf, x
