#!/usr/bin/env python3
"""What a FEATURE split (every rank keeps the whole operator and F / W columns of the operand: no exchange at all in the
SpMM) would cost per rank: the single-device c4 plan at widths 200 / 100 / 50 / 25, contiguous operands.  Informational:
north_star names the 1-D ROW partition, which is what pytextgcn_amd.sharded implements."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
N, E = 2_000_000, 50_000_000
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


base = None
for F in (200, 100, 64, 52, 50, 28, 25):
    x = torch.randn(N, F, device=dev)
    b = torch.randn(F, device=dev)
    ms = timed(lambda: plan.spmm(x, b))
    base = base or ms
    print(json.dumps({"F": F, "ms": round(ms, 3), "vs_F200": round(base / ms, 2), "ideal": round(200 / F, 2)}), flush=True)
