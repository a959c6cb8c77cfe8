#!/usr/bin/env python3
"""Experiment: how much of the c4 SpMM time is spent on the K longest rows?  Times the operator with
those rows emptied (upper bound of what a separate dense/push treatment of the hot rows could save)
and a streaming pass over the operand (lower bound of what that treatment would cost)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
N, E, F = 2_000_000, 50_000_000, 200
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
full = GraphPlan(g.edge_index, g.edge_attr, N)
rp, col, val = full.export_csr(False)
deg = (rp[1:] - rp[:-1]).long()
row = torch.repeat_interleave(torch.arange(N, device=dev), deg)
x = torch.randn(N, F, device=dev)
y = torch.empty(N, F, device=dev)


def timed(plan, reps=10):
    for _ in range(3):
        plan.spmm(x, None, out=y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        plan.spmm(x, None, out=y)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


print(f"full operator: {timed(full):.3f} ms, nnz {int(deg.sum())}")
order = torch.argsort(deg, descending=True)
for K in (16, 32, 48, 64, 128, 256, 1024):
    hot = torch.zeros(N, dtype=torch.bool, device=dev)
    hot[order[:K]] = True
    keep = ~hot[row]
    p = GraphPlan.from_coo(row[keep], col.long()[keep], val[keep], N, N, with_transpose=False)
    removed = int((~keep).sum())
    print(f"K={K:5d}: removed {removed} entries ({removed * (8 + 4 * F) / 1e9:.2f} GB algorithmic), "
          f"remaining operator {timed(p):.3f} ms")
    p.close()
