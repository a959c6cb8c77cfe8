#!/usr/bin/env python3
"""Whole-graph plans below the 24 M-entry cut of `item_weight_for` (plan.hip): work items of 128 (the default there since
round 4, measured on the per-rank operators of the partition) against 384, for F = 200 and F = 64, on c2 and on two
real-corpus-sized graphs.  The knob is read when a plan is created, so one process builds both plans."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"c2 (100 k nodes, 2 M edges)": (100_000, 2_000_000), "20NG-sized (60 k nodes, 5 M edges)": (60_000, 5_000_000),
          "R8-sized (15 k nodes, 1 M edges)": (15_000, 1_000_000), "400 k nodes, 10 M edges": (400_000, 10_000_000)}


def time_ms(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for name, (N, E) in shapes.items():
    g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
    plans = {}
    for w in ("128", "384"):
        os.environ["TGCN_ITEM_WEIGHT"] = w
        plans[w] = GraphPlan(g.edge_index, g.edge_attr, N)
    os.environ.pop("TGCN_ITEM_WEIGHT", None)
    for F in (200, 64):
        x = torch.randn(N, F, device=dev)
        y = torch.empty(N, F, device=dev)
        res = {w: [] for w in plans}
        for rnd in range(5):                               # interleaved rounds: drift hits both alike
            for w, p in plans.items():
                res[w].append(time_ms(lambda: p.spmm(x, out=y)))
        med = {w: sorted(v)[len(v) // 2] for w, v in res.items()}
        print(f"{name}  F={F}: item weight 128 {med['128']:.4f} ms, 384 {med['384']:.4f} ms  "
              f"(128 / 384 = {med['128'] / med['384']:.3f}; items {plans['128'].stats()['items']} / {plans['384'].stats()['items']})",
              flush=True)
    for p in plans.values():
        p.close()
