#!/usr/bin/env python3
"""One-off costs: plan construction (what replaces 4 gcn_norm calls per epoch in the reference) on
the benchmark graphs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

for name, N, E in [("c2", 100_000, 2_000_000), ("c4", 2_000_000, 50_000_000)]:
    g = synth.word_doc_graph(N, E, seed=44, device="cuda:0", features="none")
    for mode in ("reference", "accurate"):               # the package default first
        GraphPlan(g.edge_index, g.edge_attr, N, degree_sum=mode).close()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum=mode)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            st = p.stats()
            p.close()
        print(f"{name} [{mode}]: plan build {min(ts) * 1e3:.1f} ms  {st}", flush=True)
