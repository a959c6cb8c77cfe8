#!/usr/bin/env python3
"""SpMM at the layer-2 widths (F = C) on config c4: sub-group kernel vs the full-wave kernel
(TGCN_SPMM_NARROW=0), one subprocess per setting."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import torch
    from pytextgcn_amd import synth
    from pytextgcn_amd.plan import GraphPlan
    from tools.sweep_spmm import time_spmm
    g = synth.word_doc_graph(2_000_000, 50_000_000, seed=44, device="cuda:0", features="none")
    plan = GraphPlan(g.edge_index, g.edge_attr, 2_000_000)
    for F in (8, 32, 64, 100, 128):
        x = torch.randn(2_000_000, F, device="cuda:0")
        med, _ = time_spmm(plan, x)
        print(json.dumps({"narrow": os.environ.get("TGCN_SPMM_NARROW", "1"), "F": F, "ms": round(med, 3),
                          "alg_TBps": round(plan.algorithmic_bytes(F) / med / 1e9, 2)}), flush=True)
else:
    for v in ("1", "0"):
        subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, TGCN_SPMM_NARROW=v))
