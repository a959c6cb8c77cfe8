#!/usr/bin/env python3
"""Experiment driver for the SpMM kernel on config c4 (run on the GPU box).

  python tools/sweep_spmm.py phases          word rows vs document rows, timed separately
  python tools/sweep_spmm.py variants        item weights / column blocks,
                                             one subprocess per setting (the knobs are read once)
"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_spmm(plan, x, reps=10, transpose=False):
    y = plan.spmm(x, transpose=transpose)
    for _ in range(2):
        plan.spmm(x, out=y, transpose=transpose)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        plan.spmm(x, out=y, transpose=transpose)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], ts[0]


def graph(config):
    from pytextgcn_amd import synth
    N, E, F = {"c4": (2_000_000, 50_000_000, 200), "c2": (100_000, 2_000_000, 200)}[config]
    g = synth.word_doc_graph(N, E, seed=44, device="cuda:0", features="none")
    return g, N, E, F


def one(config="c4", F=None):
    from pytextgcn_amd.plan import GraphPlan
    g, N, E, F0 = graph(config)
    F = F or F0
    plan = GraphPlan(g.edge_index, g.edge_attr, N)
    x = torch.randn(N, F, device="cuda:0")
    med, best = time_spmm(plan, x)
    print(json.dumps({"knobs": {k: v for k, v in os.environ.items() if k.startswith("TGCN_")}, "F": F,
                      "ms_median": round(med, 4), "ms_min": round(best, 4),
                      "alg_TBps": round(plan.algorithmic_bytes(F) / med / 1e9, 3), **plan.stats()}))


def phases():
    from pytextgcn_amd.plan import GraphPlan
    g, N, E, F = graph("c4")
    V = g.n_vocab
    x = torch.randn(N, F, device="cuda:0")
    for name, rr in [("all", (0, N)), ("word rows", (0, V)), ("doc rows", (V, N))]:
        plan = GraphPlan(g.edge_index, g.edge_attr, N, row_range=rr)
        med, best = time_spmm(plan, x)
        print(json.dumps({"rows": name, "n_rows": plan.n_rows, "nnz": plan.nnz, "ms_median": round(med, 4),
                          "alg_GB": round(plan.algorithmic_bytes(F) / 1e9, 2),
                          "alg_TBps": round(plan.algorithmic_bytes(F) / med / 1e9, 3)}))
        plan.close()


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "one":
        one(*(sys.argv[2:3] or ["c4"]), F=int(sys.argv[3]) if len(sys.argv) > 3 else None)
    elif mode == "phases":
        phases()
    elif mode == "variants":
        settings = [dict(TGCN_LIB_PATH=os.path.join(ROOT, "pytextgcn_amd/lib/libtgcn_old.so")), dict(),
                    dict(TGCN_COL_BLOCK="4096"), dict(TGCN_COL_BLOCK="16384"),
                    dict(TGCN_ITEM_WEIGHT="1024"), dict(TGCN_ITEM_WEIGHT="384")]
        for kv in settings:
            env = dict(os.environ, **kv)
            r = subprocess.run([sys.executable, __file__, "one", "c4"], env=env, capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1][:330] if r.stdout.strip() else r.stderr[-500:], flush=True)
