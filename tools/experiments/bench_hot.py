#!/usr/bin/env python3
"""A/B of the dense hot block on c4: TGCN_HOT_ROWS=0 / default, F = 200 and 64.
Run each setting in its own process (the knobs are read once):  python tools/experiments/bench_hot.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from pytextgcn_amd import synth
from pytextgcn_amd.plan import GraphPlan
dev = torch.device("cuda:0")
N, E = 2_000_000, 50_000_000
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
p = GraphPlan(g.edge_index, g.edge_attr, N)
for F in (200, 64):
    x = torch.randn(N, F, device=dev); y = torch.empty(N, F, device=dev)
    for _ in range(3): p.spmm(x, None, out=y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(20): p.spmm(x, None, out=y)
    ev[1].record(); torch.cuda.synchronize()
    print(f"  F={F}: {ev[0].elapsed_time(ev[1]) / 20:.3f} ms  checksum {float(y.double().sum()):.6e}")
''' % ROOT
for label, env in [("no hot block", {"TGCN_HOT_ROWS": "0"}), ("hot block", {})]:
    print(label, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, **env}, check=True)
