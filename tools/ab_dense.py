#!/usr/bin/env python3
"""Interleaved timing of the dense kernels at the c4 layer shapes (several rounds, so that clock and
temperature drift hits every variant alike).  A/B of two library builds: run once per TGCN_LIB_PATH."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import dense  # noqa: E402

N, h, C = 2_000_000, 200, 64
if os.environ.get("AB_SHAPE"):               # AB_SHAPE=8000000,256,64: c5's layer shapes (N rows, hidden, classes)
    N, h, C = (int(v) for v in os.environ["AB_SHAPE"].split(","))
dev = "cuda:0"
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev)
G = torch.randn(N, C, device=dev)
seed = dense.new_seed(dev)
cases = {
    "nn": lambda: dense.gemm_nn(H, W),
    "nn_dropout": lambda: dense.gemm_nn(H, W, 0.5, seed),
    "nt": lambda: dense.gemm_nt(G, W),
    "nt_dropout": lambda: dense.gemm_nt(G, W, 0.5, seed),
    "nt_colsum": lambda: dense.gemm_nt(G, W, note_colsums=True),
    "nt_dropout_colsum": lambda: dense.gemm_nt(G, W, 0.5, seed, note_colsums=True),
    "tn": lambda: dense.gemm_tn(H, G),
    "tn_dropout": lambda: dense.gemm_tn(H, G, 0.5, seed),
}
if hasattr(dense.gemm_nn, "__kwdefaults__") and "record_mask" in dense.gemm_nn.__code__.co_varnames:
    _, MASK = dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
    cases["nn_dropout_record"] = lambda: dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
    cases["tn_dropout_from_record"] = lambda: dense.gemm_tn(H, G, 0.5, seed, MASK)
    if "mask" in dense.gemm_nt.__code__.co_varnames:
        cases["nt_dropout_colsum_from_record"] = lambda: dense.gemm_nt(G, W, 0.5, seed, note_colsums=True, mask=MASK)
times = {k: [] for k in cases}
for rnd in range(6):
    for name, fn in cases.items():
        fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        for i in range(5):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        times[name] += [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
print(os.environ.get("TGCN_LIB_PATH", "default library"), f"N={N} h={h} C={C}")
for name, ts in times.items():
    ts = sorted(ts[5:])          # first round = warm-up
    print(f"  {name:20s} median {ts[len(ts) // 2]:.3f} ms   min {ts[0]:.3f}")
