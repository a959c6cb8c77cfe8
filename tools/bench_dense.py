#!/usr/bin/env python3
"""Hand-written fp32 MFMA GEMMs vs torch.matmul (rocBLAS / hipBLASLt) at the c4 layer shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import dense  # noqa: E402


def t(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


N, h, C = 2_000_000, 200, 64
dev = "cuda:0"
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev)
G = torch.randn(N, C, device=dev)
for name, mine, ref, flop, byt in [
    ("nn  XW2 = H @ W      ", lambda: dense.gemm_nn(H, W), lambda: H @ W, 2 * N * h * C, 4 * N * (h + C)),
    ("nt  dH  = G @ W^T    ", lambda: dense.gemm_nt(G, W), lambda: G @ W.t(), 2 * N * h * C, 4 * N * (h + C)),
    ("tn  dW  = H^T @ G    ", lambda: dense.gemm_tn(H, G), lambda: H.t() @ G, 2 * N * h * C, 4 * N * (h + C)),
]:
    a, b = t(mine), t(ref)
    print(f"{name} mfma kernel {a:7.3f} ms ({flop / a / 1e9:6.1f} TF/s, {byt / a / 1e6:6.0f} GB/s)   torch.matmul {b:7.3f} ms")

# the same three with the dropout between the layers fused in (tgcn_gemm_*_dropout) against what they
# replace: torch's dropout kernel + GEMM (forward), mask multiply + GEMM (backward)
seed = dense.new_seed(dev)
for name, mine, ref in [
    ("nn  dropout(H) @ W   ", lambda: dense.gemm_nn(H, W, 0.5, seed), lambda: dense.gemm_nn(torch.nn.functional.dropout(H, 0.5, True), W)),
    ("nt  mask * (G @ W^T) ", lambda: dense.gemm_nt(G, W, 0.5, seed), lambda: dense.gemm_nt(G, W) * 2.0),
    ("tn  dropout(H)^T @ G ", lambda: dense.gemm_tn(H, G, 0.5, seed), lambda: dense.gemm_tn(H, G)),
]:
    a, b = t(mine), t(ref)
    print(f"{name} fused {a:7.3f} ms   unfused (separate elementwise pass where there is one) {b:7.3f} ms")
