#!/usr/bin/env python3
"""Hand-written fp32 MFMA GEMMs vs torch.matmul (rocBLAS / hipBLASLt) at the c4 layer shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

# nt with the column sums of the result taken in the epilogue (tgcn_gemm_nt_colsum) against product + tgcn_colsum
from pytextgcn_amd.plan import colsum  # noqa: E402
Wt = torch.randn(h, C, device=dev)          # [n = h, k = C]
for name, mine, ref in [
    ("nt + column sums, no mask   ", lambda: dense.gemm_nt(G, Wt, note_colsums=True), lambda: colsum(dense.gemm_nt(G, Wt))),
    ("nt + column sums, dropout   ", lambda: dense.gemm_nt(G, Wt, 0.5, seed, note_colsums=True), lambda: colsum(dense.gemm_nt(G, Wt, 0.5, seed))),
]:
    a, b = t(mine), t(ref)
    print(f"{name} fused {a:7.3f} ms   product then tgcn_colsum {b:7.3f} ms")
for F in (200, 64):
    M = torch.randn(N, F, device=dev)
    print(f"tgcn_colsum [{N} x {F}]: {t(lambda: colsum(M)):7.3f} ms")
from pytextgcn_amd.functional import masked_cross_entropy  # noqa: E402
y = torch.randint(0, C, (N,), device=dev)
mask = torch.rand(N, device=dev) < 0.7
lg = torch.randn(N, C, device=dev, requires_grad=True)
print(f"masked CE, loss + predictions (eval): {t(lambda: masked_cross_entropy(lg.detach(), y, mask, return_pred=True)):7.3f} ms")
print(f"masked CE, loss + gradient + bias gradient (train forward): {t(lambda: masked_cross_entropy(lg, y, mask)):7.3f} ms")
