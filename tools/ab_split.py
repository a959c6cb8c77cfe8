#!/usr/bin/env python3
"""Split-bf16 products (dense.enable_split_gemms) against the fp32 MFMA kernels at the c4 layer shapes:
error against float64 on a 100 k-row sample, then interleaved timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import dense  # noqa: E402

N, h, C = 2_000_000, 200, 64
dev = "cuda:0"
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev) * 0.1
G = torch.randn(N, C, device=dev)
seed = dense.new_seed(dev)


def err(got, ref):
    return ((got.double() - ref).abs().max() / ref.abs().max()).item()


ns = 100_000
ref_nn = H[:ns].double() @ W.double()
ref_nt = G[:ns].double() @ W.double().t()
ref_tn = H[:ns].double().t() @ G[:ns].double()
for on in (False, True):
    dense.enable_split_gemms(on)
    print(f"split={on}: max-norm error vs float64  nn {err(dense.gemm_nn(H[:ns], W), ref_nn):.3e}   nt {err(dense.gemm_nt(G[:ns], W), ref_nt):.3e}"
          f"   tn {err(dense.gemm_tn(H[:ns], G[:ns]), ref_tn):.3e}   tn (99 991 rows) {err(dense.gemm_tn(H[:99_991], G[:99_991]), H[:99_991].double().t() @ G[:99_991].double()):.3e}")
eye = torch.eye(200, device=dev)
b = ((torch.arange(200 * 64, device=dev, dtype=torch.float32).reshape(200, 64) % 97) - 11 * torch.arange(64, device=dev)) * 1.37
dense.enable_split_gemms(True)
print("identity products exact:", torch.equal(dense.gemm_nn(eye, b), b), torch.equal(dense.gemm_nt(torch.eye(64, device=dev), b), b.t().contiguous()),
      torch.equal(dense.gemm_tn(eye, b), b))
cases = {
    "nn": lambda: dense.gemm_nn(H, W),
    "nn_dropout": lambda: dense.gemm_nn(H, W, 0.5, seed),
    "nt": lambda: dense.gemm_nt(G, W),
    "nt_dropout_colsum": lambda: dense.gemm_nt(G, W, 0.5, seed, note_colsums=True),
    "tn": lambda: dense.gemm_tn(H, G),
    "tn_dropout": lambda: dense.gemm_tn(H, G, 0.5, seed),
}
times = {(k, on): [] for k in cases for on in (False, True)}
for rnd in range(5):
    for on in (False, True):
        dense.enable_split_gemms(on)
        for name, fn in cases.items():
            fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            ev[0].record()
            for i in range(5):
                fn()
                ev[i + 1].record()
            torch.cuda.synchronize()
            times[(name, on)] += [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
dense.enable_split_gemms(False)
for name in cases:
    a, b_ = sorted(times[(name, False)][5:]), sorted(times[(name, True)][5:])
    print(f"  {name:20s} fp32 MFMA {a[len(a) // 2]:.3f} ms   split bf16 {b_[len(b_) // 2]:.3f} ms")
