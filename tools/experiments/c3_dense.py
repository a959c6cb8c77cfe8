#!/usr/bin/env python3
"""The layer-2 products at config c3's shapes (N = 1 M, hidden 200, 219 classes: column groups / old tn kernel paths)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytextgcn_amd import dense  # noqa: E402

N, h, C = 1_000_000, 200, 219
dev = "cuda:0"
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev)
from pytextgcn_amd.plan import alloc_padded  # noqa: E402
G = alloc_padded(N, C, dev)                   # as the cross-entropy gradient arrives: rows of 220 floats, zero pad column
G.copy_(torch.randn(N, C, device=dev))
seed = dense.new_seed(dev)
cases = {"nn": lambda: dense.gemm_nn(H, W), "nn_dropout": lambda: dense.gemm_nn(H, W, 0.5, seed),
         "nt_colsum": lambda: dense.gemm_nt(G, W, note_colsums=True),
         "nt_dropout_colsum": lambda: dense.gemm_nt(G, W, 0.5, seed, note_colsums=True),
         "tn": lambda: dense.gemm_tn(H, G), "tn_dropout": lambda: dense.gemm_tn(H, G, 0.5, seed)}
_, MASK = dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
if MASK is not None:
    cases["nn_dropout_record"] = lambda: dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
    cases["tn_dropout_from_record"] = lambda: dense.gemm_tn(H, G, 0.5, seed, MASK)
ref = {"nn": H.double() @ W.double()}
print("tn max rel err vs float64:", float((dense.gemm_tn(H, G).double() - H.double().t() @ G.double()).abs().max()
                                            / (H.double().t() @ G.double()).abs().max()))
print("nn max rel err vs float64:", float((dense.gemm_nn(H, W).double() - ref["nn"]).abs().max() / ref["nn"].abs().max()))
for name, fn in cases.items():
    for _ in range(3):
        fn()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"  {name:18s} median {ts[len(ts) // 2]:.3f} ms")
