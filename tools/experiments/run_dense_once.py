#!/usr/bin/env python3
"""Five launches of each dense kernel at the c4 shapes (for counter passes: tools/prof_dense_r04.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytextgcn_amd import dense  # noqa: E402

N, h, C = 2_000_000, 200, 64
dev = "cuda:0"
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev)
G = torch.randn(N, C, device=dev)
seed = dense.new_seed(dev)
_, MASK = dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
for _ in range(5):
    dense.gemm_tn(H, G)
    dense.gemm_tn(H, G, 0.5, seed)
    dense.gemm_nt(G, W)
    dense.gemm_nt(G, W, 0.5, seed, note_colsums=True)
    dense.gemm_nn(H, W)
    # the epoch's own variants: the forward product draws and records the mask, both gradient products read the record
    dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
    dense.gemm_tn(H, G, 0.5, seed, MASK)
    dense.gemm_nt(G, W, 0.5, seed, note_colsums=True, mask=MASK)
torch.cuda.synchronize()
