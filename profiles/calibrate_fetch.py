#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration for the SpMM gather pattern (MI355X_MICROARCH.md 'HBM':
calibrate on a known byte count in your own access pattern).  The operator is a random PERMUTATION
matrix: every row of X (800 B at F = 200) is gathered exactly once, nothing can be re-used, and X
(N x 200 x 4 B = 3.2 GB at N = 4 M) is far larger than the 256 MiB Infinity Cache, so the bytes the
kernel must fetch are known: N*(800 + 8) + 4*(N+1) + 16*items."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

N, F = 4_000_000, 200
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
perm = torch.randperm(N, device=dev, generator=g)
ei = torch.stack([perm, torch.arange(N, device=dev)])           # out[i] = x[perm[i]]
plan = GraphPlan(ei, None, N, add_self_loops=False, normalize=False)
x = torch.randn(N, F, device=dev, generator=g)
y = torch.empty(N, F, device=dev)
for _ in range(3):
    plan.spmm(x, out=y)
torch.cuda.synchronize()
assert torch.equal(y, x[perm])
st = plan.stats()
known = N * (F * 4 + 8) + 4 * (N + 1) + 16 * st["items"]
print(f"known read bytes per launch: {known} ({known/1024:.0f} KiB); write bytes {N*F*4} ({N*F*4/1024:.0f} KiB)")
