#!/usr/bin/env python3
"""Does the MFMA-bound weight-gradient product (dW2 = dropout(H1)^T G2, k_gemm_tn_staged) hide under the fabric-bound backward
SpMM of layer 1 when the two run on different streams?  c4 shapes; the SpMM is the plain transposed launch and the one with
Adam in its epilogue (the epoch's).  Both orders of issue."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import dense, synth  # noqa: E402
from pytextgcn_amd.optim import Adam  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
N, E, F, C = 2_000_000, 50_000_000, 200, 64
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
del g
gen = torch.Generator(device=dev).manual_seed(1)
gout = torch.randn(N, F, device=dev, generator=gen)
H = torch.randn(N, F, device=dev, generator=gen)
G = torch.randn(N, C, device=dev, generator=gen)
W = torch.randn(F, C, device=dev, generator=gen)
seed = dense.new_seed(dev)
_, MASK = dense.gemm_nn(H, W, 0.5, seed, record_mask=True)
p = torch.nn.Parameter(torch.randn(N, F, device=dev, generator=gen) * 0.01)
opt = Adam([p], lr=0.05, amsgrad=True)
dw = torch.empty(N, F, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return round(ev[0].elapsed_time(ev[1]) / reps, 3)


def spmm():
    plan.spmm(gout, None, transpose=True, out=dw)


def spmm_adam():
    opt._fused_update(p, plan, gout)
    opt.state[p]["fused_pending"] = False


def tn():
    return dense.gemm_tn(H, G, 0.5, seed, MASK)


def nt():
    return dense.gemm_nt(G, W, 0.5, seed, note_colsums=True, mask=MASK)


def both(big, small, small_first):
    def run():
        side.wait_stream(main)
        if small_first:
            with torch.cuda.stream(side):
                small()
            big()
        else:
            big()
            with torch.cuda.stream(side):
                small()
        main.wait_stream(side)
    return run


out = {"spmm_ms": timed(spmm), "spmm_adam_ms": timed(spmm_adam), "tn_ms": timed(tn), "nt_ms": timed(nt)}
for name, big in (("spmm", spmm), ("spmm_adam", spmm_adam)):
    out[f"{name}_then_tn_one_stream_ms"] = timed(lambda: (big(), tn()))
    out[f"{name}_with_tn_on_a_side_stream_issued_first_ms"] = timed(both(big, tn, True))
    out[f"{name}_with_tn_on_a_side_stream_issued_second_ms"] = timed(both(big, tn, False))
# the chain of the real backward pass: nt -> spmm_adam on the main stream, tn beside both
out["nt_spmm_adam_tn_one_stream_ms"] = timed(lambda: (nt(), spmm_adam(), tn()))
out["nt_spmm_adam_with_tn_beside_ms"] = timed(both(lambda: (nt(), spmm_adam()), tn, True))
print(json.dumps(out, indent=1))
