#!/usr/bin/env python3
"""The backward SpMM of layer 1 with W1's Adam update in its epilogue (tgcn_spmm_adam) against the two separate
passes (tgcn_spmm transposed + tgcn_adam_step), config c4, F = 200."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.optim import Adam  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

dev = torch.device("cuda:0")
N, E, F = 2_000_000, 50_000_000, 200
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
del g
gen = torch.Generator(device=dev).manual_seed(1)
gout = torch.randn(N, F, device=dev, generator=gen)
p = torch.nn.Parameter(torch.randn(N, F, device=dev, generator=gen) * 0.01)
opt = Adam([p], lr=0.05, amsgrad=True)
st, _ = opt._state_of(p, opt.param_groups[0])
dw = torch.empty(N, F, device=dev)


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
    return ev[0].elapsed_time(ev[1]) / reps


def separate():
    plan.spmm(gout, None, transpose=True, out=dw)
    p.grad = dw
    opt.step()


def fused():
    opt._fused_update(p, plan, gout)
    opt.state[p]["fused_pending"] = False          # a timing loop: no step() between the calls


print(json.dumps({"knobs": {k: v for k, v in os.environ.items() if k.startswith("TGCN_")},
                  "spmm_ms": round(timed(lambda: plan.spmm(gout, None, transpose=True, out=dw)), 3),
                  "spmm_plus_adam_ms": round(timed(separate), 3), "fused_ms": round(timed(fused), 3)}))
