#!/usr/bin/env python3
"""Per-rank COMPUTE of the row-partitioned SpMM at world sizes 2/4/8, measured on one GPU: builds the
local operators A_r, B_r of pytextgcn_amd.sharded for a few ranks of the c4 graph (no process group, no
collectives) and times them.  Gives the compute side of the N-GPU step; the exchange (all-gather +
reduce-scatter of the V x F hub block) comes on top and can only be measured on a multi-GPU node."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.sharded import HipEngine, Partition, ShardedGraph  # noqa: E402

dev = torch.device("cuda:0")
N, E, F = 2_000_000, 50_000_000, 200
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
hubs = torch.arange(N, device=dev) < g.n_vocab
eng = HipEngine()
row, col, val, sym = eng.normalized_triplets(g.edge_index, g.edge_attr, N, True, True, False)


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


for world in (2, 4, 8):
    part = Partition(g.edge_index, N, world, hubs)
    for rank in sorted({0, world - 1}):
        sg = ShardedGraph.__new__(ShardedGraph)
        sg.world, sg.rank, sg.engine, sg.part = world, rank, eng, part
        sg.hp, sg.rp, sg.n_local = part.hp, part.rp, part.n_local
        A, B = sg._local_ops(row, col, val)
        x_local = torch.randn(sg.n_local, F, device=dev)
        xbuf = torch.randn(world * sg.hp, F, device=dev)
        tA = timed(lambda: A.spmm(x_local[sg.hp:]))
        tB = timed(lambda: B.spmm(xbuf, None, x2=x_local[sg.hp:]))
        print(json.dumps({"world": world, "rank": rank, "hub_rows_per_rank": sg.hp, "regular_rows": sg.rp,
                          "A_nnz": A.nnz, "B_nnz": B.nnz, "A_hot": A.stats()["hot_rows"], "B_hot": B.stats()["hot_rows"],
                          "A_ms": round(tA, 3), "B_ms": round(tB, 3), "compute_ms": round(tA + tB, 3),
                          "exchange_MB_each_way": round(2 * (world - 1) * sg.hp * F * 4 / 1e6, 1)}), flush=True)
        A.close(); B.close()
