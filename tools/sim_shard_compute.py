#!/usr/bin/env python3
"""Per-rank COMPUTE of the row-partitioned SpMM at world sizes 2/4/8, measured on ONE GPU: builds the local operators
A_r, B_r of one rank of the c4 graph with `ShardedGraph.for_rank` (no process group, no collectives) and times them --
A_r whole and as 4 row chunks (TGCN_RS_CHUNKS), B_r at the hidden width (split operand) and at the class width (one
buffer), and the whole local side of `ShardedGraph.spmm` (`local_step`: every launch of one distributed SpMM but the
collectives themselves).  Gives the compute side of the N-GPU step; the exchange comes on top and can only be measured
on a multi-GPU node.   python tools/sim_shard_compute.py [world ...] [--rank R]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.sharded import ShardedGraph  # noqa: E402

dev = torch.device("cuda:0")
N, E = 2_000_000, 50_000_000
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
hubs = torch.arange(N, device=dev) < g.n_vocab
argv = [a for a in sys.argv[1:]]
rank = 0
weights = [None]
if "--item-weights" in argv:             # sweep TGCN_ITEM_WEIGHT (read at plan creation): --item-weights 384,192,96
    i = argv.index("--item-weights")
    weights = [int(v) for v in argv[i + 1].split(",")]
    del argv[i:i + 2]
if "--rank" in argv:
    i = argv.index("--rank")
    rank = int(argv[i + 1])
    del argv[i:i + 2]


def timed(fn, reps=20):
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


for world, T in [(w, t) for w in ([int(w) for w in argv] or [2, 4, 8]) for t in weights]:
    if T is not None:
        os.environ["TGCN_ITEM_WEIGHT"] = str(T)
    sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, world, min(rank, world - 1), hubs=hubs, symmetric=True)
    d = sg.dirs[0]
    A, B = d.A, d.B
    hp, rp, W = sg.hp, sg.rp, world
    rec = {"world": world, "item_weight": T, "items_A": A.stats()["items"], "items_B": B.stats()["items"], "rank": sg.rank, "hub_rows_per_rank": hp, "regular_rows": rp, "A_nnz": A.nnz, "B_nnz": B.nnz,
           "A_hot": A.stats()["hot_rows"], "B_hot": B.stats()["hot_rows"]}
    # A_r as K = 4 row chunks (hub slot s -> chunk s % 4), as ShardedGraph.set_rs_chunks cuts them
    K = 4
    row, col, w = d.A_entries
    owner, slot = row // hp, row % hp
    ck = (hp + K - 1) // K
    chunks = []
    for k in range(K):
        sel = (slot % K) == k
        chunks.append(sg.engine.make_op(owner[sel] * ck + slot[sel] // K, col[sel], w[sel], W * ck, rp))
    rec["A_chunk_hot_rows"] = [c.stats()["hot_rows"] for c in chunks]
    for F in (200, 64):
        x_local = torch.randn(sg.n_local, F, device=dev)
        bias = torch.randn(F, device=dev)
        xbuf = torch.randn(W * hp + rp, F, device=dev)
        xbuf[W * hp:] = x_local[hp:]
        rec[f"A_ms_F{F}"] = round(timed(lambda: A.spmm(x_local[hp:])), 3)
        rec[f"A_4chunks_ms_F{F}"] = round(timed(lambda: [c.spmm(x_local[hp:]) for c in chunks]), 3)
        if F > 128:
            rec[f"B_ms_F{F}"] = round(timed(lambda: B.spmm(xbuf[:W * hp], None, x2=x_local[hp:])), 3)
        else:
            rec[f"B_ms_F{F}"] = round(timed(lambda: B.spmm(xbuf, None)), 3)
            rec[f"B_split_operand_ms_F{F}"] = round(timed(lambda: B.spmm(xbuf[:W * hp], None, x2=x_local[hp:])), 3)
        # everything one distributed SpMM launches on this rank except the collectives (their results stand in)
        rs_out = torch.randn(hp, F, device=dev)
        rec[f"local_step_ms_F{F}"] = round(timed(lambda: sg.local_step(d, x_local, bias, xbuf[:W * hp], rs_out)), 3)
        rec[f"exchange_MB_each_way_F{F}"] = round((W - 1) * hp * F * 4 / 1e6, 1)
    print(json.dumps(rec), flush=True)
    for op in [A, B] + chunks:
        op.close()
