#!/usr/bin/env python3
"""How many hub rows the halo form of the distributed SpMM moves, per rank and SpMM, against the whole-block forms --
computed on the host from pytextgcn_amd.sharded.Partition (no GPU, no process group): config c4 (or c5 with `c5`).
  python tools/sim_halo_rows.py [c4|c5] [world ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import sharded, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
worlds = [int(w) for w in sys.argv[2:]] or [2, 4, 8]
if cfg == "c5":
    N, E = 8_000_000, 200_000_000
    g = synth.power_law_graph(N, E, seed=44)
    hubs = None
else:
    N, E = 2_000_000, 50_000_000
    g = synth.word_doc_graph(N, E, seed=44, features="none")
    hubs = torch.arange(N) < g.n_vocab
s, t = g.edge_index[0].contiguous(), g.edge_index[1].contiguous()
for W in worlds:
    p = sharded.Partition(g.edge_index, N, W, hubs)
    hp = p.hp
    rows = {"world": W, "hub_rows_per_rank": hp, "whole_block_rows_received": (W - 1) * hp}
    gather, reduce_ = [], []
    for r in ([0, W - 1] if W > 2 else [0, 1]):
        t_mine, s_mine = p.owner[t] == r, p.owner[s] == r
        t_hub, s_hub = p.hub_mask[t], p.hub_mask[s]
        # B_r: own rows <- hub columns (+ the own hubs' loops); the columns it references in other ranks' shards
        b = t_mine & s_hub
        need = torch.zeros(W * hp, dtype=torch.bool)
        need[p.hub_col[s[b]]] = True
        own = torch.zeros(W * hp, dtype=torch.bool)
        own[r * hp:(r + 1) * hp] = True
        gather.append(int((need & ~own).sum()))
        # A_r: hub rows <- own regular columns; the partial rows it sends to other ranks
        a = t_hub & ~s_hub & s_mine
        touch = torch.zeros(W * hp, dtype=torch.bool)
        touch[p.hub_col[t[a]]] = True
        reduce_.append(int((touch & ~own).sum()))
    rows["halo_gather_rows_received"] = gather
    rows["halo_reduce_rows_sent"] = reduce_
    rows["gather_fraction"] = round(max(gather) / ((W - 1) * hp), 3)
    rows["reduce_fraction"] = round(max(reduce_) / ((W - 1) * hp), 3) if hubs is not None else None
    print(json.dumps(rows), flush=True)
