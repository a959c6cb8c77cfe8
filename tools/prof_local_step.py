#!/usr/bin/env python3
"""Only the local side of ONE rank's distributed SpMM (c4, F = 200), for rocprofv3 passes: rank R of a W-rank partition is
cut on this GPU (`ShardedGraph.for_rank`: no process group) and `local_step` -- every launch of a distributed SpMM but the
collectives -- runs STEPS times and nothing else.  tools/collect_local_step_traffic.sh puts the FETCH_SIZE / WRITE_SIZE
passes of this command into profiles/traffic.json[c4_n<W>]: what rank R's operators move over the fabric per SpMM.
  python3 tools/prof_local_step.py <world> [rank] [steps] [c4|c5]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.sharded import ShardedGraph  # noqa: E402

world = int(sys.argv[1])
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = sys.argv[4] if len(sys.argv) > 4 else "c4"
dev = torch.device("cuda:0")
if cfg == "c5":                       # power law, no word / document structure: every node a hub, B_r only (true halo)
    N, E, F = 8_000_000, 200_000_000, 256
    g = synth.power_law_graph(N, E, seed=44, device=dev)
    hubs = None
else:
    N, E, F = 2_000_000, 50_000_000, 200
    g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
    hubs = torch.arange(N, device=dev) < g.n_vocab
# the forward operators only (`symmetric=True` skips the pair of M^T: same structure, same traffic)
sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, world, rank, hubs=hubs, symmetric=True, degree_sum="reference")
d = sg.dirs[0]
hp, rp, W = sg.hp, sg.rp, world
x_local = torch.randn(sg.n_local, F, device=dev)
bias = torch.randn(F, device=dev)
gathered = torch.randn(W * hp, F, device=dev)
rs_out = torch.randn(hp, F, device=dev)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(steps):
    sg.local_step(d, x_local, bias, gathered, rs_out)
ev[1].record()
torch.cuda.synchronize()
print(f"LOCAL_STEPS {steps} world {world} rank {rank} ms_per_step {ev[0].elapsed_time(ev[1]) / steps:.4f} "
      f"A_nnz {d.A.nnz if d.A is not None else 0} B_nnz {d.B.nnz} hp {hp} rp {rp}", flush=True)
