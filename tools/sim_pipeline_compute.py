#!/usr/bin/env python3
"""Compute side of the PIPELINED exchange of a graph without hub structure (BASELINE config c5: 8 M nodes / 200 M edges,
power law, h = 256), measured on ONE GPU: builds what rank R of W would hold (`ShardedGraph.for_rank(hubs=None,
halo_lists=True)`: no process group, no collectives) and times, at the hidden and the class width,

  * B_r as ONE operator on the gathered block (what the halo / collective forms launch after the whole gather), and the
    halo form's extra kernels (pack + scatter of the referenced rows);
  * the column blocks of `_Pipeline` -- own-column block + K stage blocks added with `tgcn_spmm_acc` -- for K = 1, 2, 4, 8
    ("slices": 1 / K of every peer's rows per stage), for the per-peer scheme (W - 1 stages) and with the UNPACKED prefix
    (`prefix="auto"`: the degree-ordered slots nearly every peer reads travel as K contiguous ranges, nothing packed; one
    packed stage for the rest), each block alone and the whole sequence back to back.

With T_x the time one rank's halo rows need on the links, the pipelined step is about
    max(T_own, T_x / K) + sum_k max(T_block_k, T_x / K)   (stage k + 1 travels under block k)
against T_x + T_B for the halo form; the record carries both for T_x at link peak (7 links x 153 GB/s).
    python tools/sim_pipeline_compute.py [--world 8] [--rank 7] [--config c5|c5s] [--widths 256,64]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.sharded import ShardedGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=7)
ap.add_argument("--config", default="c5")
ap.add_argument("--widths", default="256,64")
ap.add_argument("--stages", default="1,2,4,8")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, E = {"c5": (8_000_000, 200_000_000), "c5s": (1_000_000, 25_000_000)}[args.config]
g = synth.power_law_graph(N, E, seed=44, device=dev)
W, R = args.world, args.rank
print(f"# {args.config}: N = {N}, E = {E}; rank {R} of {W}", flush=True)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return round(ev[0].elapsed_time(ev[1]) / reps, 3)


sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, W, R, hubs=None, halo_lists=True)
hp = sg.hp
link_GBps = 153.0
for di, d in enumerate(sg.dirs[:1]):
    halo_rows = sum(d.need_counts_l) - d.need_counts_l[R]
    rec = {"config": args.config, "world": W, "rank": R, "rows_per_rank": hp, "B_nnz": d.B.nnz,
           "halo_rows_received": halo_rows, "of_all_remote_rows": round(halo_rows / ((W - 1) * hp), 3),
           "auto_stages": sg.pipe_stages}
    for F in [int(v) for v in args.widths.split(",")]:
        x_local = torch.randn(hp, F, device=dev)
        bias = torch.randn(F, device=dev)
        xbuf = torch.zeros(W * hp, F, device=dev)
        xbuf[d.need_cols] = torch.randn(d.need_cols.numel(), F, device=dev)
        out = {"T_x_ms_at_link_peak": round(halo_rows * F * 4 / ((W - 1) * link_GBps * 1e6), 3)}
        out["B_one_operator_ms"] = timed(lambda: d.B.spmm(xbuf, bias))
        recv = torch.randn(halo_rows, F, device=dev)
        remote = d.need_cols[(d.need_cols // hp) != R]
        out["halo_pack_plus_scatter_ms"] = timed(lambda: (sg._rows_gather(x_local, d.send_slots),
                                                          sg._rows_scatter(xbuf, remote, recv)))
        variants = [(f"slices{k}", int(k), "slices", 0) for k in args.stages.split(",")] + [("peer", 0, "peer", 0)] + \
            [(f"slices{k}+prefix", int(k), "slices", "auto") for k in args.stages.split(",") if int(k) > 1]
        for label, K, scheme, prefix in variants:
            sg.set_pipeline(K, scheme, prefix=prefix)
            pipe = sg._pipeline(d)
            packed = [st for st in pipe.stages if st.span is None]
            bufs = [torch.randn(sum(st.recv_counts), F, device=dev) for st in pipe.stages]
            packs = timed(lambda: [sg._rows_gather(x_local, st.send_slots) for st in packed])
            own = timed(lambda: pipe.own.spmm(x_local, bias))
            y = pipe.own.spmm(x_local, bias)
            blocks = [timed(lambda st=st, b=b: st.op.spmm(b, out=y, accumulate=True)) if st.op is not None else 0.0
                      for st, b in zip(pipe.stages, bufs)]

            def whole():
                yy = pipe.own.spmm(x_local, bias)
                for st, b in zip(pipe.stages, bufs):
                    if st.op is not None:
                        st.op.spmm(b, out=yy, accumulate=True)
                return yy
            seq = timed(whole)
            # transfer at link peak: the rows that ARRIVE from the other ranks (an unread prefix row travels too)
            rows_k = [(W - 1) * (st.span[1] - st.span[0]) if st.span is not None else sum(st.recv_counts) for st in pipe.stages]
            arriving = sum(rows_k)
            Tx = arriving * F * 4 / ((W - 1) * link_GBps * 1e6)
            pers = [Tx * rk / max(1, arriving) for rk in rows_k]                     # a stage's share of the transfer
            # stage 0 travels under the own block (an unpacked one leaves at once, a packed one after its pack), stage k + 1
            # under block k; the last block has nothing left to hide
            model = max(own, pers[0]) + sum(max(b, p_next) for b, p_next in zip(blocks[:-1], pers[1:])) + \
                (blocks[-1] if blocks else 0.0)
            out[label] = {"stages": len(pipe.stages), "unpacked_stages": len(pipe.stages) - len(packed),
                          "prefix_rows_per_rank": pipe.prefix, "own_nnz": pipe.own_nnz,
                          "stage_nnz": [st.nnz for st in pipe.stages], "rows_arriving": arriving,
                          "rows_read": sum(st.rows_read for st in pipe.stages),
                          "transfer_ms_at_link_peak": round(Tx, 3), "pack_ms": packs, "own_ms": own,
                          "stage_block_ms": blocks, "sequence_ms": seq, "compute_side_ms": round(packs + seq, 3),
                          "step_ms_model_at_link_peak": round(model + packs, 3)}
            sg.drop_unused_pipelines()
        out["halo_step_ms_model_at_link_peak"] = round(out["T_x_ms_at_link_peak"] + out["B_one_operator_ms"]
                                                       + out["halo_pack_plus_scatter_ms"], 3)
        rec[f"F{F}"] = out
        del xbuf, recv
    print(json.dumps(rec), flush=True)
