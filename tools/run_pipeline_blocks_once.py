#!/usr/bin/env python3
"""One rank's local work of the pipelined exchange, a few times, for counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
rank R of W of c5 built without peers (`ShardedGraph.for_rank`), then per repetition `B_r` as ONE operator on the gathered
block, and the pipeline's packs + own-column block + K accumulated stage blocks (`tgcn_spmm_acc`), operands standing in.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/run_pipeline_blocks_once.py [K] [prefix]
Summary: python tools/run_pipeline_blocks_once.py --summarize OUT_FETCH OUT_WRITE"""
import csv
import glob
import os
import sys

REPS, F, W, R = 3, 256, 8, 7


def summarize(fetch_dir, write_dir):
    def load(d):
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        t = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
        dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t))}
        return rows, dur
    out = {}
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows, dur = load(d)
        for r in rows:
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].replace("tgcn::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            e = out.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0, "us": 0.0})
            e[name] += float(r["Counter_Value"])
            if name == "FETCH_SIZE":
                e["n"] += 1
                e["us"] += dur.get(r["Dispatch_Id"], 0.0)
    print("| kernel | dispatches per repetition | ms per repetition | fabric GB per repetition (2 x FETCH_SIZE + WRITE_SIZE) | TB/s |")
    print("|---|---|---|---|---|")
    for k, e in sorted(out.items()):
        if not any(s in k for s in ("k_spmm", "k_rows")):
            continue
        gb = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024 / 1e9 / REPS
        ms = e["us"] / 1e3 / REPS
        print(f"| `{k}` | {e['n'] / REPS:.1f} | {ms:.3f} | {gb:.2f} | {gb / ms if ms else 0:.2f} |")


if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
    summarize(sys.argv[2], sys.argv[3])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.sharded import ShardedGraph  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prefix = sys.argv[2] if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
g = synth.power_law_graph(8_000_000, 200_000_000, seed=44, device=dev)
sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, 8_000_000, W, R, hubs=None, halo_lists=True)
del g
d = sg.dirs[0]
hp = sg.hp
x_local = torch.randn(hp, F, device=dev)
xbuf = torch.zeros(W * hp, F, device=dev)
xbuf[d.need_cols] = torch.randn(d.need_cols.numel(), F, device=dev)
sg.set_pipeline(K, "slices", prefix=prefix if prefix == "auto" else int(prefix))
pipe = sg._pipeline(d)
bufs = [torch.randn(sum(st.recv_counts), F, device=dev) for st in pipe.stages]
torch.cuda.synchronize()
for _ in range(REPS):
    d.B.spmm(xbuf, None)
    for st in pipe.stages:
        if st.span is None:
            sg._rows_gather(x_local, st.send_slots)
    y = pipe.own.spmm(x_local, None)
    for st, b in zip(pipe.stages, bufs):
        if st.op is not None:
            st.op.spmm(b, out=y, accumulate=True)
torch.cuda.synchronize()
print(f"rank {R} of {W}, F = {F}, K = {K}, prefix = {pipe.prefix}: {REPS} repetitions")
