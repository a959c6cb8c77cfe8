#!/usr/bin/env python3
"""profiles/traffic.json[<cfg>_n<W>] from the passes of tools/collect_local_step_traffic.sh: bytes the local operators of rank 0 of a
W-rank partition move over the L2 <-> fabric link per distributed SpMM = sum over ALL dispatches of the step's kernels of
(2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 / steps (same correction as the single-device entries: MI355X_MICROARCH.md 'HBM'), with
the kernels' summed duration per step from the trace pass.  The exchange's own bytes (RCCL) are not in it.
  python tools/local_step_traffic.py gpurun_out/<tag> [c4|c5]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "c4"
db_path = os.path.join(ROOT, "profiles", "traffic.json")
db = json.load(open(db_path))


def in_step(rows, name_key, order_key):
    """The dispatches of the timed steps: every k_spmm_* launch and the elementwise adds AFTER the first of them (the
    construction of the operators runs before the first SpMM and launches adds of its own)."""
    rows = sorted(rows, key=lambda r: int(r[order_key]))
    first = next(i for i, r in enumerate(rows) if "k_spmm" in r[name_key])
    return [r for r in rows[first:] if "k_spmm" in r[name_key] or "CUDAFunctor_add<float>" in r[name_key]]



def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        sys.exit(f"no file matches {pattern}")
    return f[0]


for W in (2, 4, 8):
    log = open(os.path.join(src, f"w{W}.stats.log")).read()
    m = re.search(r"LOCAL_STEPS (\d+) world \d+ rank (\d+) ms_per_step ([\d.]+) A_nnz (\d+) B_nnz (\d+)", log)
    steps, rank, ms_events = int(m.group(1)), int(m.group(2)), float(m.group(3))
    dur = defaultdict(float)
    count = defaultdict(int)
    for r in in_step(list(csv.DictReader(open(one(f"{src}/w{W}/stats/**/*kernel_trace.csv")))), "Kernel_Name", "Start_Timestamp"):
        n = r["Kernel_Name"]
        k = re.sub(r"\(.*", "", n.replace("void ", "").replace("tgcn::(anonymous namespace)::", ""))[:60]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        count[k] += 1
    totals = {}
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        t = 0.0
        rows = [r for r in csv.DictReader(open(one(f"{src}/w{W}/{sub}/**/*counter_collection.csv"))) if r["Counter_Name"] == ctr]
        for r in in_step(rows, "Kernel_Name", "Dispatch_Id"):
            t += float(r["Counter_Value"])
        totals[ctr] = t / steps
    total = (2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]) * 1024.0
    kernel_ms = sum(dur.values()) / steps / 1e3
    db[f"{cfg}_n{W}"] = {
        "bytes_per_launch": total,
        "what": f"the LOCAL operators of rank {rank} of the {W}-rank partition (hot block, gather and second pass of A_r, gather and "
                "second pass of B_r, the add of the reduce-scattered rows), measured on ONE GPU with tools/prof_local_step.py: "
                "what the rank's kernels move over L2 <-> fabric per distributed SpMM.  The exchange's bytes (RCCL) are NOT in it",
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/collect_local_step_traffic.sh); sum over all "
                  "dispatches of the step's kernels / steps; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024",
        "fetch_size_kib_per_step": totals["FETCH_SIZE"], "write_size_kib_per_step": totals["WRITE_SIZE"],
        "kernel_us_per_step": {k: round(v / steps, 2) for k, v in sorted(dur.items(), key=lambda kv: -kv[1])},
        "dispatches_per_step": {k: count[k] / steps for k in dur},
        "launch_ms_rocprof_kernel_sum": kernel_ms, "launch_ms_hip_events": ms_events,
        "fabric_GBps_at_rocprof_launch_time": total / (kernel_ms * 1e-3) / 1e9,
        "round": os.path.basename(src.rstrip("/")), "kernel_sha16": bench.spmm_kernel_sha16()}
    print(f"{cfg}_n{W}: {total / 1e9:.3f} GB per SpMM over the fabric, kernels {kernel_ms:.3f} ms "
          f"({db[f'{cfg}_n{W}']['fabric_GBps_at_rocprof_launch_time']:.0f} GB/s), HIP events {ms_events:.3f} ms")
json.dump(db, open(db_path, "w"), indent=1)
