#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries committed here.

usage: python profiles/summarize.py <round-tag> <stats_dir> [<fetch_dir> <write_dir> [<calib_dir> <calibw_dir>]]
  stats_dir : rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py ...
  fetch_dir : rocprofv3 --pmc FETCH_SIZE --kernel-trace ... (own pass)
  write_dir : rocprofv3 --pmc WRITE_SIZE --kernel-trace ... (own pass)
  calib_dir : rocprofv3 --pmc FETCH_SIZE on profiles/calibrate_fetch.py (known byte count)
Writes profiles/<tag>_kernel_stats.md and prints a JSON record for profiles/traffic.json.
"""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("tgcn::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def kernel_stats(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    return list(csv.DictReader(open(f)))


def trace(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    return list(csv.DictReader(open(f)))


def counters(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def epoch_table():
    """summarize.py epoch <tag> <stats_dir> <n_epochs>: per-epoch kernel table of tools/profile_epoch.py"""
    tag, d, n = sys.argv[2], sys.argv[3], int(sys.argv[4])
    rows = kernel_stats(d)
    total = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6
    lines = [f"# Kernel time per epoch of the fused loop, {tag}", "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/profile_epoch.py"
             f"{(' ' + sys.argv[5]) if len(sys.argv) > 5 else ''}` (config c4; {n} epochs in the trace, graph "
             "generation and plan construction included in the tail rows).", "",
             f"Sum of kernel durations: {total:.2f} ms per epoch.", "",
             "| kernel | us / epoch | calls / epoch | % |", "|---|---|---|---|"]
    for r in rows[:24]:
        lines.append(f"| `{short(r['Name'])}` | {float(r['TotalDurationNs']) / n / 1e3:.1f} | "
                     f"{float(r['Calls']) / n:.1f} | {r['Percentage']} |")
    with open(os.path.join(HERE, f"{tag}_epoch_kernels.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


def main():
    if sys.argv[1] == "epoch":
        return epoch_table()
    tag, stats_dir = sys.argv[1], sys.argv[2]
    lines = [f"# rocprofv3 summary, round {tag}", ""]
    cmd_file = os.path.join(stats_dir, "..", "command.txt")
    if os.path.exists(cmd_file):
        lines += ["Command: `rocprofv3 --kernel-trace --stats --output-format csv -- "
                  + open(cmd_file).read().strip() + "`", ""]
    rows = kernel_stats(stats_dir)
    lines += ["## `--kernel-trace --stats` (top 15 kernels by total time)", "",
              "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:15]:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                     f"{float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                     f"{float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |")
    for kname in ("k_spmm_gather", "k_spmm_hot", "k_spmm_fix"):
        tr = [r for r in trace(stats_dir) if kname in r["Kernel_Name"]]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
        lines += ["", f"`{kname}` dispatches: {len(dur)}; durations (us), in launch order:", "",
                  "`" + " ".join(f"{d:.0f}" for d in dur) + "`"]
    out = {"tag": tag}
    # average duration of the SpMM kernels in the stats pass, and the launch time the bench itself measured with HIP
    # events in that very run (its JSON line in stats.log): both go into traffic.json, so that the roofline fraction can
    # be recomputed from that file alone
    out["kernel_avg_us"] = {short(r["Name"]).split("<")[0]: float(r["AverageNs"]) / 1e3 for r in rows
                            if "k_spmm" in r["Name"]}
    slog = os.path.join(stats_dir, "..", "stats.log")
    if os.path.exists(slog):
        for ln in open(slog):
            ln = ln.strip()
            if ln.startswith("{") and '"roofline"' in ln:
                try:
                    out["bench_launch_ms"] = json.loads(ln)["roofline"]["launch_ms"]
                except (ValueError, KeyError):
                    pass
    if len(sys.argv) >= 5:
        fe, wr = counters(sys.argv[3]), counters(sys.argv[4])
        lines += ["", "## PMC passes (separate runs): FETCH_SIZE / WRITE_SIZE in KiB per dispatch (TCC_* in requests)", "",
                  "| kernel | counter | dispatches | mean | min | max |", "|---|---|---|---|---|---|"]
        for agg in (fe, wr):
            for (k, c), v in sorted(agg.items()):
                if "spmm" in k or "colsum" in k:
                    lines.append(f"| `{k}` | {c} | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
        out["fetch_kib"] = {k: sum(v) / len(v) for (k, c), v in fe.items() if "spmm" in k and c == "FETCH_SIZE"}
        out["write_kib"] = {k: sum(v) / len(v) for (k, c), v in wr.items() if "spmm" in k and c == "WRITE_SIZE"}
        hits = {k: sum(v) / len(v) for (k, c), v in wr.items() if "spmm" in k and c == "TCC_HIT_sum"}
        miss = {k: sum(v) / len(v) for (k, c), v in wr.items() if "spmm" in k and c == "TCC_MISS_sum"}
        if hits:
            out["l2_hit_rate"] = {k: hits[k] / (hits[k] + miss[k]) for k in hits if k in miss and hits[k] + miss[k] > 0}
            lines += ["", "L2 hit rate (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)): " +
                      ", ".join(f"`{k}` {v:.3f}" for k, v in sorted(out["l2_hit_rate"].items()))]
    if len(sys.argv) >= 7:
        lines += ["", "## Calibration on a known byte count (profiles/calibrate_fetch.py: permutation "
                  "operator, N = 4 M, F = 200)", ""]
        for d in sys.argv[5:7]:
            for (k, c), v in sorted(counters(d).items()):
                if "spmm_gather" in k:
                    lines.append(f"- `{k}` {c}: " + " ".join(f"{x:.0f}" for x in v) + " KiB per dispatch")
                    out["calib_" + c.lower() + "_kib"] = sum(v) / len(v)
        log = os.path.join(sys.argv[5], "..", "calib.log")
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith("known read bytes"):
                    lines.append("- " + ln.strip())
    with open(os.path.join(HERE, f"{tag}_kernel_stats.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(out))
    key = os.environ.get("TGCN_TRAFFIC_KEY")          # e.g. c4_n1: record the counter traffic for bench.py
    if key and "fetch_kib" in out:
        update_traffic(key, tag, out)


def update_traffic(key, tag, out):
    """profiles/traffic.json[key] <- bytes per tgcn_spmm launch = sum over its kernels of (2 * FETCH_SIZE + WRITE_SIZE),
    stamped with the fingerprint of the kernel sources it was collected on (bench.py: spmm_kernel_sha16)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    kernels = sorted(set(out["fetch_kib"]) | set(out["write_kib"]))
    fetch = {k.split("<")[0]: out["fetch_kib"].get(k, 0.0) for k in kernels}
    write = {k.split("<")[0]: out["write_kib"].get(k, 0.0) for k in kernels}
    total = sum(2.0 * fetch[k] + write[k] for k in fetch) * 1024.0
    path = os.path.join(HERE, "traffic.json")
    with open(path) as f:
        db = json.load(f)
    db[key] = {
        "bytes_per_launch": total,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench command (tools/"
                  "collect_evidence.sh); per tgcn_spmm launch = sum over its kernels; bytes = (2*FETCH_SIZE + WRITE_SIZE)"
                  " KiB * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (calibration of round 1, profiles/"
                  "calibrate_fetch.py). Counts L2<->fabric requests, Infinity-Cache hits included: an upper bound on HBM "
                  "bytes (the live HBM figure is bench.py's roofline.hbm_activity).",
        "fetch_size_kib": fetch, "write_size_kib": write, "round": tag,
        # one launch = the sum of its kernels' average durations in the --kernel-trace --stats pass of the same command,
        # next to the HIP-event launch time bench.py measured in that pass: fabric GB/s = bytes_per_launch / launch time
        "kernel_avg_us": out.get("kernel_avg_us"),
        "launch_ms_rocprof_kernel_sum": (sum(out["kernel_avg_us"].values()) / 1e3) if out.get("kernel_avg_us") else None,
        "launch_ms_bench_hip_events": out.get("bench_launch_ms"),
        "kernel_sha16": bench.spmm_kernel_sha16()}
    t = db[key]["launch_ms_rocprof_kernel_sum"]
    if t:
        db[key]["fabric_GBps_at_rocprof_launch_time"] = total / (t * 1e-3) / 1e9
    with open(path, "w") as f:
        json.dump(db, f, indent=1)
    print(f"profiles/traffic.json[{key}] = {total / 1e9:.3f} GB per launch ({tag}, kernels {db[key]['kernel_sha16']})")


if __name__ == "__main__":
    main()
