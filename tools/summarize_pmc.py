#!/usr/bin/env python3
"""Per-kernel means of every counter found under <dir>/pass*/ (rocprofv3 CSV) + kernel durations.
usage: python tools/summarize_pmc.py <dir> [kernel-substring ...]   -> markdown on stdout"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("tgcn::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    want = sys.argv[2:] or ["spmm"]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if any(w in k for w in want):
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*_kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if any(w in k for w in want):
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(vals):
        dd = sorted(durs.get(k, [0.0]))
        print(f"### `{k}`  ({len(dd)} dispatches over all passes, median {dd[len(dd)//2]:.1f} us)\n")
        print("| counter | dispatches | mean | min | max |\n|---|---|---|---|---|")
        for c, v in sorted(vals[k].items()):
            print(f"| {c} | {len(v)} | {sum(v)/len(v):.6g} | {min(v):.6g} | {max(v):.6g} |")
        print()


if __name__ == "__main__":
    main()
