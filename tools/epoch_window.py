#!/usr/bin/env python3
"""Steady-state epoch from a rocprofv3 kernel trace of tools/profile_epoch.py: the window between the first and
the last launch of an anchor kernel (one launch per epoch) holds n - 1 whole epochs; prints per-kernel time per
epoch, the busy time and the idle time (launch gaps, host work) of that window.
  python tools/epoch_window.py <kernel_trace.csv> [anchor-substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_masked_ce"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [s for s, e, n in rows if anchor in n]
# the anchor may run more than once per epoch (train + eval): find the period from k_spmm_gather_adam or fall back
per_epoch = 1
adam = [s for s, e, n in rows if "k_spmm_gather_adam" in n]
if adam and anchor not in "k_spmm_gather_adam":
    per_epoch = max(1, round(len(marks) / len(adam)))
    marks = adam
t0, t1 = marks[0], marks[-1]
n_ep = len(marks) - 1
agg = defaultdict(lambda: [0, 0])
busy = 0
last_end = t0
last_name = "(window start)"
idle = 0
gaps = defaultdict(lambda: [0, 0])       # (kernel before, kernel after) -> [ns, count]


def _short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("tgcn::", "").split("(")[0][:110]


for s, e, n in rows:
    if s < t0 or s >= t1:
        continue
    short = _short(n)
    agg[short][0] += e - s
    agg[short][1] += 1
    busy += e - s
    if s > last_end:
        idle += s - last_end
        g = gaps[(last_name[:60], short[:60])]
        g[0] += s - last_end
        g[1] += 1
    if e >= last_end:
        last_name = short
    last_end = max(last_end, e)
print(f"window: {n_ep} epochs, {(t1 - t0) / n_ep / 1e6:.3f} ms per epoch, kernels {busy / n_ep / 1e6:.3f} ms, "
      f"idle between kernels {idle / n_ep / 1e6:.3f} ms")
print("largest idle gaps (kernel before -> kernel after): us per epoch, occurrences per epoch")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {t / n_ep / 1e3:8.1f} us  x{c / n_ep:.1f}   `{a}` -> `{b}`")
print("| kernel | us / epoch | calls / epoch |\n|---|---|---|")
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"| `{k}` | {t / n_ep / 1e3:.1f} | {c / n_ep:.1f} |")
