#!/usr/bin/env python3
"""Timeline of ONE rank of an N > 1 bench run from its rocprofv3 kernel trace: which of our kernels ran while an RCCL
kernel was resident, and how long an RCCL kernel had to wait behind the SpMM that was launched before it.
  python tools/rccl_overlap.py <kernel_trace.csv> [window_ms]
The window ends with the last exchange kernel of the trace (the timed steps come last) and reaches back window_ms."""
import csv
import sys

path = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("tgcn::", "")
    return n.split("(")[0][:70]


is_rccl = lambda n: "nccl" in n.lower() or "rccl" in n.lower()      # noqa: E731
ours = [r for r in rows if r[2].lstrip("void ").startswith(("tgcn::", "(anonymous")) or "k_spmm" in r[2] or "k_rows" in r[2]]
comm = [r for r in rows if is_rccl(r[2])]
print(f"{len(rows)} kernel launches, {len(comm)} of RCCL, {len(ours)} of libtgcn")
if not comm:
    sys.exit("no RCCL kernel in the trace")
# the window: it ends with the last exchange kernel (the long ones; the bench's own small all-reduces around the timed
# region are not of interest)
longest = max(e - s for s, e, n, q in comm)
big = [r for r in comm if r[1] - r[0] > longest / 50]
t1 = big[-1][1] + 200_000
t0 = t1 - int(window_ms * 1e6)
last = [r for r in big if r[0] >= t0]
print(f"\ntimeline of the last {window_ms:.0f} ms of exchange (us from the window start; queue = HSA queue id):")
print("| start | end | dur | queue | kernel |\n|---|---|---|---|---|")
for s, e, n, q in rows:
    if s >= t0 and e <= t1 and (is_rccl(n) or "k_spmm" in n or "k_rows" in n):
        print(f"| {(s - t0) / 1e3:9.1f} | {(e - t0) / 1e3:9.1f} | {(e - s) / 1e3:8.1f} | {q} | `{short(n)}` |")
print("\nper RCCL kernel of that window: resident time, of which a k_spmm_* kernel of this rank ran concurrently")
for s, e, n, q in last:
    both = 0
    names = set()
    for s2, e2, n2, q2 in ours:
        if "k_spmm" not in n2 or e2 <= s or s2 >= e:
            continue
        both += min(e, e2) - max(s, s2)
        names.add(short(n2).split("<")[0])
    print(f"  `{short(n)[:48]}`  {(e - s) / 1e3:9.1f} us resident, {both / 1e3:8.1f} us with {sorted(names)} running")
