#!/usr/bin/env python3
"""What actually reaches HBM: memory-controller activity while a kernel loops (round-2 measurement).

rocprofv3's FETCH_SIZE / WRITE_SIZE count requests at the L2 <-> fabric boundary, Infinity-Cache hits
included (MI355X_MICROARCH.md, HBM), so they bound the HBM bytes of the SpMM from above only.  The
memory controllers themselves report a busy percentage through the driver (`mem_busy_percent` in sysfs =
the SMU's average UMC activity, what `rocm-smi --showmemuse` prints).  This tool loops a workload for a
few seconds, samples that file from a second thread, and calibrates the percentage against streams whose
HBM traffic is known exactly (a 1 GiB device copy, a read-only sum and a fill, all far beyond the
256 MiB Infinity Cache):

    HBM bytes/s of the workload  ~=  activity(workload) / activity(copy) * bytes/s(copy)

Output: one JSON record per workload on stdout (and in --out), with the samples' median / mean.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def find_sysfs(dev_index: int):
    """sysfs directory of the torch device (matched by PCI bus id; the only amdgpu card otherwise)."""
    cards = sorted(glob.glob("/sys/class/drm/card*/device/mem_busy_percent"))
    if not cards:
        return None
    try:
        p = torch.cuda.get_device_properties(dev_index)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
        for c in cards:
            real = os.path.realpath(os.path.dirname(c))
            if want in real:
                return os.path.dirname(c)
    except Exception:                                            # noqa: BLE001
        pass
    return os.path.dirname(cards[0]) if len(cards) == 1 else None


class Sampler(threading.Thread):
    def __init__(self, sysdir, period=0.01):
        super().__init__(daemon=True)
        self.files = {k: os.path.join(sysdir, k) for k in ("mem_busy_percent", "gpu_busy_percent")}
        self.period = period
        self.samples = []
        self.stop_flag = False

    def run(self):
        while not self.stop_flag:
            row = [time.perf_counter()]
            for k, f in self.files.items():
                try:
                    with open(f) as fh:
                        row.append(float(fh.read().strip()))
                except (OSError, ValueError):
                    row.append(float("nan"))
            self.samples.append(row)
            time.sleep(self.period)


def measure(name, fn, sysdir, seconds, bytes_known=None, extra=None):
    """Loop fn() for `seconds` (batches of launches with a sync in between), sampling the activity."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler(sysdir)
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    s.stop_flag = True
    s.join()
    lo, hi = t0 + 0.25 * (t1 - t0), t0 + 0.95 * (t1 - t0)       # steady part of the window
    mem = sorted(r[1] for r in s.samples if lo <= r[0] <= hi and r[1] == r[1])
    gpu = sorted(r[2] for r in s.samples if lo <= r[0] <= hi and r[2] == r[2])
    rec = {"workload": name, "launches": n, "ms_per_launch": (t1 - t0) / n * 1e3,
           "mem_busy_percent_median": mem[len(mem) // 2] if mem else None,
           "mem_busy_percent_mean": sum(mem) / len(mem) if mem else None,
           "mem_busy_percent_min_max": [mem[0], mem[-1]] if mem else None,
           "gpu_busy_percent_median": gpu[len(gpu) // 2] if gpu else None,
           "n_samples": len(mem)}
    if bytes_known is not None:
        rec["known_hbm_bytes_per_launch"] = bytes_known
        rec["known_hbm_GBps"] = bytes_known / ((t1 - t0) / n) / 1e9
    if extra:
        rec.update(extra)
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--config", default="c4")
    ap.add_argument("--widths", default="200,64")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sysdir = find_sysfs(0)
    if sysdir is None:
        print(json.dumps({"error": "no mem_busy_percent in sysfs: memory-controller activity is not readable here"}))
        return 1
    recs = []
    # ---- calibration streams (1 GiB each way: 4x the Infinity Cache) ----------------------------------
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    recs.append(measure("idle", lambda: None, sysdir, 1.0))
    recs.append(measure("copy 1 GiB -> 1 GiB (read + write)", lambda: b.copy_(a), sysdir, args.seconds,
                        bytes_known=2 * n * 4))
    recs.append(measure("sum of 1 GiB (read only)", lambda: a.sum(), sysdir, args.seconds, bytes_known=n * 4))
    recs.append(measure("fill 1 GiB (write only)", lambda: b.fill_(1.0), sysdir, args.seconds, bytes_known=n * 4))
    del a, b
    # ---- the SpMM of the benchmark configuration -------------------------------------------------------
    from bench import CONFIGS
    from pytextgcn_amd import synth
    from pytextgcn_amd.plan import GraphPlan
    N, E, _, C = CONFIGS[args.config]
    kw = dict(vocab_frac=0.03, doc_word_share=0.9) if args.config == "c3" else {}
    g = synth.word_doc_graph(N, E, seed=44, device=dev, n_classes=C, features="none", **kw)
    plan = GraphPlan(g.edge_index, g.edge_attr, N)
    del g
    for F in [int(w) for w in args.widths.split(",")]:
        gen = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(N, F, device=dev, generator=gen)
        y = torch.empty(N, F, device=dev)
        recs.append(measure(f"{args.config} tgcn_spmm F={F}", lambda: plan.spmm(x, None, out=y), sysdir, args.seconds,
                            extra={"algorithmic_bytes_per_launch": plan.algorithmic_bytes(F),
                                   "compulsory_bytes_per_launch": 8 * plan.nnz + 4 * N + 8 * N * F,
                                   "plan": plan.stats()}))
        del x, y
    # ---- derived: HBM rate of each SpMM from the copy calibration --------------------------------------
    cal = next(r for r in recs if r["workload"].startswith("copy"))
    idle = recs[0]["mem_busy_percent_median"] or 0.0
    for r in recs:
        if "tgcn_spmm" in r["workload"] and r["mem_busy_percent_median"] is not None:
            scale = cal["known_hbm_GBps"] / max(cal["mem_busy_percent_median"] - idle, 1e-9)
            r["hbm_GBps_estimate"] = (r["mem_busy_percent_median"] - idle) * scale
            r["hbm_bytes_per_launch_estimate"] = r["hbm_GBps_estimate"] * 1e9 * r["ms_per_launch"] * 1e-3
    summary = {"method": "sysfs mem_busy_percent (UMC activity) sampled at 100 Hz, calibrated on a 1 GiB device copy",
               "sysfs": sysdir, "records": recs}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "records"}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
