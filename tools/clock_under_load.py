#!/usr/bin/env python3
"""Shader clock and board power while one kernel loops (c4 shapes): do the dense kernels run at the clock their MFMA floor
assumes?  Samples sysfs (hwmon freq1_input = sclk, power1_average / power1_input) from a second thread at 100 Hz while each
workload loops for ~1.5 s.   python tools/clock_under_load.py > profiles/r04_clock_under_load.log"""
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import dense, synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.hbm_activity import find_sysfs  # noqa: E402

dev = torch.device("cuda:0")
sysdir = find_sysfs(0)
hw = sorted(glob.glob(os.path.join(sysdir, "hwmon", "hwmon*"))) if sysdir else []
files = {}
if hw:
    for key, names in (("sclk_MHz", ["freq1_input"]), ("power_W", ["power1_average", "power1_input"])):
        for nme in names:
            f = os.path.join(hw[0], nme)
            if os.path.exists(f):
                files[key] = f
                break
print(json.dumps({"sysfs": sysdir, "files": files}), flush=True)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            row = {"t": time.perf_counter()}
            for k, f in files.items():
                try:
                    row[k] = float(open(f).read().strip()) / 1e6
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(0.01)


def measure(name, fn, seconds=1.5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
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
    lo, hi = t0 + 0.3 * (t1 - t0), t0 + 0.95 * (t1 - t0)
    rec = {"case": name, "ms_per_launch": round((t1 - t0) / n * 1e3, 4)}
    for k in files:
        v = sorted(r[k] for r in s.rows if lo <= r["t"] <= hi and k in r)
        if v:
            rec[k + "_median"] = round(v[len(v) // 2], 1)
            rec[k + "_min"] = round(v[0], 1)
            rec[k + "_max"] = round(v[-1], 1)
    print(json.dumps(rec), flush=True)


N, h, C = 2_000_000, 200, 64
H = torch.randn(N, h, device=dev)
W = torch.randn(h, C, device=dev)
G = torch.randn(N, C, device=dev)
seed = dense.new_seed(dev)
measure("idle", lambda: None, 0.5)
measure("device copy 1.6 GB", lambda: H.clone())
measure("nn", lambda: dense.gemm_nn(H, W))
measure("nn_dropout", lambda: dense.gemm_nn(H, W, 0.5, seed))
measure("nt", lambda: dense.gemm_nt(G, W))
measure("nt_dropout_colsum", lambda: dense.gemm_nt(G, W, 0.5, seed, note_colsums=True))
measure("tn", lambda: dense.gemm_tn(H, G))
measure("tn_dropout", lambda: dense.gemm_tn(H, G, 0.5, seed))
g = synth.word_doc_graph(N, 50_000_000, seed=44, device=dev, features="none")
plan = GraphPlan(g.edge_index, g.edge_attr, N)
measure("spmm F=200", lambda: plan.spmm(H))
measure("spmm F=64", lambda: plan.spmm(G))
