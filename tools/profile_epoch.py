#!/usr/bin/env python3
"""Runs only the fused epoch loop of bench.py (c4) so that a rocprofv3 kernel trace of this process is
that loop and nothing else:
  rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_epoch.py [reuse|collapse|best|rows]   (best = activation reuse + W1 update in the backward SpMM; rows = best + the last layer computes only the rows that are read)
  python profiles/summarize.py epoch <tag> OUT 5         # 5 = epochs in the trace (1 warm-up + 4 timed)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytextgcn_amd import synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else ""
N, E, F, C = bench.CONFIGS["c4"]
g = synth.word_doc_graph(N, E, seed=44, device=torch.device("cuda:0"), n_classes=C)
ms = bench.epoch_time_ms(g, F, C, fused=True, reps=4 if mode not in ("best", "rows") else 12, reuse=mode in ("reuse", "best", "rows"),
                          collapse=mode == "collapse", fuse_w1=mode in ("best", "rows"), needed_rows=mode == "rows")
print(f"epoch_ms_fused{('_' + mode) if mode else ''} = {ms:.3f}")
