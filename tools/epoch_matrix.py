#!/usr/bin/env python3
"""Epoch time of the fused loop at c4 under the bitwise-neutral switches (activation reuse, W1 update in the
backward SpMM), more repetitions than bench.py takes.   python tools/epoch_matrix.py [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytextgcn_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
N, E, F, C = bench.CONFIGS["c4"]
dev = torch.device("cuda:0")
g = synth.word_doc_graph(N, E, seed=44, device=dev, n_classes=C)
for round_ in range(2):
    for reuse in (False, True):
        for w1 in (False, True):
            ms = bench.epoch_time_ms(g, F, C, fused=True, reps=reps, reuse=reuse, fuse_w1=w1)
            print(json.dumps({"round": round_, "activation_reuse": reuse, "w1_update_in_backward": w1, "epoch_ms": round(ms, 3)}), flush=True)
    # + the opt-in split-bf16 mode of the layer-2 products (not bitwise neutral: fp32-accurate)
    for reuse, w1 in ((False, False), (True, True)):
        ms = bench.epoch_time_ms(g, F, C, fused=True, reps=reps, reuse=reuse, fuse_w1=w1, split_gemms=True)
        print(json.dumps({"round": round_, "activation_reuse": reuse, "w1_update_in_backward": w1, "split_bf16_gemms": True,
                          "epoch_ms": round(ms, 3)}), flush=True)
