#!/usr/bin/env python3
"""Epoch time on SMALL graphs (the sizes real TextGCN corpora produce), where the epoch is bound by
kernel launches rather than by bytes: eager loop (torch CE / Adam), eager loop with the fused
kernels, and HIP-graph replays (pytextgcn_amd.train)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytextgcn_amd as pkg  # noqa: E402
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.functional import masked_cross_entropy  # noqa: E402
from pytextgcn_amd.train import GraphedEval, GraphedTrainStep  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, N, E, h, C in [("7k nodes", 7_000, 760_000, 100, 6), ("c2: 100k nodes", 100_000, 2_000_000, 200, 64),
                         ("400k nodes", 400_000, 10_000_000, 200, 64)]:
    g = synth.word_doc_graph(N, E, seed=44, device=dev, n_classes=C)
    crit = torch.nn.CrossEntropyLoss()
    res = {}
    for mode in ("torch CE/Adam", "fused", "graphed"):
        torch.manual_seed(0)
        m = pkg.GCN(N, C, n_hidden_gcn=h, dropout=0.5).to(dev)
        if mode == "graphed":
            opt = pkg.optim.Adam(m.parameters(), lr=0.05, amsgrad=True, capturable=True)
            step, ev = GraphedTrainStep(m, g, opt, g.train_mask), GraphedEval(m, g)

            def epoch():
                loss = step()
                logits = ev()
                pred = logits[g.val_mask].argmax(1).cpu()
                return loss.item()
        else:
            Opt = pkg.optim.Adam if mode == "fused" else torch.optim.Adam
            opt = Opt(m.parameters(), lr=0.05, amsgrad=True)

            def epoch():
                m.train()
                if mode == "fused":
                    loss = masked_cross_entropy(m(g), g.y, g.train_mask)
                else:
                    loss = crit(m(g)[g.train_mask], g.y[g.train_mask])
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                m.eval()
                with torch.no_grad():
                    logits = m(g)
                    pred = logits[g.val_mask].argmax(1).cpu()
                return loss.item()
        res[mode] = timed(epoch)
    print(f"{name:16s} (N={N}, E={E}, h={h}): " + "   ".join(f"{k} {v:7.2f} ms" for k, v in res.items()), flush=True)
