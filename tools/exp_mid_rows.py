#!/usr/bin/env python3
"""Experiment (round 2): what do the MID-frequency word rows cost in the c4 SpMM, and what would an
L2-resident treatment of their document-column entries cost?

Row classes (by degree, after the 32 hot rows are set aside): long (> T = 512), mid (>= m for m in
--mid), short word rows, document rows.  Entries are split by column class: word columns (< V, the
160 MB word block) and document columns (>= V, the 1.44 GB document block whose rows are each used only
~5 times by non-hot rows).  For each threshold m the tool times
  (a) the operator as is,
  (b) the operator with the document-column entries of rows with degree >= m REMOVED  (upper bound on
      what a separate treatment of those entries can save), and
  (c) those entries ALONE with their columns folded onto 4096 rows (every gather L2-resident: what the
      same number of gathers costs when a column block sits in L2 -- the floor for a column-sweep kernel).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402
from tools.sweep_spmm import time_spmm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--F", type=int, default=200)
    ap.add_argument("--mid", default="512,128,32")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, E, F = 2_000_000, 50_000_000, args.F
    g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
    V = g.n_vocab
    full = GraphPlan(g.edge_index, g.edge_attr, N)
    rp, col, val = full.export_csr()
    deg = (rp[1:] - rp[:-1]).long()
    row = torch.repeat_interleave(torch.arange(N, device=dev), deg)
    col = col.long()
    x = torch.randn(N, F, device=dev)
    order = torch.argsort(deg, descending=True)
    hot = torch.zeros(N, dtype=torch.bool, device=dev)
    hot[order[:32]] = True
    is_word_row = row < V
    doc_col = col >= V
    t_full = time_spmm(full, x)[0]
    print(json.dumps({"case": "c4 as is", "ms": round(t_full, 3), "nnz": int(deg.sum()), "V": V, **full.stats()}), flush=True)
    # census
    for name, sel in [("hot rows (32 longest)", hot[row]),
                      ("word rows deg > 512, not hot", is_word_row & ~hot[row] & (deg[row] > 512)),
                      ("word rows 128 <= deg <= 512", is_word_row & (deg[row] >= 128) & (deg[row] <= 512)),
                      ("word rows 32 <= deg < 128", is_word_row & (deg[row] >= 32) & (deg[row] < 128)),
                      ("word rows deg < 32", is_word_row & (deg[row] < 32)),
                      ("document rows", ~is_word_row)]:
        rows_n = int(torch.unique(row[sel]).numel())
        print(json.dumps({"class": name, "rows": rows_n, "entries": int(sel.sum()),
                          "word_col_entries": int((sel & ~doc_col).sum()), "doc_col_entries": int((sel & doc_col).sum())}),
              flush=True)
    for m in [int(s) for s in args.mid.split(",")]:
        mid = is_word_row & ~hot[row] & (deg[row] >= m)
        take = mid & doc_col
        n_take = int(take.sum())
        keep = ~take
        p = GraphPlan.from_coo(row[keep], col[keep], val[keep], N, N)
        t_b = time_spmm(p, x)[0]
        p.close()
        # the removed entries alone, columns folded onto 4096 document rows (L2-resident block)
        q = GraphPlan.from_coo(row[take], V + (col[take] - V) % 4096, val[take], V, N)   # word rows only: no 1.6 GB of zero rows
        t_c = time_spmm(q, x)[0]
        q.close()
        # the same entries alone, true columns (what they cost on their own today)
        q = GraphPlan.from_coo(row[take], col[take], val[take], V, N)
        t_d = time_spmm(q, x)[0]
        q.close()
        print(json.dumps({"mid_threshold": m, "rows": int(torch.unique(row[mid]).numel()), "doc_col_entries": n_take,
                          "ms_without_them": round(t_b, 3), "saving_upper_bound_ms": round(t_full - t_b, 3),
                          "ms_alone_folded_L2_resident": round(t_c, 3), "ms_alone_true_columns": round(t_d, 3)}), flush=True)


if __name__ == "__main__":
    main()
