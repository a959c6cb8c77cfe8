#!/usr/bin/env python3
"""Does the order of the document nodes matter to the SpMM once the corpus has TOPICAL locality?  (the only
admissible SpMM experiment is one that removes fabric bytes on a corpus with topical locality; the benchmark generator
has none.)  `synth.word_doc_graph(..., n_topics=T)` draws 60 % of a document's words from its topic's slice of the
vocabulary; the same graph is then laid out three ways:

    by_topic    documents of a topic adjacent (a corpus file sorted by class)
    shuffled    document nodes re-labelled at random (what the plain generator amounts to)
    reordered   the shuffled graph with its documents re-labelled by their RAREST word (document frequency, ties by word
                id): a plan-level heuristic that needs no topic labels
    reordered_mid   ... by their most frequent word among those in fewer than 2 % of the documents (a topic's head word)
    propagated      ... by clusters found from the document-word edges alone (`by_propagation`: alternating votes of
                    documents and words, a few passes over the edges on the GPU)

and timed at F = 200 and F = 64 (one tgcn_spmm launch each, median of interleaved rounds).  With `--one VARIANT F` it runs
five launches of one variant only, for a `rocprofv3 --pmc FETCH_SIZE` pass (tools/exp_topical_order.sh).
  python tools/exp_topical_order.py [c3|c4] [--one by_topic 200]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

SHAPES = {"c3": dict(n_nodes=1_000_000, n_edges=24_000_000, vocab_frac=0.03, doc_word_share=0.9, n_topics=219, n_classes=219),
          "c4": dict(n_nodes=2_000_000, n_edges=50_000_000, n_topics=64, n_classes=64)}
dev = torch.device("cuda:0")


def by_rarest_word(g, mid_frequency=False):
    """The graph with its document nodes re-labelled so that documents sharing their rarest word are adjacent -- or, with
    `mid_frequency`, their MOST frequent word among those that occur in fewer than 2 % of the documents (a topic's head
    word rather than a one-off from the global tail or a stop-word-like hub)."""
    V, N = g.n_vocab, g.y.numel()
    ei = g.edge_index
    m = (ei[0] >= V) & (ei[1] < V)                                  # doc -> word
    d, w = ei[0][m] - V, ei[1][m]
    df = torch.bincount(w, minlength=V)
    if mid_frequency:
        score = torch.where(df[w] < (N - V) // 50, df[w], torch.zeros_like(df[w]))
        key = torch.zeros(N - V, dtype=torch.int64, device=ei.device).scatter_reduce_(0, d, score * V + w, "amax")
    else:
        key = torch.full((N - V,), 1 << 62, dtype=torch.int64, device=ei.device).scatter_reduce_(0, d, df[w] * V + w, "amin")
    return relabelled(g, torch.argsort(key, stable=True))


def relabelled(g, order):
    """The graph with document `order[i]` (old id, counted from the first document) re-labelled as document i."""
    V, ei = g.n_vocab, g.edge_index
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(order.numel(), device=order.device)
    coo = ei.t().clone()                                             # (ei IS the transposed view of a contiguous [E, 2] array)
    for c in (0, 1):
        is_doc = coo[:, c] >= V
        coo[:, c] = torch.where(is_doc, new_id[(coo[:, c] - V).clamp(min=0)] + V, coo[:, c])
    return coo.t(), g.edge_attr


def by_propagation(g, K=None, report=None):
    """Documents laid out by the clusters `pytextgcn_amd.reorder.cluster_documents` finds from the document-word edges alone
    (the shipped routine: alternating votes of documents and words; no topic labels)."""
    from pytextgcn_amd.reorder import cluster_documents
    V, N = g.n_vocab, g.y.numel()
    lab = cluster_documents(g.edge_index, g.edge_attr, V, N, n_clusters=K)
    if report is not None:
        K_ = int(lab.max()) + 1
        sizes = torch.bincount(lab, minlength=K_)
        t = g.y[V:].long()                       # the generator's topic of every document (its label): for the report only
        T_ = int(t.max()) + 1
        joint = torch.zeros(K_ * T_, device=lab.device).index_add_(0, lab * T_ + t, torch.ones(N - V, device=lab.device))
        report.update(clusters=K_, largest=int(sizes.max()), purity=round(float(joint.view(K_, T_).max(1).values.sum() / (N - V)), 3))
    return relabelled(g, torch.argsort(lab, stable=True))


def words_too(g, K=None):
    """`by_propagation` and, on top, the WORD nodes laid out by the cluster that holds most of each word's mass (what
    `reorder_documents(..., words=True)` does)."""
    from pytextgcn_amd.reorder import cluster_documents
    V, N = g.n_vocab, g.y.numel()
    lab = cluster_documents(g.edge_index, g.edge_attr, V, N, n_clusters=K)
    K_ = int(lab.max()) + 1
    ei = g.edge_index
    m = (ei[0] >= V) & (ei[1] < V)
    d, w, a = ei[0][m] - V, ei[1][m], g.edge_attr[m].float()
    ws = torch.zeros(V * K_, device=ei.device).index_add_(0, w * K_ + lab[d], a).view(V, K_)
    wlab = ws.argmax(1)
    order = torch.cat([torch.argsort(wlab, stable=True), torch.argsort(lab, stable=True) + V])     # old id at each new place
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(N, device=order.device)
    return new_id[ei], g.edge_attr


def graphs(cfg):
    kw = dict(SHAPES[cfg], seed=44, device=dev, features="none")
    gt = synth.word_doc_graph(**kw, doc_order="by_topic")
    gs = synth.word_doc_graph(**kw, doc_order="shuffled")
    out = {"by_topic": (gt.edge_index, gt.edge_attr), "shuffled": (gs.edge_index, gs.edge_attr),
           "reordered": by_rarest_word(gs), "reordered_mid (most frequent word below 2 % of the documents)": by_rarest_word(gs, True)}
    T = SHAPES[cfg]["n_topics"]
    for name, K in (("propagated (K = topics)", T), ("propagated (K = 4 x topics)", 4 * T), ("propagated (K by default)", None)):
        rep = {}
        out[name] = by_propagation(gs, K, report=rep)
        print(f"{cfg} {name}: {rep}", flush=True)
    out["propagated, word nodes by cluster as well (words=True)"] = words_too(gs, None)
    plain = dict(kw)
    plain.pop("n_topics")
    gp = synth.word_doc_graph(**plain)
    out["no_topics (the benchmark generator)"] = (gp.edge_index, gp.edge_attr)
    out["no_topics, propagated (nothing to find: must not hurt)"] = by_propagation(gp, None)
    return out, kw["n_nodes"]


def time_ms(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in SHAPES else "c3"
    gs, N = graphs(cfg)
    if "--one" in sys.argv:
        name, F = sys.argv[sys.argv.index("--one") + 1], int(sys.argv[sys.argv.index("--one") + 2])
        ei, w = gs[next(k for k in gs if k.startswith(name))]
        plan = GraphPlan(ei, w, N)
        x, y = torch.randn(N, F, device=dev), torch.empty(N, F, device=dev)
        for _ in range(5):
            plan.spmm(x, out=y)
        torch.cuda.synchronize()
        print(f"ONE {cfg} {name} F={F}", flush=True)
        return
    plans = {k: GraphPlan(ei, w, N) for k, (ei, w) in gs.items()}
    for F in (200, 64):
        x, y = torch.randn(N, F, device=dev), torch.empty(N, F, device=dev)
        res = {k: [] for k in plans}
        for _ in range(5):
            for k, p in plans.items():
                res[k].append(time_ms(lambda: p.spmm(x, out=y)))
        base = sorted(res["shuffled"])[2]
        for k, v in res.items():
            med = sorted(v)[2]
            print(f"{cfg} F={F:3d}  {k:70s} {med:7.3f} ms per launch   ({med / base:.3f} x shuffled; "
                  f"hot rows {plans[k].stats()['hot_rows']})", flush=True)


if __name__ == "__main__":
    main()
