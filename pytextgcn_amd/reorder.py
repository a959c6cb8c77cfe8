"""Document order for corpora with topical locality (opt-in; nothing in the reference corresponds to it).

`Text2GraphTransformer` numbers the document nodes in the order the corpus file lists them (text2graph.py:162-171).  The
propagate step gathers a word row for every (document, word) entry; which word rows a stretch of consecutive document rows
asks for decides how many of those gathers the L2 serves.  A file sorted by class keeps documents of one topic adjacent; a
shuffled file does not, and costs 11-14 % of the SpMM launch on a topical corpus (profiles/r05_exp_topical_order.log).
`reorder_documents` recovers the order from the graph alone: clusters of documents found by alternating votes over the
document-word edges (label propagation in the manner of spherical k-means: no labels, no text), documents laid out cluster
by cluster.  On the synthetic topical corpora this brings a shuffled graph to within 1 % of the sorted one (2.09 against 2.09
/ 2.41 ms at the DBpedia shape, 4.88 against 4.84 / 5.46 at the benchmark shape); on a corpus without topical structure it
changes nothing to speak of.

The result is the SAME graph under another numbering of its document nodes: every per-node tensor of the `Data` object
(`y`, the masks, the feature rows) is permuted along, the edge list keeps its order (so the reference-order normalisation
sums every degree in the same sequence and the weights are the same bits), and a training script that addresses nodes
through the masks (flat_amazon.py:101,109-114,130-131) runs unchanged.  `perm[i]` is the old id of new node i, for whoever
needs to map per-node results back.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .data import Data


def cluster_documents(edge_index: Tensor, edge_attr: Optional[Tensor], n_vocab: int, num_nodes: int,
                      n_clusters: Optional[int] = None, iters: int = 8, beta: float = 1.0,
                      max_doc_share: float = 0.05) -> Tensor:
    """Cluster id of every document node [num_nodes - n_vocab], from the document -> word edges alone.
    P(cluster | word) is the share of the word's TF-IDF mass that sits in the cluster's documents; a document's score for a
    cluster is sum over its words of weight x P(cluster | word) / P(cluster)^beta (the lift keeps the clusters from
    collapsing into one); arg-max; repeat until fewer than 0.1 % of the documents move or `iters` passes.  Words in more than
    `max_doc_share` of the documents do not vote.  The start is deterministic: documents that share their most frequent
    word among those in fewer than 2 % of the documents (a topic's head word) start together."""
    V, N = int(n_vocab), int(num_nodes)
    D = N - V
    if D <= 0 or V <= 0:
        raise ValueError("cluster_documents needs word nodes [0, n_vocab) and document nodes behind them")
    K = int(n_clusters) if n_clusters else max(16, min(512, D // 8192))
    K = max(1, min(K, D))
    dev = edge_index.device
    m = (edge_index[0] >= V) & (edge_index[1] < V)                   # document -> word entries
    d, w = edge_index[0][m] - V, edge_index[1][m]
    a = edge_attr[m].float() if edge_attr is not None else torch.ones(d.numel(), device=dev)
    df = torch.bincount(w, minlength=V)
    use = df[w] < max(2, int(D * max_doc_share))
    d, w, a = d[use], w[use], a[use]
    score = torch.where(df[w] < max(2, D // 50), df[w], torch.zeros_like(df[w]))
    head = torch.zeros(D, dtype=torch.int64, device=dev).scatter_reduce_(0, d, score * V + w, "amax") % V
    lab = (head * 2654435761 % 1000003) % K
    # Memory: the D x K score matrix (4 D K bytes: 16 GB at D = 7.8 M documents and K = 512 -- choose `n_clusters` for the
    # device) plus, per slab of edges, two [slab, K] temporaries (the gathered word rows and their product with the edge
    # weights).  The slab is sized so that those two stay within `slab_bytes` (256 MB) whatever K is.
    slab_bytes = 256 << 20
    chunk = max(1 << 14, slab_bytes // (8 * K))
    for _ in range(max(1, int(iters))):
        ws = torch.zeros(V * K, device=dev).index_add_(0, w * K + lab[d], a).view(V, K)
        ws = ws / ws.sum(1, keepdim=True).clamp_min(1e-20)           # P(cluster | word)
        prior = torch.bincount(lab, minlength=K).float().clamp_min(1.0) / D
        ws = ws / prior.pow(beta)
        ds = torch.zeros(D, K, device=dev)
        for lo in range(0, d.numel(), chunk):                        # (the D x K scores are built a slab of edges at a time)
            sl = slice(lo, lo + chunk)
            ds.index_add_(0, d[sl], a[sl].unsqueeze(1) * ws[w[sl]])
        new = ds.argmax(1)
        moved = int((new != lab).sum())
        lab = new
        if moved < max(1, D // 1000):
            break
    return lab


def _permute_features(x, perm: Tensor, new_id: Tensor, N: int):
    if x is None:
        return None
    if not torch.is_tensor(x) or x.size(0) != N:
        raise ValueError("reorder_documents: g.x must have one row per node")
    if not x.is_sparse:
        if x.size(1) == N:                                           # dense one-hot (text2graph.py:151 without sparse features)
            return x[perm][:, perm]
        return x[perm]
    xc = x.coalesce()
    (r, c), v = xc.indices(), xc.values()
    c2 = torch.where(c < N, new_id[c.clamp(max=N - 1)], c)           # the identity block's columns ARE node ids
    return torch.sparse_coo_tensor(torch.stack([new_id[r], c2]), v, xc.shape).coalesce()


def word_clusters(edge_index: Tensor, edge_attr: Optional[Tensor], n_vocab: int, doc_labels: Tensor) -> Tensor:
    """For every word node the cluster (of `doc_labels`) that holds most of the word's TF-IDF mass."""
    V = int(n_vocab)
    K = int(doc_labels.max()) + 1
    m = (edge_index[0] >= V) & (edge_index[1] < V)
    d, w = edge_index[0][m] - V, edge_index[1][m]
    a = edge_attr[m].float() if edge_attr is not None else torch.ones(d.numel(), device=edge_index.device)
    ws = torch.zeros(V * K, device=edge_index.device).index_add_(0, w * K + doc_labels[d], a).view(V, K)
    return ws.argmax(1)


def reorder_documents(g, n_clusters: Optional[int] = None, iters: int = 8, labels: Optional[Tensor] = None,
                      words: bool = False) -> Tuple[Data, Tensor]:
    """(g', perm): `g` with its document nodes laid out cluster by cluster (`cluster_documents`, or the given per-document
    `labels`); word nodes keep their ids unless `words` (then they are laid out by the cluster that holds most of each
    word's mass, inside [0, n_vocab): another 3-4 % on the topical corpora, but a word's node id no longer is its column in
    the transformer's vocabulary -- `perm` maps back).  Every tensor attribute of `g` with one entry per node is permuted
    along, `x` (sparse identity, [I_N | H] or a dense matrix) consistently with it; `edge_index` is re-labelled in place of
    its old ids and keeps its order.  `perm[i]` = old id of new node i."""
    V = int(getattr(g, "n_vocab", 0) or 0)
    ei = g.edge_index
    N = g.x.size(0) if getattr(g, "x", None) is not None else (g.y.numel() if getattr(g, "y", None) is not None
                                                                else int(ei.max()) + 1)
    if V <= 0 or V >= N:
        raise ValueError("reorder_documents needs g.n_vocab: the word nodes [0, n_vocab) stay, the documents behind them move")
    if labels is None:
        labels = cluster_documents(ei, getattr(g, "edge_attr", None), V, N, n_clusters=n_clusters, iters=iters)
    elif labels.numel() != N - V:
        raise ValueError("labels: one entry per document node")
    dev = ei.device
    labels = labels.to(dev)
    order = torch.argsort(labels, stable=True)                       # old document (counted from V) at each new place
    head = torch.argsort(word_clusters(ei, getattr(g, "edge_attr", None), V, labels), stable=True) if words else \
        torch.arange(V, device=dev)
    perm = torch.cat([head, order + V])
    new_id = torch.empty_like(perm)
    new_id[perm] = torch.arange(N, device=dev)
    out = Data()
    for k, v in g.__dict__.items():
        if k == "edge_index":
            out.edge_index = new_id[ei]                              # same shape and edge order ([2, E], whatever its strides)
        elif k == "x":
            out.x = _permute_features(v, perm, new_id, N)
        elif torch.is_tensor(v) and v.dim() >= 1 and v.size(0) == N and k != "edge_attr":
            setattr(out, k, v[perm.to(v.device)])
        else:
            setattr(out, k, v)
    return out, perm
