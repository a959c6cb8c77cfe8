"""Synthetic TextGCN-shaped graphs; the real Amazon / DBpedia CSVs are
absent from the reference tree (.MISSING_LARGE_BLOBS:1-3), so every benchmark and large parity
case is generated here.  Deterministic for a given seed WHATEVER the device: every random number is
drawn from a CPU `torch.Generator` (torch's mt19937 stream, the same on every machine) and moved
to `device`, where only deterministic work happens (searches, sorts, de-duplication) -- so the
50 M-edge configuration is still built on the GPU in seconds, and a graph generated there is the one
a CPU-only host regenerates from the same seed.  (SURVEY.md 8(d) names numpy's default_rng(44); the
torch CPU generator plays that role: seed 44, no device-dependent stream.)

Layout reproduced from textgcn/lib/text2graph.py:
  * node numbering: words [0, V), documents [V, V + D)                       (:169-170,183,191)
  * edge order: [word-word pairs interleaved (i,j),(j,i) | doc->word | word->doc] (:162-171)
  * `edge_index = coo.T`, a NON-contiguous view of an [E, 2] int64 array          (:192)
  * both directions carry the same weight; no self loops; no duplicates
  * features: sparse COO identity (:179,226-246); labels int64 with pseudo-label 0 on word nodes
    (:189-191); boolean train/val/test masks that are False on word nodes (:180-188)
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .data import Data


def _zipf_cdf(V: int, s: float, device) -> torch.Tensor:
    p = torch.arange(1, V + 1, dtype=torch.float64).pow_(-s)     # on the host: a device scan may round differently
    cdf = torch.cumsum(p, 0)
    return (cdf / cdf[-1]).float().to(device)


def _rand(n: int, gen: torch.Generator, device) -> torch.Tensor:
    """n uniform floats in [0, 1) from the CPU stream, on `device`."""
    return torch.rand(n, generator=gen).to(device)


def _randint(hi: int, n: int, gen: torch.Generator, device) -> torch.Tensor:
    return torch.randint(0, hi, (n,), generator=gen).to(device)


def _randperm(n: int, gen: torch.Generator, device) -> torch.Tensor:
    return torch.randperm(n, generator=gen).to(device)


def _sample(cdf: torch.Tensor, n: int, gen: torch.Generator) -> torch.Tensor:
    u = _rand(n, gen, cdf.device)
    return torch.searchsorted(cdf, u).clamp_(max=cdf.numel() - 1)


def _pick(keys: torch.Tensor, n: int, gen: torch.Generator) -> torch.Tensor:
    """n of the (unique, sorted) keys, chosen at random, returned sorted."""
    if keys.numel() == n:
        return keys
    sel = _randperm(keys.numel(), gen, keys.device)[:n]
    return keys[sel].sort().values


def word_doc_graph(n_nodes: int, n_edges: int, seed: int = 44, device="cpu", n_classes: int = 64,
                   vocab_frac: float = 0.1, doc_word_share: float = 0.7, zipf_s: float = 1.07,
                   features: str = "sparse_identity", n_topics: int = 0, topic_mass: float = 0.6,
                   doc_order: str = "by_topic") -> Data:
    """PMI / TF-IDF shaped word-document heterograph with exactly `n_edges` directed edges.

    `n_topics > 0` gives the corpus TOPICAL LOCALITY, which the plain generator has none of (every document draws from
    one global Zipf distribution): document d belongs to topic d * n_topics // D, a share `topic_mass` of its words comes
    from the topic's own slice of the vocabulary (Zipf within the slice), the rest from the global distribution; word-word
    pairs likewise fall inside one topic's slice with probability `topic_mass`; the label of a document is its topic
    (mod n_classes).  `doc_order="by_topic"` keeps the documents of a topic adjacent (a corpus file sorted by class, as
    the reference's DBpedia loader concatenates them, flat_dbpedia.py:41-71), `"shuffled"` re-labels them at random (same
    graph up to the numbering of the document nodes).  With `n_topics == 0` nothing changes: not one extra random number
    is drawn, the seed-44 benchmark graphs are bit for bit what they were."""
    if n_edges % 2:
        raise ValueError("n_edges must be even (every edge is emitted in both directions)")
    device = torch.device(device)
    gen = torch.Generator()                    # CPU stream: the graph does not depend on where it is built
    gen.manual_seed(seed)
    V = max(2, int(n_nodes * vocab_frac))
    D = n_nodes - V
    if D < 1:
        raise ValueError("n_nodes too small")
    n_dw = int(round(doc_word_share * n_edges / 2))
    n_ww = n_edges // 2 - n_dw
    n_ww_max = V * (V - 1) // 2
    if n_ww > n_ww_max:                       # tiny vocabularies: move the surplus to doc-word
        n_dw += n_ww - n_ww_max
        n_ww = n_ww_max
    if n_dw > D * V:
        raise ValueError("graph too dense for the requested shape")
    cdf = _zipf_cdf(V, zipf_s, device)
    word_perm = _randperm(V, gen, device)      # vocabulary order is not rank order
    if n_topics:
        if doc_order not in ("by_topic", "shuffled") or not 0.0 <= topic_mass <= 1.0 or V // n_topics < 2:
            raise ValueError("topics: doc_order is 'by_topic' or 'shuffled', 0 <= topic_mass <= 1, >= 2 words per topic")
        Vt = V // n_topics
        cdf_t = _zipf_cdf(Vt, zipf_s, device)
        topic_perm = _randperm(V, gen, device)             # topic t owns topic_perm[t * Vt : (t + 1) * Vt]

        def topical(words, topic):
            """Replace a share `topic_mass` of the globally drawn words by draws from the topic's own slice."""
            own = _rand(words.numel(), gen, device) < topic_mass
            return torch.where(own, topic_perm[topic * Vt + _sample(cdf_t, words.numel(), gen)], words)

    # ---- document-word incidences: k_d ~ clip(LogNormal, 4, 400) distinct Zipf words per doc ----
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = n_dw
    for round_ in range(64):
        if keys.numel() >= n_dw:
            break
        if round_ < 48:
            mean = max(1.0, 1.25 * want / D)
            sigma = 0.8
            mu = math.log(mean) - 0.5 * sigma * sigma
            k = torch.empty(D).log_normal_(mu, sigma, generator=gen).to(device)
            k = k.clamp_(min(4.0, mean), 400.0).round_().long().clamp_(max=V)
            docs = torch.repeat_interleave(torch.arange(D, device=device), k)
            words = word_perm[_sample(cdf, docs.numel(), gen)]
            if n_topics:
                words = topical(words, docs * n_topics // D)
        else:                                  # very dense requests: top up uniformly
            m = 2 * want + 16
            docs = _randint(D, m, gen, device)
            words = _randint(V, m, gen, device)
        keys = torch.unique(torch.cat([keys, docs * V + words]))
        want = max(n_dw - keys.numel(), 1)
    if keys.numel() < n_dw:
        raise RuntimeError("could not draw enough distinct doc-word pairs")
    keys = _pick(keys, n_dw, gen)
    dw_doc, dw_word = keys // V, keys % V

    # ---- word-word pairs: i < j sampled ~ Zipf x Zipf ------------------------------------------
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = n_ww
    for round_ in range(64):
        if keys.numel() >= n_ww:
            break
        m = 2 * want + 16
        if round_ < 48:
            a, b = word_perm[_sample(cdf, m, gen)], word_perm[_sample(cdf, m, gen)]
            if n_topics:                       # a pair falls inside ONE topic's slice with probability topic_mass
                t_pair = _randint(n_topics, m, gen, device)
                own = _rand(m, gen, device) < topic_mass
                a = torch.where(own, topic_perm[t_pair * Vt + _sample(cdf_t, m, gen)], a)
                b = torch.where(own, topic_perm[t_pair * Vt + _sample(cdf_t, m, gen)], b)
        else:
            a = _randint(V, m, gen, device)
            b = _randint(V, m, gen, device)
        ok = a != b
        lo, hi = torch.minimum(a, b)[ok], torch.maximum(a, b)[ok]
        keys = torch.unique(torch.cat([keys, lo * V + hi]))
        want = max(n_ww - keys.numel(), 1)
    if keys.numel() < n_ww:
        raise RuntimeError("could not draw enough distinct word-word pairs")
    keys = _pick(keys, n_ww, gen)
    ww_i, ww_j = keys // V, keys % V

    # ---- weights --------------------------------------------------------------------------------
    # PMI-like: Exp(1) clipped to (1e-10, 12]; TF-IDF-like: u / ||u||_2 per document, u ~ U(0.05, 1]
    w_ww = torch.empty(n_ww).exponential_(1.0, generator=gen).to(device).clamp_(1e-10, 12.0)
    u = _rand(n_dw, gen, device) * 0.95 + 0.05
    # per-document norms on the host in float64 (a device index_add_ adds in atomic, i.e. varying, order)
    sq = torch.zeros(D, dtype=torch.float64).index_add_(0, dw_doc.cpu(), (u.cpu().double()) ** 2)
    w_dw = u / sq.sqrt().float().to(device)[dw_doc]

    coo = torch.empty(n_edges, 2, dtype=torch.int64, device=device)
    coo[0:2 * n_ww:2, 0], coo[0:2 * n_ww:2, 1] = ww_i, ww_j
    coo[1:2 * n_ww:2, 0], coo[1:2 * n_ww:2, 1] = ww_j, ww_i
    a, b = 2 * n_ww, 2 * n_ww + n_dw
    coo[a:b, 0], coo[a:b, 1] = dw_doc + V, dw_word            # doc -> word
    coo[b:, 0], coo[b:, 1] = dw_word, dw_doc + V              # word -> doc
    edge_attr = torch.cat([w_ww.repeat_interleave(2), w_dw, w_dw]).float()

    # ---- labels, masks, features ---------------------------------------------------------------
    N = n_nodes
    y = torch.zeros(N, dtype=torch.int64, device=device)
    if n_topics:
        y[V:] = (torch.arange(D, device=device) * n_topics // D) % max(n_classes, 1)
        if doc_order == "shuffled":            # the same graph with the document nodes re-labelled at random
            relabel = _randperm(D, gen, device)
            is_doc0, is_doc1 = coo[:, 0] >= V, coo[:, 1] >= V
            coo[:, 0] = torch.where(is_doc0, relabel[(coo[:, 0] - V).clamp_(min=0)] + V, coo[:, 0])
            coo[:, 1] = torch.where(is_doc1, relabel[(coo[:, 1] - V).clamp_(min=0)] + V, coo[:, 1])
            y_docs = y[V:].clone()
            y[V + relabel] = y_docs
    else:
        y[V:] = _randint(n_classes, D, gen, device)
    order = _randperm(D, gen, device) + V
    n_test, n_val = D // 10, D // 10
    masks = [torch.zeros(N, dtype=torch.bool, device=device) for _ in range(3)]
    masks[0][order[:n_test]] = True
    masks[1][order[n_test:n_test + n_val]] = True
    masks[2][order[n_test + n_val:]] = True
    if features == "sparse_identity":
        ar = torch.arange(N, device=device)
        x = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N, device=device), (N, N))
        x = x.coalesce()
    elif features == "none":
        x = None
    else:
        raise ValueError(features)
    g = Data(x=x, edge_index=coo.T, edge_attr=edge_attr, y=y, test_mask=masks[0],
             val_mask=masks[1], train_mask=masks[2], n_vocab=V)
    g.n_classes = n_classes
    return g


def power_law_graph(n_nodes: int, n_edges: int, seed: int = 44, device="cpu", alpha: float = 2.1,
                    symmetric: bool = True, n_classes: int = 0, features: str = "none") -> Data:
    """Generic power-law graph (config c5 of BASELINE.json): endpoint i drawn with probability
    ~ degree weight d_i ~ Zipf(alpha) clipped to [1, 1e6]; weights U(0, 1]; no loops, no dups.
    `n_classes` > 0 adds uniform labels and 80/10/10 masks over ALL nodes (there are no word nodes here);
    `features="sparse_identity"` the one-hot feature matrix of text2graph.py:179."""
    device = torch.device(device)
    gen = torch.Generator()                    # CPU stream (see the module docstring)
    gen.manual_seed(seed)
    N = n_nodes
    uu = _rand(N, gen, device).clamp_(min=1e-12)
    deg_w = uu.cpu().double().pow(-1.0 / (alpha - 1.0)).clamp_(1.0, 1e6)  # inverse-CDF Pareto; host scan (see _zipf_cdf)
    cdf = torch.cumsum(deg_w, 0)
    cdf = (cdf / cdf[-1]).float().to(device)
    n_pairs = n_edges // 2 if symmetric else n_edges
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = n_pairs
    for round_ in range(64):
        if keys.numel() >= n_pairs:
            break
        m = int(1.2 * want) + 16
        if round_ < 48:
            a, b = _sample(cdf, m, gen), _sample(cdf, m, gen)
        else:
            a = _randint(N, m, gen, device)
            b = _randint(N, m, gen, device)
        ok = a != b
        a, b = a[ok], b[ok]
        if symmetric:
            a, b = torch.minimum(a, b), torch.maximum(a, b)
        keys = torch.unique(torch.cat([keys, a * N + b]))
        want = max(n_pairs - keys.numel(), 1)
    keys = _pick(keys, n_pairs, gen)
    a, b = keys // N, keys % N
    w = 1.0 - _rand(n_pairs, gen, device)
    if symmetric:
        coo = torch.empty(2 * n_pairs, 2, dtype=torch.int64, device=device)
        coo[0::2, 0], coo[0::2, 1] = a, b
        coo[1::2, 0], coo[1::2, 1] = b, a
        w = w.repeat_interleave(2)
    else:
        coo = torch.stack([a, b], 1)
    g = Data(x=None, edge_index=coo.T, edge_attr=w.float(), n_vocab=0)
    if n_classes > 0:
        g.y = _randint(n_classes, N, gen, device)
        u = _rand(N, gen, device)
        g.train_mask, g.val_mask, g.test_mask = u < 0.8, (u >= 0.8) & (u < 0.9), u >= 0.9
        g.n_classes = n_classes
    if features == "sparse_identity":
        ar = torch.arange(N, device=device)
        g.x = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N, device=device), (N, N)).coalesce()
    elif features != "none":
        raise ValueError(features)
    return g


def random_graph(n_nodes: int, n_edges: int, seed: int = 0, device="cpu", self_loops: int = 0,
                 duplicates: int = 0, weighted: bool = True) -> Data:
    """Unstructured ASYMMETRIC test graph: arbitrary directed edges, optional self loops (some
    repeated on one node) and duplicate edges -- everything gcn_norm has to cope with."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    src = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    dst = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if duplicates and src.numel():
        pick = torch.randint(0, src.numel(), (duplicates,), generator=gen)
        src, dst = torch.cat([src, src[pick]]), torch.cat([dst, dst[pick]])
    if self_loops:
        ls = torch.randint(0, n_nodes, (self_loops,), generator=gen)
        ls = torch.cat([ls, ls[: max(1, self_loops // 3)]])       # several loops on one node
        src, dst = torch.cat([src, ls]), torch.cat([dst, ls])
    perm = torch.randperm(src.numel(), generator=gen)
    src, dst = src[perm], dst[perm]
    w = (torch.rand(src.numel(), generator=gen) * 2 + 0.01) if weighted else None
    ei = torch.stack([src, dst]).to(device)
    return Data(x=None, edge_index=ei, edge_attr=None if w is None else w.to(device), n_vocab=0)


def synthetic_corpus(n_docs: int = 1000, vocab_size: int = 2000, n_classes: int = 4, seed: int = 44,
                     min_len: int = 12, max_len: int = 60):
    """A labelled toy corpus standing in for the absent Amazon CSVs (.MISSING_LARGE_BLOBS:1-3;
    BASELINE.json config c1): pseudo-words drawn from a Zipf law that is tilted per class, so a
    TextGCN can actually learn the labels.  Returns (list of str, list of int)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    syll = ["ka", "lo", "mi", "ren", "tu", "sa", "vi", "dor", "na", "pel", "qu", "zan", "fi", "gro", "hu", "bex"]
    words = []
    i = 0
    while len(words) < vocab_size:
        n = 2 + i % 3
        w, k = "", i
        for _ in range(n):
            w += syll[k % len(syll)]
            k //= len(syll)
        words.append(w + (str(i // 4096) if i >= 4096 else ""))
        i += 1
    words = list(dict.fromkeys(words))[:vocab_size]
    base = 1.0 / np.arange(1, len(words) + 1) ** 1.05
    docs, labels = [], []
    for d in range(n_docs):
        c = int(rng.integers(0, n_classes))
        p = base.copy()
        p[c::n_classes] *= 4.0                       # class-specific words are four times as likely
        p /= p.sum()
        n = int(rng.integers(min_len, max_len + 1))
        toks = rng.choice(len(words), size=n, p=p)
        text = " ".join(words[t] for t in toks)
        if d % 7 == 0:
            text = text.replace(" ", ", ", 2).capitalize() + "."      # punctuation / case for the tokenizer
        docs.append(text)
        labels.append(c)
    return docs, labels
