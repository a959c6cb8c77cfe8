"""pytextgcn_amd.reorder: the same graph under another numbering of its document nodes (host logic + the oracle)."""
import copy

import pytest
import torch

from oracle import gcn_oracle as O
from pytextgcn_amd import synth
from pytextgcn_amd.reorder import cluster_documents, reorder_documents


def _topical(n=4000, e=60000, topics=8):
    return synth.word_doc_graph(n, e, seed=17, n_classes=topics, n_topics=topics, doc_order="shuffled")


def test_reordered_graph_is_the_same_graph_under_another_numbering():
    g = _topical()
    N, V = g.y.numel(), g.n_vocab
    g2, perm = reorder_documents(g, n_clusters=8)
    assert torch.equal(torch.sort(perm).values, torch.arange(N)) and torch.equal(perm[:V], torch.arange(V))
    # every edge, in its old place in the list, with its old weight: (s', t') is (s, t) under the new names
    assert g2.edge_index.shape == g.edge_index.shape and torch.equal(perm[g2.edge_index], g.edge_index)
    assert g2.edge_attr is g.edge_attr and g2.n_vocab == V
    for k in ("y", "train_mask", "val_mask", "test_mask"):
        assert torch.equal(getattr(g2, k), getattr(g, k)[perm]), k
    assert g2.x.is_sparse and torch.equal(g2.x.to_dense(), torch.eye(N))
    # documents of one cluster are adjacent, and the clusters found from the edges alone follow the generator's topics
    lab = cluster_documents(g.edge_index, g.edge_attr, V, N, n_clusters=8)
    assert torch.equal(lab[perm[V:] - V], torch.sort(lab, stable=True).values)
    topic = g.y[V:]
    joint = torch.zeros(8, 8).index_put_((lab, topic), torch.ones(N - V), accumulate=True)
    assert float(joint.max(1).values.sum() / (N - V)) > 0.5                 # purity (a random labelling: 1 / 8)
    # deterministic
    assert torch.equal(lab, cluster_documents(g.edge_index, g.edge_attr, V, N, n_clusters=8))
    with pytest.raises(ValueError):
        reorder_documents(g, labels=torch.zeros(3, dtype=torch.int64))
    g_no = copy.copy(g)
    g_no.n_vocab = 0
    with pytest.raises(ValueError):
        reorder_documents(g_no)


def test_reordered_graph_gives_the_oracle_the_same_network():
    """The reference formulation (oracle) on the reordered graph with W1's rows permuted along: logits, loss and gradients
    are the original's under the permutation (the edge order is kept, so every sum runs over the same terms in the same
    order: bit for bit)."""
    g = _topical(1500, 20000, 5)
    _check_oracle_network_under(g, *reorder_documents(g, n_clusters=5))
    g3, perm3 = reorder_documents(g, n_clusters=5, words=True)     # word nodes by cluster as well, inside [0, n_vocab)
    V = g.n_vocab
    assert torch.equal(torch.sort(perm3[:V]).values, torch.arange(V)) and not torch.equal(perm3[:V], torch.arange(V))
    assert torch.equal(perm3[g3.edge_index], g.edge_index) and g3.n_vocab == V
    _check_oracle_network_under(g, g3, perm3)


def _check_oracle_network_under(g, g2, perm):
    N = g.y.numel()
    torch.manual_seed(3)
    a = O.GCNOracle(N, 5, n_hidden_gcn=16, dropout=0.0)
    b = copy.deepcopy(a)
    with torch.no_grad():
        b.layers[0].weight.copy_(a.layers[0].weight[perm])
    crit = torch.nn.CrossEntropyLoss()
    za, zb = a(g), b(g2)
    assert torch.equal(zb, za[perm])
    la, lb = crit(za[g.train_mask], g.y[g.train_mask]), crit(zb[g2.train_mask], g2.y[g2.train_mask])
    la.backward(), lb.backward()
    assert abs(la.item() - lb.item()) < 1e-6 * abs(la.item())      # (the mean runs over the rows in another order)
    assert torch.allclose(b.layers[0].weight.grad, a.layers[0].weight.grad[perm], rtol=1e-5, atol=1e-9)
    assert torch.allclose(b.layers[1].weight.grad, a.layers[1].weight.grad, rtol=1e-4, atol=1e-8)


def test_hierarchy_features_are_permuted_consistently():
    g = _topical(1200, 15000, 4)
    N, V = g.y.numel(), g.n_vocab
    gen = torch.Generator().manual_seed(1)
    H = torch.rand(N, 3, generator=gen) * (torch.rand(N, 3, generator=gen) < 0.5)
    H[:V] = 0
    ar = torch.arange(N)
    hr, hc = torch.nonzero(H, as_tuple=True)
    g.x = torch.sparse_coo_tensor(torch.stack([torch.cat([ar, hr]), torch.cat([ar, N + hc])]),
                                  torch.cat([torch.ones(N), H[hr, hc]]), (N, N + 3)).coalesce()
    g2, perm = reorder_documents(g, n_clusters=4)
    x2 = g2.x.to_dense()
    assert torch.equal(x2[:, :N], torch.eye(N)) and torch.equal(x2[:, N:], H[perm])
    g.x = torch.eye(N)                                          # dense one-hot features (sparse_features=False)
    g3, perm3 = reorder_documents(g, n_clusters=4)
    assert torch.equal(g3.x, torch.eye(N)) and torch.equal(perm3, perm)
