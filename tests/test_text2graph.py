"""Text2GraphTransformer.  CPU: the oracle against its committed fixture and the
product's host-side pieces (tokeniser / encoder, constructor surface).  GPU: the product against the
oracle on identical corpora, and BASELINE.json config c1 end to end (1k-doc corpus -> graph ->
2-layer GCN, hidden 64) against the CPU oracle."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import gcn_oracle as O, text2graph_oracle as TO
import pytextgcn_amd as pkg
from pytextgcn_amd import synth, text2graph

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "text2graph.npz"))
STOP = ["the", "a", "is", "and", "of"]


def check_same(g, b, exact_weights=True):
    assert g.n_vocab == b.n_vocab
    assert g.edge_index.dtype == torch.int64 and tuple(g.edge_index.shape) == tuple(b.edge_index.shape)
    assert torch.equal(g.edge_index.cpu(), b.edge_index)
    assert g.edge_attr.dtype == torch.float32
    if exact_weights:
        assert torch.equal(g.edge_attr.cpu(), b.edge_attr)
    else:
        assert torch.allclose(g.edge_attr.cpu(), b.edge_attr, rtol=1e-6)
    for k in ("y", "test_mask", "val_mask", "train_mask"):
        assert torch.equal(getattr(g, k).cpu(), getattr(b, k)), k
    assert torch.equal(g.x.cpu().to_dense(), b.x.to_dense())


def test_oracle_reproduces_its_fixture():
    b = TO.fit_transform(list(GOLD["docs"]), y=GOLD["y"].tolist(), test_idx=[6, 7], val_idx=[5], min_df=2,
                         window_size=5, max_df=0.9, stop_words=STOP)
    assert b.n_vocab == int(GOLD["n_vocab"]) and sorted(b.vocabulary) == list(GOLD["vocab_words"])
    np.testing.assert_equal(b.tokens, GOLD["tokens"])
    np.testing.assert_equal(b.edge_index.numpy(), GOLD["edge_index"])
    np.testing.assert_allclose(b.edge_attr.numpy(), GOLD["edge_attr"], rtol=1e-6)
    V, D = b.n_vocab, len(GOLD["docs"])
    # layout contract of text2graph.py:162-171,180-191
    ei = b.edge_index
    n_ww = int(((ei[0] < V) & (ei[1] < V)).sum())
    assert (ei[:, :n_ww] < V).all() and torch.equal(ei[0, :n_ww:2], ei[1, 1:n_ww:2])
    half = (ei.shape[1] - n_ww) // 2
    assert (ei[0, n_ww:n_ww + half] >= V).all() and (ei[1, n_ww + half:] >= V).all()
    assert torch.equal(ei[:, n_ww:n_ww + half], ei[:, n_ww + half:].flip(0))
    assert b.y[:V].sum() == 0 and b.y[V:].tolist() == GOLD["y"].tolist()
    assert b.test_mask.nonzero().flatten().tolist() == [V + 6, V + 7]
    assert b.val_mask.nonzero().flatten().tolist() == [V + 5]
    assert not b.train_mask[:V].any() and b.train_mask.sum() == D - 3
    # doc-word weights are sklearn's default TF-IDF: rows of unit L2 norm
    w = b.edge_attr[n_ww:n_ww + half].double()
    docs = ei[0, n_ww:n_ww + half] - V
    norms = torch.zeros(D, dtype=torch.float64).index_add_(0, docs, w * w)
    assert torch.allclose(norms[norms > 0], torch.ones(int((norms > 0).sum()), dtype=torch.float64), atol=1e-6)


def test_encoder_and_surface_match_the_reference_contract():
    vocab = {"cat": 0, "dog": 1, "fish": 2}
    X, L = text2graph._encode_input(["The CAT, the dog & a bird", "fish fish cat dog dog", "", "Cat-fish!"], 1,
                                    vocab, 0, 4, None)
    assert L == 5 and X.dtype == np.int32
    assert X.tolist() == [[0, 1, -1, -1, -1], [2, 2, 0, 1, 1], [-1] * 5, [0, 2, -1, -1, -1]]
    Xo, Lo = TO.encode_input(["The CAT, the dog & a bird", "fish fish cat dog dog", "", "Cat-fish!"], vocab, None)
    assert np.array_equal(X, Xo) and L == Lo
    X3, L3 = text2graph._encode_input(["fish fish cat dog dog"], 1, vocab, 0, 1, 3)      # max_length
    assert X3.tolist() == [[2, 2, 0]] and L3 == 3
    t = pkg.Text2GraphTransformer(rm_stopwords=False)
    assert t.get_params() == dict(min_df=5, window_size=20, save_path=None, n_jobs=1, max_df=1.0, verbose=0,
                                  rm_stopwords=False, sparse_features=True, max_length=None)
    assert t.stop_words is None
    sw = pkg.Text2GraphTransformer().stop_words
    assert len(sw) == 179 and {"the", "wouldn't", "yourselves"} <= sw
    with pytest.raises(FileNotFoundError):
        pkg.Text2GraphTransformer.load_graph("/nonexistent/graph.p")


def test_synthetic_corpus_is_deterministic_and_labelled():
    d1, y1 = synth.synthetic_corpus(50, 300, seed=3)
    d2, y2 = synth.synthetic_corpus(50, 300, seed=3)
    assert d1 == d2 and y1 == y2 and len(set(y1)) == 4 and all(len(d.split()) >= 12 for d in d1)


@pytest.mark.gpu
def test_gpu_transformer_matches_fixture_and_oracle(cuda, tmp_path):
    docs = list(GOLD["docs"])
    t = pkg.Text2GraphTransformer(min_df=2, window_size=5, max_df=0.9, rm_stopwords=False, save_path=str(tmp_path))
    t.stop_words = set(STOP)
    g = t.fit_transform(docs, y=GOLD["y"].tolist(), test_idx=[6, 7], val_idx=[5])
    assert np.array_equal(g.edge_index.numpy(), GOLD["edge_index"]) and g.edge_index.stride() == (1, 2)
    np.testing.assert_allclose(g.edge_attr.numpy(), GOLD["edge_attr"], rtol=1e-6)
    assert sorted(t.vocabulary) == list(GOLD["vocab_words"]) and g.n_vocab == int(GOLD["n_vocab"])
    b = TO.fit_transform(docs, y=GOLD["y"].tolist(), test_idx=[6, 7], val_idx=[5], min_df=2, window_size=5,
                         max_df=0.9, stop_words=STOP)
    check_same(g, b)
    saved = [f for f in os.listdir(tmp_path) if f.startswith("TGData_")]
    assert len(saved) == 1
    g2 = pkg.Text2GraphTransformer.load_graph(os.path.join(tmp_path, saved[0]))
    assert torch.equal(g2.edge_index, g.edge_index)
    # hierarchy features + max_length (perlevel_amazon.py:122; flat_dbpedia.py:34,70-71)
    hf = GOLD["hierarchy_feats"]
    th_ = pkg.Text2GraphTransformer(min_df=1, window_size=20, rm_stopwords=False, max_length=6)
    gh = th_.fit_transform(docs, y=GOLD["y"].tolist(), test_idx=[0], hierarchy_feats=torch.from_numpy(hf))
    assert np.array_equal(gh.edge_index.numpy(), GOLD["h_edge_index"])
    np.testing.assert_allclose(gh.edge_attr.numpy(), GOLD["h_edge_attr"], rtol=1e-6)
    xh = gh.x.coalesce()
    assert list(xh.shape) == GOLD["h_x_shape"].tolist()
    assert np.array_equal(xh.indices().numpy(), GOLD["h_x_indices"])
    assert np.array_equal(xh.values().numpy(), GOLD["h_x_values"])
    dense = pkg.Text2GraphTransformer(min_df=2, rm_stopwords=False, sparse_features=False).fit_transform(
        docs, y=GOLD["y"].tolist(), test_idx=[1])
    assert torch.equal(dense.x, torch.eye(dense.x.shape[0]))


@pytest.mark.gpu
def test_config_c1_text_to_logits_end_to_end(cuda):
    """BASELINE.json configs[0]: 1k documents -> Text2GraphTransformer -> 2-layer GCN (hidden 64):
    the product on the GPU against the oracle chain on the CPU, then a few training epochs."""
    docs, labels = synth.synthetic_corpus(1000, 1500, n_classes=4, seed=44)
    rng = np.random.default_rng(0)
    perm = rng.permutation(1000)
    test_idx, val_idx = perm[:100].tolist(), perm[100:200].tolist()
    t = pkg.Text2GraphTransformer(min_df=5, window_size=20, rm_stopwords=False)       # flat_amazon.py:66
    g = t.fit_transform(docs, y=labels, test_idx=test_idx, val_idx=val_idx)
    b = TO.fit_transform(docs, y=labels, test_idx=test_idx, val_idx=val_idx, min_df=5, window_size=20)
    check_same(g, b)
    N = g.x.shape[0]
    assert g.n_vocab > 300 and g.edge_index.shape[1] > 50000
    torch.manual_seed(0)
    ref = O.GCNOracle(N, 4, n_hidden_gcn=64, dropout=0.0)
    mine = pkg.GCN(N, 4, n_hidden_gcn=64, dropout=0.0)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(cuda).float()
    gd = g.to(cuda)
    err = (mine(gd).cpu() - ref(b)).abs().max().item() / ref(b).abs().max().item()
    assert err < 1e-5, err
    o_r = torch.optim.Adam(ref.parameters(), lr=0.05, amsgrad=True)
    o_m = pkg.optim.Adam(mine.parameters(), lr=0.05, amsgrad=True)
    for step in range(5):
        l_r, z_r = O.train_step(ref, b, o_r)
        mine.train()
        loss = pkg.functional.masked_cross_entropy(mine(gd), gd.y, gd.train_mask)
        o_m.zero_grad(set_to_none=True)
        loss.backward()
        o_m.step()
        assert abs(loss.item() - l_r.item()) < 1e-4 * abs(l_r.item()), step
    mine.eval()
    pred = mine(gd)[gd.test_mask].argmax(1).cpu()
    pred_ref = ref(b)[b.test_mask].argmax(1)
    assert (pred == pred_ref).float().mean() > 0.97
    pickle.loads(pickle.dumps(mine.cpu()))                   # th.save(gcn, ...) (flat_amazon.py:128)
