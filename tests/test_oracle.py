"""CPU tests of the oracle itself: the restatement of PyG-1.6.3 GCNConv (oracle/gcn_oracle.py) is
checked against the SURVEY.md 8(a) known-answer vector, against an independent float64 dense
formulation D^-1/2 (A + I') D^-1/2, against its own committed golden fixtures, and the C CSR
restatement (oracle/csr_spmm.c) is checked against the gather/scatter formulation."""
import os

import numpy as np
import pytest
import torch

from oracle import csr_oracle, gcn_oracle as O
from pytextgcn_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def test_known_answer_vector():
    # asymmetric weights pin target/source direction
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    w = torch.tensor([2.0, 3.0, 0.5, 0.25])
    W = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], requires_grad=True)
    b = torch.tensor([0.1, -0.1])
    _, what = O.gcn_norm(ei, w, 3)
    exp_w = torch.tensor([.5547002, .8320503, .2264554, .1132277, .25, .3076923, .6666666])
    assert torch.allclose(what, exp_w, atol=1e-6)
    out = O.gcn_conv(torch.eye(3), ei, w, W, b)
    exp = torch.tensor([[2.8461509, 3.7282014], [2.1439157, 2.9195361], [4.1126990, 4.8058214]])
    assert torch.allclose(out, exp, atol=2e-6)
    out.backward(torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]))
    exp_dw = torch.tensor([[.25, .5547002], [1.0585058, .5341477], [.6666666, .7798944]])
    assert torch.allclose(W.grad, exp_dw, atol=2e-6)
    gold = np.load(os.path.join(GOLD, "known_answer.npz"))
    assert np.allclose(gold["out"], exp.numpy(), atol=2e-6)
    assert np.allclose(gold["dW"], exp_dw.numpy(), atol=2e-6)


@pytest.mark.parametrize("seed,n,e,loops,dups,weighted,add_loops", [
    (0, 1, 0, 0, 0, True, True), (1, 7, 20, 0, 0, True, True), (2, 33, 200, 5, 9, True, True),
    (3, 64, 500, 8, 20, False, True), (4, 40, 150, 6, 5, True, False), (5, 64, 40, 0, 0, True, False),
])
def test_restatement_equals_dense_formulation(seed, n, e, loops, dups, weighted, add_loops):
    g = synth.random_graph(n, e, seed=seed, self_loops=loops, duplicates=dups, weighted=weighted)
    ei, w = g.edge_index, g.edge_attr
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 6, generator=gen)
    W = torch.randn(6, 5, generator=gen)
    b = torch.randn(5, generator=gen)
    out = O.gcn_conv(x, ei, w, W, b, add_self_loops=add_loops)
    M = O.dense_norm_adj(ei, w, n, add_self_loops=add_loops)
    ref = (M @ (x.double() @ W.double()) + b.double()).float()
    assert rel_err(out, ref) < 2e-6


def test_sparse_identity_features_equal_dense():
    g = synth.word_doc_graph(200, 1200, seed=3)
    W = torch.randn(200, 9)
    a = O.gcn_conv(g.x, g.edge_index, g.edge_attr, W, None)
    b = O.gcn_conv(torch.eye(200), g.edge_index, g.edge_attr, W, None)
    assert torch.equal(a, b)


@pytest.mark.parametrize("transpose", [False, True])
def test_csr_restatement_equals_gather_scatter(transpose):
    g = synth.random_graph(300, 4000, seed=9, self_loops=10, duplicates=30)
    ei, w = g.edge_index, g.edge_attr
    x = torch.randn(300, 37)
    b = torch.randn(37)
    nei, nw = O.gcn_norm(ei, w, 300)
    if transpose:
        nei = nei.flip(0)
    ref = O.propagate(nei, x, nw, 300) + b
    rp, c, v = csr_oracle.normalized_csr(ei, w, 300, transpose=transpose)
    assert rp[-1].item() == nw.numel()
    got = csr_oracle.csr_spmm(rp, c, v, x, b)
    got64 = csr_oracle.csr_spmm(rp, c, v, x, b, acc64=True)
    assert rel_err(got, ref) < 1e-6 and rel_err(got64, ref) < 1e-6
    assert rel_err(csr_oracle.colsum(x), x.double().sum(0).float()) < 1e-6


def test_restatement_equals_scipy_formulation_at_config_c2_size():
    """A second, independent check of the restatement at benchmark scale (config c2: 100 k nodes, 2 M edges):
    M = D^-1/2 (A + I) D^-1/2 assembled with scipy.sparse in float64 (no code shared with the oracle) against
    gcn_norm + propagate in float32, forward and transposed, and against the C CSR restatement."""
    import scipy.sparse as sp
    N, E, F = 100_000, 2_000_000, 16
    g = synth.word_doc_graph(N, E, seed=44)
    ei, w = g.edge_index, g.edge_attr
    src, dst = ei[0].numpy(), ei[1].numpy()
    A = sp.coo_matrix((w.double().numpy(), (dst, src)), shape=(N, N)).tocsr()      # M[target, source]
    A = A + sp.identity(N, dtype=np.float64, format="csr")                          # no input loops in this graph
    deg = np.asarray(A.sum(axis=1)).ravel()                                         # weighted in-degree at the target
    dis = 1.0 / np.sqrt(deg)
    M = sp.diags(dis) @ A @ sp.diags(dis)
    x = torch.randn(N, F, generator=torch.Generator().manual_seed(7))
    nei, nw = O.gcn_norm(ei, w, N)
    for transpose in (False, True):
        ref = torch.from_numpy((M.T if transpose else M) @ x.double().numpy()).float()
        got = O.propagate(nei.flip(0) if transpose else nei, x, nw, N)
        assert rel_err(got, ref) < 2e-6, transpose
        rp, c, v = csr_oracle.normalized_csr(ei, w, N, transpose=transpose)
        assert rel_err(csr_oracle.csr_spmm(rp, c, v, x), ref) < 2e-6, transpose


def test_oracle_autograd_is_transposed_operator():
    g = synth.random_graph(50, 300, seed=4, self_loops=3, duplicates=4)
    x = torch.randn(50, 8, requires_grad=True)
    nei, nw = O.gcn_norm(g.edge_index, g.edge_attr, 50)
    out = O.propagate(nei, x, nw, 50)
    dout = torch.randn(50, 8)
    out.backward(dout)
    assert rel_err(x.grad, O.propagate(nei.flip(0), dout, nw, 50)) < 1e-6


def test_gcn_composition_has_no_activation_and_dropout_between_layers():
    g = synth.word_doc_graph(120, 800, seed=5, n_classes=4)
    torch.manual_seed(0)
    m = O.GCNOracle(120, 4, n_hidden_gcn=16, dropout=0.0).eval()
    l0, l1 = m.layers
    h = O.gcn_conv(g.x, g.edge_index, g.edge_attr, l0.weight, l0.bias)
    z = O.gcn_conv(h, g.edge_index, g.edge_attr, l1.weight, l1.bias)     # no ReLU in between
    assert torch.equal(m(g), z)
    assert (h < 0).any()
    m3 = O.GCNOracle(120, 4, n_gcn=3, n_hidden_gcn=16)
    assert [tuple(l.weight.shape) for l in m3.layers] == [(120, 16), (16, 16), (16, 4)]
    assert sorted(m.state_dict()) == ["layers.0.bias", "layers.0.weight", "layers.1.bias",
                                      "layers.1.weight"]


@pytest.mark.parametrize("name", ["random53", "random53_noloops_unweighted"])
def test_golden_conv_fixtures_reproduce(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ei = torch.from_numpy(z["edge_index"])
    w = torch.from_numpy(z["edge_weight"]) if z["edge_weight"].size else None
    n, loops = int(z["n"]), bool(z["add_self_loops"])
    x = torch.from_numpy(z["x"]).requires_grad_()
    W = torch.from_numpy(z["W"]).requires_grad_()
    b = torch.from_numpy(z["b"]).requires_grad_()
    out = O.gcn_conv(x, ei, w, W, b, add_self_loops=loops)
    out.backward(torch.from_numpy(z["dout"]))
    for got, key in [(out, "out"), (x.grad, "dx"), (W.grad, "dW"), (b.grad, "db")]:
        assert rel_err(got.detach(), torch.from_numpy(z[key])) < 1e-6, key
    M = O.dense_norm_adj(ei, w, n, add_self_loops=loops)
    ref = (M @ (x.detach().double() @ W.detach().double()) + b.detach().double()).float()
    assert rel_err(torch.from_numpy(z["out"]), ref) < 2e-6


def test_golden_tiny_textgcn_reproduces():
    z = np.load(os.path.join(GOLD, "tiny_textgcn.npz"))

    class G:
        pass
    g = G()
    N = int(z["y"].shape[0])
    g.edge_index = torch.from_numpy(z["edge_index"])
    g.edge_attr = torch.from_numpy(z["edge_attr"])
    ar = torch.arange(N)
    g.x = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N), (N, N))
    g.y = torch.from_numpy(z["y"])
    g.train_mask = torch.from_numpy(z["train_mask"])
    m = O.GCNOracle(N, 3, n_hidden_gcn=8, dropout=0.0)
    m.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init.")})
    opt = torch.optim.Adam(m.parameters(), lr=0.05, amsgrad=True)
    losses = [O.train_step(m, g, opt)[0].item() for _ in range(3)]
    assert np.allclose(losses, z["losses"], rtol=1e-5)
    assert np.allclose(m(g).detach().numpy(), z["final_logits"], rtol=1e-4, atol=1e-6)
    # the word-word block of this fixture is the reference Cython module's own output
    # (textgcn/test/test_cfunc.py:105-108 input; SURVEY.md section 4 capture): symmetric pairs
    ei = z["edge_index"][:, :8]
    assert (ei[:, 0::2] == ei[::-1, 1::2]).all()
