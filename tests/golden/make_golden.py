"""Generates the golden fixtures of tests/golden/ from the CPU oracle (oracle/gcn_oracle.py).

Run in the build container:  python tests/golden/make_golden.py
The reference's own implementation of this path cannot be imported here (torch_geometric and nltk
are not installed: ordinary ModuleNotFoundError), and its tests hold no golden for
the GCN path, so these vectors come from the restatement, cross-checked in tests/test_oracle.py
against an independent float64 dense formulation.  Fixtures are data only (inputs + expected
outputs).  The graph-builder inputs reuse the reference's own test data: the 2x8 token matrix of
textgcn/test/test_cfunc.py:83-87 and the edges it yields (SURVEY.md section 4 capture).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import gcn_oracle as O  # noqa: E402


class Bag:
    pass


def conv_case(name, ei, w, n, fin, fout, seed, add_self_loops=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, fin, generator=g)
    W = torch.randn(fin, fout, generator=g).requires_grad_()
    b = torch.randn(fout, generator=g).requires_grad_()
    x.requires_grad_()
    out = O.gcn_conv(x, ei, w, W, b, add_self_loops=add_self_loops)
    dout = torch.randn(n, fout, generator=g)
    out.backward(dout)
    tgt, src, what = O.normalized_coo(ei, w, n, add_self_loops)
    np.savez(os.path.join(HERE, name + ".npz"), edge_index=ei.numpy(),
             edge_weight=np.zeros(0, np.float32) if w is None else w.numpy(), n=n,
             add_self_loops=int(add_self_loops), x=x.detach().numpy(), W=W.detach().numpy(),
             b=b.detach().numpy(), out=out.detach().numpy(), dout=dout.numpy(),
             dx=x.grad.numpy(), dW=W.grad.numpy(), db=b.grad.numpy(),
             norm_target=tgt.numpy(), norm_source=src.numpy(), norm_weight=what.numpy())


def main():
    # (1) SURVEY.md 8(a) known-answer vector: asymmetric weights pin the direction convention
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    w = torch.tensor([2.0, 3.0, 0.5, 0.25])
    W = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], requires_grad=True)
    b = torch.tensor([0.1, -0.1])
    out = O.gcn_conv(torch.eye(3), ei, w, W, b)
    dout = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    out.backward(dout)
    np.savez(os.path.join(HERE, "known_answer.npz"), edge_index=ei.numpy(), edge_weight=w.numpy(),
             W=W.detach().numpy(), b=b.numpy(), out=out.detach().numpy(), dout=dout.numpy(),
             dW=W.grad.numpy())

    # (2) ~50-node random ASYMMETRIC graph with self loops (one node carries several) + duplicates
    g = torch.Generator().manual_seed(7)
    n = 53
    src = torch.randint(0, n, (400,), generator=g)
    dst = torch.randint(0, n, (400,), generator=g)
    loops = torch.tensor([3, 3, 17, 40, 3])
    src, dst = torch.cat([src, loops, src[:9]]), torch.cat([dst, loops, dst[:9]])
    perm = torch.randperm(src.numel(), generator=g)
    ei = torch.stack([src[perm], dst[perm]])
    w = torch.rand(ei.size(1), generator=g) * 3 + 0.05
    conv_case("random53", ei, w, n, 11, 24, seed=11)
    conv_case("random53_noloops_unweighted", ei, None, n, 5, 7, seed=12, add_self_loops=False)

    # (3) tiny TextGCN-shaped graph: V = 6 words from the reference's Cython golden input
    #     (test_cfunc.py:105-108 -> word-word edges of SURVEY.md section 4) + 2 documents, through
    #     the 2-layer GCN with dropout = 0: logits, loss, all gradients, and 3 Adam(amsgrad) steps.
    X = np.array([[0, 1, 2, 3, 4, -1, -1, -1], [5, 3, 4, 1, 2, 0, 5, 1]], dtype=np.int32)   # test_cfunc.py:105-108
    ww = np.array([[0, 1, 0, 2, 0, 5, 3, 4], [1, 0, 2, 0, 5, 0, 4, 3]], dtype=np.int64)
    ww_w = np.array([0.11778303, 0.11778303, 0.11778303, 0.11778303, 0.4054651, 0.4054651,
                     0.52324814, 0.52324814], dtype=np.float32)
    V, D = 6, 2
    occ = np.zeros((D, V))
    for d in range(D):
        for t in X[d]:
            if t >= 0:
                occ[d, t] += 1
    # sklearn TfidfTransformer defaults: smooth idf, l2 norm
    df = (occ > 0).sum(0)
    idf = np.log((1 + D) / (1 + df)) + 1
    tfidf = occ * idf
    tfidf /= np.linalg.norm(tfidf, axis=1, keepdims=True)
    dd, wd = np.nonzero(occ)
    coo = np.concatenate([ww.T, np.stack([dd + V, wd], 1), np.stack([wd, dd + V], 1)])
    wts = np.concatenate([ww_w, tfidf[dd, wd], tfidf[dd, wd]]).astype(np.float32)
    N = V + D
    bag = Bag()
    bag.edge_index = torch.from_numpy(coo).T
    bag.edge_attr = torch.from_numpy(wts)
    ar = torch.arange(N)
    bag.x = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N), (N, N))
    bag.y = torch.tensor([0, 0, 0, 0, 0, 0, 1, 2])
    bag.train_mask = torch.tensor([False] * V + [True, True])
    torch.manual_seed(44)
    model = O.GCNOracle(N, 3, n_hidden_gcn=8, dropout=0.0)
    init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=0.05, amsgrad=True)
    model.train()
    logits = model(bag)
    loss = torch.nn.CrossEntropyLoss()(logits[bag.train_mask], bag.y[bag.train_mask])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    grads = {k: p.grad.detach().clone().numpy() for k, p in model.named_parameters()}
    logits0 = logits.detach().numpy().copy()
    loss0 = loss.item()
    opt.step()
    losses = [loss0]
    for _ in range(2):
        l, _ = O.train_step(model, bag, opt)
        losses.append(l.item())
    model.eval()
    final_logits = model(bag).detach().numpy()
    save = dict(edge_index=coo.T.copy(), edge_attr=wts, y=bag.y.numpy(),
                train_mask=bag.train_mask.numpy(), n_vocab=V, logits0=logits0,
                losses=np.array(losses), final_logits=final_logits, tokens=X)
    for k, v in init.items():
        save["init." + k] = v
    for k, v in grads.items():
        save["grad." + k] = v
    np.savez(os.path.join(HERE, "tiny_textgcn.npz"), **save)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
