"""Worker side of tests/test_sharded.py: runs on every rank of a gloo (CPU) process group.  The
partition / exchange logic under test is pytextgcn_amd.sharded; the per-rank local operators are
provided by a TEST-ONLY engine built on the CPU oracle (the package ships only the HIP engine)."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import csr_oracle, gcn_oracle as O  # noqa: E402
from pytextgcn_amd import sharded, synth  # noqa: E402


class _OracleOp:
    def __init__(self, row, col, val, n_rows, n_cols):
        self.n_rows, self.n_cols = n_rows, n_cols
        self.csr = csr_oracle.coo_to_csr(row, col, val, n_rows)

    def export_csr(self):
        return self.csr

    def spmm(self, x, bias=None, x2=None, out=None, accumulate=False):
        if x2 is not None:
            x = torch.cat([x, x2])
        assert x.shape[0] == self.n_cols
        y = csr_oracle.csr_spmm(*self.csr, x.contiguous(), bias)
        if accumulate:                       # GraphPlan.spmm(accumulate=True): added to `out` on the rows with entries
            assert bias is None and out is not None
            has = (self.csr[0][1:] > self.csr[0][:-1])
            out[has] += y[has]
            return out
        if out is not None:
            out.copy_(y)
            return out
        return y


class OracleEngine:
    def gcn_norm(self, edge_index, edge_weight, num_nodes, add_self_loops, degree_sum="reference"):
        """(deg^-1/2, loop weight) per node, from the oracle's add_remaining_self_loops + degree sum ("reference": one
        fp32 accumulator per node in edge order, as PyG's scatter_add on the CPU; "accurate": float64 sums)."""
        w = edge_weight if edge_weight is not None else torch.ones(edge_index.size(1))
        fill = float(add_self_loops)
        loop_w = torch.zeros(num_nodes)
        if add_self_loops:
            ei, w = O.add_remaining_self_loops(edge_index, w, fill, num_nodes)
            loop_w = w[-num_nodes:].clone()
        else:
            ei = edge_index
        if degree_sum == "accurate":
            deg = torch.zeros(num_nodes, dtype=torch.float64).index_add_(0, ei[1], w.double()).float()
        else:
            deg = torch.zeros(num_nodes).index_add_(0, ei[1], w)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        return dis, loop_w

    def make_op(self, row, col, val, n_rows, n_cols):
        return _OracleOp(row, col, val, n_rows, n_cols)

    def colsum(self, g):
        return csr_oracle.colsum(g.contiguous())

    def xw(self, x, w):
        return torch.matmul(x, w)                 # test-only engine: the dense layers of the CPU rehearsal

    # the three layer-2 products of the narrow exchange (pytextgcn_amd/narrow.py), without dropout: the fused mask is a
    # device hash this CPU engine does not have
    def gemm_nn(self, a, b, p=0.0, seed=None, record_mask=False, keys=None, out=None):
        assert seed is None and not record_mask
        c = torch.matmul(a, b)
        if out is not None:
            out.copy_(c)
            c = out
        return c

    def gemm_nt(self, a, b, p=0.0, seed=None, mask=None, keys=None):
        assert seed is None and mask is None
        return torch.matmul(a, b.t())

    def gemm_tn(self, a, g, p=0.0, seed=None, mask=None, keys=None):
        assert seed is None and mask is None
        return torch.matmul(a.t(), g)


def rel_err(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


def make_graph(kind):
    if kind == "wordoc":
        g = synth.word_doc_graph(600, 7000, seed=11, n_classes=5)
        return g, torch.arange(600) < g.n_vocab
    if kind == "wordoc_big":          # long hub rows (segments) + F = 200 on the GPU
        g = synth.word_doc_graph(30000, 600000, seed=14, n_classes=5)
        return g, torch.arange(30000) < g.n_vocab
    if kind == "wordoc_allhubs":
        return synth.word_doc_graph(400, 4000, seed=12, n_classes=5), None
    if kind == "powerlaw_allhubs":    # sparse enough that a rank's rows reference only part of the other ranks' nodes
        g = synth.power_law_graph(900, 2400, seed=15, n_classes=5, features="sparse_identity")
        return g, None
    if kind == "powerlaw_big_allhubs":   # the GPU form of the same situation: long rows (segments) + F = 200, halo < operand
        g = synth.power_law_graph(30000, 300000, seed=16, n_classes=5, features="sparse_identity")
        return g, None
    if kind in ("asym", "asym_keep_loops", "asym_raw"):
        g = synth.random_graph(300, 2500, seed=13, self_loops=7, duplicates=11)
        g.y = torch.randint(0, 5, (300,), generator=torch.Generator().manual_seed(1))
        g.train_mask = torch.rand(300, generator=torch.Generator().manual_seed(2)) < 0.5
        ar = torch.arange(300)
        g.x = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(300), (300, 300))
        return g, None
    raise ValueError(kind)


def check(kind, device="cpu"):
    """`kind` may carry "@accurate": the opt-in normalisation mode (bitwise symmetric operator, one pair of local operators
    for M and M^T); without it the package default runs (the reference-order mode: M^T gets operators of its own)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    kind, _, mode = kind.partition("@")
    mode = mode or None
    g, hubs = make_graph(kind)
    N = g.y.numel()
    if device != "cpu":
        return check_hip(kind, g, hubs, N, torch.device(device))
    if kind in ("asym_keep_loops", "asym_raw"):
        # GCNConv(add_self_loops=False): input loops stay ordinary edges, none are added;
        # GCNConv(normalize=False): the raw weights, no loops (PyG adds them inside gcn_norm only)
        norm = kind == "asym_keep_loops"
        sg = sharded.ShardedGraph(g.edge_index, g.edge_attr, N, hubs=None, engine=OracleEngine(),
                                  add_self_loops=False, normalize=norm)
        assert not sg.symmetric
        gen = torch.Generator().manual_seed(6)
        x = torch.randn(N, 9, generator=gen)
        if norm:
            nei, nw = O.gcn_norm(g.edge_index, g.edge_attr, N, add_self_loops=False)
        else:
            nei, nw = g.edge_index, g.edge_attr
        for transpose in (False, True):
            ref = O.propagate(nei.flip(0) if transpose else nei, x, nw, N)
            got = sg.gather_rows(sg.spmm(sg.scatter_rows(x), None, transpose=transpose))
            assert rel_err(got, ref) < 1e-5, (kind, transpose, rel_err(got, ref))
        return
    if kind == "wordoc":
        sg = sharded.ShardedGraph.from_data(g, engine=OracleEngine(), degree_sum=mode)   # hubs = words, from n_vocab
        assert torch.equal(sg.part.hub_mask, hubs)
    else:
        sg = sharded.ShardedGraph(g.edge_index, g.edge_attr, N, hubs=hubs, engine=OracleEngine(), degree_sum=mode)
    assert sg.degree_sum == (mode or "reference")
    # every node is owned exactly once
    own = [sg.part.owned(q) for q in range(world)]
    allids = torch.cat([o[o >= 0] for o in own])
    assert allids.numel() == N and allids.unique().numel() == N
    # PyG's association (dis[s] * w) * dis[t] rounds (i, j) and (j, i) apart: only the accurate mode keeps a symmetric
    # graph's operator bitwise symmetric (then one pair of local operators serves both directions)
    assert sg.symmetric == (kind != "asym" and sg.degree_sum == "accurate") and len(sg.dirs) == (1 if sg.symmetric else 2)
    # balance: rows equal by construction, non-zeros within 25 %
    nnz = torch.tensor([float(sum(op.csr[0][-1].item() for op in sg.ops[0] if op is not None))])
    lo, hi = nnz.clone(), nnz.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN), dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert hi.item() <= 1.25 * lo.item() + 64, (lo, hi)

    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, 12, generator=gen)
    b = torch.randn(12, generator=gen)
    nei, nw = O.gcn_norm(g.edge_index, g.edge_attr, N)
    for transpose in (False, True):
        ref = O.propagate(nei.flip(0) if transpose else nei, x, nw, N) + b
        got = sg.gather_rows(sg.spmm(sg.scatter_rows(x), b, transpose=transpose))
        assert rel_err(got, ref) < 1e-5, (kind, transpose, rel_err(got, ref))

    check_exchange_forms(sg, x, b)
    check_offline_construction(sg, g, hubs, N)
    if sg.rp > 0:
        check_narrow_exchange(sg, g, N)
        for narrow in (False, True):
            check_rows_option(sg, g, N, narrow=narrow)
        check_hierarchy_block(sg, g, N)
        for narrow in (False, True):
            check_flat_loop(sg, g, N, narrow=narrow)
        sg.set_rs_chunks(2)                                  # A'_r's kept columns cut per row chunk
        try:
            for narrow in (False, True):
                check_rows_option(sg, g, N, narrow=narrow, steps=1)
            for form in ("p2p", "halo"):                     # the chunks' reduce lists serve the restricted operators
                sg.exchange = form
                check_rows_option(sg, g, N, narrow=False, steps=1)
        finally:
            sg.exchange = "collective"
            sg.set_rs_chunks(1)

    # model level: ShardedGCN vs the oracle GCN, 3 Adam(amsgrad) steps, dropout off
    torch.manual_seed(3)
    ref = O.GCNOracle(N, 5, n_gcn=3, n_hidden_gcn=16, dropout=0.0)
    mine = sharded.ShardedGCN(sg, N, 5, n_gcn=3, n_hidden_gcn=16, dropout=0.0)
    mine.load_full_state_dict(ref.state_dict())
    o_r = torch.optim.Adam(ref.parameters(), lr=0.05, amsgrad=True)
    o_m = torch.optim.Adam(mine.parameters(), lr=0.05, amsgrad=True)
    y_l, m_l = sg.scatter_rows(g.y), sg.scatter_rows(g.train_mask)
    crit = torch.nn.CrossEntropyLoss()
    for step in range(3):
        ref.train(), mine.train()
        lo_r = ref(g)
        loss_r = crit(lo_r[g.train_mask], g.y[g.train_mask])
        o_r.zero_grad(set_to_none=True)
        loss_r.backward()
        lo_m = mine()
        loss_m = sharded.sharded_cross_entropy(sg, lo_m, y_l, m_l)
        o_m.zero_grad(set_to_none=True)
        loss_m.backward()
        mine.sync_grads()
        total = loss_m.detach().clone().reshape(1)
        dist.all_reduce(total)
        assert abs(total.item() - loss_r.item()) < 1e-5 * abs(loss_r.item()), (step, total, loss_r)
        assert rel_err(sg.gather_rows(lo_m.detach()), lo_r.detach()) < 1e-4, step
        if step == 0:
            gw1 = sg.gather_rows(mine.weights[0].grad)
            assert rel_err(gw1, ref.layers[0].weight.grad) < 5e-5
            for i in (1, 2):
                assert rel_err(mine.weights[i].grad, ref.layers[i].weight.grad) < 5e-5, i
            for i in (0, 1, 2):
                assert rel_err(mine.biases[i].grad, ref.layers[i].bias.grad) < 5e-5, i
        o_r.step(), o_m.step()
    sd = mine.full_state_dict()
    for k, v in ref.state_dict().items():
        assert rel_err(sd[k], v) < 1e-3, k

    # opt-in activation reuse: eval forward, then a training forward on the same weights re-uses layer 1
    # (one distributed SpMM less), bitwise the same logits and gradients; an optimizer step invalidates
    import pytextgcn_amd as pkg
    calls = {"n": 0}
    real_spmm = sg.spmm

    def counting_spmm(*a, **kw):
        calls["n"] += 1
        return real_spmm(*a, **kw)
    sg.spmm = counting_spmm
    try:
        def fwd_bwd():
            mine.train()
            out = mine()
            loss = sharded.sharded_cross_entropy(sg, out, y_l, m_l)
            o_m.zero_grad(set_to_none=True)
            loss.backward()
            return out.detach().clone(), mine.weights[0].grad.clone()
        plain_out, plain_g = fwd_bwd()
        pkg.enable_activation_reuse(True)
        mine.eval()
        with torch.no_grad():
            ev = mine()
        calls["n"] = 0
        out, g1 = fwd_bwd()
        assert calls["n"] == 3 + 3 - 1, calls              # 3 layers forward + 3 backward, layer-1 forward re-used
        assert torch.equal(out, plain_out) and torch.equal(g1, plain_g)
        mine.sync_grads()
        o_m.step()                                           # bumps the version counters
        calls["n"] = 0
        mine.eval()
        with torch.no_grad():
            ev2 = mine()
        assert calls["n"] == 3 and not torch.equal(ev2, ev)
    finally:
        pkg.enable_activation_reuse(False)
        sg.spmm = real_spmm


def check_flat_loop(sg, g, N, dev=None, hidden=16, classes=8, narrow=False):
    """sharded.FlatLoop against its loop body written out by hand (same switches: bit for bit on one engine) and its
    global losses against the sum of the ranks' shares."""
    import pytextgcn_amd as pkg
    dev = dev if dev is not None else torch.device("cpu")
    torch.manual_seed(23)
    init = O.GCNOracle(N, classes, n_hidden_gcn=hidden, dropout=0.0).state_dict()
    y_l = sg.scatter_rows(g.y.to(dev) % classes)
    tr_l, va_l = sg.scatter_rows(g.train_mask.to(dev)), sg.scatter_rows(g.val_mask.to(dev))

    def make():
        m = sharded.ShardedGCN(sg, N, classes, n_hidden_gcn=hidden, dropout=0.0, narrow_exchange=narrow).to(dev)
        m.load_full_state_dict(init)
        o = torch.optim.Adam(m.parameters(), lr=0.02, amsgrad=True) if dev.type == "cpu" else \
            pkg.optim.Adam(m.parameters(), lr=0.02, amsgrad=True)
        return m, o
    a, oa = make()
    with sharded.FlatLoop(a, y_l, tr_l, va_l, optimizer=oa) as loop:
        got = [loop.epoch() for _ in range(3)]
    from pytextgcn_amd import conv as conv_
    assert not conv_._REUSE and loop.epochs == 3
    b, ob = make()
    if hasattr(ob, "fuse_into_backward"):
        ob.fuse_into_backward(b.weights[0])
    rows_eval = tr_l | va_l
    sg.prepare_rows(tr_l), sg.prepare_rows(rows_eval)        # collective, once; forward(rows=...) only looks them up
    pkg.enable_activation_reuse(True)
    try:
        for step in range(3):
            b.train()
            loss = sharded.sharded_cross_entropy(sg, b(rows=tr_l), y_l, tr_l)
            ob.zero_grad(set_to_none=True)
            loss.backward()
            b.sync_grads()
            ob.step()
            b.eval()
            with torch.no_grad():
                vl, pred = sharded.sharded_cross_entropy(sg, b(rows=rows_eval), y_l, va_l, return_pred=True)
            both = torch.stack([loss.detach(), vl.detach()]).float()
            dist.all_reduce(both)
            assert got[step][0] == both[0].item() and got[step][1] == both[1].item(), (step, got[step][:2], both)
            assert (got[step][2] == pred[va_l].cpu().numpy()).all() and (got[step][3] == pred[tr_l].cpu().numpy()).all()
    finally:
        pkg.enable_activation_reuse(False)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def check_hierarchy_block(sg, g, N, dev=None, hidden=16, classes=5, F_h=6):
    """ShardedGCN with the features [I_N | H] of the hierarchical scripts (text2graph.py:237-241): against the oracle GCN on
    the whole graph with the same sparse feature matrix -- loss, logits and every gradient (W1's identity rows gathered, the
    replicated rows W1[N:] summed by sync_grads) over two Adam(amsgrad) steps; the state dict round trip."""
    import copy
    dev = dev if dev is not None else torch.device("cpu")
    gen = torch.Generator().manual_seed(31)
    H = torch.rand(N, F_h, generator=gen) * (torch.rand(N, F_h, generator=gen) < 0.4)
    H[:int(getattr(g, "n_vocab", 0) or 0)] = 0                    # word rows carry no hierarchy features (:238-241)
    ar = torch.arange(N)
    hr, hc = torch.nonzero(H, as_tuple=True)
    x = torch.sparse_coo_tensor(torch.stack([torch.cat([ar, hr]), torch.cat([ar, N + hc])]),
                                torch.cat([torch.ones(N), H[hr, hc]]), (N, N + F_h)).coalesce()
    g2 = copy.copy(g)
    g2.x = x
    torch.manual_seed(8)
    ref = O.GCNOracle(N + F_h, classes, n_hidden_gcn=hidden, dropout=0.0)
    H_l = sg.scatter_rows(H.to(dev))
    for feats in (H_l, H_l.to_sparse()):
        mine = sharded.ShardedGCN(sg, N + F_h, classes, n_hidden_gcn=hidden, dropout=0.0, hierarchy_feats=feats).to(dev)
        mine.load_full_state_dict(ref.state_dict())
        ref_i = copy.deepcopy(ref)
        o_r = torch.optim.Adam(ref_i.parameters(), lr=0.05, amsgrad=True)
        if dev.type == "cpu":
            o_m = torch.optim.Adam(mine.parameters(), lr=0.05, amsgrad=True)
        else:
            import pytextgcn_amd as pkg
            o_m = pkg.optim.Adam(mine.parameters(), lr=0.05, amsgrad=True)
        y_l, m_l = sg.scatter_rows(g.y.to(dev) % classes), sg.scatter_rows(g.train_mask.to(dev))
        crit = torch.nn.CrossEntropyLoss()
        for step in range(2):
            ref_i.train(), mine.train()
            lo_r = ref_i(g2)
            loss_r = crit(lo_r[g.train_mask], g.y[g.train_mask] % classes)
            o_r.zero_grad(set_to_none=True)
            loss_r.backward()
            lo_m = mine()
            loss_m = sharded.sharded_cross_entropy(sg, lo_m, y_l, m_l)
            o_m.zero_grad(set_to_none=True)
            loss_m.backward()
            mine.sync_grads()
            total = loss_m.detach().clone().reshape(1)
            dist.all_reduce(total)
            assert abs(total.item() - loss_r.item()) < 1e-5 * abs(loss_r.item()), (step, total, loss_r)
            assert rel_err(sg.gather_rows(lo_m.detach()).cpu(), lo_r.detach()) < 1e-5, step
            gw = ref_i.layers[0].weight.grad
            assert rel_err(sg.gather_rows(mine.weights[0].grad).cpu(), gw[:N]) < 2e-5, step
            assert rel_err(mine.weight_h.grad.cpu(), gw[N:]) < 2e-5, (step, rel_err(mine.weight_h.grad.cpu(), gw[N:]))
            assert rel_err(mine.weights[1].grad.cpu(), ref_i.layers[1].weight.grad) < 2e-5
            for i in (0, 1):
                assert rel_err(mine.biases[i].grad.cpu(), ref_i.layers[i].bias.grad) < 2e-5
            o_r.step(), o_m.step()
        sd = mine.full_state_dict()
        assert sd["layers.0.weight"].shape == (N + F_h, hidden)
        for k, v in ref_i.state_dict().items():
            assert rel_err(sd[k].cpu(), v) < 1e-3, k
    # built from the graph object, as the reference builds its model from g.x.shape[1] (flat_amazon.py:80)
    g2d = copy.copy(g2)
    g2d.x = x.to(dev)
    auto = sharded.ShardedGCN.for_data(sg, g2d, classes, n_hidden_gcn=hidden, dropout=0.0).to(dev)
    auto.load_full_state_dict(ref.state_dict())
    plain = copy.copy(g)
    ar_d = torch.arange(N, device=dev)
    plain.x = torch.sparse_coo_tensor(torch.stack([ar_d, ar_d]), torch.ones(N, device=dev), (N, N))
    assert sharded.ShardedGCN.for_data(sg, plain, classes).weight_h is None
    ref.eval(), auto.eval()
    with torch.no_grad():
        assert rel_err(sg.gather_rows(auto()).cpu(), ref(g2)) < 1e-5
    for bad in (dict(hierarchy_feats=H_l[:-1]), dict(hierarchy_feats=H_l, narrow_exchange=True)):
        try:
            sharded.ShardedGCN(sg, N + F_h, classes if "narrow_exchange" not in bad else 8, n_hidden_gcn=hidden, **bad)
            raise AssertionError(f"accepted {list(bad)}")
        except ValueError:
            pass


def check_rows_option(sg, g, N, dev=None, dropout=0.0, hidden=16, classes=8, narrow=False, fuse_w1=False, steps=2):
    """ShardedGCN.forward(rows=mask) against the same model without the option, same exchange form, weights and keyed masks:
    the logits of the rows that are read, the loss and every gradient at 1e-5 (the restricted operators are plans of
    their own: same entries, possibly another order of summation); the unread rows hold the last bias; the collectives
    that must have gone are counted; a mask that reads a hub row falls back to the whole operators (bit for bit)."""
    import pytextgcn_amd as pkg
    from pytextgcn_amd.narrow import exchange_floats_per_hub_row
    dev = dev if dev is not None else torch.device("cpu")
    torch.manual_seed(22)
    init = O.GCNOracle(N, classes, n_hidden_gcn=hidden, dropout=0.0).state_dict()
    y_l, m_l = sg.scatter_rows(g.y.to(dev) % classes), sg.scatter_rows(g.train_mask.to(dev))
    assert not bool(m_l[:sg.hp].any())                       # the training rows are documents
    names = ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor")
    real = {k: getattr(dist, k) for k in names}

    def counting(calls):
        def wrap(name):
            def f(*a, **kw):
                calls[name] += 1
                return real[name](*a, **kw)
            return f
        for k in names:
            setattr(dist, k, wrap(k))
    models, opts = [], []
    for _ in range(2):
        m = sharded.ShardedGCN(sg, N, classes, n_hidden_gcn=hidden, dropout=dropout, narrow_exchange=narrow,
                               keyed_dropout=True).to(dev)
        m.load_full_state_dict(init)
        m._seed_base = 7654321
        if dev.type == "cpu":
            o = torch.optim.Adam(m.parameters(), lr=0.02, amsgrad=True)
        else:
            o = pkg.optim.Adam(m.parameters(), lr=0.02, amsgrad=True)
            if fuse_w1:
                o.fuse_into_backward(m.weights[0])
        models.append(m), opts.append(o)
    view = sg.prepare_rows(m_l)                              # (collective; what forward(rows=m_l) will look up)
    assert sg.rows_view(m_l) is view
    sharded.sharded_cross_entropy(sg, torch.zeros(sg.n_local, classes, device=dev), y_l, m_l)   # (the mask's row count: one all-reduce, once)
    assert view is not None and view.kept_entries["B"] < sg.dirs[0].B.export_csr()[1].numel()
    for step in range(steps):
        outs = []
        for m, o, rows in zip(models, opts, (None, m_l)):
            m.train()
            calls = {k: 0 for k in names}
            if sg.exchange == "collective":
                counting(calls)
            try:
                lo = m(rows=rows)
                loss = sharded.sharded_cross_entropy(sg, lo, y_l, m_l)
                o.zero_grad(set_to_none=True)
                loss.backward()
            finally:
                for k in names:
                    setattr(dist, k, real[k])
            if sg.exchange == "collective":
                K = sg.rs_chunks
                if narrow:     # forward AG(h) AR(C) [RS(C)] | backward [AG(C)] AR(C) RS(h)
                    want = {"all_reduce": 2, "all_gather_into_tensor": 2 if rows is None else 1,
                            "reduce_scatter_tensor": (2 if rows is None else 1) * K}
                else:          # forward AG RS | AG [RS]; backward [AG] RS | AG RS
                    want = {"all_reduce": 0, "all_gather_into_tensor": 4 if rows is None else 3,
                            "reduce_scatter_tensor": (4 if rows is None else 3) * K}
                assert calls == want, (narrow, rows is not None, calls, want)
            m.sync_grads()
            grads = [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]
            outs.append((loss.detach().clone(), lo.detach().clone(), grads))
            o.step()
        (l0, z0, g0), (l1, z1, g1) = outs
        assert abs(l0.item() - l1.item()) <= 1e-5 * abs(l0.item()) + 1e-9, (step, l0, l1)
        read = m_l.cpu()
        assert rel_err(z1.cpu()[read], z0.cpu()[read]) < 1e-5, (step, rel_err(z1.cpu()[read], z0.cpu()[read]))
        for a, b in zip(g0, g1):
            assert (a is None) == (b is None)
            if a is not None and a.numel() and float(a.abs().max()) > 0:
                assert rel_err(b.cpu(), a.cpu()) < 1e-5, (step, tuple(a.shape), rel_err(b.cpu(), a.cpu()))
    # evaluation: rows = everything evaluation reads (here: all documents); unread rows hold the bias
    for m in models:
        m.eval()
    docs = sg.real.clone()
    docs[:sg.hp] = False
    try:                                                     # a mask nobody prepared: refused locally, no collective entered
        models[1](rows=docs)
        raise AssertionError("an unprepared rows mask was accepted")
    except RuntimeError as e:
        assert "prepare_rows" in str(e)
    sg.prepare_rows(docs)
    with torch.no_grad():
        full, part = models[1](), models[1](rows=docs)
    assert rel_err(part.cpu()[docs.cpu()], full.cpu()[docs.cpu()]) < 1e-5
    unread = ~docs.cpu()
    assert torch.equal(part.cpu()[unread], models[1].biases[-1].detach().cpu().expand(int(unread.sum()), -1))
    # a mask that reads a hub row on ONE rank: every rank falls back to the whole operators
    hubby = docs.clone()
    if dist.get_rank() == dist.get_world_size() - 1:
        hubby[0] = bool(sg.real[0])
    assert sg.prepare_rows(hubby) is None
    with torch.no_grad():
        assert torch.equal(models[1](rows=hubby), full) and sg.rows_view(hubby) is None
    docs_edit = docs.clone()
    sg.prepare_rows(docs_edit)
    docs_edit[sg.hp] = ~docs_edit[sg.hp]                     # edited in place since: the lookup refuses it
    try:
        sg.rows_view(docs_edit)
        raise AssertionError("an edited rows mask was accepted")
    except RuntimeError:
        pass
    try:
        sg.rows_view(docs[:-1])
        raise AssertionError("a mask of the wrong length was accepted")
    except ValueError:
        pass
    h, C = 200, 64
    assert exchange_floats_per_hub_row(h, C, False, rows=True) == 928 and exchange_floats_per_hub_row(h, C, True, rows=True) == 656
    assert exchange_floats_per_hub_row(h, C, False, training=False, rows=True, layer1_cached=True) == C


def check_narrow_exchange(sg, g, N, dev=None, dropout=0.0, hidden=16, classes=8, steps=3, fuse_w1=False):
    """ShardedGCN(narrow_exchange=True) against the plain exchange on the same partition, weights and (with dropout) the
    SAME keyed mask: losses, logits and every gradient over `steps` Adam(amsgrad) steps at 1e-5 (the sums associate
    differently over ranks: not bits); the narrow form's collectives are counted."""
    import pytextgcn_amd as pkg
    from pytextgcn_amd.narrow import exchange_floats_per_hub_row
    dev = dev if dev is not None else torch.device("cpu")
    torch.manual_seed(21)
    init = O.GCNOracle(N, classes, n_hidden_gcn=hidden, dropout=0.0).state_dict()
    y_l, m_l = sg.scatter_rows(g.y.to(dev) % classes), sg.scatter_rows(g.train_mask.to(dev))
    calls = {"all_reduce": 0, "all_gather_into_tensor": 0, "reduce_scatter_tensor": 0}
    real = {k: getattr(dist, k) for k in calls}

    def counted(name):
        def f(*a, **kw):
            calls[name] += 1
            return real[name](*a, **kw)
        return f
    models, opts, traces = [], [], []
    for narrow in (False, True):
        m = sharded.ShardedGCN(sg, N, classes, n_hidden_gcn=hidden, dropout=dropout, narrow_exchange=narrow,
                               keyed_dropout=True).to(dev)
        m.load_full_state_dict(init)
        m._seed_base = 1234567                                # the same masks in both models (no broadcast needed)
        if dev.type == "cpu":
            o = torch.optim.Adam(m.parameters(), lr=0.02, amsgrad=True)
        else:
            o = pkg.optim.Adam(m.parameters(), lr=0.02, amsgrad=True)
            if fuse_w1:
                o.fuse_into_backward(m.weights[0])
        models.append(m), opts.append(o)
    for step in range(steps):
        outs = []
        for m, o, narrow in zip(models, opts, (False, True)):
            m.train()
            if narrow and step == 0 and sg.exchange == "collective":
                for k in calls:
                    setattr(dist, k, counted(k))
            try:
                lo = m()
                loss = sharded.sharded_cross_entropy(sg, lo, y_l, m_l)
                o.zero_grad(set_to_none=True)
                loss.backward()
            finally:
                for k in calls:
                    setattr(dist, k, real[k])
            if narrow and step == 0 and sg.exchange == "collective":
                # forward AG(h) AR(C) RS(C) | backward AG(C) AR(C) RS(h): two all-gathers, two all-reduces, two reduce-scatters
                assert calls == {"all_reduce": 2, "all_gather_into_tensor": 2, "reduce_scatter_tensor": 2 * sg.rs_chunks}, calls
            m.sync_grads()         # (a rank's share of dW2 differs between the two forms; the sums over ranks agree)
            grads = [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]
            outs.append((loss.detach().clone(), lo.detach().clone(), grads))
            o.step()
        (l0, z0, g0), (l1, z1, g1) = outs
        assert abs(l0.item() - l1.item()) <= 1e-5 * abs(l0.item()) + 1e-9, (step, l0, l1)
        assert rel_err(z1.cpu(), z0.cpu()) < 1e-5, (step, rel_err(z1.cpu(), z0.cpu()))
        for a, b in zip(g0, g1):
            assert (a is None) == (b is None)
            if a is not None and a.numel() and float(a.abs().max()) > 0:
                assert rel_err(b.cpu(), a.cpu()) < 1e-5, (step, tuple(a.shape), rel_err(b.cpu(), a.cpu()))
    # The parameters after `steps` Adam steps: Adam divides a gradient by its own magnitude, so an element whose gradient
    # is zero to rounding can move by +-lr in one form and -+lr in the other (the two forms add a hub row's partial sums in
    # different orders).  What must hold: all but a vanishing share of the elements agree to 1e-5 of the largest, and no
    # element differs by more than the optimizer could have moved it.
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        pa, pb = pa.detach().double().cpu(), pb.detach().double().cpu()
        diff, scale = (pb - pa).abs(), float(pa.abs().max())
        off = float((diff > 1e-5 * scale).double().mean())
        assert off < 2e-3 and float(diff.max()) <= 2.0 * 0.02 * steps * 1.001, (tuple(pa.shape), off, float(diff.max()))
    # eval forward (no dropout): the narrow form against the plain one
    for m in models:
        m.eval()
    with torch.no_grad():
        assert rel_err(models[1]().cpu(), models[0]().cpu()) < (1e-5 if dropout == 0.0 else 1e-4)   # (on diverged weights)
    assert exchange_floats_per_hub_row(200, 64, True) == 784 and exchange_floats_per_hub_row(200, 64, False) == 1056


def check_offline_construction(sg, g, hubs, N, engine=None):
    """`ShardedGraph.for_rank` -- what the at-size GPU tests and tools/sim_shard_compute.py inspect -- builds, WITHOUT the
    peers, exactly what the collective constructor built on this rank: the same partition, the same local operators, the
    same referenced columns, and gather-side halo lists equal to the ones the ranks swapped."""
    off = sharded.ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, sg.world, sg.rank, hubs=hubs,
                                        engine=engine if engine is not None else OracleEngine(), halo_lists=True,
                                        degree_sum=sg.degree_sum)
    assert (off.hp, off.rp, off.symmetric, len(off.dirs)) == (sg.hp, sg.rp, sg.symmetric, len(sg.dirs))
    assert torch.equal(off.owned, sg.owned)
    for d_off, d_on in zip(off.dirs, sg.dirs):
        assert torch.equal(d_off.need_cols, d_on.need_cols)
        assert d_off.need_counts_l == d_on.need_counts_l and d_off.send_counts_l == d_on.send_counts_l
        assert torch.equal(d_off.send_slots.cpu(), d_on.send_slots.cpu())
        for a, b in ((d_off.A, d_on.A), (d_off.B, d_on.B)):
            assert (a is None) == (b is None)
            if a is not None and hasattr(a, "csr"):
                assert all(torch.equal(u, v) for u, v in zip(a.csr, b.csr))
            elif a is not None:
                assert all(torch.equal(u, v) for u, v in zip(a.export_csr(), b.export_csr()))


def check_exchange_forms(sg, x_full, bias):
    """Every form of the exchange gives the same distributed SpMM: the pairwise and the halo form (index lists,
    pruned reduce-scatter), with A_r in one or several row chunks, are bit-for-bit alike (both add the ranks' partial
    rows in rank order); RCCL's / gloo's reduce-scatter may add in another order, and a row of a chunk operator may be
    summed in another order than the same row of the whole A_r (the kernels' work partition depends on the operator): 1e-6."""
    x_l = sg.scatter_rows(x_full.to(sg.device))
    bias = bias.to(sg.device)
    keep = (sg.exchange, sg.rs_chunks)
    try:
        for transpose in (False, True):
            whole = None
            for K in (1, 3):
                sg.set_rs_chunks(K)
                sg.exchange = "p2p"
                base = sg.spmm(x_l, bias, transpose=transpose).clone()
                whole = base if whole is None else whole
                assert rel_err(base.cpu(), whole.cpu()) < 1e-6, (K, transpose)
                for form in sg.EXCHANGES:
                    sg.exchange = form
                    got = sg.spmm(x_l, bias, transpose=transpose)
                    if form == "collective":
                        assert rel_err(got.cpu(), base.cpu()) < 1e-6, (form, K, transpose)
                    else:
                        assert torch.equal(got, base), (form, K, transpose, rel_err(got.cpu(), base.cpu()))
        # graphs without hub structure: the pipelined exchange (own-column block at once, K stage blocks accumulated as
        # their rows land) against the halo form -- the same entries, the blocks' partial sums added in another order
        if sg.rp == 0:
            kept = (sg.pipe_stages, sg.pipe_scheme, sg.pipe_prefix)
            for transpose in (False, True):
                sg.exchange = "halo"
                base = sg.spmm(x_l, bias, transpose=transpose).clone()
                auto_a = None
                for K, scheme, prefix in ((1, "slices", 0), (2, "slices", 0), (5, "slices", 0), (0, "peer", 0),
                                          (2, "slices", 37), (3, "slices", "auto"), (2, "slices", sg.hp), (0, "peer", sg.hp)):
                    sg.set_pipeline(K, scheme, prefix=prefix)       # ("auto": one all-reduce -- every rank is here)
                    if prefix == "auto":
                        auto_a = sg.pipe_prefix
                        assert 0 <= auto_a <= sg.hp and auto_a % 256 == 0
                    sg.exchange = "pipeline"
                    got = sg.spmm(x_l, bias, transpose=transpose)
                    assert rel_err(got.cpu(), base.cpu()) < 1e-6, (K, scheme, transpose, rel_err(got.cpu(), base.cpu()))
                    assert torch.equal(got, sg.spmm(x_l, bias, transpose=transpose))      # run to run: the same bits
                    d = sg.dirs[1 if (transpose and not sg.symmetric) else 0]
                    pipe = sg._pipeline(d)
                    unpacked = [st for st in pipe.stages if st.span is not None]
                    packed = [st for st in pipe.stages if st.span is None]
                    want_a = 0 if scheme == "peer" else (min(prefix, sg.hp) if prefix != "auto" else auto_a)
                    assert pipe.prefix == want_a and (scheme == "peer" or sg.pipe_prefix == want_a)
                    assert len(packed) == (sg.world - 1 if scheme == "peer" else (1 if want_a else max(1, K)))
                    assert len(unpacked) == (max(1, K) if want_a else 0)
                    # every entry of B_r sits in exactly one block; every halo row arrives exactly once -- in an unpacked
                    # range of the prefix (where unread rows travel too) or in one packed stage
                    assert pipe.own_nnz + sum(st.nnz for st in pipe.stages) == d.B.export_csr()[1].numel()
                    halo = sum(d.need_counts_l) - d.need_counts_l[sg.rank]
                    assert sum(st.rows_read for st in unpacked) + sum(sum(st.recv_counts) for st in packed) == halo
                    assert pipe.rows_received() == (sg.world - 1) * pipe.prefix + sum(sum(st.recv_counts) for st in packed)
                    assert [st.span for st in unpacked] == [(want_a * k // max(1, K), want_a * (k + 1) // max(1, K))
                                                            for k in range(len(unpacked))]
                    if want_a == sg.hp:                              # everything travels unpacked: the packed stage is empty
                        assert all(sum(st.recv_counts) == 0 and st.op is None for st in packed)
                    if scheme == "peer" and prefix == 0:
                        for k, st in enumerate(pipe.stages):       # stage k: everything from rank - k - 1, nothing else
                            src = (sg.rank - k - 1) % sg.world
                            assert all(n == 0 for q, n in enumerate(st.recv_counts) if q != src)
                            assert st.recv_counts[src] == d.need_counts_l[src]
            sg.drop_unused_pipelines()
            assert all(len(d.pipes) <= 1 for d in sg.dirs)
            sg.set_pipeline(*kept[:2], prefix=kept[2])
            try:
                sg.exchange = "nonsense"
                raise AssertionError("an unknown exchange form was accepted")
            except ValueError:
                pass
        else:
            try:
                sg.exchange = "pipeline"
                raise AssertionError("the pipelined exchange was accepted on a hub partition")
            except ValueError:
                pass
        # the halo lists prune: never more rows than the whole block, and the reduce side only rows A_r touches
        rows = sg.exchange_rows()
        assert rows["gather_halo"] <= rows["gather_all"]
        if sg.num_nodes == 900:                # the sparse power-law graph: the halo is a fraction of the operand
            assert rows["gather_halo"] < 0.8 * rows["gather_all"], rows
        if "reduce_halo" in rows:
            assert rows["reduce_halo"] <= rows["reduce_all"]
    finally:
        sg.exchange = keep[0]
        sg.set_rs_chunks(keep[1])
    # the chunk operators of the counts that are not in use can be released (bench.py does after its trial steps);
    # another count is cut again from the kept entries when asked for
    if sg.dirs[0].A is not None:
        sg.drop_unused_chunks()
        assert all(set(d.chunks) <= {1, sg.rs_chunks} for d in sg.dirs)
        before = sg.spmm(x_l, bias).clone()
        sg.set_rs_chunks(3)
        sg.exchange = "p2p"
        again = sg.spmm(x_l, bias)
        assert rel_err(again.cpu(), before.cpu()) < 1e-6
        sg.exchange = keep[0]
        sg.set_rs_chunks(keep[1])


def check_hip(kind, g, hubs, N, dev):
    """The same checks with the product engine (libtgcn.so) on a GPU; all ranks may share one card
    (gloo moves the collectives' payload through the host)."""
    import pytextgcn_amd as pkg
    F = 200
    gd = pkg.Data(**{k: getattr(g, k) for k in g.keys}).to(dev)
    hubs_d = None if hubs is None else hubs.to(dev)
    # the package default: the reference-order normalisation through the partition -- PyG's sequential fp32 degree sums
    # and its association (not symmetric: M^T gets operators of its own); the weights are the oracle's bits, so what is
    # left against the oracle is the summation order of the SpMM alone
    sg = sharded.ShardedGraph(gd.edge_index, gd.edge_attr, N, hubs=hubs_d)
    assert sg.degree_sum == "reference" and len(sg.dirs) == 2 and not sg.symmetric
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, F, generator=gen)
    b = torch.randn(F, generator=gen)
    nei, nw = O.gcn_norm(g.edge_index, g.edge_attr, N)
    for transpose in (False, True):
        ref = O.propagate(nei.flip(0) if transpose else nei, x, nw, N) + b
        got = sg.gather_rows(sg.spmm(sg.scatter_rows(x.to(dev)), b.to(dev), transpose=transpose))
        assert rel_err(got.cpu(), ref) < 2e-6, (kind, transpose, rel_err(got.cpu(), ref))
    check_exchange_forms(sg, x, b)
    check_offline_construction(sg, gd, hubs_d, N, engine=sharded.HipEngine())
    # the opt-in accurate mode (float64 degree sums, symmetric association): one pair of operators for M and M^T of a
    # symmetric graph, within the oracle's own accuracy of it
    sga = sharded.ShardedGraph(gd.edge_index, gd.edge_attr, N, hubs=hubs_d, degree_sum="accurate")
    assert sga.symmetric == (kind != "asym") and len(sga.dirs) == (1 if sga.symmetric else 2)
    for transpose in (False, True):
        ref = O.propagate(nei.flip(0) if transpose else nei, x, nw, N) + b
        got = sga.gather_rows(sga.spmm(sga.scatter_rows(x.to(dev)), b.to(dev), transpose=transpose))
        assert rel_err(got.cpu(), ref) < 1e-5, (kind, transpose, rel_err(got.cpu(), ref))
    check_offline_construction(sga, gd, hubs_d, N, engine=sharded.HipEngine())
    del sga
    if sg.rp > 0 and dist.get_world_size() <= 3:
        # (groups of two and three ranks: the four-rank runs keep to the exchange itself -- the suite's wall time; world 4
        # and 8 run every one of these checks on the CPU engine, tests/test_sharded.py)
        # the narrow exchange against the plain one on the SAME keyed dropout mask (p = 0.5), plain and with W1's Adam
        # update inside the backward SpMM
        check_narrow_exchange(sg, gd, N, dev, dropout=0.5, hidden=F, classes=8)
        check_narrow_exchange(sg, gd, N, dev, dropout=0.5, hidden=F, classes=8, fuse_w1=True)
        # forward(rows=...): the last propagate step on the rows that are read, both exchanges
        check_rows_option(sg, gd, N, dev, dropout=0.5, hidden=F, classes=8, narrow=False)
        check_rows_option(sg, gd, N, dev, dropout=0.5, hidden=F, classes=8, narrow=True, fuse_w1=True)
        # the features [I | H] of the hierarchical scripts on the partition
        check_hierarchy_block(sg, g, N, dev, hidden=F)
        check_flat_loop(sg, g, N, dev, hidden=F, narrow=False)
        check_flat_loop(sg, g, N, dev, hidden=F, narrow=True)
    torch.manual_seed(3)
    ref = O.GCNOracle(N, 5, n_hidden_gcn=F, dropout=0.0)
    mine = sharded.ShardedGCN(sg, N, 5, n_hidden_gcn=F, dropout=0.0).to(dev)
    mine.load_full_state_dict(ref.state_dict())
    y_l, m_l = sg.scatter_rows(gd.y), sg.scatter_rows(gd.train_mask)
    lo_r = ref(g)
    loss_r = torch.nn.CrossEntropyLoss()(lo_r[g.train_mask], g.y[g.train_mask])
    loss_r.backward()
    lo_m = mine()
    loss_m = sharded.sharded_cross_entropy(sg, lo_m, y_l, m_l)
    loss_m.backward()
    mine.sync_grads()
    total = loss_m.detach().clone().reshape(1)
    dist.all_reduce(total)
    assert abs(total.item() - loss_r.item()) < 1e-5 * abs(loss_r.item())
    assert rel_err(sg.gather_rows(lo_m.detach()).cpu(), lo_r.detach()) < 1e-5
    assert rel_err(sg.gather_rows(mine.weights[0].grad).cpu(), ref.layers[0].weight.grad) < 5e-5
    assert rel_err(mine.weights[1].grad.cpu(), ref.layers[1].weight.grad) < 5e-5
    for i in (0, 1):
        assert rel_err(mine.biases[i].grad.cpu(), ref.layers[i].bias.grad) < 5e-5, i
    check_fused_w1(sg, gd, N, dev, F)


def check_fused_w1(sg, gd, N, dev, F=200):
    """optim.Adam.fuse_into_backward on the rank's W1 shard: three training steps with the update of the regular rows
    inside the backward SpMM (split operand) and the hub slice after its reduce-scatter, against the same three steps
    with backward + step.  The two roads sum a row's products in the same kernels' orders up to the cut of B_r at row hp
    (each part picks its own work partition), so: a tight tolerance, not bits.  Also: the first layer's weight gets no
    .grad, a second backward before step() raises, every exchange form serves."""
    import pytextgcn_amd as pkg
    torch.manual_seed(11)
    init = O.GCNOracle(N, 5, n_hidden_gcn=F, dropout=0.0).state_dict()
    y_l, m_l = sg.scatter_rows(gd.y), sg.scatter_rows(gd.train_mask)

    def run(fused, steps=3):
        model = sharded.ShardedGCN(sg, N, 5, n_hidden_gcn=F, dropout=0.0).to(dev)
        model.load_full_state_dict(init)
        opt = pkg.optim.Adam(model.parameters(), lr=0.02, amsgrad=True)
        if fused:
            opt.fuse_into_backward(model.weights[0])
        losses = []
        for _ in range(steps):
            loss = sharded.sharded_cross_entropy(sg, model(), y_l, m_l)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if fused:
                assert model.weights[0].grad is None
            model.sync_grads()
            opt.step()
            losses.append(loss.detach().clone())
        return model, opt, torch.stack(losses)
    plain, _, l_plain = run(False)
    fused, opt_f, l_fused = run(True)
    assert sg.dirs[0 if sg.symmetric else 1].B_reg is not None or sg.rp == 0
    assert torch.allclose(l_fused, l_plain, rtol=1e-5, atol=1e-8), (l_fused, l_plain)
    for a, b in zip(fused.parameters(), plain.parameters()):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 2e-5, rel_err(a.detach().cpu(), b.detach().cpu())
    st_f = opt_f.state[fused.weights[0]]
    assert st_f["step"] == 3 and float(st_f["exp_avg"].abs().max()) > 0
    # a second backward before step() would apply the update twice: refused
    loss = sharded.sharded_cross_entropy(sg, fused(), y_l, m_l)
    loss.backward()
    loss2 = sharded.sharded_cross_entropy(sg, fused(), y_l, m_l)
    try:
        loss2.backward()
        raise AssertionError("second backward before step() was accepted")
    except RuntimeError as e:
        assert "second backward" in str(e)
    fused.sync_grads()
    opt_f.step()


def main(rank, world, port, kinds, errfile, backend="gloo", device="cpu"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    try:
        if device != "cpu":
            if backend == "nccl" and world > 1 and "{rank}" not in device:
                sharded.let_rccl_ranks_share_a_device(rank)   # RCCL ranks on ONE card: socket transport
            device = device.format(rank=rank)              # "cuda:{rank}": one GPU per rank
            torch.cuda.set_device(torch.device(device))
        # a bounded timeout: a mismatched collective must fail the test, not hang the box
        # (the product's own group constructor: RCCL on a high-priority stream, every collective bounded)
        sharded.init_process_group(backend, torch.device(device) if backend == "nccl" else None, timeout_s=240,
                                   rank=rank, world_size=world)
        for kind in kinds:
            check(kind, device)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        with open(errfile + f".{rank}", "w") as f:
            f.write(traceback.format_exc())
        raise
