"""BASELINE-size checks that need the oracle's normalisation of the whole graph (tests/_bigcase.py, shared per session):

  * the reference-order normalisation mode (`degree_sum="reference"`) on config c4: the plan's weights are the fp32
    oracle's BIT FOR BIT and the whole two-layer eval forward meets BASELINE.json's 1e-5 against the fp32 oracle;
  * the 1-D partition at size (c4 over 8 ranks with the word nodes as hubs; c5 over 8 ranks without hub structure):
    the local operators of the first and the last rank, cut exactly as the collective constructor cuts them
    (`ShardedGraph.for_rank`), against the oracle's operator entry by entry and row by row, the nnz balance, the index
    lists of the halo exchange against a recomputation from the edge list, and the accurate mode's weights against the
    single-device plan's bit for bit.

The reference is single-device (flat_amazon.py:84-86): the partition has nothing there to mirror, so the oracle of the
partitioned form is the oracle of the whole operator, restricted to the rows and columns a rank owns."""
import pytest
import torch

from oracle import csr_oracle
import pytextgcn_amd as pkg
from pytextgcn_amd.plan import GraphPlan
from pytextgcn_amd.sharded import ShardedGraph

from test_gpu_parity import TOL, _report, rel_err, row_rel_err

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------
# the reference-order mode at c4
# ------------------------------------------------------------------------------------------------
def test_config_c4_reference_order_mode_reproduces_the_oracles_weights_and_meets_1e5(cuda, c4case):
    """`degree_sum="reference"`: PyG's CPU arithmetic (one fp32 accumulator per node, weights in edge order, the loop
    last; (dis[src] * w) * dis[dst]) -- what textgcn/lib/models.py:11-20 executes.  All 52 M weights of the c4 plan
    equal the oracle's bit for bit, M^T is stored (PyG's association is not symmetric), and the eval forward of
    GCN(N -> 200 -> 64) over all 2 M rows is within 1e-5 of the fp32 oracle's network (the accurate mode is 1.3e-5 from
    it because the oracle's hub degrees are, HISTORY.md 2.2)."""
    N, F, C = c4case.N, 200, 64
    g = c4case.g
    plan = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum="reference")
    rp_ref, col_ref, val_ref, _ = c4case.oracle_csr()
    rp, col, val = plan.export_csr()
    assert torch.equal(rp.cpu().long(), rp_ref) and torch.equal(col.cpu(), col_ref)
    v = val.cpu()
    n_diff = int((v.view(torch.int32) != val_ref.view(torch.int32)).sum())
    worst = float(((v.double() - val_ref.double()).abs() / val_ref.double()).max())
    _report("c4_reference_mode_weights", entries=int(v.numel()), entries_differing_in_any_bit=n_diff,
            worst_relative_difference=worst)
    assert n_diff == 0, (n_diff, worst)
    assert not plan.symmetric and plan.has_transpose           # (dis[s] * w) * dis[t] rounds (i, j) and (j, i) apart
    del rp, col, val, v
    # the transposed block holds the same weights, transposed: <M x, y> = <x, M^T y>
    gen = torch.Generator(device=cuda).manual_seed(12)
    x = torch.randn(N, 8, device=cuda, generator=gen)
    y = torch.randn(N, 8, device=cuda, generator=gen)
    lhs = (plan.spmm(x).double() * y.double()).sum().item()
    rhs = (x.double() * plan.spmm(y, transpose=True).double()).sum().item()
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), abs(rhs), 1.0) + 1e-6
    del x, y
    # the whole eval forward against the fp32 ORACLE's network, all rows, at the 1e-5 bar
    prev = pkg.set_degree_sum("reference")
    try:
        torch.manual_seed(7)
        model = pkg.GCN(N, C, n_hidden_gcn=F, dropout=0.5).to(cuda).float().eval()
        with torch.no_grad():
            model.layers[0].bias.normal_(0, 0.1)
            model.layers[1].bias.normal_(0, 0.1)
            ar = torch.arange(N, device=cuda)
            eye = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N, device=cuda), (N, N)).coalesce()
            logits = model(pkg.Data(x=eye, edge_index=g.edge_index, edge_attr=g.edge_attr)).cpu()
            w1, b1, w2, b2 = (t.detach().cpu() for t in (model.layers[0].weight, model.layers[0].bias,
                                                         model.layers[1].weight, model.layers[1].bias))
    finally:
        pkg.set_degree_sum(prev)
    h1 = csr_oracle.csr_spmm(rp_ref, col_ref, val_ref, w1, b1, acc64=True)
    xw2 = (h1.double() @ w2.double()).float()
    want = csr_oracle.csr_spmm(rp_ref, col_ref, val_ref, xw2, b2, acc64=True)
    e, e_row = rel_err(logits, want), row_rel_err(logits, want)
    _report("c4_eval_forward_reference_mode", plan_vs_fp32_oracle=e, plan_vs_fp32_oracle_row_relative=e_row)
    assert e < TOL and e_row < TOL, (e, e_row)


def _training_step_parity(cuda, g, N, F, C, n_vocab, oracle_coo, case):
    """One training step (flat_amazon.py:99-105: forward, CrossEntropyLoss(mean) over the training rows, backward) of GCN(N ->
    F -> C) in the package-default mode against the ORACLE's operator end to end: the C CSR oracle (float64 accumulation)
    on the oracle's normalisation, float64 GEMMs and loss on the host.  Loss, logits on the training rows and ALL FOUR
    gradients at 1e-5; dW1 additionally row by row on the 64 heaviest rows.  Both forms of the step: the plain forward, and
    `gcn(g, rows=train_mask)`; the backward propagate step skips the gradient rows the loss leaves zero in both."""
    from pytextgcn_amd.functional import masked_cross_entropy
    gen = torch.Generator().manual_seed(21)
    y_cpu = torch.randint(0, C, (N,), generator=gen)
    mask_cpu = (torch.rand(N, generator=gen) < 0.8) & (torch.arange(N) >= n_vocab)        # documents only (text2graph.py:180-188)
    y, mask = y_cpu.to(cuda), mask_cpu.to(cuda)
    torch.manual_seed(9)
    model = pkg.GCN(N, C, n_hidden_gcn=F, dropout=0.0).to(cuda).float()
    with torch.no_grad():
        model.layers[0].bias.normal_(0, 0.1)
        model.layers[1].bias.normal_(0, 0.1)
    ar = torch.arange(N, device=cuda)
    eye = torch.sparse_coo_tensor(torch.stack([ar, ar]), torch.ones(N, device=cuda), (N, N)).coalesce()
    data = pkg.Data(x=eye, edge_index=g.edge_index, edge_attr=g.edge_attr)
    w1, b1, w2, b2 = (t.detach().cpu() for t in (model.layers[0].weight, model.layers[0].bias,
                                                 model.layers[1].weight, model.layers[1].bias))
    # ---- the oracle's step (host): M and M^T as CSR of the oracle's own weights
    tgt, src, nw = oracle_coo
    csr = []
    for a_, b_ in ((tgt, src), (src, tgt)):
        order = torch.argsort(a_ * N + b_, stable=True)
        rp_ = torch.zeros(N + 1, dtype=torch.int64)
        rp_[1:] = torch.bincount(a_, minlength=N).cumsum(0)
        csr.append((rp_, b_[order].to(torch.int32), nw[order].contiguous()))
        del order
    (rp, col, val), (rpt, colt, valt) = csr
    h1 = csr_oracle.csr_spmm(rp, col, val, w1, b1, acc64=True)
    xw2 = (h1.double() @ w2.double()).float()
    logits_ref = csr_oracle.csr_spmm(rp, col, val, xw2, b2, acc64=True)
    lg = logits_ref[mask_cpu].double().requires_grad_()
    loss_ref = torch.nn.functional.cross_entropy(lg, y_cpu[mask_cpu])
    loss_ref.backward()
    dlogits = torch.zeros(N, C, dtype=torch.float32)
    dlogits[mask_cpu] = lg.grad.float()
    db2_ref = dlogits.double().sum(0)
    dxw2 = csr_oracle.csr_spmm(rpt, colt, valt, dlogits, acc64=True)
    dw2_ref = h1.double().t() @ dxw2.double()
    dh1 = (dxw2.double() @ w2.double().t()).float()
    db1_ref = dh1.double().sum(0)
    dw1_ref = csr_oracle.csr_spmm(rpt, colt, valt, dh1, acc64=True)
    del dh1, dxw2, xw2
    heavy = (rpt[1:] - rpt[:-1]).topk(64).indices
    for use_rows in (False, True):
        model.zero_grad(set_to_none=True)
        model.train()
        out = model(data, rows=mask) if use_rows else model(data)
        loss = masked_cross_entropy(out, y, mask)
        loss.backward()
        torch.cuda.synchronize()
        tag = "rows" if use_rows else "plain"
        e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
        e_out = rel_err(out.detach()[mask], logits_ref[mask_cpu])
        grads = {"dW1": (model.layers[0].weight.grad, dw1_ref), "db1": (model.layers[0].bias.grad, db1_ref),
                 "dW2": (model.layers[1].weight.grad, dw2_ref), "db2": (model.layers[1].bias.grad, db2_ref)}
        errs = {k: rel_err(a_, b_) for k, (a_, b_) in grads.items()}
        e_rows = row_rel_err(model.layers[0].weight.grad[heavy.to(cuda)], dw1_ref[heavy])
        _report(f"{case}_training_step_{tag}", loss_rel=e_loss, logits_on_training_rows=e_out,
                dW1_heaviest_rows_row_relative=e_rows, **errs)
        assert e_loss < TOL and e_out < TOL, (tag, e_loss, e_out)
        assert max(errs.values()) < TOL and e_rows < TOL, (tag, errs, e_rows)


def test_config_c4_training_step_gradients_against_the_oracle_at_full_size(cuda, c4case):
    """BASELINE.json's headline configuration (2 M nodes, 50 M edges), GCN(N -> 200 -> 64): dW1 is 2 M x 200."""
    _training_step_parity(cuda, c4case.g, c4case.N, 200, 64, c4case.N // 10, c4case.oracle_coo(), "c4")


def test_config_c3_training_step_gradients_against_the_oracle_at_full_size(cuda):
    """The DBpedia-shaped configuration (1 M nodes, 30 k words, 24 M edges) with its 219 classes (flat_dbpedia.py:80): the
    class width is no multiple of 4 (zero-padded buffers), the layer-2 products run as column groups."""
    from oracle import gcn_oracle as O
    from pytextgcn_amd import synth
    N, E = 1_000_000, 24_000_000
    g = synth.word_doc_graph(N, E, seed=44, device=cuda, vocab_frac=0.03, doc_word_share=0.9, features="none")
    coo = O.normalized_coo(g.edge_index.cpu(), g.edge_attr.cpu(), N)
    _training_step_parity(cuda, g, N, 200, 219, g.n_vocab, coo, "c3")


# ------------------------------------------------------------------------------------------------
# the 1-D partition at size
# ------------------------------------------------------------------------------------------------
def _expected_local_csr(sg, tgt, src, val, which, transpose=False):
    """Rank sg.rank's local operator written down directly from the WHOLE operator's entries (target, source, value
    -- device tensors, PyG order): A_r = hub rows (gathered numbering) x own regular columns, B_r = own rows x (all hubs
    + own regular columns), module docstring of pytextgcn_amd/sharded.py.  CSR with a row's entries by column, ties in
    the order given."""
    p, r, hp, rp, W = sg.part, sg.rank, sg.hp, sg.rp, sg.world
    if transpose:
        tgt, src = src, tgt
    t_hub, s_hub = p.hub_mask[tgt], p.hub_mask[src]
    t_mine, s_mine = p.owner[tgt] == r, p.owner[src] == r
    if which == "A":
        sel = t_hub & ~s_hub & s_mine
        t, s = tgt[sel], src[sel]
        row, col, n_rows, n_cols = p.hub_col[t], p.slot[s], W * hp, rp
    else:
        sel = t_mine & (s_hub | (s_mine & ~t_hub))
        t, s = tgt[sel], src[sel]
        row = torch.where(p.hub_mask[t], p.slot[t], hp + p.slot[t])
        col = torch.where(p.hub_mask[s], p.hub_col[s], p.reg_col[s])
        n_rows, n_cols = hp + rp, W * hp + rp
    order = torch.argsort(row * n_cols + col, stable=True)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    rowptr[1:] = torch.bincount(row, minlength=n_rows).cumsum(0)
    return rowptr, col[order], val[sel][order]


def _assert_op_equals(op, expect, bitwise=True):
    rp, col, val = op.export_csr()
    e_rp, e_col, e_val = expect
    assert torch.equal(rp.long(), e_rp)
    assert torch.equal(col.long(), e_col)
    if bitwise:
        n_diff = int((val.view(torch.int32) != e_val.view(torch.int32)).sum())
        assert n_diff == 0, (n_diff, float((val - e_val).abs().max()))
    else:
        assert float((val.double() - e_val.double()).abs().max()) <= 2e-6 * float(e_val.abs().max())


def _rows_of_csr(rp, col, val, rows, keep_col=None):
    """Sub-CSR (host) of the listed rows of a host CSR, optionally keeping only the entries whose column passes
    `keep_col` (a bool tensor over the columns)."""
    lo, n = rp[rows], rp[rows + 1] - rp[rows]
    start = torch.zeros(rows.numel() + 1, dtype=torch.int64)
    start[1:] = n.cumsum(0)
    pos = torch.arange(int(start[-1])) - torch.repeat_interleave(start[:-1], n) + torch.repeat_interleave(lo, n)
    c, v = col[pos], val[pos]
    if keep_col is None:
        return start, c.contiguous(), v.contiguous()
    keep = keep_col[c.long()]
    rid = torch.repeat_interleave(torch.arange(rows.numel()), n)[keep]
    out_rp = torch.zeros(rows.numel() + 1, dtype=torch.int64)
    out_rp[1:] = torch.bincount(rid, minlength=rows.numel()).cumsum(0)
    return out_rp, c[keep].contiguous(), v[keep].contiguous()


def test_config_c4_eight_rank_partition_first_and_last_rank_against_the_oracle(cuda, c4case):
    """c4 over 8 ranks, word nodes as hubs (BASELINE.json configs[3]): ranks 0 and 7 in the reference-order mode --
    A_r and B_r equal the oracle's entries bit for bit; A_r @ X_reg and B_r @ [X_hub ; X_reg] against the C CSR oracle
    on the ORACLE's rows (document rows: the whole row; hub rows: the columns the operator holds) at 1e-5, also per
    row; every node owned exactly once; non-zeros balanced to 2 %.  Rank 0 once more in the default mode: its weights
    are the single-device plan's bit for bit."""
    N, F, W = c4case.N, 200, 8
    g = c4case.g
    V = g.n_vocab
    hubs = torch.arange(N, device=cuda) < V
    tgt, src, nw = (t.to(cuda) for t in c4case.oracle_coo())
    rp_ref, col_ref, val_ref, _ = c4case.oracle_csr()
    gen = torch.Generator(device=cuda).manual_seed(21)
    x = torch.randn(N, F, device=cuda, generator=gen)
    bias = torch.randn(F, device=cuda, generator=gen)
    x_cpu = x.cpu()
    nnz = {}
    for r in (0, W - 1):
        sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, W, r, hubs=hubs, degree_sum="reference")
        p, hp, rp_ = sg.part, sg.hp, sg.rp
        assert not sg.symmetric and len(sg.dirs) == 2           # PyG's association: M^T gets its own operators
        if r == 0:
            own = torch.cat([p.owned(q)[p.owned(q) >= 0] for q in range(W)])
            assert own.numel() == N and bool((torch.sort(own).values == torch.arange(N, device=cuda)).all())
        for k, d in enumerate(sg.dirs):
            _assert_op_equals(d.A, _expected_local_csr(sg, tgt, src, nw, "A", transpose=bool(k)))
            _assert_op_equals(d.B, _expected_local_csr(sg, tgt, src, nw, "B", transpose=bool(k)))
        A, B = sg.ops[0]
        nnz[r] = A.nnz + B.nnz
        # operands in the rank's layout: gathered hub block [W * hp] (padding rows zero), own regular rows [rp]
        node_of_hub_row = torch.full((W * hp,), -1, dtype=torch.int64, device=cuda)
        hub_ids = torch.nonzero(p.hub_mask).flatten()
        node_of_hub_row[p.hub_col[hub_ids]] = hub_ids
        xh = torch.zeros(W * hp, F, device=cuda)
        xh[node_of_hub_row >= 0] = x[node_of_hub_row[node_of_hub_row >= 0]]
        own_reg = sg.owned[hp:]
        xr = torch.zeros(rp_, F, device=cuda)
        xr[own_reg >= 0] = x[own_reg[own_reg >= 0]]
        yB = B.spmm(xh, bias, x2=xr).cpu()
        yA = A.spmm(xr).cpu()
        # (1) document rows of B_r: the oracle's WHOLE row (a document's entries are words and its own loop)
        docs = own_reg[own_reg >= 0].cpu()
        want = csr_oracle.csr_spmm(*_rows_of_csr(rp_ref, col_ref, val_ref, docs), x_cpu, bias.cpu(), acc64=True)
        got = yB[hp:][(own_reg >= 0).cpu()]
        assert rel_err(got, want) < TOL and row_rel_err(got, want) < TOL, (r, rel_err(got, want), row_rel_err(got, want))
        # (2) hub rows of B_r: the oracle's row restricted to hub (word) columns
        own_hub = sg.owned[:hp]
        words = own_hub[own_hub >= 0].cpu()
        is_word = torch.arange(N) < V
        want = csr_oracle.csr_spmm(*_rows_of_csr(rp_ref, col_ref, val_ref, words, keep_col=is_word), x_cpu, bias.cpu(),
                                   acc64=True)
        got = yB[:hp][(own_hub >= 0).cpu()]
        assert rel_err(got, want) < TOL and row_rel_err(got, want) < TOL, (r, rel_err(got, want), row_rel_err(got, want))
        # (3) A_r: every hub row restricted to the documents this rank owns
        mine = (~is_word) & (p.owner.cpu() == r)
        all_words = node_of_hub_row[node_of_hub_row >= 0].cpu()
        want = csr_oracle.csr_spmm(*_rows_of_csr(rp_ref, col_ref, val_ref, all_words, keep_col=mine), x_cpu, None,
                                   acc64=True)
        got = yA[(node_of_hub_row >= 0).cpu()]
        assert rel_err(got, want) < TOL and row_rel_err(got, want) < TOL, (r, rel_err(got, want), row_rel_err(got, want))
        if r == W - 1:
            # the operators of `ShardedGCN.forward(rows=...)` at this size (ShardedGraph.rows_view cuts them from the local
            # operators' own CSR): kept rows of B_r as B_r gives them, every other row the bias; A'_r and B'_r over the kept
            # columns as the whole operators give them on a gradient whose other rows are zero
            from pytextgcn_amd.sharded import _RowsView
            keepm = (torch.rand(rp_, device=cuda, generator=gen) < 0.8) & (own_reg >= 0)
            view = _RowsView(sg, keepm)
            assert view.kept_entries["B"] < B.nnz and view.kept_entries["At"] < sg.dirs[1].A.nnz
            yv, yw = view.B_rows.spmm(xh, bias, x2=xr), B.spmm(xh, bias, x2=xr)
            assert rel_err(yv[hp:][keepm], yw[hp:][keepm]) < 2e-6
            assert torch.equal(yv[:hp], bias.expand(hp, -1)) and torch.equal(yv[hp:][~keepm], bias.expand(int((~keepm).sum()), -1))
            dT = sg.dirs[1]
            gz = xr * keepm.unsqueeze(1)
            assert rel_err(view.At.spmm(xr), dT.A.spmm(gz)) < 2e-6
            whole_t = dT.B.spmm(torch.zeros(W * hp, F, device=cuda), None, x2=gz)
            assert float(whole_t[:hp].abs().max()) == 0.0          # B'_r's hub rows hold hub columns only
            assert rel_err(view.Bt_reg.spmm(xr), whole_t[hp:]) < 2e-6
            _report("c4_partition_rows_view_rank7", kept_regular_rows=int(keepm.sum()), **view.kept_entries,
                    B_entries=B.nnz, At_entries=dT.A.nnz)
            del view, yv, yw, gz, whole_t
        for d in sg.dirs:
            d.A.close(), d.B.close()
        del sg, xh, xr, yA, yB
    assert abs(nnz[0] - nnz[W - 1]) <= 0.02 * max(nnz.values()), nnz
    _report("c4_partition_8_ranks", nnz_rank0=nnz[0], nnz_rank7=nnz[W - 1])
    del tgt, src, nw
    # accurate mode: the weights of the partition are the single-device plan's, bit for bit (one degree routine)
    plan = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum="accurate")
    prp, pcol, pval = plan.export_csr()
    ptgt = torch.repeat_interleave(torch.arange(N, device=cuda), (prp[1:] - prp[:-1]).long())
    sg = ShardedGraph.for_rank(g.edge_index, g.edge_attr, N, W, 0, hubs=hubs, degree_sum="accurate")
    assert sg.symmetric and len(sg.dirs) == 1
    _assert_op_equals(sg.dirs[0].A, _expected_local_csr(sg, ptgt, pcol.long(), pval, "A"))
    _assert_op_equals(sg.dirs[0].B, _expected_local_csr(sg, ptgt, pcol.long(), pval, "B"))
    # the cut of B_r at local row hp that the fused W1 update uses (ShardedGraph.spmm_adam_w1): the two parts hold B_r's
    # entries, in B_r's order, and their products are B_r's rows at this size (each part picks its own work partition)
    d = sg.dirs[0]
    sg._split_B(d)
    hp, rp_ = sg.hp, sg.rp
    brp, bcol, bval = d.B.export_csr()
    hrp, hcol, hval = d.B_hub.export_csr()
    rrp, rcol, rval = d.B_reg.export_csr()
    cut = int(brp[hp].item())
    assert torch.equal(hrp, brp[:hp + 1]) and torch.equal(rrp, brp[hp:] - cut)
    assert torch.equal(hcol, bcol[:cut]) and torch.equal(rcol, bcol[cut:])
    assert torch.equal(hval, bval[:cut]) and torch.equal(rval, bval[cut:])
    gen = torch.Generator(device=cuda).manual_seed(77)
    xb = torch.randn(W * hp, 200, device=cuda, generator=gen)
    xo = torch.randn(rp_, 200, device=cuda, generator=gen)
    whole = d.B.spmm(xb, None, x2=xo)
    assert rel_err(d.B_hub.spmm(xb, None, x2=xo), whole[:hp]) < 1e-6
    assert rel_err(d.B_reg.spmm(xb, None, x2=xo), whole[hp:]) < 1e-6
    # ... and tgcn_spmm_adam_split on the regular rows IS tgcn_spmm_split followed by tgcn_adam_step, bit for bit
    from pytextgcn_amd import _lib
    from pytextgcn_amd.plan import _stream_ptr
    lib = _lib.load()
    p0 = torch.randn(rp_, 200, device=cuda, generator=gen) * 0.01
    st0 = [torch.rand(rp_, 200, device=cuda, generator=gen) * 1e-3 for _ in range(3)]       # exp_avg, exp_avg_sq, max
    hyper = (0.05, 0.9, 0.999, 1e-8, 0.0, 3)
    pf, stf = p0.clone(), [t.clone() for t in st0]
    d.B_reg.spmm_adam(xb, pf, stf[0], stf[1], stf[2], *hyper, transpose=False, g2=xo)
    pu, stu = p0.clone(), [t.clone() for t in st0]
    grad = d.B_reg.spmm(xb, None, x2=xo)
    _lib.check(lib.tgcn_adam_step(pu.data_ptr(), grad.data_ptr(), stu[0].data_ptr(), stu[1].data_ptr(), stu[2].data_ptr(),
                                  pu.numel(), *hyper, _stream_ptr(pu.device)))
    assert torch.equal(pf, pu) and all(torch.equal(a, b) for a, b in zip(stf, stu))


def test_config_c5_eight_rank_partition_without_hubs_and_its_halo_lists(cuda, c5case):
    """c5 (8 M nodes, 200 M edges, power law, h = 256; BASELINE.json configs[4]) over 8 ranks with `hubs=None`: every
    node is a hub, A_r is empty, B_r = own rows x all nodes, and the exchange is the halo form.  Rank 7 in the
    reference-order mode: both directions' operators equal the oracle's entries bit for bit; all its rows of M @ X at
    the class width against the C CSR oracle on the oracle's rows; sampled rows at h = 256.  Its halo lists against a
    recomputation from the edge list: the referenced columns (counts per owner, none twice, every column the operator
    holds) and the slots it is asked for.  Rank 0 in the accurate mode with the symmetry fingerprint at size, weights
    bit for bit the single-device plan's."""
    N, W = c5case.N, 8
    g = c5case.g
    ei = g.edge_index
    tgt, src, nw = c5case.oracle_coo()                      # host: 208 M entries
    r = W - 1
    sg = ShardedGraph.for_rank(ei, g.edge_attr, N, W, r, hubs=None, degree_sum="reference", halo_lists=True)
    p, hp = sg.part, sg.hp
    assert sg.rp == 0 and hp == N // W and all(d.A is None for d in sg.dirs) and len(sg.dirs) == 2
    own_cpu = p.owner.cpu()
    mine = own_cpu[tgt] == r
    t_d, s_d, w_d = tgt[mine].to(cuda), src[mine].to(cuda), nw[mine].to(cuda)
    _assert_op_equals(sg.dirs[0].B, _expected_local_csr(sg, t_d, s_d, w_d, "B"))
    mine_t = own_cpu[src] == r
    _assert_op_equals(sg.dirs[1].B, _expected_local_csr(sg, tgt[mine_t].to(cuda), src[mine_t].to(cuda), nw[mine_t].to(cuda),
                                                        "B", transpose=True))
    # halo lists, recomputed from the edge list on the host (tools/sim_halo_rows.py's logic)
    d = sg.dirs[0]
    hub_col = p.hub_col.cpu()
    ei_cpu = ei.cpu()
    e_s, e_t = ei_cpu[0], ei_cpu[1]
    ref_need = torch.unique(torch.cat([hub_col[e_s[own_cpu[e_t] == r]], hub_col[torch.nonzero(own_cpu == r).flatten()]]))
    need = d.need_cols.cpu()
    assert torch.equal(need, ref_need)                                       # sorted, none twice
    assert d.need_counts_l == torch.bincount(ref_need // hp, minlength=W).tolist()
    _, bcol, _ = d.B.export_csr()
    assert bool(torch.isin(torch.unique(bcol.long()).cpu(), need).all())     # every column the operator holds arrives
    foreign = int(need.numel()) - d.need_counts_l[r]
    _report("c5_halo_rows_rank7", referenced_rows_of_other_ranks=foreign, of=(W - 1) * hp)
    assert 0.3 < foreign / ((W - 1) * hp) < 0.7                              # about half of the operand travels
    # the slots this rank is asked for: peers q whose rows reference its nodes (plus its own loops)
    slot_cpu = p.slot.cpu()
    to_me = own_cpu[e_s] == r
    asked = torch.unique(own_cpu[e_t[to_me]] * hp + slot_cpu[e_s[to_me]])
    asked = torch.unique(torch.cat([asked, r * hp + slot_cpu[own_cpu == r]]))
    assert torch.equal(d.send_slots.cpu(), asked % hp)
    assert d.send_counts_l == torch.bincount(asked // hp, minlength=W).tolist()
    del ei_cpu, e_s, e_t, to_me, asked
    # rows of M @ X: the class width over ALL rows of the rank against the oracle's rows, h = 256 on sampled rows
    rp_l, col_l, val_l = (t.cpu() for t in _expected_local_csr(sg, t_d, s_d, w_d, "B"))
    gen = torch.Generator(device=cuda).manual_seed(31)
    node_of_row = torch.full((W * hp,), -1, dtype=torch.int64, device=cuda)
    node_of_row[p.hub_col] = torch.arange(N, device=cuda)
    for F in (64, 256):
        x = torch.randn(N, F, device=cuda, generator=gen)
        xg = torch.zeros(W * hp, F, device=cuda)
        xg[node_of_row >= 0] = x[node_of_row[node_of_row >= 0]]
        y = d.B.spmm(xg)
        if F == 64:
            want = csr_oracle.csr_spmm(rp_l, col_l.to(torch.int32), val_l, xg.cpu(), acc64=True)
            assert rel_err(y, want) < TOL and row_rel_err(y, want) < TOL, (rel_err(y, want), row_rel_err(y, want))
        else:
            deg = rp_l[1:] - rp_l[:-1]
            rows = torch.cat([deg.topk(6).indices, torch.randint(0, hp, (120,), generator=torch.Generator().manual_seed(3))])
            for i in rows.tolist():
                s, e = int(rp_l[i]), int(rp_l[i + 1])
                want = (val_l[s:e].double().unsqueeze(1) * xg[col_l[s:e].to(cuda)].double().cpu()).sum(0)
                assert rel_err(y[i], want.float()) < TOL, i
        # the PIPELINED exchange's column blocks at this size (sharded._Pipeline; the default for hubs=None): the own-column
        # block on the rank's own rows + the stage blocks ADDED (tgcn_spmm_acc) on the rows as they would land -- every row
        # of the result against B_r as one operator (same entries; the blocks' partial sums associate differently), for the
        # staged all-to-all (4 stages) and for one peer per stage (7 stages)
        bias = torch.randn(F, device=cuda, generator=gen)
        y_one = d.B.spmm(xg, bias)
        x_own = xg[r * hp:(r + 1) * hp].contiguous()
        # ... and with the unpacked prefix ("auto": the degree-ordered slots of which >= 90 % of the (row, peer) pairs are read
        # travel as 4 contiguous ranges, one packed stage for the rest)
        for K, scheme, prefix in ((4, "slices", 0), (0, "peer", 0), (4, "slices", "auto")):
            sg.set_pipeline(K, scheme, prefix=prefix)
            pipe = sg._pipeline(d)
            tag = scheme + ("+prefix" if prefix else "")
            assert len(pipe.stages) == (W - 1 if scheme == "peer" else (5 if prefix else 4))
            assert pipe.own_nnz + sum(st.nnz for st in pipe.stages) == d.B.nnz
            assert sum(st.rows_read for st in pipe.stages) == foreign          # every halo row arrives exactly once
            if prefix:
                assert 0.2 * hp < pipe.prefix < 0.8 * hp and pipe.prefix % 256 == 0
                read = sum(st.rows_read for st in pipe.stages if st.span is not None)
                assert read >= 0.88 * (W - 1) * pipe.prefix                    # >= 90 % of the prefix pairs (here: rank 7's)
            y_p = pipe.own.spmm(x_own, bias)
            for st in pipe.stages:
                assert sum(st.recv_counts) == st.rows.numel()
                assert st.span is not None or bool(((st.rows // hp) != r).all())
                if st.op is not None:
                    st.op.spmm(xg[st.rows], out=y_p, accumulate=True)
            e1, e2 = rel_err(y_p, y_one), row_rel_err(y_p, y_one)
            _report(f"c5_pipeline_blocks_rank7_F{F}_{tag}", max_norm=e1, row_relative=e2, prefix_rows_per_rank=pipe.prefix,
                    stage_entries=[st.nnz for st in pipe.stages], own_entries=pipe.own_nnz)
            assert e1 < 2e-6 and e2 < TOL, (F, tag, e1, e2)
            sg.drop_unused_pipelines()
        del x, xg, y, y_one
    for dd in sg.dirs:
        dd.B.close()
    del sg, t_d, s_d, w_d, tgt, src, nw
    # accurate mode, rank 0: symmetry decided by the fingerprint at size; weights = the single-device plan's bits
    sg = ShardedGraph.for_rank(ei, g.edge_attr, N, W, 0, hubs=None, degree_sum="accurate")
    assert sg.symmetric and len(sg.dirs) == 1
    plan = GraphPlan(ei, g.edge_attr, N, degree_sum="accurate")
    prp, pcol, pval = plan.export_csr()
    lo_hi = torch.nonzero(sg.part.owner == 0).flatten()
    ptgt = torch.repeat_interleave(torch.arange(N, device=cuda), (prp[1:] - prp[:-1]).long())
    keep = sg.part.owner[ptgt] == 0
    _assert_op_equals(sg.dirs[0].B, _expected_local_csr(sg, ptgt[keep], pcol[keep].long(), pval[keep], "B"))
    assert int(keep.sum()) == sg.dirs[0].B.nnz and lo_hi.numel() == N // W
