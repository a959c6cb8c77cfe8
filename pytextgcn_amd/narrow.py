"""The narrow exchange of the sharded two-layer GCN: hub rows cross xGMI at the CLASS width wherever the algebra allows.

No reference counterpart (the reference is single-device, flat_amazon.py:84-86).  What makes it legal is the reference's
own network, textgcn/lib/models.py:17-25: NO activation between the layers, only dropout -- so around the (known) dropout
mask D the network is linear:

    XW2 = (H1 * D) @ W2,   H1[hub rows] = sum over ranks q of P_q,     P_q = A_q @ W1_reg(q)  (+ the owner's B_q share)
    =>  XW2[hub rows] = sum_q (P_q * D_hub) @ W2                        -- the sum can be taken AFTER the product

and the mask is a stateless hash of (seed, mask row, column) that any rank can evaluate for any hub row once the hub rows'
mask rows are their positions in the gathered hub block (`tgcn_set_dropout_row_keys`) and the seed is common to the group.

One training step of `ShardedGCN` exchanges (h = hidden width, C = classes; AG all-gather, RS reduce-scatter, AR
all-reduce = RS + AG, all over the [W * hp] hub block):

    plain      forward  L1: AG(h) RS(h)     L2: AG(C) RS(C)       backward  L2: AG(C) RS(C)     L1: AG(h) RS(h)     = 4 h + 4 C
    narrow     forward  L1: AG(h)  AR(C)    L2:       RS(C)       backward  L2: AG(C) AR(C)     L1:       RS(h)     = 2 h + 6 C

  forward:  every rank keeps its partial sums P_q of ALL hub rows (the owner folds its own B_q hub rows in), forms
            Q_q = dropout(P_q) @ W2 [W * hp, C] and the block is all-reduced: every rank then holds XW2 of every hub --
            which is exactly the operand block layer 2's B_r needs, so layer 2 gathers nothing;
  backward: layer 2's hub gradient rows are all-reduced instead of reduce-scattered (G_hub of every hub on every rank);
            every rank forms dH1_hub = (G_hub @ W2^T) * D_hub itself ([W * hp, C] x [C, h]: ~5 GFLOP at c4) instead of
            all-gathering it at width h, and the hub rows' share of dW2 = dropout(P_q)^T @ G_hub, which the flat
            all-reduce of the small gradients sums.

Two of the four width-h collectives become width-C ones: at c4 (h = 200, C = 64) 1056 -> 784 floats per hub row and step.
When the caller names the rows it reads and none is a hub row (`ShardedGCN.forward(rows=...)`: the words of a TextGCN
graph are never read, flat_amazon.py:101,109-114), layer 2 has no hub rows to produce and its gradient none to gather:

    plain+rows forward  L1: AG(h) RS(h)     L2: AG(C)             backward  L2:       RS(C)     L1: AG(h) RS(h)     = 4 h + 2 C
    narrow+rows forward L1: AG(h)  AR(C)    L2:                   backward  L2:       AR(C)     L1:       RS(h)     = 2 h + 4 C

The sums associate differently over ranks than in the plain form, so the results agree with it to fp32 rounding (tests
hold them to 1e-5), not bit for bit -- hence opt-in (`ShardedGCN(..., narrow_exchange=True)`).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor


def exchange_floats_per_hub_row(h: int, C: int, narrow: bool, training: bool = True, rows: bool = False,
                                layer1_cached: bool = False) -> int:
    """Floats per hub row that one step puts through collectives (each AG / RS moves (W - 1) / W of the block per rank;
    AR counts as RS + AG): the width-units of the module docstring.  `rows`: the caller names the rows it reads and none
    is a hub row (`ShardedGCN.forward(rows=...)`): layer 2 loses its forward RS(C) and its backward AG(C).
    `layer1_cached`: the forward pass finds layer 1's value from the preceding call (`enable_activation_reuse`)."""
    l1_fwd = 0 if layer1_cached else (h if narrow else 2 * h)           # narrow: AG(h); plain: AG(h) RS(h)
    l2_fwd = (2 * C if narrow else C) if rows else (3 * C if narrow else 2 * C)
    if not training:
        return l1_fwd + l2_fwd
    l2_bwd = (2 * C if narrow else C) if rows else (3 * C if narrow else 2 * C)
    l1_bwd = h if narrow else 2 * h
    return l1_fwd + l2_fwd + l2_bwd + l1_bwd


def _own(sg) -> slice:
    return slice(sg.rank * sg.hp, (sg.rank + 1) * sg.hp)


def _reg_keys(sg):
    """Mask rows of this rank's regular rows when they are handed to a product on their own: behind all hub rows."""
    return (0, 0, sg.world * sg.hp + sg.rank * sg.rp)


def own_row_keys(sg):
    """Mask rows of a rank's whole local matrix [own hubs ; own regular rows]: hub row i is mask row rank * hp + i (its
    position in the gathered hub block), regular row j mask row W * hp + rank * rp + j."""
    return (sg.hp, sg.rank * sg.hp, sg.world * sg.hp + sg.rank * sg.rp - sg.hp)


def layer1_partials(sg, w1: Tensor, b1: Tensor):
    """(P, Yr): P [W * hp, h] = this rank's partial sums of EVERY hub row of M @ W1 + b1 (A_r @ W1_reg, plus -- in the
    own block -- B_r's hub rows, bias included: the sum over ranks of P is H1[hubs]); Yr [rp, h] = H1 of the own regular
    rows, final.  One all-gather at width h (the hub rows of W1), nothing reduced."""
    d = sg.dirs[0]
    hp = sg.hp
    xbuf, gathered = sg._start_gather(d, w1)
    P = d.A.spmm(w1[hp:])                                       # overlaps the all-gather
    whole = sg._whole_operand(d, w1)
    gathered()
    Yb = sg._apply_B(d, w1, b1, xbuf, whole)
    P[_own(sg)] += Yb[:hp]
    return P, Yb[hp:]


class NarrowGCN2(torch.autograd.Function):
    """logits_local = the two-layer GCN of textgcn/lib/models.py:17-25 on one-hot features over the 1-D partition, with
    the exchange of the module docstring.  Inputs: the rank's W1 shard [n_local, h], b1, W2 [h, C], b2; `p` = dropout
    probability (0 in eval mode), `seed` = the group-common dropout seed (device int64[1]) or None; `cache` = None or a
    dict the layer-1 partials are kept in between calls on unchanged (W1, b1) (`enable_activation_reuse`); `view` = None
    or the operators of the rows that are read (`ShardedGraph.rows_view`: no hub row among them) -- layer 2 then has no
    hub rows to produce (its A_r and RS(C) go) and its gradient none to gather (the backward AG(C) goes): 2 h + 4 C."""

    @staticmethod
    def forward(ctx, sg, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, p: float, seed: Optional[Tensor], cache,
                view=None):
        eng = sg.engine
        hp, rp, W = sg.hp, sg.rp, sg.world
        d = sg.dirs[0]
        w1d, b1d, w2d, b2d = w1.detach().contiguous(), b1.detach(), w2.detach().contiguous(), b2.detach()
        C = w2d.size(1)
        want_grad = any(ctx.needs_input_grad[1:5])
        drop = p > 0.0 and seed is not None
        # ---- layer 1: partial sums, nothing reduced (or the cached ones of the last call on these weights)
        key = (w1.data_ptr(), w1._version, b1.data_ptr(), b1._version)
        if cache is not None and cache.get("key") == key:
            P, Yr = cache["P"], cache["Yr"]
        else:
            P, Yr = layer1_partials(sg, w1d, b1d)
            if cache is not None:
                cache.update(key=key, P=P, Yr=Yr)
        # ---- the dropout + W2 product on the partial sums, written straight into layer 2's operand buffer
        narrow_w = C <= sg._NARROW                        # the sub-group SpMM kernels take [hub block ; own rows] in one piece
        buf = sg.operand_buffer(C, P.device)              # the buffer layer 2's SpMM would gather into
        Z = buf[:W * hp]
        X2r = buf[W * hp:] if narrow_w else torch.empty(rp, C, dtype=torch.float32, device=P.device)
        mP = mR = None
        if drop:
            r1 = eng.gemm_nn(P, w2d, p, seed, record_mask=want_grad, out=Z)
            r2 = eng.gemm_nn(Yr, w2d, p, seed, record_mask=want_grad, keys=_reg_keys(sg), out=X2r)
            if want_grad:
                mP, mR = r1[1], r2[1]
        else:
            eng.gemm_nn(P, w2d, out=Z)
            eng.gemm_nn(Yr, w2d, out=X2r)
        work = dist.all_reduce(Z, group=sg.group, async_op=True)      # -> XW2 of every hub row, on every rank
        # ---- layer 2: A_r on the own regular rows while the block is reduced; B_r reads the block as it is
        pending = [] if view is not None else [sg._start_reduce(ch, ch.op.spmm(X2r)) for ch in d.chunks[sg.rs_chunks]]
        work.wait()
        B = d.B if view is None else view.B_rows
        out = B.spmm(buf, b2d) if narrow_w else B.spmm(Z, b2d, x2=X2r)
        for finish in pending:
            finish(out[:hp])
        if want_grad:
            ctx.sg, ctx.p, ctx.drop, ctx.view = sg, p, drop, view
            ctx.masks = (mP is not None, mR is not None)
            ctx.fused_param = w1 if isinstance(w1, torch.nn.Parameter) else None
            ctx.save_for_backward(P, Yr, w2d, *([seed] if drop else []), *[m for m in (mP, mR) if m is not None])
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        if ctx.needs_input_grad[1] and ctx.fused_param is not None:
            # a refusal ("second backward before step") must come BEFORE the first collective of this pass is started
            from .sharded import _fused_optimizer_for
            opt_ = _fused_optimizer_for(ctx.fused_param)
            if opt_ is not None:
                opt_.assert_no_pending_update(ctx.fused_param)
        sg = ctx.sg
        eng = sg.engine
        hp, rp, W = sg.hp, sg.rp, sg.world
        saved = list(ctx.saved_tensors)
        P, Yr, w2 = saved[:3]
        rest = saved[3:]
        seed = rest.pop(0) if ctx.drop else None
        mP = rest.pop(0) if ctx.masks[0] else None
        mR = rest.pop(0) if ctx.masks[1] else None
        p = ctx.p if ctx.drop else 0.0
        kreg = _reg_keys(sg) if ctx.drop else None
        g = grad_out.contiguous()
        d_b2 = sg.colsum_real(g) if ctx.needs_input_grad[4] else None
        dT = sg.dirs[1 if not sg.symmetric else 0]
        # ---- layer 2 backward: G = M^T dOut; the hub rows are ALL-REDUCED (every rank gets every hub's row)
        view = ctx.view
        if view is None:
            xbuf, gathered = sg._start_gather(dT, g)                   # AG at width C
            PT = dT.A.spmm(g[hp:])                                    # [W * hp, C] partial sums
            whole = sg._whole_operand(dT, g)
            gathered()
            Yb = sg._apply_B(dT, g, None, xbuf, whole)
            PT[_own(sg)] += Yb[:hp]
            G_reg = Yb[hp:]
        else:
            # no hub row was read: the hub rows of g are zero on every rank -- nothing to gather, and B'_r's hub rows (hub
            # columns only) contribute nothing
            PT = view.At.spmm(g[hp:])
            G_reg = view.Bt_reg.spmm(g[hp:])
        work = dist.all_reduce(PT, group=sg.group, async_op=True)
        # ---- the regular rows' share while the block is reduced; layer 1's partial sums and their reduce-scatter (the one
        # width-h collective left in the backward pass) start as early as their operand exists
        dH1_reg = eng.gemm_nt(G_reg, w2, p, seed, mask=mR, keys=kreg)          # [rp, h]; leaves its column sums
        pending = [sg._start_reduce(ch, ch.op.spmm(dH1_reg)) for ch in dT.chunks[sg.rs_chunks]]
        d_b1 = eng.colsum(dH1_reg) if ctx.needs_input_grad[2] else None
        work.wait()
        G_hub = PT                                                    # [W * hp, C]: dXW2 of every hub row
        d_w2 = None
        if ctx.needs_input_grad[3]:
            # dropout(H1)^T @ G over the rows this rank accounts for: its partial sums of all hub rows (summed over ranks
            # by the flat all-reduce of the small gradients: sum_q dropout(P_q)^T G_hub = dropout(H1_hub)^T G_hub) + its own
            # regular rows
            d_w2 = eng.gemm_tn(P, G_hub, p, seed, mask=mP) + eng.gemm_tn(Yr, G_reg, p, seed, mask=mR, keys=kreg)
        dH1_hub = eng.gemm_nt(G_hub, w2, p, seed, mask=mP)            # [W * hp, h], formed locally: no AG at width h
        if d_b1 is not None:
            d_b1 = d_b1 + eng.colsum(dH1_hub[_own(sg)])
        # ---- layer 1 backward: dW1_own = B'_r @ [dH1_hub ; dH1_reg] + [RS(A'_r @ dH1_reg) ; 0]
        d_w1 = None
        if ctx.needs_input_grad[1]:
            from .sharded import _fused_optimizer_for
            opt = _fused_optimizer_for(ctx.fused_param) if ctx.fused_param is not None else None
            if opt is None or not opt._fused_update_sharded(ctx.fused_param, sg, dH1_reg, hub_block=dH1_hub, pending=pending):
                d_w1 = dT.B.spmm(dH1_hub, None, x2=dH1_reg)
                for finish in pending:
                    finish(d_w1[:hp])
        else:
            for finish in pending:                                    # the collective was entered: complete it on every rank
                finish(torch.empty(hp, dH1_reg.size(1), dtype=dH1_reg.dtype, device=dH1_reg.device))
        return None, d_w1, d_b1, d_w2, d_b2, None, None, None, None
