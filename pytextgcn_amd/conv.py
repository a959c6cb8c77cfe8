"""GCNConv: drop-in for `torch_geometric.nn.GCNConv` (PyG 1.6.3) as PyTextGCN uses it.

Call site replaced: textgcn/lib/models.py:20 `x = layer(x, g.edge_index, g.edge_attr)`, layers
built at models.py:11,13,15 with `GCNConv(in, out, add_self_loops=True)`.  Same parameter names
and layout as PyG 1.6.3 (`weight` (in, out) glorot-uniform, `bias` (out,) zeros), same semantics
(norm -> x @ W -> propagate -> + bias), but the propagate step and its gradient are the HIP CSR
SpMM of libtgcn.so and the normalisation is computed once per graph and cached.
"""
from __future__ import annotations

import math
import weakref
from typing import Optional

import torch
from torch import Tensor, nn

from . import dense
from .plan import GraphPlan, colsum, known_nonzero_rows, padded_base, plan_for


def _fused_optimizer_for(param):
    from .optim import fused_optimizer_for      # late import: optim imports nothing from here, but keep it lazy
    return fused_optimizer_for(param)


def _pad_cols(t: Tensor, width: int) -> Tensor:
    if t.size(-1) == width:
        return t
    if t.dim() == 1:                            # a bias: `width` floats
        out = t.new_zeros(width)
        out[:t.numel()] = t
        return out
    base = padded_base(t, width)               # a producer of this package left it padded: no copy
    if base is not None:
        return base
    return torch.nn.functional.pad(t, (0, width - t.size(-1)))


class _Propagate(torch.autograd.Function):
    """out = M @ xw + bias;  d xw = M^T @ d out;  d bias = column sums of d out."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, xw: Tensor, bias: Optional[Tensor]):
        F = xw.size(1)
        F4 = (F + 3) & ~3                     # 16-byte rows keep the float4 kernel path
        out = plan.spmm(_pad_cols(xw.detach(), F4),
                        None if bias is None else _pad_cols(bias.detach(), F4))
        ctx.plan = plan
        ctx.F = F
        ctx.has_bias = bias is not None
        # the operand IS a parameter whose optimizer asked for the update to happen in this backward
        # (pytextgcn_amd.optim.Adam.fuse_into_backward): TextGCN's W1 under one-hot features
        ctx.fused_param = xw if (isinstance(xw, nn.Parameter) and _fused_optimizer_for(xw) is not None) else None
        return out if F4 == F else out[:, :F]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        plan, F = ctx.plan, ctx.F
        F4 = (F + 3) & ~3
        g = _pad_cols(grad_out, F4).contiguous()
        d_xw = d_bias = None
        if ctx.needs_input_grad[1]:
            p = getattr(ctx, "fused_param", None)
            opt = _fused_optimizer_for(p) if p is not None else None
            if opt is not None and F4 == F and opt._fused_update(p, plan, g):
                d_xw = None                       # spent on the optimizer row by row, never stored
            else:
                # a gradient whose producer knows which rows are exactly zero (the fused cross-entropy: every row outside
                # the loss mask) runs on M^T restricted to the other columns: the same sums without the zero terms
                keep = known_nonzero_rows(g)
                op = plan.transposed_on_rows(keep) if keep is not None else None
                d_xw = op.spmm(g) if op is not None else plan.spmm(g, None, transpose=True)
                if F4 != F:
                    d_xw = d_xw[:, :F]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            d_bias = colsum(g)[:F]
        return None, d_xw, d_bias


class _PropagateCached(torch.autograd.Function):
    """_Propagate whose forward value was computed earlier from the same (plan, xw, bias) versions."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, xw: Tensor, bias: Optional[Tensor], value: Tensor):
        ctx.plan = plan
        ctx.F = xw.size(1)
        ctx.has_bias = bias is not None
        ctx.fused_param = xw if (isinstance(xw, nn.Parameter) and _fused_optimizer_for(xw) is not None) else None
        return value.detach()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        d_xw, d_bias = _Propagate.backward(ctx, grad_out)[1:]
        return None, d_xw, d_bias, None


def propagate(plan: GraphPlan, xw: Tensor, bias: Optional[Tensor]) -> Tensor:
    if plan.n_rows != plan.num_nodes:
        raise ValueError("propagate() needs a whole-graph plan; use pytextgcn_amd.sharded for row blocks")
    return _Propagate.apply(plan, xw, bias)


# Opt-in activation reuse.  In the reference's loop (flat_amazon.py:99-109) the eval forward of epoch k
# and the training forward of epoch k+1 both evaluate layer 1 on the SAME weights: M @ W1 + b1 is
# computed twice (dropout comes after it).  With reuse enabled a layer whose input is the one-hot
# feature matrix keeps its last output keyed by the version counters of W1 and b1 and the plan, and
# hands it out again while nothing changed.  Off by default; results are bitwise identical either way.
_REUSE = False


def enable_activation_reuse(on: bool = True) -> None:
    global _REUSE
    _REUSE = bool(on)


def _reuse_key(plan: GraphPlan, weight: Tensor, bias: Optional[Tensor]):
    return (id(plan), weight.data_ptr(), weight._version, tuple(weight.shape),
            None if bias is None else (bias.data_ptr(), bias._version))


# sparse feature matrices that are exactly the identity (text2graph.py:179): X @ W is W itself
# keyed by id(x) (tensors compare element-wise, so they cannot key a dict); the weakref both
# validates the entry and removes it when the tensor dies
_IDENTITY_CACHE: dict = {}


def is_sparse_identity(x: Tensor) -> bool:
    if not x.is_sparse:
        return False
    hit = _IDENTITY_CACHE.get(id(x))
    if hit is not None and hit[0]() is x:
        return hit[1]
    ok = False
    if x.size(0) == x.size(1):
        xc = x if x.is_coalesced() else x.coalesce()
        idx, val = xc.indices(), xc.values()
        if val.numel() == x.size(0):
            ok = bool(((idx[0] == idx[1]).all() & (val == 1).all()).item())
    key = id(x)
    _IDENTITY_CACHE[key] = (weakref.ref(x, lambda _, k=key: _IDENTITY_CACHE.pop(k, None)), ok)
    return ok


# [I_N | H] features of the hierarchical scripts (text2graph.py:237-241; perlevel_amazon.py:122,156):
# X @ W = W[:N] + H @ W[N:], with H the sparse N x F_h hierarchy block.
_SPLIT_CACHE: dict = {}


def split_identity_block(x: Tensor):
    """For a sparse [N, N + F_h] matrix whose first N columns are the identity, return the sparse
    [N, F_h] remainder (None when x does not have that shape).  Cached per tensor object."""
    if not x.is_sparse or x.size(1) <= x.size(0):
        return None
    hit = _SPLIT_CACHE.get(id(x))
    if hit is not None and hit[0]() is x:
        return hit[1]
    n = x.size(0)
    xc = x if x.is_coalesced() else x.coalesce()
    idx, val = xc.indices(), xc.values()
    left = idx[1] < n
    ok = bool((int(left.sum()) == n) and bool(((idx[0][left] == idx[1][left]) & (val[left] == 1)).all()))
    rest = None
    if ok:
        r = ~left
        rest = torch.sparse_coo_tensor(torch.stack([idx[0][r], idx[1][r] - n]), val[r],
                                       (n, x.size(1) - n)).coalesce()
    key = id(x)
    _SPLIT_CACHE[key] = (weakref.ref(x, lambda _, k=key: _SPLIT_CACHE.pop(k, None)), rest)
    return rest


# Sparse feature blocks that are NOT the identity -- the hierarchy block H of [I | H], or a general sparse x --
# times a dense weight: the same CSR SpMM as the propagate step, on a rectangular operator built once per feature
# tensor (tgcn_plan_create_coo with its transpose, for the weight gradient H^T @ dXW).
_FEATURE_PLANS: dict = {}


def _feature_plan(x: Tensor) -> GraphPlan:
    hit = _FEATURE_PLANS.get(id(x))
    if hit is not None and hit[0]() is x:
        return hit[1]
    xc = x if x.is_coalesced() else x.coalesce()
    idx, val = xc.indices(), xc.values()
    plan = GraphPlan.from_coo(idx[0], idx[1], val, x.size(0), x.size(1), with_transpose=True)
    key = id(x)
    _FEATURE_PLANS[key] = (weakref.ref(x, lambda _, k=key: _FEATURE_PLANS.pop(k, None)), plan)
    return plan


class _SparseTimesDense(torch.autograd.Function):
    """x_sparse @ w and d w = x_sparse^T @ d out (the features carry no gradient)."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, w: Tensor):
        ctx.plan = plan
        F = w.size(1)
        F4 = (F + 3) & ~3
        ctx.F = F
        out = plan.spmm(_pad_cols(w.detach(), F4).contiguous())
        return out if F4 == F else out[:, :F]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        F4 = (ctx.F + 3) & ~3
        dw = ctx.plan.spmm(_pad_cols(grad_out, F4).contiguous(), transpose=True)
        return None, dw if F4 == ctx.F else dw[:, :ctx.F]


def sparse_times(x: Tensor, w: Tensor) -> Tensor:
    """`x @ w` for a sparse COO feature matrix on the HIP SpMM (no torch.sparse.mm / vendor library)."""
    if not x.is_cuda:
        raise RuntimeError(f"pytextgcn_amd: sparse features live on {x.device}; the GCN path runs only on an AMD GPU "
                           "through libtgcn.so (there is no CPU fallback)")
    if w.dtype != torch.float32:
        raise TypeError("sparse features times weight: float32 weights only (flat_amazon.py:85 casts the model)")
    return _SparseTimesDense.apply(_feature_plan(x), w)


def glorot_(t: Tensor) -> Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-a, a)


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True, normalize: bool = True,
                 bias: bool = True, degree_sum: Optional[str] = None, **kwargs):
        """`degree_sum` (not a PyG argument): "accurate" | "reference" -- how gcn_norm's degrees are summed
        (pytextgcn_amd.plan); None = the package default, `pytextgcn_amd.set_degree_sum`."""
        super().__init__()
        self.degree_sum = degree_sum
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached                  # accepted for signature parity; plans are always cached
        self.add_self_loops = add_self_loops
        self.normalize = normalize
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        glorot_(self.weight)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None,
                input_dropout: float = 0.0, rows: Optional[Tensor] = None) -> Tensor:
        """`input_dropout` > 0 (an extension used by pytextgcn_amd.models.GCN when fused dropout is
        enabled): the layer sees dropout(x, p) -- training-mode inverted dropout -- fused into x @ W.
        `rows` (an extension; a bool mask over the nodes, a tensor the caller KEEPS: the restricted operator is cached
        under it): only these rows of the result will be read -- the propagate step runs on the operator restricted to
        them (GraphPlan.on_rows); every other row comes back holding the bias."""
        plan = self.plan(x, edge_index, edge_weight)
        if rows is not None:
            plan = plan.on_rows(rows) or plan
        if input_dropout > 0.0:
            if x.is_sparse:
                raise ValueError("input_dropout applies to dense activations")
            return propagate(plan, dense.xw_dropout(x, self.weight, input_dropout), self.bias)
        if _REUSE and x.is_sparse and x.size(1) == self.in_channels and is_sparse_identity(x):
            # layer 1 of TextGCN: one-hot features, X @ W1 is W1; keep M @ W1 + b1 while W1, b1 are unchanged
            key = _reuse_key(plan, self.weight, self.bias)
            hit = getattr(self, "_reuse_cache", None)
            if hit is not None and hit[0] == key and hit[2] is plan:   # `is`: an id() can be recycled
                if torch.is_grad_enabled() and (self.weight.requires_grad or
                                                (self.bias is not None and self.bias.requires_grad)):
                    return _PropagateCached.apply(plan, self.weight, self.bias, hit[1])
                return hit[1].detach()
            out = propagate(plan, self.weight, self.bias)
            self._reuse_cache = (key, out.detach(), plan)
            return out
        return propagate(plan, self.features_times(x, self.weight), self.bias)

    def plan(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> GraphPlan:
        loops = (2 if self.improved else 1) if self.add_self_loops else 0     # fill weight of added loops
        return plan_for(edge_index, edge_weight, x.size(0), loops, self.normalize, getattr(self, "degree_sum", None))

    def features_times(self, x: Tensor, w: Tensor) -> Tensor:
        """X @ w for the feature formats of text2graph.py:226-246 (`w` has `in_channels` rows)."""
        if not x.is_sparse:
            return dense.xw(x, w)             # fp32 MFMA kernels for tall-skinny shapes
        if x.size(1) != self.in_channels:
            raise ValueError(f"x has {x.size(1)} features, the layer expects {self.in_channels}")
        if is_sparse_identity(x):
            return w
        h = split_identity_block(x)
        n = x.size(0)
        return w[:n] + sparse_times(h, w[n:]) if h is not None else sparse_times(x, w)

    def __getstate__(self):
        # th.save(gcn, ...) pickles the whole module (flat_amazon.py:128): cached activations and the
        # plan they point to (a ctypes handle) stay behind
        state = self.__dict__.copy()
        state.pop("_reuse_cache", None)
        return state

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
