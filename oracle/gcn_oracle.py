"""CPU oracle for the PyTextGCN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module.  Nothing under `pytextgcn_amd/` imports it; the product path is the HIP library
behind `include/tgcn.h` and fails loudly when that library is missing.

PARITY UNPINNED (by the reference): the arithmetic of this path lives in a third-party
dependency that is absent from /root/reference -- `torch-geometric==1.6.3`
(requirements.yml:87; `torch_geometric.nn.GCNConv`, `gcn_norm`, `MessagePassing.propagate`)
over `torch-scatter==2.0.5` (requirements.yml:88) and `pytorch=1.7.0` (requirements.yml:44).
The reference's only test of the path (textgcn/test/test_model.py:10-41) has no assertions, so
there is no reference-held golden vector to pin against, and `import textgcn` fails here with an
ordinary ModuleNotFoundError (torch_geometric, nltk).  What this file restates, op for op, is the
published PyG-1.6.3 algorithm as the reference calls it:

  call sites      textgcn/lib/models.py:11,13,15 (construction: add_self_loops=True, defaults
                  normalize=True, cached=False, bias=True) and models.py:20 (invocation
                  `layer(x, g.edge_index, g.edge_attr)`)
  composition     textgcn/lib/models.py:17-25 (dropout between layers, NO activation: the call
                  at models.py:22 is commented out)
  training step   flat_amazon.py:82,89,99-106 (CrossEntropyLoss(mean) on train_mask rows,
                  Adam(lr, amsgrad=True), zero_grad(set_to_none=True))

It is cross-checked inside the test-suite against an independent float64 dense formulation
`D^-1/2 (A + I) D^-1/2` (Kipf & Welling) and against the known-answer vector of SURVEY.md
section 8(a); golden fixtures under tests/golden/ are generated from it by
tests/golden/make_golden.py.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn


# --------------------------------------------------------------------------------------
# PyG 1.6.3 torch_geometric/utils/loop.py: add_remaining_self_loops
# --------------------------------------------------------------------------------------
def add_remaining_self_loops(edge_index: Tensor, edge_weight: Optional[Tensor], fill_value: float,
                             num_nodes: int) -> Tuple[Tensor, Optional[Tensor]]:
    """Existing self-loops are removed from the edge list and re-appended (keeping their weight;
    when a node carries several, the last one in edge order wins -- index assignment on CPU);
    every other node gets a loop of weight `fill_value`.  The N loop entries sit at the tail."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_index = torch.arange(0, num_nodes, dtype=row.dtype, device=row.device)
    loop_index = loop_index.unsqueeze(0).repeat(2, 1)
    new_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    new_weight = None
    if edge_weight is not None:
        inv_mask = ~mask
        loop_weight = torch.full((num_nodes,), fill_value, dtype=edge_weight.dtype,
                                 device=edge_weight.device)
        remaining = edge_weight[inv_mask]
        if remaining.numel() > 0:
            # sequential assignment so that "last one wins" is deterministic
            idx = row[inv_mask]
            for i in range(idx.numel()):
                loop_weight[idx[i]] = remaining[i]
        new_weight = torch.cat([edge_weight[mask], loop_weight], dim=0)
    return new_index, new_weight


# --------------------------------------------------------------------------------------
# PyG 1.6.3 torch_geometric/nn/conv/gcn_conv.py: gcn_norm (dense edge_index branch)
# --------------------------------------------------------------------------------------
def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
             add_self_loops: bool = True, dtype=torch.float32, improved: bool = False) -> Tuple[Tensor, Tensor]:
    """deg is the weighted IN-degree at the target (`edge_index[1]`), loops included;
    `w_hat = deg^-1/2[row] * w * deg^-1/2[col]` with inf -> 0."""
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    if add_self_loops:
        fill_value = 2.0 if improved else 1.0                     # gcn_conv.py: `fill_value = 2. if improved else 1.`
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype, device=edge_weight.device)
    deg.index_add_(0, col, edge_weight)                      # scatter_add(edge_weight, col)
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
    return edge_index, deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]


def propagate(edge_index: Tensor, x: Tensor, edge_weight: Tensor, num_nodes: int) -> Tensor:
    """flow=source_to_target, aggr=add: out[col_e] += w_e * x[row_e].  Materialises the
    nnz x F message tensor exactly as the reference formulation does (k6-k8 of SURVEY 2a)."""
    row, col = edge_index[0], edge_index[1]
    x_j = x.index_select(0, row)                             # k6
    msg = edge_weight.view(-1, 1) * x_j                      # k7
    out = torch.zeros(num_nodes, x.size(1), dtype=x.dtype, device=x.device)
    out.index_add_(0, col, msg)                              # k8 (scatter_add)
    return out


def gcn_conv(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], weight: Tensor,
             bias: Optional[Tensor], add_self_loops: bool = True, normalize: bool = True) -> Tensor:
    """PyG 1.6.3 GCNConv.forward: norm (recomputed each call, cached=False) -> x @ W ->
    propagate -> + bias.  `x` may be a sparse COO tensor (text2graph.py:179,246)."""
    n = x.size(0)
    if normalize:
        edge_index, edge_weight = gcn_norm(edge_index, edge_weight, n, add_self_loops, weight.dtype)
    elif edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=weight.dtype)
    xw = torch.sparse.mm(x, weight) if x.is_sparse else torch.matmul(x, weight)   # k5
    out = propagate(edge_index, xw, edge_weight, n)
    if bias is not None:
        out = out + bias                                     # k9
    return out


def glorot_(t: Tensor, generator: Optional[torch.Generator] = None) -> Tensor:
    """PyG inits.glorot: U(-a, a), a = sqrt(6 / (fan_in + fan_out)) over the last two dims."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a, generator=generator)
    return t


class GCNConvOracle(nn.Module):
    """Parameter layout of PyG 1.6.3: `weight` (in, out), `bias` (out,)."""

    def __init__(self, in_channels: int, out_channels: int, add_self_loops: bool = True):
        super().__init__()
        self.add_self_loops = add_self_loops
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        glorot_(self.weight)

    def forward(self, x, edge_index, edge_weight=None):
        return gcn_conv(x, edge_index, edge_weight, self.weight, self.bias, self.add_self_loops)


class GCNOracle(nn.Module):
    """textgcn/lib/models.py:6-25, op for op (no activation: models.py:22 is a comment)."""

    def __init__(self, in_channels, out_channels, n_gcn=2, n_hidden_gcn=64, activation=nn.ReLU,
                 dropout=0.5):
        super().__init__()
        self.activation = activation()        # constructed, never applied (models.py:9,22)
        self.dropout = dropout
        self.layers = nn.ModuleList([GCNConvOracle(in_channels, n_hidden_gcn)])
        for _ in range(n_gcn - 2):
            self.layers.append(GCNConvOracle(n_hidden_gcn, n_hidden_gcn))
        self.layers.append(GCNConvOracle(n_hidden_gcn, out_channels))

    def forward(self, g):
        x = g.x
        for i, layer in enumerate(self.layers):
            x = layer(x, g.edge_index, g.edge_attr)
            if i < len(self.layers) - 1:
                x = nn.functional.dropout(x, p=self.dropout, training=self.training)
        return x


def train_step(model: nn.Module, g, optimizer: torch.optim.Optimizer) -> Tuple[Tensor, Tensor]:
    """flat_amazon.py:100-106 then :107-109: one optimisation step followed by the eval
    forward.  Returns (train loss, eval logits)."""
    criterion = nn.CrossEntropyLoss(reduction="mean")
    model.train()
    outputs = model(g)[g.train_mask]
    loss = criterion(outputs, g.y[g.train_mask])
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    model.eval()
    with torch.no_grad():
        logits = model(g)
    return loss.detach(), logits


# --------------------------------------------------------------------------------------
# Independent cross-checks (float64, dense) used by the tests to validate the restatement
# --------------------------------------------------------------------------------------
def dense_norm_adj(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                   add_self_loops: bool = True) -> Tensor:
    """M = D^-1/2 (A + I') D^-1/2 in float64, M[target, source]; I' puts 1.0 only where the
    diagonal of A is empty (the add_remaining rule).  Duplicate off-diagonal edges add up."""
    n = num_nodes
    row, col = edge_index[0].long(), edge_index[1].long()
    w = (torch.ones(row.numel(), dtype=torch.float64) if edge_weight is None
         else edge_weight.double())
    a = torch.zeros(n, n, dtype=torch.float64)
    off = row != col
    a.index_put_((col[off], row[off]), w[off], accumulate=True)
    if add_self_loops:
        diag = torch.ones(n, dtype=torch.float64)
        for i in torch.nonzero(~off).flatten().tolist():       # last one wins
            diag[row[i]] = w[i]
        a = a + torch.diag(diag)
    else:
        a.index_put_((col[~off], row[~off]), w[~off], accumulate=True)
    deg = a.sum(dim=1)                                       # in-degree at target
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    return dis.view(-1, 1) * a * dis.view(1, -1)


def normalized_coo(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                   add_self_loops: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """(target, source, w_hat) of the normalised operator, float32, in PyG's edge order."""
    ei, w = gcn_norm(edge_index, edge_weight, num_nodes, add_self_loops)
    return ei[1], ei[0], w
