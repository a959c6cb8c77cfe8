"""GCN: drop-in for `textgcn.lib.models.GCN` (textgcn/lib/models.py:6-25).

Same constructor signature and defaults, same `layers` ModuleList (state_dict keys
`layers.{i}.weight` (in, out) / `layers.{i}.bias`), same forward: dropout between layers, none
after the last, and NO activation -- the reference's activation call is commented out
(models.py:22) although `self.activation` is constructed (models.py:9); both facts are kept.
"""
from __future__ import annotations

import torch
from torch import nn

from . import dense
from .conv import GCNConv, propagate

# Opt-in: the reference network has no non-linearity (models.py:22 is commented out) and dropout is the
# identity in eval mode, so without autograd the L-layer forward
#     x_i = M (x_{i-1} W_i) + b_i
# equals   z_0 = X (W_1 W_2 ... W_L);  z_i = M z_{i-1} + b_i (W_{i+1} ... W_L)
# -- L propagations at the width of the OUTPUT (C classes) instead of one at every hidden width, and no
# N x h intermediate.  Same value up to fp32 rounding (different association), hence off by default
# and excluded from the bitwise tests; parity against the oracle is checked at the 1e-5 bar.
_COLLAPSE = False


def enable_linear_collapse(on: bool = True) -> None:
    global _COLLAPSE
    _COLLAPSE = bool(on)


# Opt-in: the dropout between the layers (models.py:23) fused into the next layer's x @ W and its
# autograd (libtgcn.so tgcn_gemm_*_dropout): neither the dropped activation nor its mask is stored, the
# three GEMMs regenerate the mask from an 8-byte seed.  Same distribution as torch's dropout, but the
# random stream is the library's own (seeded from torch's generator), hence off by default.
_FUSED_DROPOUT = False


def enable_fused_dropout(on: bool = True) -> None:
    global _FUSED_DROPOUT
    _FUSED_DROPOUT = bool(on)


class GCN(nn.Module):
    def __init__(self, in_channels, out_channels, n_gcn=2, n_hidden_gcn=64, activation=nn.ReLU,
                 dropout=0.5):
        super().__init__()
        self.activation = activation()
        self.dropout = dropout
        self.layers = nn.ModuleList([GCNConv(in_channels, n_hidden_gcn, add_self_loops=True)])
        for _ in range(n_gcn - 2):
            self.layers.append(GCNConv(n_hidden_gcn, n_hidden_gcn, add_self_loops=True))
        self.layers.append(GCNConv(n_hidden_gcn, out_channels, add_self_loops=True))

    def _collapsed_forward(self, g, rows=None):
        layers = list(self.layers)
        tail = [None] * len(layers)            # tail[i] = W_{i+1} ... W_L (None = identity)
        for i in range(len(layers) - 2, -1, -1):
            w_next = layers[i + 1].weight
            tail[i] = w_next if tail[i + 1] is None else dense.xw(w_next, tail[i + 1])
        z = layers[0].features_times(g.x, dense.xw(layers[0].weight, tail[0]))
        for i, layer in enumerate(layers):
            b = layer.bias
            if b is not None and tail[i] is not None:
                b = dense.xw(b.unsqueeze(0), tail[i]).squeeze(0)      # a 1-row product: still no vendor GEMM
            plan = layer.plan(g.x, g.edge_index, g.edge_attr)
            if rows is not None and i == len(layers) - 1:
                plan = plan.on_rows(rows) or plan
            z = propagate(plan, z, b)
        return z

    def forward(self, g, rows=None):
        """`rows` (an extension of the reference's signature; a bool mask over the nodes that the caller keeps): the rows of
        the logits that will be READ -- `g.train_mask` in the training step (flat_amazon.py:101 indexes the output with it),
        the validation and training rows in evaluation (:109-114).  The LAST layer's propagate step then runs on the
        operator restricted to them; every other row of the result holds the last layer's bias.  In a TextGCN graph the
        word rows, which nobody reads, hold two thirds of the operator's entries."""
        if (_COLLAPSE and len(self.layers) > 1 and not torch.is_grad_enabled()
                and (not self.training or self.dropout == 0)):
            return self._collapsed_forward(g, rows)
        x = g.x
        pending = 0.0                          # dropout still owed to x (fused into the next layer)
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            kw = {"rows": rows} if (rows is not None and i == last) else {}
            x = layer(x, g.edge_index, g.edge_attr, input_dropout=pending, **kw) if pending > 0.0 \
                else layer(x, g.edge_index, g.edge_attr, **kw)
            pending = 0.0
            if i < len(self.layers) - 1:
                if _FUSED_DROPOUT and self.training and 0.0 < self.dropout < 1.0 and not x.is_sparse:
                    pending = float(self.dropout)
                else:
                    x = nn.functional.dropout(x, p=self.dropout, training=self.training)
        return x
