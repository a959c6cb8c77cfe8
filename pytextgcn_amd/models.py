"""GCN: drop-in for `textgcn.lib.models.GCN` (textgcn/lib/models.py:6-25).

Same constructor signature and defaults, same `layers` ModuleList (state_dict keys
`layers.{i}.weight` (in, out) / `layers.{i}.bias`), same forward: dropout between layers, none
after the last, and NO activation -- the reference's activation call is commented out
(models.py:22) although `self.activation` is constructed (models.py:9); both facts are kept.
"""
from __future__ import annotations

from torch import nn

from .conv import GCNConv


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

    def forward(self, g):
        x = g.x
        for i, layer in enumerate(self.layers):
            x = layer(x, g.edge_index, g.edge_attr)
            if i < len(self.layers) - 1:
                x = nn.functional.dropout(x, p=self.dropout, training=self.training)
        return x
