"""Data: the attribute bag `Text2GraphTransformer` returns (text2graph.py:192-193) and
`GCN.forward` reads (`g.x`, `g.edge_index`, `g.edge_attr`, models.py:18,20).

`torch_geometric.data.Data` is used by the reference; it is not a dependency here.  This class
keeps the part of its surface the reference scripts touch: keyword construction, attribute access,
`.to(device)` (flat_amazon.py:86), `num_nodes`, `keys`, pickling (text2graph.py:195-217).
A real `torch_geometric.data.Data` object works equally well with `GCN` (duck typing).
"""
from __future__ import annotations

import torch


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.y = y
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    @property
    def num_nodes(self):
        if self.x is not None:
            return self.x.size(0)
        if self.edge_index is not None and self.edge_index.numel() > 0:
            return int(self.edge_index.max()) + 1
        return None

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else self.edge_index.size(1)

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    def apply(self, func):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, func(v))
        return self

    def to(self, device, *args, **kwargs):
        return self.apply(lambda t: t.to(device, *args, **kwargs))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def __repr__(self):
        parts = []
        for k in self.keys:
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v}")
        return f"Data({', '.join(parts)})"
