"""pytextgcn_amd -- MI355X-native implementation of PyTextGCN's GCN hot path.

Mirrors the reference's public surface (textgcn/__init__.py:1-4 exports `Text2GraphTransformer`
and `models`): `models.GCN(in, out, n_hidden_gcn=..., dropout=...)(graph)` runs the two GCNConv
layers as hand-written HIP kernels (libtgcn.so, include/tgcn.h) on an AMD Instinct MI355X.
"""
from . import functional, models, optim, train
from .conv import GCNConv, enable_activation_reuse
from .data import Data
from .dense import enable_split_gemms
from .models import GCN, enable_fused_dropout, enable_linear_collapse
from .reorder import cluster_documents, reorder_documents
from .plan import GraphPlan, clear_plan_cache, colsum, enable_zero_row_skipping, plan_for, set_degree_sum
from .text2graph import Text2GraphTransformer

__all__ = ["Text2GraphTransformer", "models", "functional", "optim", "train", "GCN", "GCNConv", "Data", "GraphPlan", "plan_for", "colsum",
           "clear_plan_cache", "set_degree_sum", "enable_zero_row_skipping", "enable_activation_reuse", "enable_linear_collapse", "enable_fused_dropout", "enable_split_gemms", "reorder_documents", "cluster_documents"]
