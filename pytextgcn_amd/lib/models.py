"""`textgcn.lib.models` import path (flat_amazon.py:14 `from textgcn.lib.models import *`)."""
from ..models import GCN

__all__ = ["GCN"]
