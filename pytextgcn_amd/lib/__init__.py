"""Import-path mirror of `textgcn.lib` (textgcn/lib/__init__.py:1-4), so that a reference script or
test changes only the package name:

    from textgcn.lib.models import *                         ->  from pytextgcn_amd.lib.models import *
    from textgcn.lib import sliding_window_tester, test_sym_matrix, compute_word_word_edges
                                                             ->  from pytextgcn_amd.lib import ...

(This directory also holds the built libtgcn.so.)
"""
from ..graphbuilder import compute_word_word_edges, sliding_window_tester, test_sym_matrix
from ..text2graph import Text2GraphTransformer

__all__ = ["Text2GraphTransformer", "compute_word_word_edges", "sliding_window_tester", "test_sym_matrix"]
