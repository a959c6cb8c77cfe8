"""GraphPlan: the cached, device-resident normalised operator behind GCNConv.

Host-side owner of an opaque `tgcn_plan` (include/tgcn.h).  It replaces what PyG-1.6.3
`gcn_norm` recomputes on every layer call in the reference (textgcn/lib/models.py:11-15 build the
layers with cached=False; models.py:20 calls them 4x per epoch) by one device-side build per graph.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"pytextgcn_amd: `{name}` lives on {t.device}; the GCN path runs only on an AMD GPU "
            "through libtgcn.so (there is no CPU fallback)")


# How gcn_norm's arithmetic is carried out (include/tgcn.h, `normalize` of tgcn_plan_create):
#   "reference"  THE DEFAULT: the bits of the reference's CPU path (PyG-1.6.3 gcn_norm as textgcn/lib/models.py:11-20 runs
#                it): one fp32 accumulator per node, weights added sequentially in edge order, the loop last, and PyG's
#                association (dis[src] * w) * dis[dst].  Every weight equals oracle/gcn_oracle.py's bit for bit, so a
#                network built by the import swap of INTEGRATION.md returns the reference's numbers to BASELINE.json's 1e-5
#                (the only rounding left is the summation order inside the SpMM).  The association is not symmetric, so
#                M^T is stored beside M (twice the plan memory; the transposed launch reads its own block).
#   "accurate"   opt-in: float64 degree sums rounded to fp32 once, and w * (dis[src] * dis[dst]).  Within fp32 rounding of
#                the EXACT normalisation (the reference's sequential fp32 sum over a hub's ~10^6 edges is a few 1e-5
#                off it), a symmetric graph stays bitwise symmetric, and one stored block serves M and M^T.
_DEGREE_SUM = "reference"


def set_degree_sum(mode: str) -> str:
    """Package default for plans built from now on ("reference" | "accurate"); returns the previous one."""
    global _DEGREE_SUM
    if mode not in _lib.DEGREE_SUMS:
        raise ValueError(f"degree_sum must be one of {sorted(_lib.DEGREE_SUMS)}")
    prev, _DEGREE_SUM = _DEGREE_SUM, mode
    return prev


def default_degree_sum() -> str:
    return _DEGREE_SUM


class GraphPlan:
    """M = D^-1/2 (A + I') D^-1/2 with M[target, source], rows [row_begin, row_end), kept in HBM
    as CSR with interleaved (col, val) pairs, plus M^T unless the operator is symmetric."""

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                 add_self_loops=True, normalize: bool = True,
                 row_range: Optional[Tuple[int, int]] = None, degree_sum: Optional[str] = None):
        lib = _lib.load()
        degree_sum = _DEGREE_SUM if degree_sum is None else degree_sum
        if degree_sum not in _lib.DEGREE_SUMS:
            raise ValueError(f"degree_sum must be one of {sorted(_lib.DEGREE_SUMS)}")
        self.degree_sum = degree_sum if normalize else None
        norm_mode = (_lib.NORM_REFERENCE if degree_sum == "reference" else _lib.NORM_ACCURATE) if normalize else _lib.NORM_OFF
        _require_cuda(edge_index, "edge_index")
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.long()
        self.device = edge_index.device
        n_edges = edge_index.size(1)
        if edge_weight is not None:
            _require_cuda(edge_weight, "edge_weight")
            if edge_weight.numel() != n_edges:
                raise ValueError(f"edge_weight has {edge_weight.numel()} entries for {n_edges} edges")
            edge_weight = edge_weight.detach().reshape(-1).float().contiguous()
        src, dst = edge_index[0], edge_index[1]           # views; strides are passed through
        row_begin, row_end = (0, num_nodes) if row_range is None else row_range
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.tgcn_plan_create(
                num_nodes, n_edges,
                src.data_ptr() if n_edges else None, src.stride(0) if n_edges else 1,
                dst.data_ptr() if n_edges else None, dst.stride(0) if n_edges else 1,
                edge_weight.data_ptr() if edge_weight is not None else None,
                int(add_self_loops), norm_mode, row_begin, row_end,
                self.device.index if self.device.index is not None else torch.cuda.current_device(),
                _stream_ptr(self.device), ctypes.byref(handle)))
        self._adopt(lib, handle, n_cols=num_nodes, n_cols_t=num_nodes)
        self.row_begin, self.row_end = int(row_begin), int(row_end)

    def _adopt(self, lib, handle, n_cols: int, n_cols_t: int) -> None:
        self._h = handle
        self._lib = lib
        self.num_nodes = int(n_cols)                  # rows of the forward operand X
        self.n_cols, self.n_cols_t = int(n_cols), int(n_cols_t)
        self.row_begin, self.row_end = 0, self.query(_lib.Q_N_ROWS)
        self.n_rows = self.query(_lib.Q_N_ROWS)        # rows of the forward result
        self.n_rows_t = self.query(_lib.Q_N_ROWS_T)    # rows of the transposed result
        self.nnz = self.query(_lib.Q_NNZ)
        self.nnz_t = self.query(_lib.Q_NNZ_T)
        self.symmetric = bool(self.query(_lib.Q_SYMMETRIC))
        self.has_transpose = bool(self.query(_lib.Q_HAS_TRANSPOSE))

    @classmethod
    def from_coo(cls, row: Tensor, col: Tensor, val: Optional[Tensor], n_rows: int, n_cols: int,
                 with_transpose: bool = False) -> "GraphPlan":
        """Plan of an explicit n_rows x n_cols operator M[row[i], col[i]] = val[i] (duplicates add
        up, nothing is normalised): the per-rank local operators of pytextgcn_amd.sharded."""
        lib = _lib.load()
        _require_cuda(row, "row")
        self = cls.__new__(cls)
        self.device = row.device
        row = row.long().contiguous()
        col = col.long().contiguous()
        if val is not None:
            val = val.detach().float().contiguous()
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.tgcn_plan_create_coo(
                n_rows, n_cols, row.numel(), row.data_ptr() if row.numel() else None,
                col.data_ptr() if row.numel() else None,
                val.data_ptr() if val is not None else None, int(with_transpose),
                self.device.index if self.device.index is not None else torch.cuda.current_device(),
                _stream_ptr(self.device), ctypes.byref(handle)))
        self._adopt(lib, handle, n_cols=n_cols, n_cols_t=n_rows)
        return self

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            self._lib.tgcn_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection ----------------------------------------------------------------------
    def query(self, what: int) -> int:
        out = ctypes.c_int64()
        _lib.check(self._lib.tgcn_plan_query(self._h, what, ctypes.byref(out)))
        return int(out.value)

    def stats(self) -> dict:
        return {"n_nodes": self.num_nodes, "n_rows": self.n_rows, "nnz": self.nnz,
                "symmetric": self.symmetric, "items": self.query(_lib.Q_ITEMS),
                "long_rows": self.query(_lib.Q_LONG_ROWS), "segments": self.query(_lib.Q_SEGMENTS),
                "hot_rows": self.query(_lib.Q_HOT_ROWS), "device_bytes": self.query(_lib.Q_DEVICE_BYTES)}

    def algorithmic_bytes(self, F: int, bias: bool = False, transpose: bool = False) -> int:
        """SURVEY.md 8(d) / BASELINE.md gather model, unpadded F, no cache reuse assumed:
        nnz*(4 + 4 + 4F) + n_rows*(4 + 4F) (+4F for the bias)."""
        nnz = self.nnz_t if transpose else self.nnz
        n_out = self.n_rows_t if transpose else self.n_rows
        return nnz * (8 + 4 * F) + n_out * (4 + 4 * F) + (4 * F if bias else 0)

    def export_csr(self, transpose: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
        nnz = self.nnz_t if transpose else self.nnz
        n_out = self.n_rows_t if transpose else self.n_rows
        rowptr = torch.empty(n_out + 1, dtype=torch.int32, device=self.device)
        col = torch.empty(nnz, dtype=torch.int32, device=self.device)
        val = torch.empty(nnz, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.tgcn_plan_export(self._h, int(transpose), rowptr.data_ptr(),
                                              col.data_ptr(), val.data_ptr(),
                                              _stream_ptr(self.device)))
        return rowptr, col, val

    def _workspace(self, transpose: int, F: int, device) -> Optional[Tensor]:
        """Carry workspace of one SpMM, kept per (stream, direction, width): the layer widths alternate every
        call, and two streams (or threads on their own streams) must never share one (include/tgcn.h).  The
        buffer only ever serves launches on the stream it is keyed by, so reuse is ordered by that stream."""
        key = (torch.cuda.current_stream(device).cuda_stream, transpose, F)
        cache = self.__dict__.setdefault("_ws_cache", {})
        ws = cache.get(key)
        if ws is None:
            n = self._lib.tgcn_spmm_workspace_bytes(self._h, transpose, F)
            ws = torch.empty(n, dtype=torch.uint8, device=device) if n else False
            if len(cache) >= 16:
                cache.clear()
            cache[key] = ws
        return ws if ws is not False else None

    def transposed_on_rows(self, keep: Tensor, max_share: float = 0.85) -> Optional["GraphPlan"]:
        """M^T restricted to the COLUMNS (operand rows) `keep` selects: the operator that gives M^T @ g for every g whose
        other rows are exactly zero.  Built once per (plan, mask) from this plan's own transposed CSR (same entries, same
        order within a row) and kept; None when the mask keeps more than `max_share` of the entries (nothing to gain), or
        while a HIP graph is being captured and the operator is not there yet (it allocates and synchronises)."""
        cache = self.__dict__.setdefault("_restricted_t", {})
        key = (keep.data_ptr(), keep._version, keep.numel())
        hit = cache.get(key)
        if hit is not None:
            return hit[0]
        if torch.cuda.is_current_stream_capturing() or keep.dtype != torch.bool or keep.device != self.device \
                or keep.numel() != (self.n_cols_t):
            return None
        rowptr, col, val = self.export_csr(transpose=True)
        sel = keep[col.long()]
        kept = int(sel.sum().item())
        op = None
        if kept <= max_share * max(1, col.numel()):
            row = torch.repeat_interleave(torch.arange(self.n_rows_t, device=self.device), (rowptr[1:] - rowptr[:-1]).long())
            op = GraphPlan.from_coo(row[sel], col[sel].long(), val[sel], self.n_rows_t, self.n_cols_t)
        if len(cache) >= 4:
            cache.clear()
        cache[key] = (op, keep)                      # (the entry holds the mask: its address cannot be recycled under the key)
        return op

    def on_rows(self, keep: Tensor, max_share: float = 0.85) -> Optional["GraphPlan"]:
        """M restricted to the ROWS `keep` selects (with its transpose): the operator whose product equals M @ x on those
        rows and leaves every other row at the bias.  For a LAST layer whose other rows nobody reads -- the loss sees
        `out[g.train_mask]` only (flat_amazon.py:101), the metrics the validation and training rows (:109-114); the word
        rows, which nobody reads, hold two thirds of a TextGCN operator's entries.  Built once per (plan, mask) from this
        plan's own CSR (same entries, same order within a row) and kept; None when the mask keeps more than `max_share` of
        the entries, or while a HIP graph is being captured and the operator is not there yet."""
        cache = self.__dict__.setdefault("_restricted_rows", {})
        key = (keep.data_ptr(), keep._version, keep.numel())
        hit = cache.get(key)
        if hit is not None:
            return hit[0]
        if torch.cuda.is_current_stream_capturing():
            return None
        if keep.dtype != torch.bool or keep.device != self.device or keep.numel() != self.n_rows:
            raise ValueError(f"rows: a bool mask of {self.n_rows} entries on {self.device} is required")
        builds = self.__dict__["_restricted_builds"] = self.__dict__.get("_restricted_builds", 0) + 1
        if builds == 4:
            import warnings
            warnings.warn("pytextgcn_amd: `rows=` has been given four different mask tensors for one graph; every new tensor "
                          "costs the construction of a restricted operator -- compute the mask once and pass the SAME "
                          "tensor every call (e.g. rows_eval = g.val_mask | g.train_mask before the loop)")
        rowptr, col, val = self.export_csr()
        row = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), (rowptr[1:] - rowptr[:-1]).long())
        sel = keep[row]
        op = None
        if int(sel.sum().item()) <= max_share * max(1, col.numel()):
            op = GraphPlan.from_coo(row[sel], col[sel].long(), val[sel], self.n_rows, self.n_cols, with_transpose=True)
        if len(cache) >= 4:
            cache.clear()
        cache[key] = (op, keep)                      # (the entry holds the mask: its address cannot be recycled under the key)
        return op

    # -- compute ----------------------------------------------------------------------------
    def spmm(self, x: Tensor, bias: Optional[Tensor] = None, transpose: bool = False,
             out: Optional[Tensor] = None, x2: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
        """out[r] = sum_j M(^T)[row_begin + r, j] x[j] (+ bias); x is [num_nodes, F] fp32.
        With `x2` the operand is split: columns [0, len(x)) read x, the remaining ones x2.
        `accumulate` (`tgcn_spmm_acc`): the sums are ADDED to `out` (required, no bias) on the rows that hold entries;
        rows without entries are not touched."""
        _require_cuda(x, "x")
        if x.dtype != torch.float32 or x.dim() != 2:
            raise TypeError(f"spmm operand must be a 2-D float32 tensor, got {x.dtype} {tuple(x.shape)}")
        need = self.n_cols_t if transpose else self.n_cols
        if x2 is not None and x.size(0) == 0:             # everything lives in the second buffer
            x, x2 = x2, None
        if x2 is not None and x2.size(0) == 0:
            x2 = None
        split = x.size(0)
        if x2 is not None:
            _require_cuda(x2, "x2")
            if x2.dtype != torch.float32 or x2.dim() != 2 or x2.size(1) != x.size(1):
                raise TypeError("x2 must be float32 with the same number of columns as x")
            if x2.stride(1) != 1:
                x2 = x2.contiguous()
        if x.size(0) + (0 if x2 is None else x2.size(0)) != need:
            raise ValueError(f"operand has {x.size(0) + (0 if x2 is None else x2.size(0))} rows, "
                             f"the operator has {need} columns")
        n_out = self.n_rows_t if transpose else self.n_rows
        if x.stride(1) != 1:
            x = x.contiguous()
        F = x.size(1)
        if bias is not None:
            bias = bias.detach().float().contiguous()
            if bias.numel() != F:
                raise ValueError(f"bias has {bias.numel()} entries for F={F}")
        if accumulate and (out is None or bias is not None):
            raise ValueError("accumulate=True adds into `out` (required) and takes no bias")
        if out is None:
            out = torch.empty(n_out, F, dtype=torch.float32, device=x.device)
        elif out.shape != (n_out, F) or out.dtype != torch.float32 or out.stride(1) != 1 or out.device != x.device:
            raise ValueError("`out` must be float32 [n_rows, F] with unit column stride on the operand's device")
        ws = self._workspace(int(transpose), F, x.device)
        ws_bytes = ws.numel() if ws is not None else 0
        if accumulate:
            _lib.check(self._lib.tgcn_spmm_acc(
                self._h, int(transpose), x.data_ptr(), x.stride(0),
                x2.data_ptr() if x2 is not None else None, x2.stride(0) if x2 is not None else 0, split, F,
                out.data_ptr(), out.stride(0), ws.data_ptr() if ws is not None else None, ws_bytes, _stream_ptr(x.device)))
            return out
        _lib.check(self._lib.tgcn_spmm_split(
            self._h, int(transpose), x.data_ptr(), x.stride(0),
            x2.data_ptr() if x2 is not None else None, x2.stride(0) if x2 is not None else 0, split, F,
            bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
            ws.data_ptr() if ws is not None else None, ws_bytes, _stream_ptr(x.device)))
        return out


def _spmm_adam(self, g: Tensor, param: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, max_exp_avg_sq: Optional[Tensor],
               lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               scalars: Optional[Tensor] = None, transpose: bool = True, g2: Optional[Tensor] = None) -> None:
    """param <- Adam(param, grad = M(^T) @ g) row by row, the gradient never stored (`tgcn_spmm_adam`).  `g2`: the
    operand's rows from g.size(0) on (a split operand, as GraphPlan.spmm's x2: `tgcn_spmm_adam_split`)."""
    _require_cuda(g, "g")
    F = g.size(1)
    n_out = self.n_rows_t if transpose else self.n_rows
    need = self.n_cols_t if transpose else self.n_cols
    have = g.size(0) + (g2.size(0) if g2 is not None else 0)
    ok = (g.dtype == torch.float32 and g.dim() == 2 and have == need and g.stride(1) == 1
          and param.shape == (n_out, F) and param.is_contiguous() and param.dtype == torch.float32)
    if g2 is not None:
        ok = ok and (g2.dtype == torch.float32 and g2.dim() == 2 and g2.size(1) == F and g2.stride(1) == 1
                     and g2.device == g.device)
    for t in (exp_avg, exp_avg_sq, max_exp_avg_sq):
        ok = ok and (t is None or (t.shape == param.shape and t.is_contiguous() and t.dtype == torch.float32))
    if not ok:
        raise ValueError("spmm_adam: operand / parameter / state shapes do not fit the operator")
    ws = self._workspace(int(transpose), F, g.device)
    _lib.check(self._lib.tgcn_spmm_adam_split(
        self._h, int(transpose), g.data_ptr(), g.stride(0), g2.data_ptr() if g2 is not None else None,
        g2.stride(0) if g2 is not None else 0, g.size(0) if g2 is not None else 0, F, param.data_ptr(), exp_avg.data_ptr(),
        exp_avg_sq.data_ptr(), max_exp_avg_sq.data_ptr() if max_exp_avg_sq is not None else None, param.stride(0),
        lr, beta1, beta2, eps, weight_decay, int(step), scalars.data_ptr() if scalars is not None else None,
        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_ptr(g.device)))


GraphPlan.spmm_adam = _spmm_adam


# Row-padded buffers.  The float4 SpMM path wants rows of 4 k floats; a width that is not a multiple of 4 (DBpedia's 219
# classes) used to be padded by a copy in front of every propagate step.  Producers inside this package (the nn GEMM,
# the cross-entropy gradient) instead allocate their [n, F] result as the first F columns of a zero-padded [n, F4] buffer
# and hand out the view; `padded_base` gives the propagate step the buffer back -- no copy.  Only this package ever holds
# the base, and its kernels write the first F columns only, so the pad columns stay zero.
_PADDED: dict = {}


def alloc_padded(n: int, F: int, device) -> Tensor:
    """A float32 [n, F] tensor that is the leading part of a zero-padded [n, round_up(F, 4)] buffer (a plain contiguous
    tensor when F is a multiple of 4)."""
    import weakref
    F4 = (F + 3) & ~3
    if F4 == F:
        return torch.empty(n, F, dtype=torch.float32, device=device)
    base = torch.empty(n, F4, dtype=torch.float32, device=device)
    base[:, F:].zero_()
    view = base[:, :F]
    key = base.data_ptr()
    _PADDED[key] = (weakref.ref(base, lambda _, k=key: _PADDED.pop(k, None)), F)
    return view


def padded_base(t: Tensor, width: int) -> Optional[Tensor]:
    """The zero-padded [n, width] buffer `t` is the leading part of, if it came from `alloc_padded` (else None)."""
    hit = _PADDED.get(t.data_ptr())
    base = hit[0]() if hit is not None else None
    if base is None or t.dim() != 2 or t.size(1) != hit[1] or base.shape != (t.size(0), width) \
            or t.stride() != (width, 1) or base.data_ptr() != t.data_ptr() or t.dtype != torch.float32:
        return None
    return base


# Column sums a producer kernel already took while it wrote the matrix (tgcn_masked_ce_grad: the gradient of the
# logits; tgcn_gemm_nt*: the gradient of the hidden activation).  Keyed by storage address, validated by a weak
# reference to the producing tensor (alive => the address was not recycled), its shape and its version counter.
_KNOWN_COLSUMS: dict = {}


def note_colsum(t: Tensor, sums: Tensor) -> None:
    """Record `sums == t.sum(0)` for the tensor `t` as it is now (any later in-place edit invalidates the note)."""
    import weakref
    key = t.data_ptr()
    _KNOWN_COLSUMS[key] = (weakref.ref(t, lambda _, k=key: _KNOWN_COLSUMS.pop(k, None)), tuple(t.shape), t._version, sums,
                           sums._version)


def _known_colsum(g: Tensor) -> Optional[Tensor]:
    hit = _KNOWN_COLSUMS.get(g.data_ptr())
    if hit is None:
        return None
    t = hit[0]()
    if t is None or t.data_ptr() != g.data_ptr() or hit[2] != g._version or t._version != g._version \
            or hit[3]._version != hit[4]:
        return None                        # (the last test: somebody scaled or clipped the noted sums in place)
    if hit[1] == tuple(g.shape) and g.is_contiguous():
        return hit[3]
    # the leading columns of a noted zero-padded buffer (alloc_padded): the same sums, cut to width
    if g.dim() == 2 and len(hit[1]) == 2 and g.size(0) == hit[1][0] and g.size(1) < hit[1][1] and g.stride() == t.stride():
        return hit[3][:g.size(1)]
    return None


# Rows a producer kernel knows to be EXACTLY zero.  `tgcn_masked_ce_grad` writes 0.f into every row of the logits'
# gradient that the loss mask does not select (flat_amazon.py:101-102: the loss sees `out[g.train_mask]` only) -- at c4 that
# is every word node and the validation / test documents, 28 % of the rows -- and the propagate step that consumes the
# gradient, dXW2 = M^T dOut, gathers those rows once per entry like any other.  The note lets it use M^T restricted to the
# columns that can contribute (GraphPlan.transposed_on_rows).  Keyed and validated like the column-sum notes.
_KNOWN_ZERO_ROWS: dict = {}
_SKIP_ZERO_ROWS = __import__("os").environ.get("TGCN_SKIP_ZERO_ROWS", "1") != "0"      # (A/B runs: TGCN_SKIP_ZERO_ROWS=0)


def enable_zero_row_skipping(on: bool = True) -> bool:
    """The backward propagate step of a gradient whose zero rows are known (the fused cross-entropy's) runs on the
    operator restricted to the non-zero rows (default: on).  Returns the previous setting."""
    global _SKIP_ZERO_ROWS
    prev, _SKIP_ZERO_ROWS = _SKIP_ZERO_ROWS, bool(on)
    return prev


def note_zero_rows(t: Tensor, keep: Tensor) -> None:
    """Record that row r of `t` is exactly zero wherever `keep[r]` is False, for `t` and `keep` as they are now."""
    import weakref
    key = t.data_ptr()
    _KNOWN_ZERO_ROWS[key] = (weakref.ref(t, lambda _, k=key: _KNOWN_ZERO_ROWS.pop(k, None)), t.size(0), t._version, keep,
                             keep._version)


def known_nonzero_rows(g: Tensor) -> Optional[Tensor]:
    """The bool mask of the rows of `g` that may be non-zero, if its producer left one (else None)."""
    if not _SKIP_ZERO_ROWS:
        return None
    hit = _KNOWN_ZERO_ROWS.get(g.data_ptr())
    if hit is None:
        return None
    t = hit[0]()
    if t is None or t.data_ptr() != g.data_ptr() or g.dim() != 2 or g.size(0) != hit[1] or hit[2] != g._version \
            or t._version != g._version or hit[3]._version != hit[4] or hit[3].numel() != g.size(0):
        return None
    return hit[3]


def colsum(g: Tensor) -> Tensor:
    """Column sums of a float32 [n, F] device matrix (the bias gradient), deterministic.  A matrix whose producer
    kernel left its column sums (`note_colsum`) is not read again."""
    known = _known_colsum(g)
    if known is not None:
        return known
    lib = _lib.load()
    _require_cuda(g, "g")
    if g.dtype != torch.float32 or g.dim() != 2:
        raise TypeError("colsum operand must be a 2-D float32 tensor")
    if g.stride(1) != 1:
        g = g.contiguous()
    n, F = g.shape
    out = torch.empty(F, dtype=torch.float32, device=g.device)
    ws_bytes = lib.tgcn_colsum_workspace_bytes(n, F)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    _lib.check(lib.tgcn_colsum(g.data_ptr(), g.stride(0), n, F, out.data_ptr(), ws.data_ptr(),
                               ws.numel(), _stream_ptr(g.device)))
    return out


# ------------------------------------------------------------------------------------------
# Plan cache: the graph is static across layers and epochs, so both
# GCNConv layers and every epoch share one plan.  Entries keep the tensors alive (so a data_ptr
# cannot be recycled under a live key) and are invalidated by in-place edits via `_version`.
# ------------------------------------------------------------------------------------------
_PLAN_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()
_PLAN_CACHE_MAX = 4


def _key(edge_index: Tensor, edge_weight: Optional[Tensor], n: int, loops: bool, norm: bool, degree_sum: str):
    k = (edge_index.device, edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
         tuple(edge_index.stride()), edge_index.dtype, n, loops, norm, degree_sum if norm else None)
    if edge_weight is not None:
        k += (edge_weight.data_ptr(), edge_weight._version, tuple(edge_weight.shape),
              edge_weight.dtype)
    return k


def plan_for(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
             add_self_loops: bool = True, normalize: bool = True, degree_sum: Optional[str] = None) -> GraphPlan:
    degree_sum = _DEGREE_SUM if degree_sum is None else degree_sum
    key = _key(edge_index, edge_weight, num_nodes, add_self_loops, normalize, degree_sum)
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        _PLAN_CACHE.move_to_end(key)
        return hit[0]
    plan = GraphPlan(edge_index, edge_weight, num_nodes, add_self_loops, normalize, degree_sum=degree_sum)
    _PLAN_CACHE[key] = (plan, edge_index, edge_weight)
    while len(_PLAN_CACHE) > _PLAN_CACHE_MAX:
        _PLAN_CACHE.popitem(last=False)
    return plan


def clear_plan_cache() -> None:
    _PLAN_CACHE.clear()
