"""Dense X @ W of the GCN layers on the fp32 matrix cores (libtgcn.so `tgcn_gemm_*`,
pytextgcn_amd/csrc/dense.hip): replaces `torch.matmul(x, self.weight)` of PyG-1.6.3
GCNConv.forward (reference call site textgcn/lib/models.py:20) and its autograd for tall-skinny
shapes -- millions of rows, layer widths of a few hundred (one launch while the small operand fits the
LDS, column groups / k chunks beyond: DBpedia's 219 classes at hidden width 200 take two groups)."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib
from .plan import _stream_ptr, alloc_padded


def enable_split_gemms(on: bool = True) -> bool:
    """Opt-in numerical mode of the layer-2 products (`tgcn_set_gemm_split`): fp32 products formed from an exact
    three-way bf16 split on the bf16 matrix cores -- fp32-accurate, faster, but not the fp32 FMA chain bit for bit.
    Returns the previous setting."""
    return bool(_lib.load().tgcn_set_gemm_split(1 if on else 0))


def _require(x: Tensor, w: Tensor) -> None:
    """The dense layers run on libtgcn.so only: anything else is an error, never a silent vendor fallback."""
    if not x.is_cuda or not w.is_cuda:
        raise RuntimeError(f"pytextgcn_amd: dense x @ W needs both operands on an AMD GPU (x on {x.device}, W on "
                           f"{w.device}); there is no CPU fallback")
    if x.dtype != torch.float32 or w.dtype != torch.float32 or x.dim() != 2 or w.dim() != 2:
        raise TypeError(f"pytextgcn_amd: dense x @ W takes 2-D float32 operands, got {x.dtype} {tuple(x.shape)} and "
                        f"{w.dtype} {tuple(w.shape)} (the reference casts the model with .float(), flat_amazon.py:85)")
    if x.size(1) != w.size(0) or w.size(0) == 0 or w.size(1) == 0:
        raise ValueError(f"x @ W: shapes {tuple(x.shape)} and {tuple(w.shape)} do not fit")


def _rowmajor4(t: Tensor) -> Tensor:
    """Unit column stride, row stride a multiple of 4 and a 16-byte aligned base (float4 loads)."""
    if t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % 16 != 0:
        if t.size(1) % 4 == 0:
            return t.contiguous()
        pad = (-t.size(1)) % 4
        return torch.nn.functional.pad(t, (0, pad))[:, :t.size(1)]
    return t


def _check_seed(seed: Tensor, dev) -> None:
    if seed.dtype != torch.int64 or seed.numel() != 1 or seed.device != dev:
        raise TypeError("dropout seed must be a one-element int64 tensor on the operand's device")


class _row_keys:
    """`with _row_keys(lib, keys):` -- the dropout products launched inside take mask row i + key0 for row i < split and
    i + key1 otherwise (`tgcn_set_dropout_row_keys`, keys = (split, key0, key1)); the thread's state is put back to the
    identity on the way out, so no call outside this module ever sees keys it did not ask for."""

    def __init__(self, lib, keys):
        self.lib, self.keys = lib, keys

    def __enter__(self):
        if self.keys is not None:
            _lib.check(self.lib.tgcn_set_dropout_row_keys(*(int(v) for v in self.keys)))

    def __exit__(self, *exc):
        if self.keys is not None:
            self.lib.tgcn_set_dropout_row_keys(0, 0, 0)
        return False


def gemm_nn(a: Tensor, b: Tensor, p: float = 0.0, seed: Tensor = None, record_mask: bool = False, keys=None,
            out: Tensor = None):
    """a [N, k] @ b [k, n]; with `seed`: dropout(a, p) @ b, the mask regenerated from the seed.
    `record_mask` (with `seed`): returns (product, mask) where `mask` is the kernel's record of its keep decisions
    ([N, words] int32, `tgcn_gemm_nn_dropout_mask`) for `gemm_tn(..., mask=...)`, or None when a product of this shape
    cannot record it.  `keys` = (split, key0, key1): which mask row a row of `a` is (see `_row_keys`).  `out`: a
    float32 [N, n] result buffer of the caller's (unit column stride, rows 16-byte aligned)."""
    lib = _lib.load()
    a, b = _rowmajor4(a), b.contiguous()
    N, k = a.shape
    n = b.size(1)
    if out is None:
        c = alloc_padded(N, n, a.device)      # rows of 4 j floats (zero pad columns): the propagate step takes it as it is
    else:
        c = out
        if c.shape != (N, n) or c.dtype != torch.float32 or c.device != a.device or (N and c.stride(1) != 1) \
                or c.stride(0) % 4 != 0 or c.data_ptr() % 16 != 0:
            raise ValueError("gemm_nn: `out` must be a float32 [N, n] device tensor with unit column stride and 16-byte rows")
    args = (a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), N, k, n)
    mask = None
    if seed is None:
        _lib.check(lib.tgcn_gemm_nn(*args, _stream_ptr(a.device)))
    else:
        _check_seed(seed, a.device)
        words = int(lib.tgcn_dropout_mask_words(k, n)) if record_mask else 0
        with _row_keys(lib, keys):
            if words:
                mask = torch.empty(N, words, dtype=torch.int32, device=a.device)
                _lib.check(lib.tgcn_gemm_nn_dropout_mask(*args, float(p), seed.data_ptr(), mask.data_ptr(), mask.stride(0),
                                                         _stream_ptr(a.device)))
            else:
                _lib.check(lib.tgcn_gemm_nn_dropout(*args, float(p), seed.data_ptr(), _stream_ptr(a.device)))
    return (c, mask) if record_mask else c


def gemm_nt(a: Tensor, b: Tensor, p: float = 0.0, seed: Tensor = None, note_colsums: bool = False,
            mask: Tensor = None, keys=None) -> Tensor:
    """a [N, k] @ b[n, k]^T; with `seed` the [N, n] result is masked and scaled (dropout backward).
    `note_colsums`: the kernel also sums the columns of the result it stores and the sums are recorded for
    `plan.colsum` (the result is a gradient on its way to a layer with a bias).  `mask` (with `seed` and `note_colsums`):
    the record `gemm_nn(..., record_mask=True)` left of the same dropout -- same bits, no hashing."""
    lib = _lib.load()
    a, b = _rowmajor4(a), b.contiguous()
    N, k = a.shape
    n = b.size(0)
    c = torch.empty(N, n, dtype=torch.float32, device=a.device)
    args = (a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), N, k, n)
    if note_colsums:
        from .plan import note_colsum
        if seed is not None:
            _check_seed(seed, a.device)
        sums = torch.empty(n, dtype=torch.float32, device=a.device)
        ws_bytes = lib.tgcn_gemm_nt_colsum_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
        with _row_keys(lib, keys if seed is not None else None):
            if mask is not None and seed is not None:
                if mask.dtype != torch.int32 or mask.device != a.device or mask.dim() != 2 or mask.size(0) != N or \
                        mask.stride(1) != 1:
                    raise TypeError("mask must be the [N, words] int32 record of gemm_nn(..., record_mask=True)")
                _lib.check(lib.tgcn_gemm_nt_colsum_mask(*args, float(p), seed.data_ptr(), mask.data_ptr(), mask.stride(0),
                                                        sums.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr(a.device)))
            else:
                _lib.check(lib.tgcn_gemm_nt_colsum(*args, float(p), seed.data_ptr() if seed is not None else None,
                                                   sums.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr(a.device)))
        note_colsum(c, sums)
        return c
    if seed is None:
        _lib.check(lib.tgcn_gemm_nt(*args, _stream_ptr(a.device)))
    else:
        _check_seed(seed, a.device)
        with _row_keys(lib, keys):
            _lib.check(lib.tgcn_gemm_nt_dropout(*args, float(p), seed.data_ptr(), _stream_ptr(a.device)))
    return c


def gemm_tn(a: Tensor, g: Tensor, p: float = 0.0, seed: Tensor = None, mask: Tensor = None, keys=None) -> Tensor:
    """a[N, k]^T @ g[N, n]; with `seed`: dropout(a, p)^T @ g; `mask`: the record `gemm_nn(..., record_mask=True)` left
    of the same dropout (same bits as hashing from the seed, without the hashing)."""
    lib = _lib.load()
    if a.stride(1) != 1:
        a = a.contiguous()
    if g.stride(1) != 1:
        g = g.contiguous()
    N, k = a.shape
    n = g.size(1)
    c = torch.empty(k, n, dtype=torch.float32, device=a.device)
    ws_bytes = lib.tgcn_gemm_tn_workspace_bytes(N, k, n)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=a.device)
    args = (a.data_ptr(), a.stride(0), g.data_ptr(), g.stride(0), c.data_ptr(), c.stride(0), N, k, n)
    if seed is None:
        _lib.check(lib.tgcn_gemm_tn(*args, ws.data_ptr(), ws.numel(), _stream_ptr(a.device)))
    else:
        _check_seed(seed, a.device)
        with _row_keys(lib, keys):
            if mask is not None:
                if mask.dtype != torch.int32 or mask.device != a.device or mask.dim() != 2 or mask.size(0) != N or \
                        mask.stride(1) != 1:
                    raise TypeError("mask must be the [N, words] int32 record of gemm_nn(..., record_mask=True)")
                _lib.check(lib.tgcn_gemm_tn_dropout_mask(*args, float(p), seed.data_ptr(), mask.data_ptr(), mask.stride(0),
                                                         ws.data_ptr(), ws.numel(), _stream_ptr(a.device)))
            else:
                _lib.check(lib.tgcn_gemm_tn_dropout(*args, float(p), seed.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    _stream_ptr(a.device)))
    return c


class _XW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor):
        ctx.save_for_backward(x, w)
        return gemm_nn(x.detach(), w.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w = ctx.saved_tensors
        dx = gemm_nt(g, w, note_colsums=True) if ctx.needs_input_grad[0] else None        # g @ w^T
        dw = gemm_tn(x, g) if ctx.needs_input_grad[1] else None        # x^T @ g
        return dx, dw


class _XWDropout(torch.autograd.Function):
    """dropout(x, p) @ w with the mask a stateless hash of (seed, row, column) in all three GEMMs: the dropped
    activation is never stored.  The forward product leaves its keep decisions as 4 bytes per 32 elements (1 / 32 of
    x) when a weight gradient will be wanted; dropout(x)^T @ g then tests bits instead of hashing every element a
    second time -- the same decisions, so the same bits in the result."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, p: float, seed: Tensor, keys=None):
        ctx.p, ctx.keys = p, keys
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            out, mask = gemm_nn(x.detach(), w.detach(), p, seed, record_mask=True, keys=keys)
        else:
            out, mask = gemm_nn(x.detach(), w.detach(), p, seed, keys=keys), None
        ctx.has_mask = mask is not None
        ctx.save_for_backward(x, w, seed, *([mask] if mask is not None else []))
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w, seed = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        k = ctx.keys
        dx = gemm_nt(g, w, ctx.p, seed, note_colsums=True, mask=mask, keys=k) if ctx.needs_input_grad[0] else None     # mask * (g @ w^T) / (1 - p)
        dw = gemm_tn(x, g, ctx.p, seed, mask, keys=k) if ctx.needs_input_grad[1] else None     # dropout(x)^T @ g
        return dx, dw, None, None, None


def new_seed(device) -> Tensor:
    """A fresh 64-bit seed drawn ON the device from torch's generator (no host sync; under HIP-graph
    capture every replay draws a new one)."""
    return torch.empty(1, dtype=torch.int64, device=device).random_()


def xw_dropout(x: Tensor, w: Tensor, p: float, seed: Tensor = None, keys=None) -> Tensor:
    """dropout(x, p) @ w (training-mode inverted dropout, textgcn/lib/models.py:23 followed by the next
    layer's x @ W) as ONE pass over x.  `keys` = (split, key0, key1): which row of the mask a row of x is (the 1-D
    partition keys a node's mask by its position in the partition, not by the local row: `tgcn_set_dropout_row_keys`)."""
    if p <= 0.0:
        return xw(x, w)
    _require(x, w)
    return _XWDropout.apply(x, w, float(p), new_seed(x.device) if seed is None else seed, keys)


def xw(x: Tensor, w: Tensor) -> Tensor:
    """x @ w with gradients, on the hand-written MFMA kernels."""
    _require(x, w)
    return _XW.apply(x, w)
