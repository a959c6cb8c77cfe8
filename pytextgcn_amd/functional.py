"""Fused training-step operators around the GCN (the caller of the path, flat_amazon.py:82,99-106).

`masked_cross_entropy(logits, target, mask)` equals
`CrossEntropyLoss(reduction='mean')(logits[mask], target[mask])` -- what the reference computes at
flat_amazon.py:101-102 (train) and :110 (validation) -- but runs as ONE HIP kernel over the full
[N, C] logits (libtgcn.so `tgcn_masked_ce`) that also produces the gradient, instead of boolean-mask
indexing + log_softmax + nll_loss and their backward kernels.
"""
from __future__ import annotations

import ctypes

import torch
from torch import Tensor

from . import _lib
from .plan import _require_cuda, _stream_ptr, alloc_padded, note_colsum, note_zero_rows, padded_base

_COUNT_CACHE: dict = {}


def _mask_count(mask: Tensor) -> int:
    """Rows selected by a (static) mask; one device sync per distinct mask OBJECT and version.  Keyed
    by id() and validated through a weak reference: a data pointer is no identity (the allocator hands
    the storage of a dead temporary to the next one)."""
    import weakref
    key = id(mask)
    hit = _COUNT_CACHE.get(key)
    if hit is not None and hit[0]() is mask and hit[1] == mask._version:
        return hit[2]
    count = int(mask.sum().item())
    try:
        ref = weakref.ref(mask, lambda _, k=key: _COUNT_CACHE.pop(k, None))
    except TypeError:
        return count
    _COUNT_CACHE[key] = (ref, mask._version, count)
    return count


_TARGET_OK: dict = {}


def _check_targets(target: Tensor, mask: Tensor, C: int) -> None:
    """Class indices of the selected rows must lie in [0, C): torch raises for anything else (its
    `ignore_index` is not implemented here), and the kernel would otherwise index outside the row.
    Labels and masks are static across epochs (text2graph.py:180-191), so the check -- one device sync
    -- runs once per (target, mask) object pair and version."""
    import weakref
    key = (id(target), id(mask), C)
    hit = _TARGET_OK.get(key)
    if hit is not None and hit[0]() is target and hit[1]() is mask and hit[2] == (target._version, mask._version):
        return
    bad = ((target < 0) | (target >= C)) & mask
    if bool(bad.any().item()):
        t = int(target[bad][0].item())
        raise IndexError(f"Target {t} is out of bounds for {C} classes (on a row selected by the mask)")
    try:
        _TARGET_OK[key] = (weakref.ref(target, lambda _, k=key: _TARGET_OK.pop(k, None)),
                           weakref.ref(mask, lambda _, k=key: _TARGET_OK.pop(k, None)),
                           (target._version, mask._version))
    except TypeError:
        pass


def _launch(logits: Tensor, target: Tensor, mask: Tensor, want_grad: bool, count=None, want_pred: bool = False):
    lib = _lib.load()
    _require_cuda(logits, "logits")
    if logits.dtype != torch.float32 or logits.dim() != 2:
        raise TypeError("logits must be a 2-D float32 tensor")
    n, C = logits.shape
    if target.shape != (n,) or mask.shape != (n,):
        raise ValueError("target and mask must have one entry per logits row")
    if mask.dtype != torch.bool:
        raise TypeError("mask must be a bool tensor")
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    if torch.cuda.is_current_stream_capturing():
        pass                                   # no sync inside a HIP-graph capture: checked on the eager warm-up step
    else:
        _check_targets(target, mask, C)
    target = target.long().contiguous()
    mask = mask.contiguous()
    if count is None:
        count = _mask_count(mask)
    inv = 1.0 / count if count else float("nan")          # torch: mean over an empty selection = nan
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    pred = torch.empty(n, dtype=torch.int64, device=logits.device) if want_pred else None
    if want_grad:
        # loss, gradient and the column sums of the gradient (the last layer's bias gradient) in one pass
        # rows of 4 j floats (zero pad columns) when C is not a multiple of 4: the backward propagate step then takes the
        # gradient as it is instead of padding a copy; its column sums are kept at the padded width for the same reason
        dlogits = alloc_padded(n, C, logits.device)
        C4 = (C + 3) & ~3
        dbias = torch.zeros(C4, dtype=torch.float32, device=logits.device)[:C] if C4 != C else \
            torch.empty(C, dtype=torch.float32, device=logits.device)
        ws_bytes = lib.tgcn_masked_ce_grad_workspace_bytes(n, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logits.device)
        _lib.check(lib.tgcn_masked_ce_grad(
            logits.data_ptr(), logits.stride(0), n, C, target.data_ptr(), mask.data_ptr(),
            ctypes.c_float(inv), loss.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), dbias.data_ptr(),
            pred.data_ptr() if pred is not None else None,
            ws.data_ptr(), ws_bytes, _stream_ptr(logits.device)))
        return loss, (dlogits, dbias), pred
    ws_bytes = lib.tgcn_masked_ce_workspace_bytes()
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logits.device)
    _lib.check(lib.tgcn_masked_ce_pred(
        logits.data_ptr(), logits.stride(0), n, C, target.data_ptr(), mask.data_ptr(),
        ctypes.c_float(inv), loss.data_ptr(), None, C,
        pred.data_ptr() if pred is not None else None,
        ws.data_ptr(), ws_bytes, _stream_ptr(logits.device)))
    return loss, None, pred


def _scale_by_device_scalar(x: Tensor, scale: Tensor) -> None:
    """x *= scale for a one-element float32 device tensor `scale`; a no-op on the device when it is exactly 1."""
    _lib.check(_lib.load().tgcn_scale_by_device_scalar(x.data_ptr(), x.numel(), scale.data_ptr(),
                                                       _stream_ptr(x.device)))
    torch.autograd.graph.increment_version(x)


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, mask: Tensor, count, want_pred: bool):
        loss, grads, pred = _launch(logits.detach(), target, mask, ctx.needs_input_grad[0], count, want_pred)
        ctx.save_for_backward(*(grads if grads is not None else ()))
        ctx.mask, ctx.mask_version = mask, mask._version      # (the zero-row note below is only true for THIS version)
        if not want_pred:
            return loss
        ctx.mark_non_differentiable(pred)
        return loss, pred

    @staticmethod
    def backward(ctx, grad_out: Tensor, *_):
        dlogits, dbias = ctx.saved_tensors
        # dlogits is ours: scale it in place (no second N x C pass) -- which makes this node single-use:
        # a second backward through it (retain_graph=True) would compound the scale, so it is refused
        if getattr(ctx, "_tgcn_used", False):
            raise RuntimeError("masked_cross_entropy: backward was already run through this loss (its gradient "
                               "buffer is scaled in place); recompute the loss instead of retain_graph=True")
        ctx._tgcn_used = True
        base = padded_base(dlogits, (dlogits.size(1) + 3) & ~3)       # the zero-padded buffer behind an odd class width
        if grad_out.is_cuda and grad_out.dtype == torch.float32 and grad_out.numel() == 1:
            # `loss.backward()` seeds this node with 1: the kernel reads the scalar and leaves at once.  (The kernel
            # walks a flat buffer: the padded one where there is one -- its pad columns are and stay zero.)
            _scale_by_device_scalar(base if base is not None else dlogits, grad_out)
            _scale_by_device_scalar(dbias, grad_out)
        else:
            dlogits.mul_(grad_out)
            dbias.mul_(grad_out)
        # the layer that produced the logits finds its bias gradient ready (plan.colsum) -- under the padded buffer too
        if base is not None:                  # (view and buffer share one address: one note, under the padded form)
            note_colsum(base, torch.as_strided(dbias, (base.size(1),), (1,)))     # dbias was cut from zeros(C4)
        else:
            note_colsum(dlogits, dbias)
        # every row the mask does not select is exactly zero (the kernel wrote 0.f there; scaling keeps it): the propagate
        # step that takes this gradient may skip those operand rows (plan.known_nonzero_rows)
        # -- unless the caller edited the mask in place since the forward pass: the gradient's zero pattern is the OLD
        # mask's, so no note is left and the consumer gathers every row
        if ctx.mask._version == ctx.mask_version:
            note_zero_rows(base if base is not None else dlogits, ctx.mask)
        return dlogits, None, None, None, None


def masked_cross_entropy(logits: Tensor, target: Tensor, mask: Tensor, count=None, return_pred: bool = False):
    """`count` overrides the divisor (default: rows selected by `mask`); the sharded path passes the
    GLOBAL count so that per-rank losses and gradients add up to the single-device ones.
    `return_pred=True` returns `(loss, pred)` with `pred = logits.argmax(1)` for EVERY row, taken in the
    same pass: `pred[mask]` is what the reference computes on the host at flat_amazon.py:111-114."""
    if logits.requires_grad and torch.is_grad_enabled():
        return _MaskedCE.apply(logits, target, mask, count, return_pred)
    loss, _, pred = _launch(logits, target, mask, False, count, return_pred)
    return (loss, pred) if return_pred else loss
