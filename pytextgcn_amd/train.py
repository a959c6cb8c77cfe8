"""HIP-graph captured training / evaluation steps for the loop of flat_amazon.py:99-117.

Real TextGCN graphs are small (10^4-10^5 nodes): every kernel of an epoch then runs for microseconds
and the ~80 launches per epoch, each dressed in Python and dispatcher overhead, ARE the epoch time.
`GraphedTrainStep` captures one whole optimisation step -- forward through libtgcn.so, fused masked
cross-entropy, backward, fused Adam with a device-side step counter -- into one HIP graph
(`torch.cuda.CUDAGraph`, which on ROCm is a hipGraph) and replays it per epoch; `GraphedEval` does the
same for the eval forward.  Every compute entry point of include/tgcn.h only enqueues on the caller's
stream, which is what makes the capture legal (tests/test_gpu_parity.py checks it).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import conv, models
from .functional import masked_cross_entropy, _mask_count
from .optim import Adam


class GraphedTrainStep:
    """loss = masked CE(model(g), g.y, mask); backward; optimizer.step() -- as one graph replay.

    The optimizer must be `pytextgcn_amd.optim.Adam(..., capturable=True)` (or any torch optimizer
    built with capturable=True).  `warmup` eager steps run first (they are real training steps); the
    graph owns the gradient buffers afterwards, so do not call `zero_grad()` yourself.  `mask` must not change
    afterwards: its row count and the backward operator restricted to its rows (plan.transposed_on_rows, built during the
    warm-up) are part of the captured step -- the masks of a TextGCN graph are static (text2graph.py:180-191).
    `needed_rows_only`: the step reads the logits of the mask's rows only (flat_amazon.py:101), so the last layer computes
    only those (`GCN.forward(g, rows=mask)`; the restricted operator is built during the warm-up as well).
    """

    def __init__(self, model, g, optimizer, mask: Tensor, warmup: int = 3, needed_rows_only: bool = False):
        if not any(gr.get("capturable", False) for gr in optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs an optimizer created with capturable=True")
        self.model, self.g, self.opt, self.mask = model, g, optimizer, mask
        self._kw = {"rows": mask} if needed_rows_only else {}
        conv.enable_activation_reuse(False)          # a replay must not depend on Python-side caches
        _mask_count(mask)                            # the one host sync of the loss, done up front
        model.train()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = masked_cross_entropy(model(g, **self._kw), g.y, mask)
            self.loss.backward()
            optimizer.step()
        self.steps = max(1, warmup)                  # the capture itself does not execute the step

    def _eager_step(self):
        loss = masked_cross_entropy(self.model(self.g, **self._kw), self.g.y, self.mask)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self) -> Tensor:
        """Replay one optimisation step; returns the (static) loss tensor of that step."""
        self.graph.replay()
        self.steps += 1
        return self.loss


class GraphedEval:
    """`with no_grad: logits = model(g)` in eval mode as one graph replay; returns static logits.  `rows` (a bool mask
    the caller keeps unchanged): the rows of the logits that will be read (`GCN.forward(g, rows=...)`: validation and
    training rows, flat_amazon.py:109-114); every other row holds the last bias."""

    def __init__(self, model, g, rows: Tensor = None):
        self.model, self.g = model, g
        kw = {"rows": rows} if rows is not None else {}
        was_training = model.training
        model.eval()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            model(g, **kw)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = model(g, **kw)
        model.train(was_training)

    def __call__(self) -> Tensor:
        self.graph.replay()
        return self.logits


class FlatLoop:
    """The epoch of flat_amazon.py:99-117 (= flat_dbpedia.py:99-117) with every switch of this package that leaves the
    loop's numbers alone, behind one object -- what the reference writes as

        optimizer = th.optim.Adam(gcn.parameters(), lr=lr, amsgrad=True)           # :89
        for epoch in range(epochs):                                                # :99
            gcn.train(); loss = criterion(gcn(g)[g.train_mask], g.y[g.train_mask])
            optimizer.zero_grad(); loss.backward(); optimizer.step()
            gcn.eval(); logits = gcn(g); ... np.argmax(logits[g.val_mask].cpu().numpy(), axis=1) ...

    becomes

        loop = pytextgcn_amd.train.FlatLoop(gcn, g, lr=lr)
        for epoch in range(epochs):
            loss, val_loss, pred_val, pred_train = loop.epoch()                    # floats, numpy class ids
            f1_val = f1_score(y_val, pred_val, average="macro")                    # host work, as in the reference

    Inside: the fused masked cross-entropy and Adam(amsgrad) kernels, the dropout between the layers fused into the layer-2
    products, W1's update inside the backward SpMM and layer 1's activation kept from the evaluation pass for the next
    training pass (both bit for bit the plain kernels' results), the last layer computed on the rows that are read only
    (`GCN.forward(g, rows=...)`: fp32-equal on those rows), predictions taken on the device and shipped as the narrowest
    integer type behind ONE synchronisation per epoch.  At the benchmark size this is the 14 ms epoch where the import swap
    alone takes 36 (`bench.py`: `epoch_ms_fused_w1_reuse_needed_rows_only` against `epoch_ms`).
    The dropout mask comes from the library's own hash stream (seeded from torch's generator), and the package-wide switches
    `enable_fused_dropout` / `enable_activation_reuse` are ON while the loop lives (`close()`, or leaving its `with` block,
    restores them).  The masks of `g` must not change (text2graph.py:180-191: they are static)."""

    def __init__(self, model, g, lr: float, amsgrad: bool = True, weight_decay: float = 0.0, betas=(0.9, 0.999),
                 eps: float = 1e-8, needed_rows_only: bool = True):
        self.model, self.g = model, g
        self.optimizer = Adam(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        # the first layer's weight gradient on one-hot features IS the backward SpMM's result: spent on the update row
        # by row (other feature formats / widths take the ordinary road by themselves: `Adam._fused_update` declines)
        self.optimizer.fuse_into_backward(model.layers[0].weight)
        self._saved = (models._FUSED_DROPOUT, conv._REUSE)
        models.enable_fused_dropout(True)
        conv.enable_activation_reuse(True)
        self.rows_train = g.train_mask if needed_rows_only else None
        self.rows_eval = (g.val_mask | g.train_mask) if needed_rows_only else None
        n_classes = model.layers[-1].out_channels
        self._n_val = int(g.val_mask.sum().item())
        self._rows = torch.cat([g.val_mask.nonzero().flatten(), g.train_mask.nonzero().flatten()])
        dtype = torch.uint8 if n_classes <= 256 else torch.int16 if n_classes <= 32767 else torch.int32
        self._pred_host = torch.empty(self._rows.numel(), dtype=dtype).pin_memory()
        self._loss_host = torch.empty(2, dtype=torch.float32).pin_memory()
        self.epochs = 0

    def _fwd(self, rows):
        return self.model(self.g, rows=rows) if rows is not None else self.model(self.g)

    def epoch(self):
        """One training step + one evaluation pass.  Returns (training loss, validation loss, predicted classes of the
        validation rows, of the training rows) -- the numpy arrays are views of a pinned buffer that the next call
        overwrites."""
        model, g, opt = self.model, self.g, self.optimizer
        model.train()
        loss = masked_cross_entropy(self._fwd(self.rows_train), g.y, g.train_mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        model.eval()
        with torch.no_grad():
            val_loss, pred = masked_cross_entropy(self._fwd(self.rows_eval), g.y, g.val_mask, return_pred=True)
            self._pred_host.copy_(pred.index_select(0, self._rows).to(self._pred_host.dtype), non_blocking=True)
            self._loss_host.copy_(torch.stack([loss.detach(), val_loss]), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self.epochs += 1
        p = self._pred_host.numpy()
        return float(self._loss_host[0]), float(self._loss_host[1]), p[:self._n_val], p[self._n_val:]

    def test(self):
        """Predicted classes of the test rows (flat_amazon.py:130-131), as numpy."""
        self.model.eval()
        with torch.no_grad():
            return self.model(self.g, rows=self.g.test_mask)[self.g.test_mask].argmax(1).cpu().numpy()

    def close(self) -> None:
        if self._saved is not None:
            models.enable_fused_dropout(self._saved[0])
            conv.enable_activation_reuse(self._saved[1])
            self._saved = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
