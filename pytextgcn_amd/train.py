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

from . import conv
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
