"""Adam with torch.optim.Adam's signature and arithmetic (the reference uses
`th.optim.Adam(gcn.parameters(), lr=lr, amsgrad=True)`, flat_amazon.py:89,106), whose step is ONE
fused HIP pass per parameter (libtgcn.so `tgcn_adam_step`: 36 bytes per element with amsgrad)
instead of torch's ~10 multi-tensor kernels.  W1 is N x h -- the largest tensor of the model
 -- so the optimizer is the largest cost of an epoch after the SpMMs."""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import _lib
from .plan import _stream_ptr


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 capturable=False):
        """`capturable=True` keeps the step counter on the device (as torch.optim.Adam's flag of the
        same name), so that `step()` can be captured in a HIP graph and replayed."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      amsgrad=amsgrad, capturable=capturable))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("pytextgcn_amd.optim.Adam handles contiguous float32 GPU "
                                       "parameters only (there is no CPU fallback)")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                cap = group.get("capturable", False)
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.int64, device=p.device) if cap else 0
                    if cap:
                        st["scalars"] = torch.zeros(2, dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                vmax = st.get("max_exp_avg_sq")
                if cap:
                    _lib.check(lib.tgcn_adam_step_capturable(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        vmax.data_ptr() if vmax is not None else None, p.numel(), group["lr"], b1, b2,
                        group["eps"], group["weight_decay"], st["step"].data_ptr(),
                        st["scalars"].data_ptr(), _stream_ptr(p.device)))
                else:
                    st["step"] += 1
                    _lib.check(lib.tgcn_adam_step(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        vmax.data_ptr() if vmax is not None else None, p.numel(), group["lr"], b1, b2,
                        group["eps"], group["weight_decay"], st["step"], _stream_ptr(p.device)))
                # the kernel wrote through raw pointers: tell autograd (and the activation cache of
                # pytextgcn_amd.conv) that the parameter changed, as an in-place torch op would
                torch.autograd.graph.increment_version(p)
        return loss
