"""Adam with torch.optim.Adam's signature and arithmetic (the reference uses
`th.optim.Adam(gcn.parameters(), lr=lr, amsgrad=True)`, flat_amazon.py:89,106), whose step is ONE
fused HIP pass per parameter (libtgcn.so `tgcn_adam_step`: 36 bytes per element with amsgrad)
instead of torch's ~10 multi-tensor kernels.  W1 is N x h -- the largest tensor of the model
 -- so the optimizer is the largest cost of an epoch after the SpMMs."""
from __future__ import annotations

import weakref

import torch
from torch.optim import Optimizer

from . import _lib
from .plan import _stream_ptr


# parameter -> the optimizer that updates it inside the backward pass (kept OUTSIDE the tensor's __dict__: the
# reference pickles whole modules, flat_amazon.py:128, and a weak reference does not pickle)
# keyed by id() and validated through a weak reference to the tensor (tensors compare element-wise, so they
# cannot key a dict themselves; an id can be recycled once its tensor is gone)
_FUSED: dict = {}


def fused_optimizer_for(param):
    """The live optimizer registered for `param` by Adam.fuse_into_backward, or None."""
    hit = _FUSED.get(id(param))
    if hit is None or hit[0]() is not param:
        return None
    return hit[1]()


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 capturable=False):
        """`capturable=True` keeps the step counter on the device (as torch.optim.Adam's flag of the
        same name), so that `step()` can be captured in a HIP graph and replayed."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      amsgrad=amsgrad, capturable=capturable))

    def fuse_into_backward(self, param: torch.Tensor) -> None:
        """Opt in to updating `param` INSIDE the backward pass: for a GCNConv layer on one-hot features
        (TextGCN's first layer, text2graph.py:179) the weight gradient is the result of the transposed SpMM,
        dW1 = M^T dH1, and `tgcn_spmm_adam` spends each finished row on this optimizer's update of that row
        instead of storing it -- the N x h gradient and the optimizer's own pass over W1 disappear.  The
        parameter then never receives a `.grad`; `step()` skips it and updates the others as usual.  Same bits
        as backward + step.  Only for loops that call step() after every backward (flat_amazon.py:104-106): a
        second backward through the layer before step() (gradient accumulation, retain_graph, two uses of the layer
        in one loss) would apply the update twice, so it raises instead."""
        if not any(param is p for g in self.param_groups for p in g["params"]):
            raise ValueError("fuse_into_backward: the tensor is not one of this optimizer's parameters")
        key = id(param)
        _FUSED[key] = (weakref.ref(param, lambda _, k=key: _FUSED.pop(k, None)), weakref.ref(self))

    def _state_of(self, p, group):
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
        return st, cap

    @torch.no_grad()
    def _fused_update(self, p, plan, g, transpose: bool = True) -> bool:
        """Called from the backward of the propagate step whose operand is `p` (pytextgcn_amd.conv): apply this
        step's update of p with grad = M(^T) @ g.  False = not applicable here (caller takes the plain path)."""
        group = next((gr for gr in self.param_groups if any(p is q for q in gr["params"])), None)
        if group is None or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.dim() != 2 \
                or p.size(1) % 4 != 0 or p.size(1) <= 128 or g.stride(1) != 1 or g.stride(0) % 4 != 0:
            return False                      # (widths <= 128 run the sub-group SpMM kernels, which sum in another order)
        lib = _lib.load()
        st, cap = self._state_of(p, group)
        if st.get("fused_pending", False):
            raise RuntimeError(
                "pytextgcn_amd.optim.Adam: a parameter registered with fuse_into_backward() received a second backward "
                "before step() (gradient accumulation / retain_graph / the layer used twice in one loss): its update "
                "would be applied twice.  Call step() after every backward, or do not fuse this parameter.")
        st["fused_pending"] = True
        b1, b2 = group["betas"]
        vmax = st.get("max_exp_avg_sq")
        if cap:
            # advance the device-side step and its two factors (n = 0: no elementwise pass), then the fused update
            _lib.check(lib.tgcn_adam_step_capturable(
                p.data_ptr(), p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                vmax.data_ptr() if vmax is not None else None, 0, group["lr"], b1, b2, group["eps"],
                group["weight_decay"], st["step"].data_ptr(), st["scalars"].data_ptr(), _stream_ptr(p.device)))
            plan.spmm_adam(g, p, st["exp_avg"], st["exp_avg_sq"], vmax, group["lr"], b1, b2, group["eps"],
                           group["weight_decay"], 0, scalars=st["scalars"], transpose=transpose)
        else:
            st["step"] += 1
            plan.spmm_adam(g, p, st["exp_avg"], st["exp_avg_sq"], vmax, group["lr"], b1, b2, group["eps"],
                           group["weight_decay"], st["step"], transpose=transpose)
        torch.autograd.graph.increment_version(p)
        return True

    @torch.no_grad()
    def assert_no_pending_update(self, p) -> None:
        """Raises when `p` (registered with fuse_into_backward) already received its update of this step: a second backward
        before step() would apply it twice.  The 1-D partition calls this BEFORE the backward pass starts its first
        collective, so that a refusal never leaves peers inside one."""
        st = self.state.get(p)
        if st and st.get("fused_pending", False):
            raise RuntimeError(
                "pytextgcn_amd.optim.Adam: a parameter registered with fuse_into_backward() received a second backward "
                "before step(): its update would be applied twice.")

    def _fused_update_sharded(self, p, sg, g, hub_block=None, pending=None) -> bool:
        """The same for a rank's row shard of W1 in the 1-D partition (pytextgcn_amd.sharded._ShardedPropagate): the
        regular rows are updated inside the backward SpMM (split operand), the hub slice by one Adam pass once its
        gradient rows are reduced.  False = not applicable (caller takes the plain path)."""
        group = next((gr for gr in self.param_groups if any(p is q for q in gr["params"])), None)
        if group is None or group.get("capturable", False) or not p.is_cuda or p.dtype != torch.float32 \
                or not p.is_contiguous() or p.dim() != 2 or p.size(1) % 4 != 0 or p.size(1) <= 128 \
                or g.stride(1) != 1 or g.stride(0) % 4 != 0 or not hasattr(sg, "spmm_adam_w1") \
                or not hasattr(sg.dirs[0].B, "export_csr"):
            return False
        lib = _lib.load()
        st, _ = self._state_of(p, group)
        if st.get("fused_pending", False):
            raise RuntimeError(
                "pytextgcn_amd.optim.Adam: a parameter registered with fuse_into_backward() received a second backward "
                "before step(): its update would be applied twice.")
        st["fused_pending"] = True
        st["step"] += 1
        b1, b2 = group["betas"]
        vmax = st.get("max_exp_avg_sq")
        hyper = (group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"])

        def adam(rows, grad, op, g1, g2):
            pr, m, v = p[rows], st["exp_avg"][rows], st["exp_avg_sq"][rows]
            vm = vmax[rows] if vmax is not None else None
            if pr.numel() == 0:
                return
            if op is not None:                 # rows whose gradient is the operator's product: spent inside the launch
                op.spmm_adam(g1, pr, m, v, vm, *hyper, transpose=False, g2=g2)
            else:
                _lib.check(lib.tgcn_adam_step(pr.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
                                              vm.data_ptr() if vm is not None else None, pr.numel(), *hyper,
                                              _stream_ptr(p.device)))
        sg.spmm_adam_w1(g, adam, hub_block=hub_block, pending=pending)
        torch.autograd.graph.increment_version(p)
        return True

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
                    st_ = self.state.get(p)
                    if st_:
                        st_["fused_pending"] = False       # its update of this step happened in the backward pass
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("pytextgcn_amd.optim.Adam handles contiguous float32 GPU "
                                       "parameters only (there is no CPU fallback)")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st, cap = self._state_of(p, group)
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
