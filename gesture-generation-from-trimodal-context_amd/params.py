"""Flat parameter slabs.

All trainable parameters of one network live in ONE contiguous fp32 buffer (and their gradients, Adam moments in
three more of the same shape): the nn.Parameters of the drop-in modules are views into it.  That makes the optimiser
a single fused launch, zero_grad a single memset, and the data-parallel gradient exchange a handful of large
bucketed all-reduces instead of one per tensor.  state_dict()/load_state_dict() see ordinary named tensors.
"""
from collections import OrderedDict

import torch

from . import ops

ALIGN = 4   # floats: every tensor starts 16-byte aligned so vectorised kernels can take any parameter directly


class ParamSlab:
    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.names, self.params = [], []
        seen = set()
        named = []
        for name, p in module.named_parameters():      # named_parameters() already de-duplicates shared tensors
            if id(p) in seen:
                continue
            seen.add(id(p))
            named.append((name, p))
        # frozen parameters (requires_grad=False, e.g. freeze_wordembed: multimodal_context_net.py:40-41) go to the END of the slab:
        # the optimiser and the gradient exchange work on the trainable prefix [0, n_train) only, like optim.Adam(parameters())
        # which skips parameters without a gradient
        named.sort(key=lambda np_: not np_[1].requires_grad)          # stable: module order within each class
        self.names, self.params = [n for n, _ in named], [p for _, p in named]
        self.offsets, off = [], 0
        self.n_train = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if p.requires_grad:
                self.n_train = off
        self.numel = off
        self.frozen = {n for n, p in named if not p.requires_grad}
        self.flat = self.grad = self.m = self.v = None
        self.step = None
        self._ptr = None
        self.rebuild()

    def rebuild(self):
        """(Re)create the slabs on the parameters' current device and re-point the nn.Parameters into them."""
        dev = self.params[0].device
        flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        grad = torch.zeros_like(flat)
        old_m, old_v, old_step = self.m, self.v, self.step
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.detach().to(torch.float32))
                p.data = view
                p.grad = grad[off:off + p.numel()].view(p.shape) if p.requires_grad else None
        self.flat, self.grad = flat, grad
        self.m = torch.zeros_like(flat) if old_m is None else old_m.to(dev)
        self.v = torch.zeros_like(flat) if old_v is None else old_v.to(dev)
        self.step = torch.zeros((), device=dev, dtype=torch.int32) if old_step is None else old_step.to(dev)
        self._ptr = self.params[0].data_ptr()

    def ensure(self):
        """Module.to()/cuda() replaces .data and breaks the views: detect and rebuild."""
        p0 = self.params[0]
        if {n for n, p in zip(self.names, self.params) if not p.requires_grad} != self.frozen:
            raise RuntimeError("requires_grad of a parameter changed after the slab was laid out (the trainable prefix, Adam and the "
                               "gradient exchange are sized at construction): build the module with its final freeze settings")
        if p0.data_ptr() != self._ptr or p0.device != self.flat.device or any(
                p.requires_grad and (p.grad is None or p.grad.device != self.flat.device) for p in self.params[:2]):
            self.rebuild()
        return self

    def views(self):
        """name -> parameter view, name -> gradient view (aliases such as the TCN's net.0/net.4 excluded)."""
        P, G = OrderedDict(), OrderedDict()
        for n, p, off in zip(self.names, self.params, self.offsets):
            P[n] = self.flat[off:off + p.numel()].view(p.shape)
            G[n] = self.grad[off:off + p.numel()].view(p.shape)
        return P, G

    def zero_grad(self):
        if self.n_train:
            ops.zero_(self.grad[:self.n_train])
