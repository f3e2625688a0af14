"""Flat-buffer parameters and the fused Adam step (C ABI: lhg_adam_step).

All parameters of a module are re-bound as views of ONE contiguous fp32 buffer and their
gradients as views of a second one, so that (a) Adam is a single kernel launch over the whole
model instead of ~200 per-tensor launches (ref: torch.optim.Adam call sites watermelon.py:137-138)
and (b) data-parallel gradient reduction works on a handful of large contiguous buckets
(distributed.GradSynchronizer).  ``state_dict()`` / ``load_state_dict()`` keep working because
they copy through the views.
"""

from __future__ import annotations

import torch

from . import hip_ops


class FlatParams:
    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # 64-float alignment keeps every view 256-byte aligned
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.data = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.data[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + p.numel()].view_as(p)
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        """Keep .grad bound to the flat buffer (set_to_none would drop the views)."""
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def grad_view(self, i):
        o = self.offsets[i]
        return self.grad[o:o + self.params[i].numel()]


class FusedAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) semantics on a FlatParams buffer."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def step(self):
        self.step_count += 1
        hip_ops.adam_step_(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                           self.eps, self.step_count)
        for p in self.flat.params:
            hip_ops.bump_version(p)

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr, betas=self.betas, eps=self.eps)

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
