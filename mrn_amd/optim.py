"""Optimiser side of the training step on flat buffers (HIP kernels) + the host-side LR schedule.

Reference: Adam(lr) + OneCycleLR(div_factor=20, final_div_factor=1000, cos) and clip_grad_norm_(params, 5)
(il_modules/base.py:72-114,255-262; il_modules/mrn.py:52-94,362-371).
"""
import math

import torch

from . import ops


class OneCycle:
    """torch.optim.lr_scheduler.OneCycleLR (two-phase cosine) as a plain host function of the step counter."""

    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=20.0, final_div_factor=1000.0):
        if total_steps <= 0:
            raise ValueError("total_steps must be positive")
        self.max_lr, self.total = max_lr, total_steps
        self.initial = max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.up_end = float(pct_start * total_steps) - 1
        self.down_end = float(total_steps - 1)

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def lr_at(self, step):
        """learning rate used by the optimiser step number `step` (0-based)"""
        if step >= self.total:
            raise ValueError(f"Tried to step {step + 1} times. The specified number of total steps is {self.total}")
        if step <= self.up_end:
            return self._cos(self.initial, self.max_lr, step / self.up_end)
        return self._cos(self.max_lr, self.min_lr, (step - self.up_end) / (self.down_end - self.up_end))


class FlatAdam:
    """Adam over one flat fp32 buffer.  The parameters' storage is moved into the buffer (each .data becomes a view),
    gradients accumulate into views of a second flat buffer, so clip + update are two kernels and the
    data-parallel all-reduce is a single collective on `self.grad`."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam got no trainable parameters")
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]     # keep every view 16-byte aligned
        n = sum(sizes)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p, sz in zip(self.params, sizes):
            view = self.flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            off += sz
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.param_groups = [{"lr": lr}]      # what the learners' logging reads
        self.last_norm = None

    def zero_grad(self):
        self.grad.zero_()

    def step(self, lr=None, max_norm=None):
        """clip_grad_norm_(max_norm) (skipped when None) followed by the Adam update"""
        if lr is not None:
            self.lr = lr
            self.param_groups[0]["lr"] = lr
        self.step_count += 1
        nc = ops.grad_norm_clip(self.grad, max_norm) if max_norm is not None else None
        self.last_norm = nc
        ops.adam_step(self.flat, self.grad, self.m, self.v, nc, self.lr, self.step_count, self.betas, self.eps)
        return nc
