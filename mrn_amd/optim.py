"""Optimiser side of the training step on flat buffers (HIP kernels) + the host-side LR schedules.

Reference: Adam(lr) / SGD(momentum, weight decay) / Adadelta(rho, eps) + OneCycleLR(div_factor=20, final_div_factor=1000,
cos; momentum cycled for SGD) or the stepwise `adjust_learning_rate`, and clip_grad_norm_(params, 5)
(il_modules/base.py:72-114,255-269; il_modules/mrn.py:52-94,362-371; tools/utils.py:169-178).
"""
import math

import torch

from . import ops


class OneCycle:
    """torch.optim.lr_scheduler.OneCycleLR (two-phase cosine) as a plain host function of the step counter.  With
    cycle_momentum (the reference sets it for SGD, il_modules/base.py:90-93) the momentum runs the inverse cycle between
    max_momentum 0.95 and base_momentum 0.85 (torch defaults)."""

    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=20.0, final_div_factor=1000.0, cycle_momentum=False,
                 base_momentum=0.85, max_momentum=0.95):
        if total_steps <= 0:
            raise ValueError("total_steps must be positive")
        self.max_lr, self.total = max_lr, total_steps
        self.initial = max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.up_end = float(pct_start * total_steps) - 1
        self.down_end = float(total_steps - 1)
        self.cycle_momentum, self.base_momentum, self.max_momentum = cycle_momentum, base_momentum, max_momentum

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def _phase(self, step):
        if step >= self.total:
            raise ValueError(f"Tried to step {step + 1} times. The specified number of total steps is {self.total}")
        if step <= self.up_end:
            return 0, step / self.up_end
        return 1, (step - self.up_end) / (self.down_end - self.up_end)

    def lr_at(self, step):
        """learning rate used by the optimiser step number `step` (0-based)"""
        ph, pct = self._phase(step)
        if ph == 0:
            return self._cos(self.initial, self.max_lr, pct)
        return self._cos(self.max_lr, self.min_lr, pct)

    def momentum_at(self, step):
        """SGD momentum of optimiser step `step` when cycle_momentum is on, else None"""
        if not self.cycle_momentum:
            return None
        ph, pct = self._phase(step)
        if ph == 0:
            return self._cos(self.max_momentum, self.base_momentum, pct)
        return self._cos(self.base_momentum, self.max_momentum, pct)


class FlatOptimizer:
    """Optimiser state over one flat fp32 buffer.  The parameters' storage is moved into the buffer (each .data becomes a
    view), gradients accumulate into views of a second flat buffer, so clip + update are two kernels and the
    data-parallel all-reduce is a single collective on `self.grad`."""

    n_state = 0

    def __init__(self, params, lr):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError(f"{type(self).__name__} got no trainable parameters")
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]     # keep every view 16-byte aligned
        n = sum(sizes)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.state = [torch.zeros(n, device=dev, dtype=torch.float32) for _ in range(self.n_state)]
        self.offsets = []
        off = 0
        for p, sz in zip(self.params, sizes):
            view = self.flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            self.offsets.append(off)
            off += sz
        self.lr = lr
        self.step_count = 0
        self.param_groups = [{"lr": lr}]      # what the learners' logging and adjust_learning_rate() touch
        self.last_norm = None
        self.norm_state = torch.zeros(3, device=dev, dtype=torch.float32)      # [norm, clip coefficient, skipped steps] (ops.grad_norm_clip)

    def zero_grad(self):
        self.grad.zero_()

    def skipped_steps(self):
        """steps the update kernels skipped because the clipped gradient's norm was not finite (device counter: this call synchronises;
        the learners read it when they log, bench.py after its timed region).  The host's step_count -- bias correction, schedule --
        advances on such a step too."""
        return int(self.norm_state[2].item())

    def view_of(self, flat_like, i):
        """the slice of a flat-buffer-shaped tensor that belongs to parameter i, shaped like it"""
        p = self.params[i]
        return flat_like[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)

    def step(self, lr=None, max_norm=None, momentum=None):
        """clip_grad_norm_(max_norm) (skipped when None) followed by the update.  lr None: param_groups[0]["lr"], i.e.
        whatever adjust_learning_rate() last wrote there (the reference's stepwise schedule)"""
        if lr is None:
            lr = self.param_groups[0]["lr"]
        self.lr = lr
        self.param_groups[0]["lr"] = lr
        self.step_count += 1
        nc = ops.grad_norm_clip(self.grad, max_norm, out=self.norm_state) if max_norm is not None else None
        self.last_norm = nc
        self._update(nc, lr, momentum)
        # the kernels wrote the parameters through raw pointers: tell torch, so that every repacked-weight cache keyed on
        # (data_ptr, _version) (conv OHWI / HL32 packs, fragment-major recurrent weights) is rebuilt from the new values
        torch.autograd.graph.increment_version(self.params)
        return nc


class FlatAdam(FlatOptimizer):
    n_state = 2

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr)
        self.betas, self.eps = betas, eps
        self.m, self.v = self.state

    def _update(self, nc, lr, momentum):
        ops.adam_step(self.flat, self.grad, self.m, self.v, nc, lr, self.step_count, self.betas, self.eps)


class FlatSGD(FlatOptimizer):
    n_state = 1

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, lr)
        self.momentum, self.weight_decay = momentum, weight_decay

    def _update(self, nc, lr, momentum):
        mu = self.momentum if momentum is None else momentum
        ops.sgd_step(self.flat, self.grad, self.state[0], nc, lr, mu, self.weight_decay)


class FlatAdadelta(FlatOptimizer):
    n_state = 2

    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6):
        super().__init__(params, lr)
        self.rho, self.eps = rho, eps

    def _update(self, nc, lr, momentum):
        ops.adadelta_step(self.flat, self.grad, self.state[0], self.state[1], nc, lr, self.rho, self.eps)
