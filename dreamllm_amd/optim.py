"""AdamW on the fused HIP kernel + device-side global-norm clipping (no host sync in the step).

Mirrors what the reference trainer does around the hot path (omni/train/trainer.py:392-465 create_optimizer -> torch AdamW;
:799-807 clip_grad_norm_ -> optimizer.step): decoupled weight decay, bias-corrected moments, fp32 math.  Optimizer state
dtype defaults to the parameter dtype (bf16 params => bf16 moments, as in the reference recipe that casts the whole
model to bf16, projects/dreamllm/train.py:66-71,170); pass state_dtype=torch.float32 to keep fp32 moments (288 GB HBM
has the room: +27 GB for Vicuna-7B).
"""
from __future__ import annotations

import torch

from . import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, state_dtype=None, max_grad_norm=None,
                 multi_tensor=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.multi_tensor = multi_tensor   # False: one launch per tensor (round-3 path; A/B and tests)
        self.state_dtype = state_dtype
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None  # device scalar (fp32) of the last step, for logging without a sync

    def _clip_coef(self):
        """coef = min(1, max_norm / (||g||_2 + 1e-6)) as a device scalar (torch.nn.utils.clip_grad_norm_ semantics).  Round 4: the
        bf16, 16-byte-aligned gradients (every weight matrix) go through the multi-tensor kernel -- 7 launches instead of 295 for
        the 7B stage-II step; per-chunk partials in a fixed order, so replicas still derive identical norms."""
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not grads:
            return None
        multi = [g for g in grads if self.multi_tensor and ops._multi_ok(g)]
        rest = [g for g in grads if not (self.multi_tensor and ops._multi_ok(g))]
        pieces = []
        if multi:
            pieces.append(ops.sumsq_multi(multi))
        if rest:
            parts = torch.zeros(len(rest), ops.SUMSQ_PARTS, dtype=torch.float32, device=grads[0].device)
            for i, g in enumerate(rest):
                ops.sumsq_partials_(g.contiguous(), parts[i])
            pieces.append(parts.view(-1))
        allp = pieces[0] if len(pieces) == 1 else torch.cat(pieces)
        norm = ops.reduce_sum_f32(allp).sqrt()
        self.last_grad_norm = norm
        return torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        coef = self._clip_coef() if self.max_grad_norm is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            batches = {}   # step number -> ([p], [g], [m], [v]) of the tensors the multi-tensor kernel takes
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    sd = self.state_dtype or p.dtype
                    st["exp_avg"] = torch.zeros_like(p, dtype=sd, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=sd, memory_format=torch.preserve_format)
                    # per-parameter step, kept in `state` like torch.optim.AdamW does (a CPU scalar tensor): it is part of
                    # state_dict(), so the bias correction resumes where a checkpoint left off
                    st["step"] = torch.tensor(0.0)
                st["step"] += 1
                g = p.grad
                if (self.multi_tensor and g.dtype == p.dtype and ops._multi_ok(p) and ops._multi_ok(g) and ops._multi_ok(st["exp_avg"])
                        and ops._multi_ok(st["exp_avg_sq"])):
                    b = batches.setdefault(int(st["step"]), ([], [], [], []))
                    b[0].append(p); b[1].append(g); b[2].append(st["exp_avg"]); b[3].append(st["exp_avg_sq"])
                    continue
                ops.adamw_(p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                           group["weight_decay"], int(st["step"]), 1.0, coef)
            for stp, (ps, gs, ms, vs) in batches.items():
                ops.adamw_multi_(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], group["weight_decay"], stp, 1.0, coef)
        return loss
