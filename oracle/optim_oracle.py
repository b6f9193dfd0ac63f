"""CPU restatement of the reference's optimiser + schedule.  TEST INFRASTRUCTURE ONLY (same import
rule as the other oracle modules).

`transformers.AdamW` - used at /root/reference/CLIP/train.py:143 (`AdamW(model.parameters(), lr=lr,
no_deprecation_warning=True)`) and /root/reference/CLIP_prefix_caption/train.py:336 - was removed from
transformers >= 5, so the class cannot be imported here: its published update rule
(transformers 4.x optimization.py, AdamW.step; defaults betas=(0.9,0.999), eps=1e-6, weight_decay=0.0,
correct_bias=True) is restated below.  `get_linear_schedule_with_warmup` still exists
(HF:optimization.py:107-131) and pins linear_schedule().
"""
from __future__ import annotations

import math
from typing import Dict

import torch


class HFAdamW:
    def __init__(self, params: Dict[str, torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
                 correct_bias=True):
        self.params, self.lr, self.betas, self.eps, self.wd, self.correct_bias = params, lr, betas, eps, weight_decay, correct_bias
        self.state = {k: dict(step=0, exp_avg=torch.zeros_like(v), exp_avg_sq=torch.zeros_like(v)) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        b1, b2 = self.betas
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            st = self.state[k]
            st["step"] += 1
            st["exp_avg"].mul_(b1).add_(g, alpha=1.0 - b1)
            st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = st["exp_avg_sq"].sqrt().add_(self.eps)
            step_size = self.lr
            if self.correct_bias:
                step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
            p.addcdiv_(st["exp_avg"], denom, value=-step_size)
            if self.wd > 0.0:
                p.add_(p, alpha=-self.lr * self.wd)


def linear_schedule(step: int, warmup: int, total: int) -> float:
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))
