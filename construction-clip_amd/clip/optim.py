"""Fused optimiser for models whose parameters live in a cclip_hip ParamArena.

`AdamW` reproduces `transformers.AdamW(lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
correct_bias=True)` - what the reference optimises with (/root/reference/CLIP/train.py:143,
/root/reference/CLIP_prefix_caption/train.py:336; the class no longer exists in transformers >= 5) -
as ONE launch over the flat arena that also rewrites the bf16 weight shadows.
`get_linear_schedule_with_warmup` restates the schedule of CLIP/train.py:145-147.
Any torch.optim optimiser also works on the model's parameters (they are ordinary nn.Parameters);
this one is simply ~400 launches cheaper per step.
"""
from __future__ import annotations

import torch

from cclip_hip import ops


class AdamW:
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0,
                 correct_bias: bool = True, torch_semantics: bool = False):
        self.model = model
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        self.param_groups = [dict(self.defaults, params=list(model.parameters()))]
        self.mode = 1 if torch_semantics else 0
        self.step_count = 0
        self._m = self._v = None

    def _state(self):
        arena = self.model.arena
        if self._m is None or self._m.numel() != arena.total or self._m.device != arena.flat.device:
            self._m = torch.zeros_like(arena.flat)
            self._v = torch.zeros_like(arena.flat)
        return arena

    def step(self, grad_scale: float = 1.0, pending=None) -> None:
        """pending: [(start, end, work)] from parallel.allreduce_gradients_async - the arena is then updated range by range,
        each as soon as its all-reduce has finished (the ranges must cover the arena; same result as one pass)."""
        arena = self._state()
        if not pending:
            arena.adopt_foreign_grads()      # (with pending reductions the reducer adopted them before it reduced)
        g = self.param_groups[0]
        self.step_count += 1
        # parameters that never received a gradient keep a zero slot -> no update beyond decay (as HF skips them).
        # (With pending reductions every rank takes the same branch: p.grad is None depends on the model, not on the data.)
        none = [n for n, p in arena.params.items() if p.grad is None]
        if none and pending:
            for _, _, w in pending:
                w.wait()
        for n in none:
            arena.g[n].zero_()
        kw = dict(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                  step=self.step_count, correct_bias=g["correct_bias"], grad_scale=grad_scale, mode=self.mode)
        if pending:
            assert pending[0][0] == 0 and pending[-1][1] == arena.total and all(a[1] == b[0] for a, b in zip(pending, pending[1:]))
            for s, e, w in pending:
                w.wait()
                ops.adamw_step(arena.flat[s:e], arena.gflat[s:e], self._m[s:e], self._v[s:e], bf16_shadow=arena.bflat[s:e], **kw)
        else:
            ops.adamw_step(arena.flat, arena.gflat, self._m, self._v, bf16_shadow=arena.bflat, **kw)
        arena.mark_shadows_fresh()            # the kernel rewrote masters and shadows together

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.model.parameters():
            p.grad = None

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self._m, exp_avg_sq=self._v, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self._m, self._v = sd["exp_avg"], sd["exp_avg_sq"]
        self.param_groups[0].update(sd["param_groups"][0])


class LinearWarmupSchedule:
    """get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps): lr * min(step/warmup,
    (total-step)/(total-warmup)) (CLIP/train.py:145-147)."""

    def __init__(self, optimizer: AdamW, num_warmup_steps: int, num_training_steps: int):
        self.optimizer, self.warmup, self.total = optimizer, num_warmup_steps, num_training_steps
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_step = 0
        self._apply()

    def _factor(self, step: int) -> float:
        if step < self.warmup:
            return float(step) / float(max(1, self.warmup))
        return max(0.0, float(self.total - step) / float(max(1, self.total - self.warmup)))

    def _apply(self):
        self.optimizer.param_groups[0]["lr"] = self.base_lr * self._factor(self.last_step)

    def step(self):
        self.last_step += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int):
    return LinearWarmupSchedule(optimizer, num_warmup_steps, num_training_steps)
