"""Flat parameter arena: every parameter of a module is a view into ONE fp32 device buffer, with
a parallel flat fp32 gradient buffer and a flat bf16 "shadow" (the MFMA operand copy).

Why (MI355X-first): the optimiser step, the bf16 re-cast and the data-parallel gradient
all-reduce each become a single launch / a handful of large RCCL collectives over contiguous
memory instead of ~400 per-tensor calls; 288 GB of HBM makes the three copies (1.5 GB for
ViT-B/32) irrelevant.  nn.Parameter objects stay ordinary parameters (state_dict / load_state_dict
/ torch optimisers keep working, as /root/reference/CLIP/train.py:111,143,214 needs).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

_ALIGN = 64  # elements; keeps every view 256-byte (fp32) / 128-byte (bf16) aligned


class ParamArena:
    def __init__(self, module: nn.Module, device: torch.device, shadow_dtype: torch.dtype = torch.bfloat16):
        named: List[Tuple[str, nn.Parameter]] = list(module.named_parameters())
        self.device = device
        self.names = [n for n, _ in named]
        self.offsets: Dict[str, int] = {}
        total = 0
        for n, p in named:
            self.offsets[n] = total
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = total
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.gflat = torch.zeros(total, device=device, dtype=torch.float32)
        self.shadow_dtype = shadow_dtype          # 16-bit MFMA operand type: bfloat16 (default) or float16
        self.bflat = torch.zeros(total, device=device, dtype=shadow_dtype)
        self.params: Dict[str, nn.Parameter] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for n, p in named:
                off, k = self.offsets[n], p.numel()
                view = self.flat[off:off + k].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                self.params[n] = p
                self.g[n] = self.gflat[off:off + k].view(p.shape)
                self.b[n] = self.bflat[off:off + k].view(p.shape)
        self._stamp = None
        self.shadow_gen = 0                      # bumped whenever the 16-bit shadows are rewritten (cast or fused optimiser)
        self.tflat: Optional[torch.Tensor] = None  # transposed 16-bit shadows of the registered matrices (same offsets)
        self.t: Dict[str, torch.Tensor] = {}
        self._t_table = None
        self._t_gen = -1
        self.accumulating: Dict[int, bool] = {}
        # set by clip.parallel.GradReducer: called as listener(grad_views, producer_streams) from inside the hand-written
        # backward whenever a group of gradient slots has had its last kernel ENQUEUED (not finished: the streams say where)
        self.grad_listener = None

    # ---- bf16 shadows ----
    def _current_stamp(self):
        return tuple(p._version for p in self.params.values())

    def refresh_shadows(self, force: bool = False) -> None:
        """Re-cast fp32 masters -> bf16 shadows (one launch) if any parameter was modified in place since
        the last cast (optimizer.step / load_state_dict bump the version counters)."""
        stamp = self._current_stamp()
        if force or stamp != self._stamp:
            ops.cast_f32_to_bf16(self.flat, self.bflat)
            self._stamp = stamp
            self.shadow_gen += 1

    def mark_shadows_fresh(self) -> None:
        self._stamp = self._current_stamp()
        self.shadow_gen += 1

    # ---- transposed shadows (dgrad operands) ----
    def register_transposed(self, names) -> None:
        """Keep a transposed 16-bit copy of these 2-D parameters ([out, in] -> [in, out]) in `self.t[name]`, rebuilt by ONE
        batched launch whenever the shadows changed (refresh_transposed).  The backward of an nn.Linear layer then reads its
        weight K-contiguously, like the forward (csrc/transpose16.hip)."""
        names = [n for n in names if self.params[n].dim() == 2]
        if not names:
            return
        if self.tflat is None:
            self.tflat = torch.zeros(self.total, device=self.device, dtype=self.shadow_dtype)
        rows, tiles = [], 0
        for n in names:
            r, c = self.params[n].shape
            off = self.offsets[n]
            self.t[n] = self.tflat[off:off + r * c].view(c, r)
            rows.append([off, off, r, c])
            tiles = max(tiles, ((r + 63) // 64) * ((c + 63) // 64))
        old = [] if self._t_table is None else self._t_table[0].tolist()
        table = torch.tensor(old + rows, dtype=torch.int64, device=self.device)
        self._t_table = (table, max(tiles, 0 if self._t_table is None else self._t_table[1]))
        self._t_gen = -1

    def refresh_transposed(self) -> None:
        if self._t_table is None:
            return
        self.refresh_shadows()
        if self._t_gen != self.shadow_gen:
            ops.transpose16_batched(self.bflat, self.tflat, self._t_table[0], self._t_table[1])
            self._t_gen = self.shadow_gen

    def intact(self) -> bool:
        """False if someone re-pointed a parameter away from the arena (module.to(), .half(), ...)."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * self.offsets[n] and p.dtype == torch.float32
                   for n, p in self.params.items())

    # ---- gradients ----
    def begin_backward(self) -> Dict[int, bool]:
        """For every parameter decide overwrite vs accumulate: p.grad None -> the kernels overwrite the
        arena slot; otherwise they add to it (a foreign .grad tensor is first copied into the slot)."""
        acc: Dict[int, bool] = {}
        for n, p in self.params.items():
            gv = self.g[n]
            if p.grad is None:
                acc[id(gv)] = False
            else:
                if p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
                    p.grad = gv
                acc[id(gv)] = True
        return acc

    def adopt_foreign_grads(self) -> int:
        """A parameter reached through ordinary autograd edges (CLIP's `logit_scale`: the loss node returns its gradient
        to autograd, which allocates `.grad` itself) has its gradient OUTSIDE the arena.  Everything that works on the flat
        gradient buffer - the fused optimiser, the data-parallel all-reduce - calls this first: such gradients are copied
        into their slots and `.grad` is re-pointed at the slot.  Returns the number adopted."""
        n_adopted = 0
        for n, p in self.params.items():
            gr = p.grad
            if gr is not None and gr.data_ptr() != self.g[n].data_ptr():
                self.g[n].copy_(gr)
                p.grad = self.g[n]
                n_adopted += 1
        return n_adopted

    # ---- static loss scale of the fp16 operand mode ----
    # The hand-written backward is linear in the upstream gradient.  With fp16 operands its 16-bit dY / dX stream carries
    # the 1/N of a mean loss unscaled, and at realistic N (1024 pairs; 256 x 40 caption tokens) most of it would be fp16
    # subnormals.  The backward therefore runs on S x the upstream gradient (S = LOSS_SCALE_FP16, a power of two: exact) and
    # the slots it wrote are multiplied by 1/S afterwards; slots that are being accumulated into are pre-multiplied by S.
    LOSS_SCALE_FP16 = 4096.0

    def loss_scale(self) -> float:
        return self.LOSS_SCALE_FP16 if self.shadow_dtype == torch.float16 else 1.0

    def slot_ranges(self, names):
        """maximal contiguous [start, end) element ranges of the flat buffers covering the named parameters' slots"""
        spans = sorted((self.offsets[n], self.offsets[n] + (self.params[n].numel() + _ALIGN - 1) // _ALIGN * _ALIGN) for n in names)
        out = []
        for s, e in spans:
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return [(s, e) for s, e in out]

    def scale_grads(self, names, alpha: float) -> None:
        if alpha == 1.0:
            return
        for s, e in self.slot_ranges(names):
            ops.scale_f32(self.gflat[s:e], alpha)

    def notify_grads(self, grad_views, streams=()) -> None:
        if self.grad_listener is not None:
            self.grad_listener(grad_views, streams)

    def publish_grads(self, names) -> None:
        """Point .grad of the parameters whose slots were just written at the arena views."""
        for n in names:
            p = self.params[n]
            if p.requires_grad and p.grad is None:
                p.grad = self.g[n]
        if self.grad_listener is not None:
            import torch as _t
            st = [_t.cuda.current_stream()] if self.gflat.is_cuda else []
            self.grad_listener([self.g[n] for n in names if self.params[n].requires_grad], st)
