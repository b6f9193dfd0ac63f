"""Data parallelism for the contrastive fine-tune: one process per GPU, `torch.distributed`
("nccl" = RCCL over xGMI on ROCm).  The reference has no distributed code at all;
the semantics are fixed by its single-GPU loss (CLIP/train_caption.py:124-129) evaluated on the
global batch - see clip/loss.py for the embedding all-gather / reduce-scatter.

Gradients: every parameter gradient lives in the model's flat arena, so the DP reduction is a few
large SUM all-reduces over contiguous fp32 ranges (per tower, so the first tower's reduction can
run on a side stream under the second tower's backward) instead of ~400 per-tensor collectives.
xGMI is point-to-point (7 links/GPU): large messages are what RCCL's direct algorithms want.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).
    Returns (rank, world, local_rank); no-op for world 1."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1:
        from cclip_hip import ops
        ops.sync_tuned_table()          # every rank runs rank 0's GEMM tile choices: same tiles, same summation order
    return rank, world, local


def collectives_active(group=None) -> bool:
    """True when the data-parallel collectives must run: a process group with more than one rank - or with ONE rank under
    CCLIP_DP_FORCE_COLLECTIVES=1, which sends every collective of the path through RCCL unchanged (identity results); that is
    how the one-GPU box executes the real `backend="nccl"` code path (tests/test_rccl_gpu.py)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("CCLIP_DP_FORCE_COLLECTIVES") == "1"


def grad_buckets(arena, max_bucket_elems: int = 64 << 20) -> List[Tuple[int, int]]:
    """Contiguous [start, end) element ranges of the flat gradient buffer, cut at parameter boundaries."""
    cuts = [0]
    for n in arena.names:
        off = arena.offsets[n]
        if off - cuts[-1] >= max_bucket_elems:
            cuts.append(off)
    cuts.append(arena.total)
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]


def allreduce_gradients(model, group=None, max_bucket_elems: int = 64 << 20, async_op: bool = False):
    """SUM the arena's gradient buffer over ranks, in large buckets.  (SUM, not mean: clip.loss already
    differentiates the global-mean loss.)  Returns the work handles when async_op."""
    if not collectives_active(group):
        return []
    arena = model.arena if hasattr(model, "arena") else model
    arena.adopt_foreign_grads()
    works = []
    for s, e in grad_buckets(arena, max_bucket_elems):
        works.append(dist.all_reduce(arena.gflat[s:e], group=group, async_op=async_op))
    return works if async_op else []


def allreduce_gradients_async(model, group=None, max_bucket_elems: int = 32 << 20):
    """The same bucketed SUM all-reduce, issued asynchronously: returns [(start, end, work)] in bucket order for
    `optim.AdamW.step(pending=...)`, which waits for bucket i, updates its range and lets RCCL reduce bucket i+1 meanwhile
    (the fused optimiser is one ~1 ms pass over the arena: most of it then runs under the collective).  World size 1: []."""
    if not collectives_active(group):
        return []
    arena = model.arena if hasattr(model, "arena") else model
    arena.adopt_foreign_grads()
    return [(s, e, dist.all_reduce(arena.gflat[s:e], group=group, async_op=True))
            for s, e in grad_buckets(arena, max_bucket_elems)]


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    if collectives_active(group):
        arena = model.arena if hasattr(model, "arena") else model
        dist.broadcast(arena.flat, src=src, group=group)
        arena.refresh_shadows(force=True)


class _WireCast:
    """work handle of a bucket reduced in a 16-bit wire type: wait() = wait for the collective, then widen back in place"""

    def __init__(self, work, dst: torch.Tensor, buf: torch.Tensor, stream):
        self.work, self.dst, self.buf, self.stream = work, dst, buf, stream

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.buf)          # dtype widening copy (plumbing), on the waiting stream


class GradReducer:
    """Gradient all-reduce OVERLAPPED with backward ("bucketed ... overlapped with backward").

    The arena is laid out in `named_parameters()` order, so the parameters of consecutive transformer blocks are contiguous.
    The hand-written backward reports every group of gradient slots whose last kernel has been enqueued
    (`ParamArena.notify_grads`: per block from `BlockStack.backward`, the tower's remaining tensors from `publish_grads`),
    together with the stream(s) those kernels run on.  A bucket (a contiguous arena range cut at parameter boundaries)
    is all-reduced as soon as all of its parameters are reported: the collective is issued on a communication stream that
    waits on events recorded on the producing streams - it starts when THAT bucket's gradients are final and runs under the
    backward of the blocks in front of it.  `finish()` reduces what is left (buckets holding a parameter that only the
    end of backward completes, e.g. logit_scale, whose gradient autograd itself delivers) and returns the
    [(start, end, work)] list `optim.AdamW.step(pending=...)` consumes bucket by bucket.

    Issue order is the host's program order - identical on every rank - as RCCL requires.  `wire_dtype=torch.bfloat16`
    sends 16-bit buckets (half the bytes: 303 MB instead of 605 MB for ViT-B/32); every gradient element is then rounded to
    8 significant bits before the sum (relative 2^-9 per element, on top of the 16-bit gradient stream's own rounding) -
    off by default, because the parity tests bound fp32 sums.
    """

    def __init__(self, model, group=None, max_bucket_elems: int = 16 << 20, wire_dtype: Optional[torch.dtype] = None):
        self.arena = model.arena if hasattr(model, "arena") else model
        self.group, self.wire_dtype = group, wire_dtype
        self.buckets = grad_buckets(self.arena, max_bucket_elems)
        ar = self.arena
        self._need = []                       # per bucket: parameter offsets it waits for
        for s, e in self.buckets:
            self._need.append({ar.offsets[n] for n in ar.names if s <= ar.offsets[n] < e and ar.params[n].requires_grad})
        self._bucket_of = {}
        for bi, need in enumerate(self._need):
            for off in need:
                self._bucket_of[off] = bi
        self._comm = None
        self._armed = False
        self.active = collectives_active(group)
        ar.grad_listener = self._on_grads
        self.fired_early = 0                  # buckets of the last step that were reduced from inside backward

    # -- one step -----------------------------------------------------------------------------------------------
    def begin(self) -> None:
        """arm for the coming backward (a gradient-accumulation micro-step that should NOT reduce simply does not arm)"""
        self._armed = self.active
        self._seen = [set() for _ in self.buckets]
        self._works = [None] * len(self.buckets)
        self._producers = []                  # every stream a gradient kernel of this backward has been reported on, in first-seen order
        self.waited = [None] * len(self.buckets)   # per bucket: the streams its all-reduce was ordered behind (tests read this)
        self.fired_early = 0

    def _on_grads(self, grad_views, streams) -> None:
        if not self._armed:
            return
        for st in streams:
            if not any(st is q or st == q for q in self._producers):
                self._producers.append(st)
        touched = set()
        for t in grad_views:
            bi = self._bucket_of.get(t.storage_offset())
            if bi is not None and self._works[bi] is None:
                self._seen[bi].add(t.storage_offset())
                touched.add(bi)
        for bi in sorted(touched):
            if self._works[bi] is None and self._seen[bi] == self._need[bi]:
                self._fire(bi, streams)
                self.fired_early += 1

    def _fire(self, bi: int, streams) -> None:
        """A bucket is cut at parameter boundaries only, so it can hold gradients written on DIFFERENT streams (the tail of the
        vision tower - s0 and its weight-gradient side stream - and the head of the text tower - s1 and its side stream - when
        both towers run backward side by side), while the notification that completes it names only the streams of its LAST
        group.  The collective is therefore ordered behind EVERY stream that has produced gradients since begin(), not just the
        notifying ones: an event recorded now on a stream covers everything enqueued on it so far, which includes this bucket's
        kernels (round 2 waited on the notifying streams only - correct by luck of the default layout)."""
        s, e = self.buckets[bi]
        view = self.arena.gflat[s:e]
        streams = list(self._producers) + [st for st in streams if not any(st is q or st == q for q in self._producers)]
        self.waited[bi] = tuple(streams)
        if not view.is_cuda:
            self._works[bi] = dist.all_reduce(view, group=self.group, async_op=True)
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=view.device)
        comm = self._comm
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            comm.wait_event(ev)
        with torch.cuda.stream(comm):
            if self.wire_dtype is not None:
                buf = view.to(self.wire_dtype)
                self._works[bi] = _WireCast(dist.all_reduce(buf, group=self.group, async_op=True), view, buf, comm)
            else:
                self._works[bi] = dist.all_reduce(view, group=self.group, async_op=True)

    def finish(self):
        """after backward(): reduce the remaining buckets; returns [(start, end, work)] in arena order ([] when inactive)"""
        if not self._armed:
            return []
        self._armed = False
        self.arena.adopt_foreign_grads()
        cur = [torch.cuda.current_stream()] if self.arena.gflat.is_cuda else []
        for bi in range(len(self.buckets)):
            if self._works[bi] is None:
                self._fire(bi, cur)
        return [(s, e, w) for (s, e), w in zip(self.buckets, self._works)]
