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
    return rank, world, local


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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    arena = model.arena if hasattr(model, "arena") else model
    works = []
    for s, e in grad_buckets(arena, max_bucket_elems):
        works.append(dist.all_reduce(arena.gflat[s:e], group=group, async_op=async_op))
    return works if async_op else []


def allreduce_gradients_async(model, group=None, max_bucket_elems: int = 32 << 20):
    """The same bucketed SUM all-reduce, issued asynchronously: returns [(start, end, work)] in bucket order for
    `optim.AdamW.step(pending=...)`, which waits for bucket i, updates its range and lets RCCL reduce bucket i+1 meanwhile
    (the fused optimiser is one ~1 ms pass over the arena: most of it then runs under the collective).  World size 1: []."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    arena = model.arena if hasattr(model, "arena") else model
    return [(s, e, dist.all_reduce(arena.gflat[s:e], group=group, async_op=True))
            for s, e in grad_buckets(arena, max_bucket_elems)]


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        arena = model.arena if hasattr(model, "arena") else model
        dist.broadcast(arena.flat, src=src, group=group)
        arena.refresh_shadows(force=True)
