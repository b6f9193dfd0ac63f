"""hipGraph capture for the launch-bound end of the path: small-batch inference.

`CLIP/predict.py` scores 1..16 images against 2 prompts, `parse_coco.py` / `application.py` encode ONE image at a time
(parse_coco.py:43, application.py:97).  At those sizes a tower forward is ~150 kernels of a few microseconds each and the
time goes to launching them.  `GraphedCallable` records one call of an inference function on fixed-shape inputs into a HIP
graph (torch.cuda.CUDAGraph = hipGraph on ROCm; every kernel here is launched on torch's current stream, so stream capture
sees them all) and afterwards replays it with one launch; a graph is kept per input shape.

Measured (MI355X, ViT-B/32, one image): 1.16 ms eager, 1.16 ms replayed - the chain of ~150 DEPENDENT kernels costs ~7 us per
kernel on the device itself, so the replay removes host work (useful when the host is the bottleneck, e.g. a busy Flask
worker) but not latency; shortening the chain needs fewer, larger kernels.

Not for training steps (the autograd nodes allocate and free per step, the autotuner and the shadow refresh are stateful).
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch


class GraphedCallable:
    def __init__(self, fn: Callable[..., torch.Tensor], warmup: int = 2):
        self.fn, self.warmup = fn, warmup
        self._graphs: Dict[Tuple, tuple] = {}

    def _key(self, args):
        return tuple((tuple(a.shape), a.dtype, a.device) for a in args)

    def __call__(self, *args: torch.Tensor) -> torch.Tensor:
        key = self._key(args)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [a.clone() for a in args]
            side = torch.cuda.Stream(device=args[0].device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():          # autotuner, attribute setting, allocator warm-up: outside capture
                for _ in range(self.warmup):
                    self.fn(*static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                static_out = self.fn(*static_in)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        for s, a in zip(static_in, args):
            s.copy_(a)
        graph.replay()
        return static_out.clone()


def graphed_encoders(model):
    """(encode_image, encode_text) replayed from HIP graphs; weights must not change in between (inference)."""
    return GraphedCallable(lambda x: model.encode_image(x)), GraphedCallable(lambda t: model.encode_text(t))
