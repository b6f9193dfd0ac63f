"""Thin torch-tensor -> raw-pointer shims over the C ABI (include/cclip_hip.h).

torch is plumbing here: it owns device memory and the stream; every FLOP happens in
libcclip_hip.so.  All functions enqueue on torch's current stream and return immediately.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import lib, check

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

ACT_NONE, ACT_QUICKGELU, ACT_TANH, ACT_GELU_NEW, ACT_RELU = 0, 1, 2, 3, 4
ACT_DQUICKGELU, ACT_DTANH, ACT_DGELU_NEW, ACT_DRELU = 16, 17, 18, 19


# Optional per-launch timing of the dominant kernel family (bench.py's roofline leg): when GEMM_EVENTS is a
# list, every cclip_gemm_bf16 launch is bracketed by HIP events on the launch stream and
# (start, end, flops, layout) is appended.  None (default) = zero overhead.
GEMM_EVENTS = None


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")


HALF_TYPES = (torch.bfloat16, torch.float16)


def _req16(t: torch.Tensor, name: str):
    if t.dtype not in HALF_TYPES or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda bfloat16/float16, got {t.device} {t.dtype}")


def _fn(base: str, *tensors):
    """Pick the bf16 or the fp16 twin of an entry point from the dtype of its 16-bit tensors (which must agree)."""
    kinds = {t.dtype for t in tensors if t is not None and t.dtype in HALF_TYPES}
    if len(kinds) > 1:
        raise TypeError(f"{base}: mixed bfloat16 / float16 operands")
    f16 = kinds == {torch.float16}
    if base == "cclip_gemm_bf16":
        return lib.cclip_gemm_f16 if f16 else lib.cclip_gemm_bf16
    if base == "cclip_cast_f32_to_bf16":
        return lib.cclip_cast_f32_to_f16 if f16 else lib.cclip_cast_f32_to_bf16
    return getattr(lib, base + "_f16") if f16 else getattr(lib, base)


def _is16(t) -> int:
    return int(t is not None and t.dtype in HALF_TYPES)


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p),
        ("a_kcontig", c_int), ("b_kcontig", c_int),
        ("lda", c_long), ("ldb", c_long),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("alpha", c_float),
        ("bias", c_void_p),
        ("act", c_int),
        ("aux", c_void_p), ("ldaux", c_long),
        ("residual", c_void_p), ("ldr", c_long),
        ("out_f32", c_void_p), ("out_bf16", c_void_p), ("out_pre_bf16", c_void_p), ("ldc", c_long),
        ("split_k", c_int), ("split_ws", c_void_p),
        ("tile_config", c_int),
        ("colsum_out", c_void_p), ("colsum_accumulate", c_int), ("colsum_of_b", c_int),
        ("ln_stats", c_void_p), ("ln_c1", c_void_p), ("rowstats_out", c_void_p),
    ]


# Tile-configuration autotuner.  The GEMM tile configurations (128x128, 256x128, 256x256, persistent 256x128 with a
# streamed epilogue - forward layout / full tiles only) win on different
# shapes (tile-count quantisation over 256 CUs, K length, epilogue weight), and a model issues ~20 distinct GEMM
# shapes thousands of times.  The choice per (shape, layout, epilogue, split) key comes from a PERSISTED table:
#   * `gemm_tune.json` next to libcclip_hip.so (committed; refreshed by tools/tune_gemm.py on a GPU box), or the file named
#     by CCLIP_TUNE_FILE, is loaded at the first GEMM call.  It is keyed by a hash of the GEMM kernel sources: a table made
#     for other kernels is ignored.
#   * only a key the table does not hold is timed (trial launches on scratch outputs, GPU to itself), added to the table,
#     and - when CCLIP_TUNE_FILE is set or CCLIP_TUNE_SAVE=1 - written back, so the next process does not tune at all.
#   * under data parallelism rank 0's table is broadcast (`sync_tuned_table`), and a key missed later is decided by rank 0
#     for everyone: all ranks run the same tiles, hence the same summation order.
# With a complete table a run is launch-for-launch reproducible: no trial launches, no timing-dependent choices.
# AUTOTUNE=False pins configuration 1.
AUTOTUNE = True


class _GenDict(dict):
    """dict that counts its mutations (the per-family index of _nearest_tuned is rebuilt when the count moves)"""
    gen = 0

    def _bump(self):
        self.gen += 1

    def __setitem__(self, k, v):
        self._bump(); dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._bump(); dict.__delitem__(self, k)

    def clear(self):
        self._bump(); dict.clear(self)

    def update(self, *a, **kw):
        self._bump(); dict.update(self, *a, **kw)

    def pop(self, *a):
        self._bump(); return dict.pop(self, *a)


_TUNED = _GenDict()        # timed / persisted entries ONLY (choices derived by _nearest_tuned live in _DERIVED)
_TUNE_MIN_FLOPS = 2.0 * (1 << 29)
_TUNE_STATE = {"loaded": False, "source_hash": None, "misses": 0, "path": None}


def _tune_paths():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return os.environ.get("CCLIP_TUNE_FILE"), os.path.join(here, "gemm_tune.json")


def kernel_source_hash() -> str:
    """sha256 over the tile-GEMM kernel sources (csrc/gemm_bf16*.hip and the generated K loops gemm_a4*.inc, except the skinny GEMV path and the fp8 kernels, neither of
    which has tile configurations to choose from; gemm_bf16_impl.h, cclip_common.h): the key a tuned table is valid for."""
    import hashlib, os
    if _TUNE_STATE["source_hash"] is None:
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
        h = hashlib.sha256()
        for f in sorted(os.listdir(csrc)):
            if ((f.startswith("gemm_bf16") or f.startswith("gemm_a4")) and "skinny" not in f and "fp8" not in f) or f == "cclip_common.h":
                h.update(f.encode())
                with open(os.path.join(csrc, f), "rb") as fh:
                    h.update(fh.read())
        _TUNE_STATE["source_hash"] = h.hexdigest()[:16]
    return _TUNE_STATE["source_hash"]


import os as _os
_CFG_REMAP = {int(a): int(b) for a, b in (kv.split(":") for kv in _os.environ.get("CCLIP_GEMM_REMAP", "").split(",") if kv)}
class _LRU(dict):
    """bounded key -> choice cache of derived tile choices (insertion-ordered dict: oldest entry evicted first)"""
    cap = 1024

    def put(self, k, v):
        if k in self:
            dict.pop(self, k)
        elif len(self) >= self.cap:
            dict.pop(self, next(iter(self)))
        dict.__setitem__(self, k, v)

    def update(self, other=()):                     # accepts a dict or an iterable of keys (tests restore a saved key set)
        for k in other:
            self.put(k, other[k] if isinstance(other, dict) else None)


_DERIVED = _LRU()     # choices filled in by _nearest_tuned for keys the table does not hold: bounded, never persisted, never searched
_FAMILY = {"gen": -1, "idx": {}}


def _key_str(key) -> str:
    return "|".join(str(int(k)) if isinstance(k, bool) else str(k) for k in key)


def _family_of(f):
    """(family key, token field index, token count) of a split key: the family is the key with its TOKEN dimension blanked -
    rows of a forward-layout GEMM, the contraction length of a weight-gradient GEMM."""
    tok = 3 if (f[4] == "0" and f[5] == "0") else 1
    return "|".join(f[:tok] + ["*"] + f[tok + 1:]), tok, int(f[tok])


def _family_index():
    """family key -> (sorted token counts, choices in that order), built from the timed / persisted entries and rebuilt only
    when _TUNED changed.  A lookup is then one split of the QUERY key and a bisect: it does not grow with the number of keys a
    run has met (round 2 scanned - and string-split - every table entry per miss and stored derived entries in the same table,
    so a run with a new packed row count every step slowed down linearly with its length)."""
    if _FAMILY["gen"] != _TUNED.gen:
        fam = {}
        for k, v in _TUNED.items():
            fk, _, t = _family_of(k.split("|"))
            fam.setdefault(fk, []).append((t, v))
        _FAMILY["idx"] = {fk: ([t for t, _ in sorted(lst)], [v for _, v in sorted(lst)]) for fk, lst in fam.items()}
        _FAMILY["gen"] = _TUNED.gen
    return _FAMILY["idx"]


def _nearest_tuned(key: str, unbounded: bool = False):
    """A shape the table does not hold whose only difference from a tuned one is its TOKEN dimension (within 16x) - a last partial
    batch, a packed text batch whose row count follows the captions' lengths - takes that entry's tile configuration instead of
    being timed in the middle of the step: deterministic (no timing), no synchronisation.  A weight gradient's split-K count is
    scaled with the contraction length (same K-tiles per split).  `unbounded`: no distance limit (data parallelism: a rank must
    never time a shape on its own, see gemm_bf16).  CCLIP_TUNE_EXACT=1 (tools/tune_gemm.sh) disables it: every shape is timed."""
    import bisect, math, os
    if os.environ.get("CCLIP_TUNE_EXACT") == "1":
        return None
    fk, tok, want = _family_of(key.split("|"))
    ent = _family_index().get(fk)
    if ent is None:
        return None
    toks, vals = ent
    i = bisect.bisect_left(toks, want)
    best, best_d = None, (float("inf") if unbounded else math.log(16.0))
    for j in (i - 1, i):
        if 0 <= j < len(toks):
            d = abs(math.log(toks[j] / want))
            if d < best_d:
                best, best_d = j, d
    if best is None:
        return None
    cfg, sp = vals[best]
    cfg, order = cfg & 255, cfg & ~255      # (bits 8..15: the tile order's column-group width, kept)
    if cfg in (4, 8, 10) and tok == 1:
        kk = int(key.split("|")[3])
        if cfg == 4 and want % 256:
            cfg = 3           # the persistent streaming configuration covers full 256-row tiles only
        if cfg in (8, 10) and (kk % 64 or kk < 192):
            cfg = 3           # the hand-scheduled configurations need whole 64-deep K-tiles
    cfg |= order
    if tok == 3 and sp > 1:
        sp = max(1, min(sp, round(sp * want / toks[best])))
    if tok == 3 and cfg == 11 and (want % 64 or want // 64 < 2 * sp):
        cfg = 2               # the hand-scheduled weight-gradient kernel needs whole 64-token K-tiles, >= 2 per split
    return (cfg, sp)


def load_tuned_table(path=None, force: bool = False) -> int:
    """Load the persisted (key -> (tile_config, split_k)) table; returns the number of entries taken."""
    import json, os
    if _TUNE_STATE["loaded"] and not force and path is None:
        return 0
    _TUNE_STATE["loaded"] = True
    env, default = _tune_paths()
    n = 0
    for cand in ([path] if path else [default, env]):       # the env file is read last: its entries win
        if cand and os.path.isfile(cand):
            try:
                with open(cand) as f:
                    blob = json.load(f)
            except (OSError, ValueError):
                continue
            if blob.get("kernel_source_hash") != kernel_source_hash():
                continue                                    # made for other kernels: tune again
            for k, v in blob.get("table", {}).items():
                _TUNED[k] = (int(v[0]), int(v[1]))
                n += 1
            _TUNE_STATE["path"] = cand
    return n


def save_tuned_table(path=None) -> str:
    import json, os
    env, default = _tune_paths()
    path = path or env or default
    blob = {"kernel_source_hash": kernel_source_hash(),
            "note": "GEMM tile configuration per (dtype, M, N, K, layout, epilogue, split) key; written by cclip_hip.ops",
            "table": {k: list(v) for k, v in sorted(_TUNED.items())}}
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(blob, f, indent=0, sort_keys=True)
    os.replace(tmp, path)
    return path


def sync_tuned_table(group=None, src: int = 0) -> None:
    """Data parallelism: every rank adopts rank `src`'s table (call once after init_process_group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    load_tuned_table()
    box = [dict(_TUNED) if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    _TUNED.clear()
    _TUNED.update(box[0])


def _collectives_live() -> bool:
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _agree_on_choice(choice):
    """a key tuned mid-run under DP: rank 0's measurement decides for every rank (same call sequence on all ranks)"""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or os.environ.get("CCLIP_TUNE_DP_SYNC", "1") == "0":
        return choice
    box = [choice]
    dist.broadcast_object_list(box, src=0)
    return tuple(box[0])


def _launch_gemm(d) -> None:
    check(d._fn(ctypes.byref(d), _stream()), "cclip_gemm_bf16")


def _time_desc(d, reps: int = 3) -> float:
    """best of three timed groups after two warm-up launches: one noisy sample must not pin a slow configuration for the run"""
    _launch_gemm(d)
    _launch_gemm(d)
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _launch_gemm(d)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def _autotune(d, key, outs, candidates):
    """candidates: list of (tile_config, split_k).  Outputs are redirected to scratch so that in-place residual
    GEMMs (x += ...) are not applied more than once."""
    saved = (d.out_f32, d.out_bf16, d.out_pre_bf16, d.tile_config, d.split_k, d.split_ws)
    saved_cs = (d.colsum_out, d.colsum_accumulate)
    # the trials must have the GPU to themselves: with two tower streams the other tower's kernels are still running when
    # this one meets a new shape, and a candidate timed next to them can lose to a slower one timed alone
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
    cs_tmp = None
    if d.colsum_out:                        # trial launches must not touch (or accumulate into) the real bias gradient
        cs_tmp = torch.empty(max(d.M, d.N), device="cuda", dtype=torch.float32)
        d.colsum_out, d.colsum_accumulate = cs_tmp.data_ptr(), 0
    # scratch outputs with the SAME row stride as the real ones (outputs are often column slices of a wider slab)
    tmp = [torch.empty((d.M, d.ldc), device=t.device, dtype=t.dtype) if t is not None else None for t in outs]
    d.out_f32 = 0 if tmp[0] is None else tmp[0].data_ptr()
    d.out_bf16 = 0 if tmp[1] is None else tmp[1].data_ptr()
    d.out_pre_bf16 = 0 if tmp[2] is None else tmp[2].data_ptr()
    best, best_t = candidates[0], float("inf")
    ws = None
    for cfg, sp in candidates:
        d.tile_config, d.split_k = cfg, sp
        if sp > 1:
            need = sp * (d.M * d.N + max(d.M, d.N))
            if ws is None or ws.numel() < need:
                ws = torch.empty(need, device=outs[0].device if outs[0] is not None else "cuda", dtype=torch.float32)
            d.split_ws = ws.data_ptr()
        try:
            t = _time_desc(d)
        except Exception:          # a configuration that does not cover this shape / epilogue (status 1: nothing launched)
            continue
        if t < best_t:
            best, best_t = (cfg, sp), t
    d.out_f32, d.out_bf16, d.out_pre_bf16, d.tile_config, d.split_k, d.split_ws = saved
    d.colsum_out, d.colsum_accumulate = saved_cs
    best = _agree_on_choice(best)
    _TUNED[key] = best
    _TUNE_STATE["misses"] += 1
    import os
    if os.environ.get("CCLIP_TUNE_FILE") or os.environ.get("CCLIP_TUNE_SAVE") == "1":
        try:
            save_tuned_table()
        except OSError:
            pass
    return best


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, *, a_kcontig: bool = True, b_kcontig: bool = True,
              alpha: float = 1.0, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
              aux: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              out_f32: Optional[torch.Tensor] = None, out_bf16: Optional[torch.Tensor] = None,
              out_pre: Optional[torch.Tensor] = None, split_k: int = 1,
              split_ws: Optional[torch.Tensor] = None, M: Optional[int] = None, tile_config: int = 0,
              split_candidates=None, scratch=None, colsum_out: Optional[torch.Tensor] = None,
              colsum_accumulate: bool = False, colsum_of_b: bool = False, ln_stats: Optional[torch.Tensor] = None,
              ln_c1: Optional[torch.Tensor] = None, rowstats_out: Optional[torch.Tensor] = None) -> None:
    """C[m][n] = epi(alpha * sum_k A(m,k) B(n,k)); see cclip_gemm_bf16 in include/cclip_hip.h.
    A: [M,K] (a_kcontig) or [K,M]; B: [N,K] (b_kcontig) or [K,N]; 2-D, inner stride 1.
    split_candidates (wgrad): list of (tile_config, split_k) to autotune over; `scratch(n)` returns an fp32
    workspace of n floats for the chosen split.
    colsum_out (wgrad layout only): fp32 [M] (+)= sum_k A(m,k), or with colsum_of_b fp32 [N] (+)= sum_k B(n,k) - the bias
    gradient, fused into the weight-gradient GEMM.
    ln_stats [M, 2] + ln_c1 [N]: LayerNorm folded into this projection; rowstats_out [N/64, M, 2]: the residual form also emits
    the 16-bit copy (out_bf16) and row statistics for the next folded projection (include/cclip_hip.h; tile configuration 8)."""
    _req16(A, "A"); _req16(B, "B")
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    Mx, K = (A.shape[0], A.shape[1]) if a_kcontig else (A.shape[1], A.shape[0])
    N, Kb = (B.shape[0], B.shape[1]) if b_kcontig else (B.shape[1], B.shape[0])
    if M is None:
        M = Mx
    assert K == Kb, (A.shape, B.shape, a_kcontig, b_kcontig)
    outs3 = (out_f32, out_bf16, out_pre)
    outs = [t for t in outs3 if t is not None]
    assert outs, "no output"
    ldc = outs[0].stride(0)
    for t in outs:
        assert t.stride(0) == ldc and t.stride(1) == 1 and t.shape[0] >= M and t.shape[1] == N
    d = GemmDesc()
    d._fn = _fn("cclip_gemm_bf16", A, B, aux, out_bf16, out_pre)
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
    d.lda, d.ldb = A.stride(0), B.stride(0)
    d.M, d.N, d.K = M, N, K
    d.alpha = alpha
    d.bias = 0 if bias is None else bias.data_ptr()
    d.act = act
    d.aux = 0 if aux is None else aux.data_ptr()
    d.ldaux = 0 if aux is None else aux.stride(0)
    d.residual = 0 if residual is None else residual.data_ptr()
    d.ldr = 0 if residual is None else residual.stride(0)
    d.out_f32 = 0 if out_f32 is None else out_f32.data_ptr()
    d.out_bf16 = 0 if out_bf16 is None else out_bf16.data_ptr()
    d.out_pre_bf16 = 0 if out_pre is None else out_pre.data_ptr()
    d.ldc = ldc
    d.split_k = split_k
    d.split_ws = 0 if split_ws is None else split_ws.data_ptr()
    d.tile_config = tile_config
    if colsum_out is not None:
        _req(colsum_out, torch.float32, "colsum_out")
        assert not a_kcontig and not b_kcontig and colsum_out.numel() == (N if colsum_of_b else M) and colsum_out.is_contiguous()
        d.colsum_out, d.colsum_accumulate, d.colsum_of_b = colsum_out.data_ptr(), int(colsum_accumulate), int(colsum_of_b)
    if ln_stats is not None or rowstats_out is not None:
        for t_, n_ in ((ln_stats, "ln_stats"), (ln_c1, "ln_c1"), (rowstats_out, "rowstats_out")):
            if t_ is not None:
                _req(t_, torch.float32, n_)
                assert t_.is_contiguous()
        d.ln_stats = 0 if ln_stats is None else ln_stats.data_ptr()
        d.ln_c1 = 0 if ln_c1 is None else ln_c1.data_ptr()
        d.rowstats_out = 0 if rowstats_out is None else rowstats_out.data_ptr()
        d.tile_config = tile_config = 8           # the only configuration with these epilogue forms
    if bias is not None:
        _req(bias, torch.float32, "bias")
    if residual is not None:
        _req(residual, torch.float32, "residual")
    if out_f32 is not None:
        _req(out_f32, torch.float32, "out_f32")
    if tile_config == 0 and AUTOTUNE and 2.0 * M * N * K >= _TUNE_MIN_FLOPS and (outs[0].is_contiguous() or True):
        if not _TUNE_STATE["loaded"]:
            load_tuned_table()
        key = _key_str((str(A.dtype).replace("torch.", ""), M, N, K, a_kcontig, b_kcontig, act, out_f32 is not None,
                        out_bf16 is not None, out_pre is not None, residual is not None, bias is not None,
                        split_k if split_candidates is None else -1, colsum_out is not None, colsum_of_b))
        choice = _TUNED.get(key)
        if choice is None:
            choice = _DERIVED.get(key)
        if choice is None:
            dp = _collectives_live()
            choice = _nearest_tuned(key, unbounded=dp)
            if choice is None and dp:
                # data parallelism: a shape only THIS rank meets (its packed row count) must not be timed here - the trial
                # launches would put a broadcast into one rank's call sequence only.  Deterministic fallback instead.
                choice = (3 if (a_kcontig and b_kcontig) else 1, split_k if split_candidates is None else split_candidates[0][1])
            if choice is not None:
                _DERIVED.put(key, choice)                      # bounded cache; never persisted, never searched
                _TUNE_STATE["derived"] = _TUNE_STATE.get("derived", 0) + 1
        if choice is None:
            cands = split_candidates if split_candidates is not None else [(c, split_k) for c in (1, 2, 3, 4, 5, 7, 8)]
            if split_candidates is None and a_kcontig and b_kcontig and split_k == 1:
                # the 256-wide configurations again with the column tiles taken in groups of g (tile_config bits 8..15): which
                # weight panels one XCD's L2 holds together - decided by timing, like the tile itself
                cands += [(c + 256 * g, split_k) for c in (3, 8) for g in (3, 4, 6) if (N + 255) // 256 > g]
            if split_candidates is None and split_k > 1:
                d.split_ws = split_ws.data_ptr()
            choice = _autotune(d, key, outs3, cands)
        if _CFG_REMAP:                                         # measurement aid (CCLIP_GEMM_REMAP="8:3,10:3"): A/B a table entry against another configuration
            choice = (_CFG_REMAP.get(choice[0], choice[0]), choice[1])
        d.tile_config, d.split_k = choice
        if d.split_k > 1 and split_candidates is not None:
            ws = scratch(d.split_k * (M * N + max(M, N)))
            d.split_ws = ws.data_ptr()
    if d.split_k > 1:
        assert d.split_ws, "split_k > 1 needs a workspace"
    if GEMM_EVENTS is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _launch_gemm(d)
        e1.record()
        GEMM_EVENTS.append((e0, e1, 2.0 * M * N * K, (int(a_kcontig), int(b_kcontig)), (M, N, K)))
        return
    _launch_gemm(d)


def rowstats_combine(partials: torch.Tensor, stats: torch.Tensor, *, rows: int, D: int, eps: float = 1e-5) -> None:
    """stats[m] = (mean, rstd) from the [nblk, rows, 2] (sum, sum of squares) partials a residual GEMM emitted (rowstats_out)"""
    _req(partials, torch.float32, "partials"); _req(stats, torch.float32, "stats")
    check(lib.cclip_rowstats_combine(_p(partials), c_int(partials.shape[0]), c_int(rows), c_int(D), c_float(eps), _p(stats), _stream()),
          "cclip_rowstats_combine")


# --------------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------------
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, rows: int, row_index=None,
                  out_bf16=None, out_f32=None, mean=None, rstd=None, eps: float = 1e-5) -> None:
    _req(x, torch.float32, "x")
    D = x.shape[-1]
    out = out_bf16 if out_bf16 is not None else out_f32
    check(_fn("cclip_layernorm_fwd", out_bf16)(_p(x), c_long(x.stride(-2)), _p(row_index), c_int(rows), c_int(D), _p(gamma), _p(beta),
                                  c_float(eps), _p(out_bf16), _p(out_f32), c_long(out.stride(-2)), _p(mean), _p(rstd),
                                  _stream()), "cclip_layernorm_fwd")


def layernorm_bwd_ws_floats(rows: int, D: int) -> int:
    return lib.cclip_layernorm_bwd_ws_floats(c_int(rows), c_int(D))


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, *,
                  rows: int, row_index=None, dx_res=None, dx_out=None, dx_out_bf16=None, dgamma=None, dbeta=None,
                  accumulate: bool = False, ws=None) -> None:
    D = x.shape[-1]
    dxo = dx_out if dx_out is not None else dx_out_bf16
    lddx = dxo.stride(-2) if dxo is not None else x.stride(-2)
    if dx_res is not None:
        assert dx_res.stride(-2) == lddx
    if dx_out is not None and dx_out_bf16 is not None:
        assert dx_out.stride(-2) == dx_out_bf16.stride(-2)
    check(_fn("cclip_layernorm_bwd", dy, dx_out_bf16)(_p(dy), c_int(_is16(dy)), c_long(dy.stride(-2)), _p(x),
                                  c_long(x.stride(-2)), _p(row_index), c_int(rows), c_int(D), _p(gamma), _p(mean),
                                  _p(rstd), _p(dx_res), _p(dx_out), _p(dx_out_bf16), c_long(lddx), _p(dgamma),
                                  _p(dbeta), c_int(int(accumulate)), _p(ws), _stream()), "cclip_layernorm_bwd")


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p),
        ("ldq", c_long), ("ldk", c_long), ("ldv", c_long),
        ("o", c_void_p), ("ldo", c_long),
        ("lse", c_void_p), ("key_keep", c_void_p),
        ("B", c_int), ("T", c_int), ("H", c_int), ("head_dim", c_int), ("causal", c_int),
        ("scale", c_float),
        ("dout", c_void_p), ("lddo", c_long),
        ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
        ("lddq", c_long), ("lddk", c_long), ("lddv", c_long),
        ("o_fp8", c_void_p), ("ldo_fp8", c_long), ("o_block_scale", c_void_p), ("ld_o_block_scale", c_long),
        ("cu_seqlens", c_void_p),
    ]


def _attn_desc(q, k, v, o, lse, B, T, H, causal, key_keep, scale, head_dim=64):
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        _req16(t, n)
        assert t.stride(-1) == 1
    d = AttnDesc()
    d.q, d.k, d.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    d.ldq, d.ldk, d.ldv = q.stride(-2), k.stride(-2), v.stride(-2)
    d.o, d.ldo = o.data_ptr(), o.stride(-2)
    d.lse = 0 if lse is None else lse.data_ptr()
    d.key_keep = 0 if key_keep is None else key_keep.data_ptr()
    d.B, d.T, d.H, d.head_dim, d.causal = B, T, H, head_dim, int(causal)
    d.scale = head_dim ** -0.5 if scale is None else scale
    return d


def _attn_cu(d, cu, B):
    if cu is not None:
        assert cu.dtype == torch.int32 and cu.is_cuda and cu.is_contiguous() and cu.numel() == B + 1
        d.cu_seqlens = cu.data_ptr()


def attention_fwd(q, k, v, o, *, B: int, T: int, H: int, causal: bool = False, key_keep=None, lse=None, scale=None, out_mx=None,
                  cu=None) -> None:
    """q/k/v/o: bf16 2-D views [B*T, >= H*64] (any row stride, inner stride 1); head h at columns h*64..
    out_mx = (o8 [B*T, H*64] uint8, block_scale [H*64/128, B*T, 4] uint8): the output as e4m3 + E8M0 block scales (the out-proj's
    block-scaled fp8 operand, see gemm_fp8) instead of 16-bit; `o` is then only the dtype witness and is not written.
    cu (int32 [B+1], T <= 128): packed batch - sequence b is rows [cu[b], cu[b+1]), T = the longest length."""
    d = _attn_desc(q, k, v, o, lse, B, T, H, causal, key_keep, scale)
    _attn_cu(d, cu, B)
    if out_mx is not None:
        o8, omx = out_mx
        R = o.shape[0] if cu is not None else B * T            # packed batch: the caller's row count
        assert o8.dtype == torch.uint8 and o8.stride(1) == 1 and o8.shape[0] >= R and o8.shape[1] >= H * 64 and H % 2 == 0
        _req_mx(omx, R, H * 64, "out_mx[1]")
        d.o_fp8, d.ldo_fp8, d.o_block_scale, d.ld_o_block_scale = o8.data_ptr(), o8.stride(0), omx.data_ptr(), omx.stride(0)
    check(_fn("cclip_attention_fwd", q, k, v, o)(ctypes.byref(d), _stream()), "cclip_attention_fwd")


def attention_bwd(q, k, v, o, lse, dout, dq, dk, dv, *, B: int, T: int, H: int, causal: bool = False, key_keep=None,
                  scale=None, cu=None) -> None:
    d = _attn_desc(q, k, v, o, lse, B, T, H, causal, key_keep, scale)
    _attn_cu(d, cu, B)
    d.dout, d.lddo = dout.data_ptr(), dout.stride(-2)
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.lddq, d.lddk, d.lddv = dq.stride(-2), dk.stride(-2), dv.stride(-2)
    check(_fn("cclip_attention_bwd", q, k, v, o, dout, dq, dk, dv)(ctypes.byref(d), _stream()), "cclip_attention_bwd")


def attention_small_fwd(q, k, v, o, *, B: int, T: int, H: int, head_dim: int, lse=None, scale=None) -> None:
    """Generic-head_dim unmasked attention (TransformerMapper): same tensor conventions as attention_fwd."""
    d = _attn_desc(q, k, v, o, lse, B, T, H, False, None, scale, head_dim)
    check(_fn("cclip_attention_small_fwd", q, k, v, o)(ctypes.byref(d), _stream()), "cclip_attention_small_fwd")


def attention_small_bwd(q, k, v, o, lse, dout, dq, dk, dv, *, B: int, T: int, H: int, head_dim: int, scale=None) -> None:
    d = _attn_desc(q, k, v, o, lse, B, T, H, False, None, scale, head_dim)
    d.dout, d.lddo = dout.data_ptr(), dout.stride(-2)
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.lddq, d.lddk, d.lddv = dq.stride(-2), dk.stride(-2), dv.stride(-2)
    check(_fn("cclip_attention_small_bwd", q, k, v, o, dout, dq, dk, dv)(ctypes.byref(d), _stream()), "cclip_attention_small_bwd")


def attention_decode(q, kcache, vcache, out, *, H: int, S: int, scale=None) -> None:
    """One new query per sequence against its KV cache.  q/out: [B, H*64] 16-bit rows (any row stride); kcache/vcache:
    [B, Smax, H*64] views (position stride = stride(1), sequence stride = stride(0)); S = cached positions incl. the new one."""
    for t, n in ((q, "q"), (kcache, "kcache"), (vcache, "vcache"), (out, "out")):
        _req16(t, n)
        assert t.stride(-1) == 1
    B = q.shape[0]
    assert kcache.dim() == 3 and vcache.shape == kcache.shape and kcache.stride() == vcache.stride() and S <= kcache.shape[1]
    check(_fn("cclip_attention_decode", q, kcache, vcache, out)(
        _p(q), c_long(q.stride(0)), _p(kcache), _p(vcache), c_long(kcache.stride(1)), c_long(kcache.stride(0)), _p(out),
        c_long(out.stride(0)), c_int(B), c_int(H), c_int(S), c_float(64 ** -0.5 if scale is None else scale), _stream()),
        "cclip_attention_decode")


# --------------------------------------------------------------------------------------------
# device-side preprocess (PIL's 8-bit bicubic resampler + crop + normalise)
# --------------------------------------------------------------------------------------------
def resample_h_u8(src_rows: torch.Tensor, bounds: torch.Tensor, kk: torch.Tensor, ksize: int, out: torch.Tensor) -> None:
    """src_rows uint8 [rows, W, 3] -> out uint8 [rows, n, 3]; bounds int32 [n, 2], kk int32 [n, ksize]."""
    assert src_rows.dtype == torch.uint8 and out.dtype == torch.uint8 and src_rows.is_cuda and src_rows.stride(2) == 1 and src_rows.stride(1) == 3
    assert bounds.dtype == torch.int32 and kk.dtype == torch.int32 and bounds.is_contiguous() and kk.is_contiguous() and out.is_contiguous()
    check(lib.cclip_resample_h_u8(_p(src_rows), c_long(src_rows.stride(0)), c_int(src_rows.shape[0]), _p(bounds), _p(kk), c_int(ksize),
                                  c_int(out.shape[1]), _p(out), c_long(out.stride(0)), _stream()), "cclip_resample_h_u8")


def resample_v_norm(tmp: torch.Tensor, row0: int, bounds: torch.Tensor, kk: torch.Tensor, ksize: int, mean, std, out: torch.Tensor) -> None:
    """tmp uint8 [rows, n, 3] (input rows row0..) -> out fp32 [3, n, n] = ((u8 / 255) - mean) / std."""
    _req(out, torch.float32, "out")
    assert tmp.dtype == torch.uint8 and tmp.is_contiguous() and out.is_contiguous()
    m3, s3 = (c_float * 3)(*mean), (c_float * 3)(*std)
    check(lib.cclip_resample_v_norm(_p(tmp), c_long(tmp.stride(0)), c_int(row0), _p(bounds), _p(kk), c_int(ksize), c_int(out.shape[1]),
                                    m3, s3, _p(out), _stream()), "cclip_resample_v_norm")


# --------------------------------------------------------------------------------------------
# fp8 (e4m3) inference projections
# --------------------------------------------------------------------------------------------
def quantize_rows_fp8(x16: torch.Tensor, out8: torch.Tensor, scale: torch.Tensor) -> None:
    """x16 [R, C] 16-bit -> out8 [R, C] uint8 (e4m3 bytes), scale [R] fp32 (x ~= scale[r] * fp8)."""
    _req16(x16, "x16"); _req(scale, torch.float32, "scale")
    assert out8.dtype == torch.uint8 and x16.dim() == 2 and out8.shape == x16.shape and x16.stride(1) == 1 and out8.stride(1) == 1
    R, C = x16.shape
    check(_fn("cclip_quantize_rows_fp8", x16)(_p(x16), c_long(x16.stride(0)), c_int(R), c_int(C), _p(out8), c_long(out8.stride(0)),
                                             _p(scale), _stream()), "cclip_quantize_rows_fp8")


def layernorm_fwd_fp8(x: torch.Tensor, gamma, beta, out8: torch.Tensor, scale: torch.Tensor, *, rows: int, eps: float = 1e-5) -> None:
    _req(x, torch.float32, "x"); _req(scale, torch.float32, "scale")
    assert out8.dtype == torch.uint8 and out8.stride(-1) == 1
    check(lib.cclip_layernorm_fwd_fp8(_p(x), c_long(x.stride(-2)), c_int(rows), c_int(x.shape[-1]), _p(gamma), _p(beta), c_float(eps),
                                      _p(out8), c_long(out8.stride(-2)), _p(scale), _stream()), "cclip_layernorm_fwd_fp8")


def quantize_mx_fp8(x16: torch.Tensor, out8: torch.Tensor, block_scale: torch.Tensor, *, rows: Optional[int] = None) -> None:
    """x16 [R, C] 16-bit (C % 32 == 0) -> out8 [R, C] uint8 (e4m3 bytes) + block_scale [ceil(C/128), R, 4] uint8 (E8M0): every
    32 consecutive columns of a row share the power-of-two scale 2^(e - 127) (the block-scaled MFMA's operand format); the
    scale of (row r, block b) is block_scale[b >> 2, r, b & 3] (K-tile major: what the consuming GEMM reads per K-tile is
    contiguous)."""
    _req16(x16, "x16")
    assert out8.dtype == torch.uint8 and x16.dim() == 2 and out8.shape[1] == x16.shape[1] and x16.stride(1) == 1 and out8.stride(1) == 1
    R, C = x16.shape
    R = R if rows is None else rows
    _req_mx(block_scale, R, C, "block_scale")
    check(_fn("cclip_quantize_mx_fp8", x16)(_p(x16), c_long(x16.stride(0)), c_int(R), c_int(C), _p(out8), c_long(out8.stride(0)),
                                           _p(block_scale), c_long(block_scale.stride(0)), _stream()), "cclip_quantize_mx_fp8")


def _req_mx(t: torch.Tensor, rows: int, cols: int, name: str):
    if not (t.dtype == torch.uint8 and t.is_cuda and t.dim() == 3 and t.shape[0] * 128 >= cols and t.shape[1] >= rows and t.shape[2] == 4
            and t.stride(2) == 1 and t.stride(1) == 4):
        raise TypeError(f"{name}: expected a cuda uint8 [>= {(cols + 127) // 128}, >= {rows}, 4] block-scale tensor, got {tuple(t.shape)} {t.dtype}")


def mx_scale_buffer(rows: int, cols: int, device) -> torch.Tensor:
    """Block-scale tensor for a [rows, cols] block-scaled e4m3 operand: uint8 [ceil(cols/128), rows, 4]."""
    return torch.empty((cols + 127) // 128, rows, 4, device=device, dtype=torch.uint8)


class Fp8GemmDesc(ctypes.Structure):
    """include/cclip_hip.h: cclip_fp8_gemm_desc"""
    _fields_ = [("A", c_void_p), ("lda", c_long), ("scale_a", c_void_p), ("block_scale_a", c_void_p), ("ld_block_scale_a", c_long),
                ("B", c_void_p), ("ldb", c_long), ("scale_b", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int),
                ("bias", c_void_p), ("act", c_int), ("out16", c_void_p), ("ldc", c_long),
                ("out_fp8", c_void_p), ("ld_out_fp8", c_long), ("out_block_scale", c_void_p), ("ld_out_block_scale", c_long),
                ("out_f32", c_void_p), ("residual", c_void_p), ("ldf", c_long)]


def gemm_fp8(A8, scale_a, B8, scale_b, out16=None, *, bias=None, act: int = ACT_NONE, M: Optional[int] = None,
             block_scale_a=None, out_mx=None, out_f32=None, residual=None, half=None) -> None:
    """act(sa[m] * sb[n] * sum_k A8[m][k] B8[n][k] + bias[n]); A8 [M,K], B8 [N,K] uint8 e4m3, sb per output channel.
    A's scale: scale_a [M] fp32 per row, or block_scale_a [K/128, M, 4] uint8 E8M0 per 32-deep k block (applied by the MFMA).
    Output (exactly one): out16 (16-bit) | out_mx = (out8 [M,N] uint8, block_scale [N/128, M, 4] uint8): e4m3 + E8M0 per 32 columns,
    the next GEMM's block-scaled A operand | out_f32 (+ residual, fp32, may alias): the residual stream.
    `half`: torch.bfloat16 / torch.float16 picks the library twin when no 16-bit tensor is among the arguments."""
    assert A8.dtype == torch.uint8 and B8.dtype == torch.uint8 and A8.stride(1) == 1 and B8.stride(1) == 1
    Mx, K = A8.shape
    N = B8.shape[0]
    M = Mx if M is None else M
    assert B8.shape[1] == K
    d = Fp8GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.scale_b = A8.data_ptr(), A8.stride(0), B8.data_ptr(), B8.stride(0), scale_b.data_ptr()
    d.M, d.N, d.K, d.act = M, N, K, act
    d.bias = 0 if bias is None else bias.data_ptr()
    if block_scale_a is not None:
        _req_mx(block_scale_a, M, K, "block_scale_a")
        d.block_scale_a, d.ld_block_scale_a = block_scale_a.data_ptr(), block_scale_a.stride(0)
    else:
        _req(scale_a, torch.float32, "scale_a")
        d.scale_a = scale_a.data_ptr()
    if out16 is not None:
        _req16(out16, "out16")
        assert out16.shape[1] == N and out16.shape[0] >= M
        d.out16, d.ldc = out16.data_ptr(), out16.stride(0)
    if out_mx is not None:
        o8, omx = out_mx
        assert o8.dtype == torch.uint8 and o8.shape[1] == N and o8.shape[0] >= M and o8.stride(1) == 1
        _req_mx(omx, M, N, "out_mx[1]")
        d.out_fp8, d.ld_out_fp8, d.out_block_scale, d.ld_out_block_scale = o8.data_ptr(), o8.stride(0), omx.data_ptr(), omx.stride(0)
    if out_f32 is not None:
        _req(out_f32, torch.float32, "out_f32"); _req(residual, torch.float32, "residual")
        assert out_f32.shape[1] == N and out_f32.stride(0) == residual.stride(0)
        d.out_f32, d.residual, d.ldf = out_f32.data_ptr(), residual.data_ptr(), out_f32.stride(0)
    ev = None
    if GEMM_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    f16 = (out16.dtype if out16 is not None else half) == torch.float16
    fn = lib.cclip_gemm_fp8_ex_f16 if f16 else lib.cclip_gemm_fp8_ex
    check(fn(ctypes.byref(d), _stream()), "cclip_gemm_fp8_ex")
    if ev is not None:
        ev[1].record()
        GEMM_EVENTS.append((ev[0], ev[1], 2.0 * M * N * K, (1, 1), (M, N, K)))


class BlockPtrs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_w", "ln2_b", "w_fc", "b_fc", "w_proj", "b_proj")]


class DecodeDesc(ctypes.Structure):
    _fields_ = [("n_layer", c_int), ("n_seq", c_int), ("width", c_int), ("heads", c_int), ("hidden", c_int), ("act", c_int),
                ("linear_layout", c_int), ("pos", c_int),
                ("blocks", ctypes.POINTER(BlockPtrs)), ("x", c_void_p), ("kcache", c_void_p), ("vcache", c_void_p),
                ("ld_layer", c_long), ("ld_seq", c_long), ("scratch16", c_void_p),
                ("lnf_w", c_void_p), ("lnf_b", c_void_p), ("wte16", c_void_p), ("vocab", c_int), ("logits", c_void_p), ("ld_logits", c_long)]


def block_ptr_array(blocks):
    """ctypes array of per-layer weight pointers for cclip_gpt2_decode_step (blocks: cclip_hip.stack.BlockWeights)."""
    arr = (BlockPtrs * len(blocks))()
    for i, w in enumerate(blocks):
        for n, _ in BlockPtrs._fields_:
            t = getattr(w, n)
            setattr(arr[i], n, 0 if t is None else t.data_ptr())
    return arr


def gpt2_decode_step(blocks_arr, n_layer, x, kcache, vcache, pos, scratch16, *, heads, hidden, act, linear_layout,
                     lnf_w=None, lnf_b=None, wte16=None, logits=None) -> None:
    """One KV-cached decode step for x.shape[0] sequences, issued natively (cclip_gpt2_decode_step).
    kcache / vcache: [n_layer, n_seq_alloc, max_len, width] 16-bit; x: fp32 [n_seq, width] (in place)."""
    _req(x, torch.float32, "x")
    nb, D = x.shape
    assert x.is_contiguous() and kcache.dim() == 4 and kcache.stride(3) == 1 and kcache.stride(2) == D and kcache.stride() == vcache.stride()
    assert scratch16.numel() >= nb * (5 * D + hidden) and scratch16.dtype == kcache.dtype
    d = DecodeDesc()
    d.n_layer, d.n_seq, d.width, d.heads, d.hidden, d.act, d.linear_layout, d.pos = n_layer, nb, D, heads, hidden, act, int(linear_layout), pos
    d.blocks = blocks_arr
    d.x, d.kcache, d.vcache = x.data_ptr(), kcache.data_ptr(), vcache.data_ptr()
    d.ld_layer, d.ld_seq = kcache.stride(0), kcache.stride(1)
    d.scratch16 = scratch16.data_ptr()
    if logits is not None:
        _req(logits, torch.float32, "logits")
        d.lnf_w, d.lnf_b, d.wte16 = lnf_w.data_ptr(), lnf_b.data_ptr(), wte16.data_ptr()
        d.vocab, d.logits, d.ld_logits = wte16.shape[0], logits.data_ptr(), logits.stride(0)
    check(_fn("cclip_gpt2_decode_step", kcache)(ctypes.byref(d), _stream()), "cclip_gpt2_decode_step")


class BeamDesc(ctypes.Structure):
    _fields_ = [("step", DecodeDesc),
                ("n_steps", c_int), ("first", c_int), ("stop_token", c_int), ("ld_tokens", c_int), ("max_len", c_int), ("grid_cap", c_int),
                ("temperature", c_float),
                ("first_logits", c_void_p), ("wte_f32", c_void_p), ("wpe_f32", c_void_p),
                ("slot_of", c_void_p), ("tokens", c_void_p),
                ("scores", c_void_p), ("seq_lengths", c_void_p), ("is_stopped", c_void_p),
                ("state", c_void_p), ("select_ws", c_void_p)]


BEAM_MAX_BEAMS, BEAM_MAX_LAYERS, BEAM_SELECT_WS_FLOATS = 8, 24, 256 * 8 * 20


class BeamState:
    """Device state of one persistent beam search (cclip_gpt2_beam_search): token rows, scores, lengths, stop flags, the
    cache-slot table and the kernel's bookkeeping words."""

    def __init__(self, n_beams: int, max_len: int, max_tokens: int, device):
        assert 1 <= n_beams <= BEAM_MAX_BEAMS
        self.n_beams, self.max_len = n_beams, max_len
        self.tokens = torch.zeros(n_beams, max_tokens, device=device, dtype=torch.int32)
        self.scores = torch.zeros(n_beams, device=device, dtype=torch.float32)
        self.seq_lengths = torch.ones(n_beams, device=device, dtype=torch.float32)
        self.is_stopped = torch.zeros(n_beams, device=device, dtype=torch.int32)
        self.slot_of = torch.zeros(max_len, BEAM_MAX_BEAMS, device=device, dtype=torch.int32)
        self.state = torch.zeros(8, device=device, dtype=torch.int32)
        self.select_ws = torch.empty(BEAM_SELECT_WS_FLOATS, device=device, dtype=torch.float32)
        self.x = None


def gpt2_beam_search(blocks_arr, n_layer, st: BeamState, kcache, vcache, pos, scratch16, n_steps, *, heads, hidden, act, lnf_w, lnf_b,
                     wte16, wte_f32, wpe_f32, temperature, stop_token, first_logits=None, logits=None, grid_cap: int = 0) -> None:
    """`n_steps` KV-cached decode steps + beam selections in ONE persistent launch (cclip_gpt2_beam_search; Conv1D layout).
    first_logits ([vocab] fp32, the prefill's last row) given: starts with the one-sequence selection of the first loop
    iteration.  The cache ([n_layer, n_beams, max_len, width]) holds the prefix in slot 0; beams are reordered through
    st.slot_of."""
    nb = st.n_beams
    D = wte16.shape[1]
    if st.x is None:
        st.x = torch.empty(nb, D, device=kcache.device, dtype=torch.float32)
    assert kcache.dim() == 4 and kcache.shape[1] == nb and kcache.stride(3) == 1 and kcache.stride(2) == D and kcache.stride() == vcache.stride()
    assert kcache.shape[2] == st.max_len and scratch16.numel() >= nb * (5 * D + hidden) and scratch16.dtype == kcache.dtype
    assert n_layer <= BEAM_MAX_LAYERS and wte_f32.dtype == torch.float32 and wpe_f32.dtype == torch.float32
    assert wte_f32.is_contiguous() and wpe_f32.is_contiguous() and wte16.is_contiguous() and wpe_f32.shape[0] >= st.max_len
    d = BeamDesc()
    s = d.step
    s.n_layer, s.n_seq, s.width, s.heads, s.hidden, s.act, s.linear_layout, s.pos = n_layer, nb, D, heads, hidden, act, 0, pos
    s.blocks = blocks_arr
    s.x, s.kcache, s.vcache = st.x.data_ptr(), kcache.data_ptr(), vcache.data_ptr()
    s.ld_layer, s.ld_seq = kcache.stride(0), kcache.stride(1)
    s.scratch16 = scratch16.data_ptr()
    s.lnf_w, s.lnf_b, s.wte16, s.vocab = lnf_w.data_ptr(), lnf_b.data_ptr(), wte16.data_ptr(), wte16.shape[0]
    if logits is not None:
        _req(logits, torch.float32, "logits")
        s.logits, s.ld_logits = logits.data_ptr(), logits.stride(0)
    d.n_steps, d.first, d.stop_token, d.ld_tokens, d.max_len, d.grid_cap = n_steps, int(first_logits is not None), stop_token, st.tokens.stride(0), st.max_len, grid_cap
    d.temperature = temperature
    if first_logits is not None:
        _req(first_logits, torch.float32, "first_logits")
        assert first_logits.is_contiguous() and first_logits.numel() == wte16.shape[0]
        d.first_logits = first_logits.data_ptr()
    d.wte_f32, d.wpe_f32 = wte_f32.data_ptr(), wpe_f32.data_ptr()
    d.slot_of, d.tokens = st.slot_of.data_ptr(), st.tokens.data_ptr()
    d.scores, d.seq_lengths, d.is_stopped = st.scores.data_ptr(), st.seq_lengths.data_ptr(), st.is_stopped.data_ptr()
    d.state, d.select_ws = st.state.data_ptr(), st.select_ws.data_ptr()
    check(_fn("cclip_gpt2_beam_search", kcache)(ctypes.byref(d), _stream()), "cclip_gpt2_beam_search")


# --------------------------------------------------------------------------------------------
# exact fp32 GEMM:  C = alpha * A @ B^T-like contraction with arbitrary strides
# --------------------------------------------------------------------------------------------
def gemm_f32(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, *, alpha: float = 1.0, beta: float = 0.0,
             alpha_log_dev: Optional[torch.Tensor] = None) -> None:
    """C[m,n] = alpha * sum_k A[m,k] * B[n,k] + beta*C.  A: [M,K], B: [N,K] as (possibly transposed) 2-D views."""
    _req(A, torch.float32, "A"); _req(B, torch.float32, "B"); _req(C, torch.float32, "C")
    M, K = A.shape
    N, Kb = B.shape
    assert K == Kb and C.shape == (M, N) and C.stride(1) == 1
    check(lib.cclip_gemm_f32(_p(A), c_long(A.stride(0)), c_long(A.stride(1)), _p(B), c_long(B.stride(0)),
                             c_long(B.stride(1)), c_int(M), c_int(N), c_int(K), c_float(alpha), _p(alpha_log_dev), c_float(beta), _p(C),
                             c_long(C.stride(0)), _stream()), "cclip_gemm_f32")


# --------------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------------
def patchify(image: torch.Tensor, out_bf16: torch.Tensor, P: int) -> None:
    _req(image, torch.float32, "image")
    assert image.is_contiguous() and image.dim() == 4 and image.shape[1] == 3 and image.shape[2] == image.shape[3]
    check(_fn("cclip_patchify", out_bf16)(_p(image), _p(out_bf16), c_int(image.shape[0]), c_int(image.shape[2]), c_int(P), _stream()),
          "cclip_patchify")


def vit_embed_ln(patch_out, cls, pos, gamma, beta, x, *, rows: int, T: int, x0=None, mean=None, rstd=None, eps=1e-5):
    D = patch_out.shape[-1]
    check(lib.cclip_vit_embed_ln(_p(patch_out), _p(cls), _p(pos), c_int(rows), c_int(T), c_int(D), _p(gamma), _p(beta),
                                 c_float(eps), _p(x0), _p(x), _p(mean), _p(rstd), _stream()), "cclip_vit_embed_ln")


def text_embed(text_i32, emb, pos, x, *, rows: int, L: int) -> None:
    _req(text_i32, torch.int32, "text")
    check(lib.cclip_text_embed(_p(text_i32), _p(emb), _p(pos), c_int(rows), c_int(L), c_int(emb.shape[1]),
                               c_int(emb.shape[0]), _p(x), _stream()), "cclip_text_embed")


SCATTER_DETERMINISTIC = True     # False: the one-launch fp32-atomics kernel (sums rows of one token id in hardware order)
_SEG_CHUNK = 64


def embed_scatter_tables(text_i32, V: int, *, rows: int, keep: Optional[torch.Tensor] = None):
    """Index tables of the deterministic embedding-gradient sum (see embed_scatter_add): the row list sorted by token id
    (stable) and, per sorted position, the chunk / run bookkeeping csrc/embed.hip walks.  Integer math on a [rows] vector that
    depends on the token ids only - a training step can build it while the forward pass runs (clip/model.py does, on a side
    stream).  The sort is the device library's; the tables come from one scan-free kernel (cclip_embed_tables: two binary
    searches per position; round 2 built them from ~25 small torch launches incl. three single-block scans, 0.6 ms)."""
    tok = text_i32[:rows].clamp(0, V - 1)
    if keep is not None:                                  # dropped rows sort to the end under a sentinel id and form no chunk
        tok = torch.where(keep[:rows], tok, torch.full_like(tok, V))
    st, perm = torch.sort(tok.to(torch.int32), stable=True)
    order = perm.to(torch.int32)
    tabs = torch.empty((3, rows), device=text_i32.device, dtype=torch.int32)
    check(lib.cclip_embed_tables(_p(st), c_int(rows), c_int(V), _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _stream()), "cclip_embed_tables")
    return (order, st, tabs[0], tabs[1], tabs[2])


def embed_scatter_add(text_i32, dx, demb, *, rows: int, L: Optional[int] = None, seq_stride: Optional[int] = None,
                      seq_off: int = 0, keep: Optional[torch.Tensor] = None, tables=None, scratch=None) -> None:
    """demb[text[r]] += dx[(r // L) * seq_stride + seq_off + r % L] for r < rows (the embedding-table gradient).
    keep: optional bool [rows] - rows known to carry a zero gradient (text positions after EOT) can be dropped up front.
    tables: embed_scatter_tables(text_i32, V, rows=rows, keep=keep) built earlier (same text / keep).

    Deterministic by default: the row list is sorted by token id (stable) and every run of equal ids is summed in list order
    by the wave that owns its embedding row, runs longer than 64 rows through ordered partials (csrc/embed.hip).  The sort /
    run bookkeeping is integer index math on a [rows] vector; every float is added by the HIP kernels."""
    L = rows if L is None else L
    seq_stride = L if seq_stride is None else seq_stride
    if not SCATTER_DETERMINISTIC:
        check(lib.cclip_embed_scatter_add(_p(text_i32), _p(dx), c_long(dx.stride(-2)), c_int(rows), c_int(demb.shape[1]),
                                          c_int(demb.shape[0]), _p(demb), c_int(L), c_int(seq_stride), c_int(seq_off),
                                          _stream()), "cclip_embed_scatter_add")
        return
    V, D = demb.shape
    order, st, cend, cidx, rlen = tables if tables is not None else embed_scatter_tables(text_i32, V, rows=rows, keep=keep)
    n = rows
    # one slot per CHUNK (a run of <= 64 equal ids), touched by multi-chunk runs only; slot = id + (chunk start >> 6)
    # (cclip_embed_tables).  `scratch(n_floats)` (the stack's grow-only scratch) keeps it out of the allocator: round 2 drew a
    # fresh [rows, D] fp32 tensor - 161 MB at bs 1024 - per backward pass.
    slots = V + n // _SEG_CHUNK + 1
    partial = scratch(slots * D) if scratch is not None else torch.empty(slots * D, device=text_i32.device, dtype=torch.float32)
    check(lib.cclip_embed_segsum(_p(order), _p(st), _p(cend), _p(cidx), _p(rlen), c_int(n), _p(dx), c_long(dx.stride(-2)), c_int(D),
                                 _p(demb), c_int(L), c_int(seq_stride), c_int(seq_off), _p(partial), _stream()), "cclip_embed_segsum")


def caption_embed(prefix_proj, ids_i32, wte, wpe, x, *, B: int, P: int, Lt: int) -> None:
    _req(prefix_proj, torch.float32, "prefix_proj")
    assert prefix_proj.is_contiguous() and prefix_proj.numel() == B * P * wte.shape[1], "prefix_proj must be dense [B, P*D]"
    check(lib.cclip_caption_embed(_p(prefix_proj), _p(ids_i32), _p(wte), _p(wpe), c_int(B), c_int(P), c_int(Lt),
                                  c_int(wte.shape[1]), c_int(wte.shape[0]), _p(x), _stream()), "cclip_caption_embed")


def add_positional(emb, wpe, x, *, rows: int, S: int) -> None:
    check(lib.cclip_add_positional(_p(emb), _p(wpe), c_int(rows), c_int(S), c_int(wpe.shape[1]), _p(x), _stream()),
          "cclip_add_positional")


def colsum_ws_floats(R: int, C: int) -> int:
    return lib.cclip_colsum_ws_floats(c_int(R), c_int(C))


def colsum(inp: torch.Tensor, out: torch.Tensor, ws: torch.Tensor, *, R: int, C: int, ld: int, accumulate: bool = False):
    check(_fn("cclip_colsum", inp)(_p(inp), c_int(_is16(inp)), c_long(ld), c_int(R), c_int(C), _p(out),
                           c_int(int(accumulate)), _p(ws), _stream()), "cclip_colsum")


# --------------------------------------------------------------------------------------------
# loss side
# --------------------------------------------------------------------------------------------
def l2norm_fwd(x, y, inv_norm) -> None:
    check(lib.cclip_l2norm_fwd(_p(x), c_long(x.stride(0)), c_int(x.shape[0]), c_int(x.shape[1]), _p(y), c_long(y.stride(0)),
                               _p(inv_norm), _stream()), "cclip_l2norm_fwd")


def l2norm_bwd(dy, y, inv_norm, dx, mul_dev=None) -> None:
    check(lib.cclip_l2norm_bwd(_p(dy), c_long(dy.stride(0)), _p(y), c_long(y.stride(0)), _p(inv_norm), c_int(y.shape[0]),
                               c_int(y.shape[1]), _p(dx), c_long(dx.stride(0)), _p(mul_dev), _stream()), "cclip_l2norm_bwd")


def xent_rows(logits, labels_i32, *, loss_row=None, pred=None, dlogits=None, grad_scale: float = 1.0,
              ignore_index: int = -100, rowdot=None) -> None:
    _req(logits, torch.float32, "logits"); _req(labels_i32, torch.int32, "labels")
    R, C = logits.shape
    check(_fn("cclip_xent_rows", dlogits)(_p(logits), c_long(logits.stride(0)), c_int(R), c_int(C), _p(labels_i32), c_int(ignore_index),
                              c_float(grad_scale), _p(loss_row), _p(pred), _p(dlogits),
                              c_int(_is16(dlogits)),
                              c_long(0 if dlogits is None else dlogits.stride(0)), _p(rowdot), _stream()), "cclip_xent_rows")


def reduce_dot(a, b, out, *, alpha: float = 1.0, mul_dev=None, accumulate: bool = False) -> None:
    check(lib.cclip_reduce_dot(_p(a), _p(b), c_long(a.numel()), c_float(alpha), _p(mul_dev), _p(out),
                               c_int(int(accumulate)), _stream()), "cclip_reduce_dot")


# --------------------------------------------------------------------------------------------
# optimiser / casts over flat buffers
# --------------------------------------------------------------------------------------------
def adamw_step(param, grad, exp_avg, exp_avg_sq, *, lr: float, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0,
               step: int = 1, correct_bias: bool = True, grad_scale: float = 1.0, mode: int = 0, bf16_shadow=None):
    n = param.numel()
    check(_fn("cclip_adamw_step", bf16_shadow)(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), c_long(n), c_float(lr), c_float(beta1),
                               c_float(beta2), c_float(eps), c_float(weight_decay), c_int(step), c_int(int(correct_bias)),
                               c_float(grad_scale), c_int(mode), _p(bf16_shadow), _stream()), "cclip_adamw_step")


def transpose16_batched(src_base: torch.Tensor, dst_base: torch.Tensor, table_dev: torch.Tensor, max_tiles: int) -> None:
    """table_dev: int64 [n, 4] on the device = (src element offset, dst element offset, rows, cols) per matrix"""
    assert src_base.dtype in HALF_TYPES and dst_base.dtype == src_base.dtype and table_dev.dtype == torch.int64 and table_dev.is_cuda
    check(lib.cclip_transpose16_batched(_p(src_base), _p(dst_base), _p(table_dev), c_int(table_dev.shape[0]), c_int(max_tiles),
                                        _stream()), "cclip_transpose16_batched")


def scale_f32(x: torch.Tensor, alpha: float) -> None:
    """x *= alpha in place (flat fp32, numel % 4 == 0, 16-byte aligned)"""
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    check(lib.cclip_scale_f32(_p(x), c_long(x.numel()), c_float(alpha), _stream()), "cclip_scale_f32")


def cast_f32_to_bf16(src, dst) -> None:
    check(_fn("cclip_cast_f32_to_bf16", dst)(_p(src), _p(dst), c_long(src.numel()), _stream()), "cclip_cast_f32_to_bf16")
