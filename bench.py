"""Headline benchmark: image-text pairs/sec of the CLIP ViT-B/32 contrastive step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode train|fwd|image|caption] [--batch 1024] [--dtype bf16|fp16|fp8]

Workload (BASELINE.json configs[1]): CLIP/train.py's step `model(image, text)` -> symmetric CE ->
backward -> AdamW, scaled to bs = 1024 pairs per GPU, synthetic 224x224 N(0,1) images + 77-token captions
(SURVEY.md 8d), seeded OpenAI-style weights, bf16 MFMA operands / fp32 accumulate / fp32 masters.
One "step" = encode_image + encode_text + logits + loss + full backward + optimiser step over one batch.
`--mode fwd` times encode + logits only, `--mode image` encode_image alone (the "40 % of the bf16 roofline"
target of BASELINE.md is quoted on it), `--mode caption` BASELINE.json configs[3] (ClipCaptionModel: MLP mapper +
GPT-2-small, V = 21128, bs = 256, 40 caption tokens -> S = 80, fwd + bwd + AdamW on pre-extracted CLIP prefixes).
Those are extras with their own metric names; the default line is always the train step.
N > 1: one process per GPU (torchrun contract), weak scaling (per-GPU batch fixed), embedding
all-gather + reduce-scatter and SUM all-reduce of the flat gradient arena over RCCL.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - dominant kernel family cclip_gemm_bf16 = gemm_bf16_kernel<*> / gemm_a4*_kernel / gemm_stream_kernel<*> (97 % of the step's
                 FLOPs, ~78 % of its time): algorithmic FLOPs of
                 every launch in the timed region (2*M*N*K) / their summed durations (HIP events on the launch
                 stream), against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline - the CPU oracle (kind "port": the reference's `clip` package is absent) timed
                 on the host cores, SURVEY.md 8d's procedure (bs 64, 3 warm-up + 5 timed, median): the train step and,
                 beside it, the forward-only encode+logits figure.
The text tower runs on PACKED rows by default (each caption's positions 0..EOT; `config.text_rows`) and the last block of each tower on its pooled rows; the `all_rows` leg times the same step with both switched off.
and, in the default (train, one GPU) run, extra objects timed after the headline steps: `encode_image`
(images/s + fraction of the bf16 MFMA peak - BASELINE.md's 40 % target), `encode_image_fp8` (the same with the block
projections in e4m3: BASELINE configs[4]'s path on the headline model), `forward_only`, `parity_mode` (the same step
with fp16 operands).  `roofline.traffic` / `mfma_busy` are REPLAYED from committed rocprofv3 --pmc summaries and only
when those were taken on the same GEMM kernel sources (see attach_replayed_pmc).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]

PAIR_FWD_FLOPS = 14_777_163_776          # encode_image 8 817 623 040 + encode_text 5 959 540 736
IMAGE_FWD_FLOPS = 8_817_623_040
CAPTION_FWD_FLOPS = 16_420_000_000 + 243_793_920   # GPT-2 prefix forward at S=80, V=21128 + mapper
PEAK_BF16 = 2.5e15                       # dense bf16 MFMA, MI355X_MICROARCH.md


def tower_flops(geo):
    """Algorithmic forward FLOPs (2*MAC, GEMM-class work, no causal discount) per image / per caption, SURVEY.md 8d's
    accounting: ViT-B/32 -> 8 817 623 040 and 5 959 540 736; ViT-L/14@336px -> 381 919 789 056 per image."""
    def layers(T, W, L):
        return L * (24 * T * W * W + 4 * T * T * W)
    T, W = geo.vision_tokens, geo.vision_width
    img = 2 * (T - 1) * 3 * geo.vision_patch_size ** 2 * W + layers(T, W, geo.vision_layers) + 2 * W * geo.embed_dim
    Tt, Wt = geo.context_length, geo.transformer_width
    txt = layers(Tt, Wt, geo.transformer_layers) + 2 * Wt * geo.embed_dim
    return img, txt


def tail_saving(geo):
    """FLOPs per image / per caption NOT executed when the last block's out-proj and MLP run on the pooled row only
    (BlockStack tail_rows, default): 18 W^2 per dropped token (2 W^2 out-proj + 16 W^2 MLP)."""
    T, W = geo.vision_tokens, geo.vision_width
    Wt = geo.transformer_width
    return 18 * W * W * (T - 1), 18 * Wt * Wt          # text: per dropped ROW (the caller multiplies by its dropped rows per caption)


def host_cores() -> int:
    """CPU share this process may actually use: min(affinity mask, cgroup quota, 16 = one GPU's share of the box)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(mode: str, bs: int = 64):
    """The CPU oracle on the host cores, SURVEY.md 8d's procedure: bs = 64, 3 warm-up + 5 timed iterations, median.
    mode "train": the fwd+bwd step (value) plus the forward-only `encode_image + encode_text + logits` figure beside it
    (the metric's own wording); "fwd": forward only; "image": encode_image only, in images/s."""
    from clip.weights import MODELS, init_state_dict, synthetic_images, synthetic_text
    from oracle import clip_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    geo = MODELS["ViT-B/32"]
    sd = init_state_dict(geo, 567)
    img, txt = synthetic_images(bs, geo, 568), synthetic_text(bs, geo, 569)

    def timed(fn, what, warm=3, reps=5, budget_s=22.0):
        times, t_all = [], time.time()
        for it in range(warm + reps):
            t0 = time.time()
            fn()
            times.append(time.time() - t0)
            log(f"cpu baseline {what} iteration {it}: {times[-1]:.2f}s")
            if time.time() - t_all > budget_s and len(times) > warm:      # a slow host: keep the run bounded, say so below
                break
        kept = times[warm:] if len(times) > warm else times[-1:]
        return sorted(kept)[len(kept) // 2], len(kept)

    def fwd():
        with torch.no_grad():
            O.clip_forward(sd, img, txt)

    def image_only():
        with torch.no_grad():
            O.encode_image(sd, img)

    def train():
        sdg = {k: v.requires_grad_(True) for k, v in sd.items()}
        loss, _ = O.contrastive_loss(*O.clip_forward(sdg, img, txt))
        loss.backward()
        for v in sdg.values():
            v.grad = None

    where = f"oracle/clip_oracle.py fp32, batch {bs}, torch CPU {cores} threads, median of {{n}} timed iterations after 3 warm-up"
    if mode == "image":
        t, n = timed(image_only, "encode_image")
        return dict(value=round(bs / t, 2), unit="images/s", cores=cores, kind="port", sample="encode_image only; " + where.format(n=n))
    tf, nf = timed(fwd, "forward")
    fwd_obj = dict(value=round(bs / tf, 2), unit="pairs/s", sample="encode_image + encode_text + logits (forward only); " + where.format(n=nf))
    if mode != "train":
        return dict(cores=cores, kind="port", **fwd_obj)
    for v in sd.values():
        v.requires_grad_(False)
    tt, nt = timed(train, "train step")
    return dict(value=round(bs / tt, 2), unit="pairs/s", cores=cores, kind="port",
                sample="contrastive step forward + backward (no optimiser); " + where.format(n=nt), forward=fwd_obj)


def gemm_roofline(ev, nprof, traffic=None, traffic_src=None):
    """Fold the (start, end, flops, layout, shape) HIP-event records of ops.GEMM_EVENTS into the `roofline` object."""
    tot_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in ev)
    tot_fl = sum(f for _, _, f, *_ in ev)
    by = {}
    for e0, e1, f, lay, _shape in ev:
        k = {(1, 1): "fwd", (1, 0): "dgrad", (0, 0): "wgrad"}[lay]
        t, fl, n = by.get(k, (0.0, 0.0, 0))
        by[k] = (t + e0.elapsed_time(e1), fl + f, n + 1)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    return dict(bound="mfma", kernel="gemm_bf16_kernel<*> + gemm_a4_kernel<*> / gemm_a4w_kernel + gemm_stream_kernel<*> (cclip_gemm_bf16: all layouts and tile configs)", achieved=round(ach, 1),
                peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=round(ach * 1e12 / PEAK_BF16, 4), traffic=traffic,
                traffic_unit="HBM-side bytes per launch (PMC, includes Infinity-Cache hits)", traffic_source=traffic_src,
                flops_per_launch=round(tot_fl / max(len(ev), 1)),
                launches_per_step=len(ev) // nprof, gemm_ms_per_step=round(tot_ms / nprof, 3),
                by_layout={k: dict(tflops=round(fl / (t * 1e-3) / 1e12, 1), ms_per_step=round(t / nprof, 3), launches=n // nprof)
                           for k, (t, fl, n) in by.items()})


def attach_replayed_pmc(roof, args, B, world, src_hash):
    """PMC counters cannot be read from inside the process.  `traffic` (FETCH_SIZE x 2 per MI355X_MICROARCH.md + WRITE_SIZE, per
    GEMM launch) and `mfma_busy` are therefore REPLAYED from the committed rocprofv3 --pmc summaries of this same command
    (profiles/r03_train_bs1024_*_pmc.json) and marked as such - and only when they describe what just ran: default train
    step, ViT-B/32, bs 1024, bf16, one GPU, and the SAME GEMM kernel sources (hash recorded when the profile was taken).
    Anything else reports null rather than a stale number."""
    roof["traffic_replayed"] = None
    if not (args.mode == "train" and B == 1024 and args.dtype == "bf16" and args.model == "ViT-B/32" and world == 1):
        return
    tp = os.path.join(ROOT, "profiles", "r03_train_bs1024_hbm_traffic_pmc.json")
    mp = os.path.join(ROOT, "profiles", "r03_train_bs1024_mfma_busy_pmc.json")
    try:
        t = json.load(open(tp))
        if t.get("kernel_source_hash") == src_hash:
            gf = t["gemm_family"]
            roof["traffic"] = round((gf["fetch_mb_per_launch_corrected"] + gf["write_mb_per_launch"]) * 1e6)
            roof["traffic_replayed"] = dict(source="profiles/r03_train_bs1024_hbm_traffic_pmc.json", command=t.get("command"),
                                            git=t.get("git"), kernel_source_hash=src_hash)
        m = json.load(open(mp))
        if m.get("kernel_source_hash") == src_hash:
            roof["mfma_busy"] = m["gemm_family_mfma_busy"]
            roof["mfma_busy_replayed"] = dict(source="profiles/r03_train_bs1024_mfma_busy_pmc.json", git=m.get("git"))
    except (OSError, KeyError, ValueError):
        pass


def caption_main(args, rank, world, dev, B, cdt):
    """BASELINE.json configs[3]: CLIP_prefix_caption/train.py step (train.py:347-361) on pre-extracted prefixes."""
    import torch.distributed as dist
    from clip import optim as coptim
    from clip import parallel
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    from cclip_hip import ops
    geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
    Lc = 40
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, 567))
    model = model.to(dev).train()
    if cdt == torch.float16:
        model.half()
    parallel.broadcast_parameters(model)
    opt = coptim.AdamW(model, lr=2e-5)                                     # train.py:336 (lr 2e-5)
    sched = coptim.get_linear_schedule_with_warmup(opt, 5000, 10 * 1000)   # train.py:338-340
    tokens, mask, prefix, attribute = [t.to(dev) for t in synthetic_caption_batch(B, geo, Lc, 567 + rank)]

    def step():
        opt.zero_grad()
        loss = model.caption_loss(tokens, prefix, attribute, mask)
        loss.backward()
        # every rank's loss is its local mean: average the summed gradients; AdamW runs bucket by bucket under the all-reduce
        opt.step(grad_scale=1.0 / world, pending=parallel.allreduce_gradients_async(model, None))
        sched.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"caption warmup step {i} done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    nprof = min(2, args.steps)
    if rank == 0:
        ops.GEMM_EVENTS = []
    os.environ["CCLIP_WGRAD_STREAM"] = "0"
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        ev, ops.GEMM_EVENTS = ops.GEMM_EVENTS, None
        S = geo.prefix_length + geo.attribute_length + Lc
        out = {"metric": f"caption samples/sec (ClipCaptionModel train step: MLP mapper + GPT-2-small, bs={B})",
               "value": round(B * world * args.steps / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"CLIP_prefix_caption/train.py step (fwd+bwd+AdamW), MLP mapper 512->7680->15360 + GPT-2-small "
                          f"V={geo.vocab_size}, bs={B}/GPU, prefix 20 + attribute 20 + {Lc} caption tokens (S={S}), seeded synthetic weights",
                          "global_batch": B * world, "parallelism": f"dp{world}", "mode": "caption"},
               "not_executed_flops_view": {"frac_of_bf16_peak": {"step": round(3 * CAPTION_FWD_FLOPS * B * world / (dt / args.steps) / (PEAK_BF16 * world), 4)},
                                           "note": "nominal FLOPs (every row of every sequence), although the packed step does not compute them all"},
               "loss": round(float(loss.item()), 5), "roofline": gemm_roofline(ev, nprof), "cpu_baseline": None}
        roof = out["roofline"]
        out["step_mfu_bf16"] = round(roof["flops_per_launch"] * roof["launches_per_step"] / (dt / args.steps) / PEAK_BF16, 4)   # executed GEMM FLOPs
        lens = ((tokens != 0) * torch.arange(1, Lc + 1, device=dev)[None, :]).amax(dim=1)
        out["config"]["rows"] = {"rows_dense": B * S, "rows_live": int((S - Lc + lens - 1).clamp(min=1).sum().item()), "packed": model._pack_rows(),
                                 "note": "GPT-2 is causal and the loss ignores zero targets (train.py:357): of a sequence only the rows up to the one "
                                         "that predicts its last non-zero token matter; caption_loss runs the stack on those rows, sequences back "
                                         "to back (CCLIP_PACK_TEXT=0: all P + A + L rows)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = caption_cpu_baseline(geo, Lc)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def caption_cpu_baseline(geo, Lc, budget_s: float = 12.0):
    from clip_caption import init_caption_state_dict, synthetic_caption_batch
    from oracle import caption_oracle as CO
    cores = host_cores()
    torch.set_num_threads(cores)
    bs = 8
    sd = init_caption_state_dict(geo, 567)
    tokens, mask, prefix, attribute = synthetic_caption_batch(bs, geo, Lc, 568)
    times, t_all = [], time.time()
    for it in range(12):
        t0 = time.time()
        sdg = {k: v.detach().requires_grad_(True) for k, v in sd.items() if k != "model.lm_head.weight"}
        sdg["model.lm_head.weight"] = sdg["model.transformer.wte.weight"]
        lg = CO.caption_forward(sdg, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
        CO.caption_loss(lg, tokens, geo.prefix_length, geo.attribute_length).backward()
        times.append(time.time() - t0)
        log(f"caption cpu baseline iteration {it}: {times[-1]:.2f}s")
        if time.time() - t_all > budget_s and it >= 2:
            break
    t = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    return dict(value=bs / t, unit="samples/s", cores=cores, kind="port",
                sample=f"oracle/caption_oracle.py fp32 fwd+bwd, batch {bs}, median of {len(times) - 1 or 1} iterations after 1 warm-up, "
                       f"torch CPU {cores} threads")


def _time_loop(fn, warm: int, reps: int) -> float:
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def extra_legs(model, image, text, geo, B, args):
    import clip
    from clip import optim as coptim
    from clip.weights import init_state_dict
    img_fl, txt_fl = tower_flops(geo)
    img_ex = img_fl - (tail_saving(geo)[0] if model._tail_rows() else 0)       # FLOPs executed per image
    out = {}
    model.eval()

    def enc_image():
        with torch.no_grad():
            model.encode_image(image)

    def fwd():
        with torch.no_grad():
            fi, ft = model.encode_image_text(image, text)
            clip.contrastive_loss(fi, ft, model.logit_scale, None)

    t = _time_loop(enc_image, 3, 20)
    nominal = {}          # fractions on FLOPs that were NOT all executed (SURVEY 8d's nominal counts): kept apart, never a roofline claim
    out["encode_image"] = dict(images_per_s=round(B / t, 1), ms=round(t * 1e3, 3), frac_of_bf16_peak=round(img_ex * B / t / PEAK_BF16, 4),
                               note="encode_image alone, same model / batch / operand type as the headline step, 20 iterations; the fraction counts the "
                                    "FLOPs executed (the last block's out-proj / MLP run on the class rows only)")
    nominal["encode_image"] = round(img_fl * B / t / PEAK_BF16, 4)
    t = _time_loop(fwd, 3, 20)
    # the text tower's executed FLOPs scale with the rows it runs on (packed: the captions' live positions only)
    live = float((text.argmax(-1) + 1).sum().item()) / (B * geo.context_length) if model._pack_text_rows() else 1.0
    txt_ex = txt_fl * live - (tail_saving(geo)[1] * (live * geo.context_length - 1) if model._tail_rows() else 0)
    out["forward_only"] = dict(pairs_per_s=round(B / t, 1), ms=round(t * 1e3, 3),
                               frac_of_bf16_peak=round((img_ex + txt_ex) * B / t / PEAK_BF16, 4),
                               note="encode_image + encode_text + logits + loss, forward only, 20 iterations; the fraction counts the FLOPs "
                                    "executed (text tower on its live rows)")
    nominal["forward_only"] = round((img_fl + txt_fl) * B / t / PEAK_BF16, 4)
    if args.dtype != "fp8":
        # the same encode_image with the block projections in e4m3 (BASELINE configs[4]'s path on the headline model): inference
        # only, accuracy bounded not matched (tests/test_fp8_gpu.py: image features ~3e-2 of the fp32 oracle)
        model.fp8_projections(True)
        t = _time_loop(enc_image, 3, 20)
        model.fp8_projections(False)
        out["encode_image_fp8"] = dict(images_per_s=round(B / t, 1), ms=round(t * 1e3, 3), frac_of_bf16_peak=round(img_ex * B / t / PEAK_BF16, 4),
                                       frac_of_fp8_peak=round(img_ex * B / t / (2 * PEAK_BF16), 4),
                                       note="encode_image with e4m3 qkv / out-proj / fc / c_proj (block-scaled fp8 MFMA), same model and batch, "
                                            "20 iterations; an extra at NARROWER arithmetic than the reference - throughput only, not a roofline claim")
        nominal["encode_image_fp8"] = round(img_fl * B / t / PEAK_BF16, 4)
    model.train()
    if model._pack_text_rows() or model._tail_rows():
        # the same train step with every row computed, as the reference's modules do: the text tower on all 77 positions of every
        # caption and the last block of each tower on every token (both switches off; identical features and loss)
        oo = coptim.AdamW(model, lr=1e-5)

        def step_dense():
            oo.zero_grad()
            fi, ft = model.encode_image_text(image, text)
            loss, stats = clip.contrastive_loss(fi, ft, model.logit_scale, None)
            loss.backward()
            oo.step()

        model.pack_text_rows = False
        prev_tail = os.environ.get("CCLIP_TAIL_ROWS")
        os.environ["CCLIP_TAIL_ROWS"] = "0"
        t = _time_loop(step_dense, 2, 6)
        model.pack_text_rows = None
        if prev_tail is None:
            del os.environ["CCLIP_TAIL_ROWS"]
        else:
            os.environ["CCLIP_TAIL_ROWS"] = prev_tail
        out["all_rows"] = dict(ms_per_step=round(t * 1e3, 3), pairs_per_s=round(B / t, 1),
                               note="the same train step computing every row like the reference's modules: text tower on all 77 positions "
                                    "of every caption (CCLIP_PACK_TEXT=0) and the last block of each tower on every token "
                                    "(CCLIP_TAIL_ROWS=0); identical features and loss, gradients equal to summation order; 6 iterations")
        del oo
    if args.dtype == "bf16":
        m16 = clip.build_model(init_state_dict(geo, 567), torch.float16).to(image.device).train()
        o16 = coptim.AdamW(m16, lr=1e-5)

        def step16():
            o16.zero_grad()
            fi, ft = m16.encode_image_text(image, text)
            loss, stats = clip.contrastive_loss(fi, ft, m16.logit_scale, None)
            loss.backward()
            o16.step()

        t = _time_loop(step16, 3, 6)
        out["parity_mode"] = dict(dtype="fp16", ms_per_step=round(t * 1e3, 3), pairs_per_s=round(B / t, 1),
                                  note="the same train step with fp16 MFMA operands (features <= 1e-3 of the fp32 oracle, bit-exact "
                                       "argmax: tests/test_clip_parity_gpu.py); 6 iterations")
        del m16, o16
    out["not_executed_flops_view"] = dict(frac_of_bf16_peak=nominal,
                                          note="the same timings divided into SURVEY 8d's NOMINAL FLOP counts (every token of the last block, all 77 "
                                               "text positions) although those rows were not computed: informational, not roofline fractions")
    return out


def log(msg: str):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # where are we, if something stalls
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default 1024; 256 for --mode caption)")
    ap.add_argument("--mode", choices=["train", "fwd", "image", "caption"], default="train")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="train mode: skip the extra encode_image / forward-only / fp16-operand legs timed after the headline steps")
    ap.add_argument("--text-prefetch", action="store_true",
                    help="announce the next token batch one step ahead (model.prefetch_text) instead of reading the packed text tower's row count "
                         "back at the start of the step; measured SLOWER (51.6-51.7 vs 50.8 ms, DESIGN.md 6.0) and therefore off")
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp8"], default="bf16",
                    help="MFMA operand type; fp8 = e4m3 projections of the image tower's blocks (inference modes only; CCLIP_FP8_WIDE=0: qkv / fc only), bf16 elsewhere")
    ap.add_argument("--tower-streams", type=int, default=int(os.environ.get("CCLIP_TOWER_STREAMS", "2")),
                    help="2 = run the image and text towers (forward and backward) on two HIP streams")
    args = ap.parse_args()

    import torch.distributed as dist
    import clip
    from clip import optim as coptim
    from clip import parallel
    from clip.weights import MODELS, init_state_dict, synthetic_text
    from cclip_hip import ops

    rank, world, local = parallel.init_distributed(os.environ.get("CCLIP_DIST_BACKEND"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if "CCLIP_FORCE_DEVICE" in os.environ:      # rehearsal of the N>1 path on a one-GPU box (gloo backend)
        local = int(os.environ["CCLIP_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    geo = MODELS[args.model]
    B = args.batch or (256 if args.mode == "caption" else 1024)
    cdt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.dtype == "fp8" and args.mode not in ("image", "fwd"):
        raise SystemExit("--dtype fp8 is an inference path: use --mode image or --mode fwd")
    if args.mode == "caption":
        return caption_main(args, rank, world, dev, B, cdt)

    model = clip.build_model(init_state_dict(geo, 567), cdt).to(dev)
    model.train()
    if args.dtype == "fp8":
        model.eval().fp8_projections(wide=os.environ.get("CCLIP_FP8_WIDE", "1") != "0")   # 0: qkv / fc only (the round-1 form)
    parallel.broadcast_parameters(model)
    opt = coptim.AdamW(model, lr=1e-5)                       # CLIP/train.py:143 (HF AdamW, lr 1e-5)
    sched = coptim.get_linear_schedule_with_warmup(opt, 5000, 1000 * 50)   # CLIP/train.py:145-147
    g = torch.Generator(device=dev).manual_seed(567 + rank)
    image = torch.randn(B, 3, geo.image_resolution, geo.image_resolution, device=dev, generator=g)
    text = synthetic_text(B, geo, 567 + rank).to(dev)
    group = None
    reducer = parallel.GradReducer(model, group)

    os.environ["CCLIP_TOWER_STREAMS"] = str(args.tower_streams)

    # --text-prefetch: a training loop holds batch k+1 while step k runs; it can announce that batch's token ids
    # (model.prefetch_text) BEFORE it launches step k, and the packed text tower then finds its live-row count in pinned memory
    # instead of reading it back at the start of the step.  Two copies of the synthetic token batch alternate so that the
    # announcement really is one step ahead of the use.  Default: one blocking read per step.
    texts = [text, text.clone()] if args.text_prefetch else [text]
    turn = [0]

    def encode_both():
        t = texts[turn[0] % len(texts)]
        if len(texts) > 1:
            model.prefetch_text(texts[(turn[0] + 1) % len(texts)])
        turn[0] += 1
        return model.encode_image_text(image, t)

    def step():
        if args.mode == "image":
            with torch.no_grad():
                return [model.encode_image(image).float().sum()]
        if args.mode == "fwd":
            with torch.no_grad():
                fi, ft = encode_both()
                loss, stats = clip.contrastive_loss(fi, ft, model.logit_scale, group)
            return stats
        opt.zero_grad()
        fi, ft = encode_both()
        loss, stats = clip.contrastive_loss(fi, ft, model.logit_scale, group)
        reducer.begin()              # N > 1: gradient buckets are all-reduced from inside backward, block group by block group
        loss.backward()
        opt.step(pending=reducer.finish())   # AdamW bucket i as soon as it is reduced, under the all-reduce of bucket i+1
        sched.step()
        return stats

    log(f"model ready on {dev}; warmup x{args.warmup}")
    for i in range(args.warmup):
        stats = step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    loss_val = float(stats[0].item())
    log(f"timed {args.steps} steps in {dt:.3f}s")

    # ---- roofline leg: same steps again with every GEMM launch bracketed by HIP events ----
    roof = None
    nprof = min(2, args.steps)
    if rank == 0:
        ops.GEMM_EVENTS = []
    os.environ["CCLIP_TOWER_STREAMS"] = "1"   # kernel durations are measured with the launches serialised on one stream
    os.environ["CCLIP_WGRAD_STREAM"] = "0"
    prev_lanes = os.environ.get("CCLIP_IMAGE_LANES")
    os.environ["CCLIP_IMAGE_LANES"] = "1"     # (inference encode_image otherwise runs two half-batch lanes on two streams)
    for _ in range(nprof):          # every rank steps (the step contains collectives); only rank 0 records events
        step()
    torch.cuda.synchronize()
    os.environ["CCLIP_TOWER_STREAMS"] = str(args.tower_streams)
    if prev_lanes is None:
        del os.environ["CCLIP_IMAGE_LANES"]
    else:
        os.environ["CCLIP_IMAGE_LANES"] = prev_lanes
    if rank == 0:
        ev = ops.GEMM_EVENTS
        ops.GEMM_EVENTS = None
        roof = gemm_roofline(ev, nprof, None, None)
        attach_replayed_pmc(roof, args, B, world, ops.kernel_source_hash())
    if world > 1:
        dist.barrier()

    # ---- extra legs of the default run (one GPU, train mode): the figures BASELINE.md's targets are quoted on, timed in
    # the driver's own run so that they are driver-observable: encode_image alone (the ">= 40 % of the bf16 MFMA roofline"
    # target), encode+logits forward only (the metric's own wording), and the same train step with fp16 operands (the
    # operand type that meets north_star's 1e-3 parity bar, tests/test_clip_parity_gpu.py) ----
    extras = None
    if args.mode == "train" and world == 1 and not args.no_extras:
        extras = extra_legs(model, image, text, geo, B, args)
        log("extra legs done")

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / dt
        img_fl, txt_fl = tower_flops(geo)
        step_flops = (img_fl if args.mode == "image" else (img_fl + txt_fl) * (3 if args.mode == "train" else 1)) * B
        out = {
            "metric": (f"images/sec encode_image {args.model} bs={B} at {world} MI355X" if args.mode == "image"
                       else f"image-text pairs/sec (encode+logits"
                            f"{' forward only' if args.mode == 'fwd' else ' + backward + AdamW: the full CLIP/train.py step'}) "
                            f"{args.model} bs={B} at {world} MI355X"),
            "value": round(value, 1), "unit": "images/s" if args.mode == "image" else "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("CLIP/train.py contrastive fine-tune step (fwd+bwd+AdamW)" if args.mode == "train"
                                    else "encode_image forward only" if args.mode == "image"
                                    else "encode_image+encode_text+logits forward only") + f", {args.model}, bs={B}/GPU, "
                       f"{geo.image_resolution}x{geo.image_resolution} N(0,1) images + 77-token captions, seeded synthetic weights",
                       "global_batch": B * world, "parallelism": f"dp{world}", "mode": args.mode},
            "loss": round(loss_val, 5),
            "roofline": roof,
        }
        # whole-step MFU from the GEMM FLOPs the step EXECUTES (the roofline leg's launches): with the text tower on packed rows
        # that is less than the dense-equivalent 3 x 14.78 GFLOP per pair, which is reported beside it and labelled as such
        exe = roof["flops_per_launch"] * roof["launches_per_step"] if roof else None
        out["step_mfu_bf16"] = round((exe if exe else step_flops) / (dt / args.steps) / PEAK_BF16, 4)
        nominal_step = round(step_flops / (dt / args.steps) / PEAK_BF16, 4)
        # the number BASELINE.json's target is about, inside the driver-parsed roofline object: encode_image ViT-B/32 bs 1024 as a
        # fraction of the bf16 MFMA peak on EXECUTED FLOPs (this run's own figure in --mode image; the extra leg's in train mode)
        if roof is not None:
            tgt = None
            if args.mode == "image" and args.dtype != "fp8":
                img_ex = img_fl - (tail_saving(geo)[0] if model._tail_rows() else 0)
                tgt = round(img_ex * B * world / (dt / args.steps) / (PEAK_BF16 * world), 4)
            elif extras is not None and "encode_image" in extras:
                tgt = extras["encode_image"]["frac_of_bf16_peak"]
            roof["target"] = {"workload": f"encode_image {args.model} bs{B}", "frac_executed": tgt, "goal": 0.40}
        if args.mode != "image":
            live = int((text.argmax(-1) + 1).sum().item())
            packed = model._pack_text_rows()
            out["config"]["text_rows"] = {
                "context_length": geo.context_length, "rows_dense": B * geo.context_length, "rows_live": live, "packed": packed,
                "row_count_prefetched": bool(packed and args.text_prefetch),   # model.prefetch_text one batch ahead (no mid-step read-back)
                "note": "captions end at their EOT token (position uniform in [2, 76], SURVEY.md 8d); the causal text tower pools the EOT "
                        "row, so later positions influence neither features nor gradients" + (
                            " - it runs on the live rows only, sequences back to back (CCLIP_PACK_TEXT=0: all 77 positions)" if packed
                            else " - run on all 77 positions (CCLIP_PACK_TEXT=0)")}
        log("roofline leg done")
        if extras is not None:
            out.update(extras)
        out.setdefault("not_executed_flops_view", {"frac_of_bf16_peak": {}})["frac_of_bf16_peak"]["step"] = nominal_step
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.mode)
            log("cpu baseline done")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
