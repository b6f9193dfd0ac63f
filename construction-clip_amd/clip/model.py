"""`CLIP` - the nn.Module the reference's scripts get from `clip.load()` and call as
`model(image, text)`, `model.encode_image`, `model.encode_text`
(/root/reference/CLIP/train.py:105,161; /root/reference/CLIP/predict.py:12,46;
/root/reference/CLIP_prefix_caption/parse_coco.py:20,43,45,50), rebuilt MI355X-first.

Same module tree and state_dict keys as openai/CLIP, so checkpoints written by
`torch.save(model.state_dict())` (CLIP/train.py:213-217) load unchanged - but the modules are
parameter holders only: all arithmetic is a hand-scheduled sequence of HIP launches
(cclip_hip.stack / libcclip_hip.so) wrapped in three autograd nodes (image tower, text tower,
logits).  There is no torch fallback: on a non-CUDA device forward raises.

Precision: fp32 master weights in a flat arena, bf16 MFMA operands with fp32 accumulation,
fp32 residual stream and LayerNorm statistics, exact-fp32 pooled projections / normalise / logits.
(The reference's CUDA path is plain fp16 weights; bf16 + fp32 masters is the
MI355X-native counterpart.)
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from cclip_hip import duet, ops
from cclip_hip.arena import ParamArena
from cclip_hip.stack import BlockStack, BlockWeights, Scratch, StackGeometry

from .weights import CLIPGeometry, geometry_from_state_dict


# ------------------------------------------------------------------------------------------------
# parameter holders (OpenAI key layout)
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; run the model through "
                           "CLIP.encode_image / encode_text / forward (HIP path)")


class LayerNorm(_Holder):
    def __init__(self, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))


class Linear(_Holder):
    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.zeros(n_out))


class MultiheadAttention(_Holder):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = Linear(width, width)


class MLP(_Holder):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = Linear(width, 4 * width)
        self.c_proj = Linear(4 * width, width)


class ResidualAttentionBlock(_Holder):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.ln_1 = LayerNorm(width)
        self.attn = MultiheadAttention(width, heads)
        self.ln_2 = LayerNorm(width)
        self.mlp = MLP(width)


class Transformer(_Holder):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])


class Conv2d(_Holder):
    def __init__(self, width: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))


class Embedding(_Holder):
    def __init__(self, vocab: int, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab, width))


class VisionTransformer(_Holder):
    def __init__(self, geo: CLIPGeometry):
        super().__init__()
        w = geo.vision_width
        self.input_resolution, self.output_dim = geo.image_resolution, geo.embed_dim
        self.conv1 = Conv2d(w, geo.vision_patch_size)
        self.class_embedding = nn.Parameter(torch.empty(w))
        self.positional_embedding = nn.Parameter(torch.empty(geo.vision_tokens, w))
        self.ln_pre = LayerNorm(w)
        self.transformer = Transformer(w, geo.vision_layers, geo.vision_heads)
        self.ln_post = LayerNorm(w)
        self.proj = nn.Parameter(torch.empty(w, geo.embed_dim))


_BLOCK_KEYS = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "w_qkv": "attn.in_proj_weight", "b_qkv": "attn.in_proj_bias",
               "w_o": "attn.out_proj.weight", "b_o": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
               "w_fc": "mlp.c_fc.weight", "b_fc": "mlp.c_fc.bias", "w_proj": "mlp.c_proj.weight", "b_proj": "mlp.c_proj.bias"}
_MATS = ("w_qkv", "w_o", "w_fc", "w_proj")


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: this CLIP runs on MI355X HIP kernels only; got a {t.device} tensor "
                           "(there is deliberately no CPU/eager fallback)")


# ------------------------------------------------------------------------------------------------
class CLIP(nn.Module):
    def __init__(self, geo: CLIPGeometry, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        assert compute_dtype in (torch.bfloat16, torch.float16)
        self.geo = geo
        self.compute_dtype = compute_dtype      # 16-bit MFMA operand type (masters, residual stream, head stay fp32)
        self.context_length = geo.context_length
        self.vocab_size = geo.vocab_size
        self.visual = VisionTransformer(geo)
        self.transformer = Transformer(geo.transformer_width, geo.transformer_layers, geo.transformer_heads)
        self.token_embedding = Embedding(geo.vocab_size, geo.transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(geo.context_length, geo.transformer_width))
        self.ln_final = LayerNorm(geo.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(geo.transformer_width, geo.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self._arena: Optional[ParamArena] = None
        self._rt: Optional[dict] = None

    # -- nn.Module plumbing ---------------------------------------------------------------------
    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)      # .to()/.cuda()/.half() re-point parameters: rebuild lazily
        self._arena, self._rt = None, None
        return out

    def float(self):          # masters are always fp32; nothing to convert
        return self

    def half(self):           # fp16 MFMA operands (the reference's CUDA dtype; 8x finer than bf16: parity mode)
        return self.set_compute_dtype(torch.float16)

    def bfloat16(self):       # bf16 MFMA operands (default; training range)
        return self.set_compute_dtype(torch.bfloat16)

    def set_compute_dtype(self, dtype: torch.dtype):
        assert dtype in (torch.bfloat16, torch.float16)
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._arena, self._rt = None, None
        return self

    @property
    def arena(self) -> ParamArena:
        self._ensure_runtime()
        return self._arena

    def _ensure_runtime(self):
        dev = self.logit_scale.device
        if dev.type != "cuda":
            raise RuntimeError("CLIP parameters are on %s; move the model to the GPU (model.to('cuda')) - "
                               "the HIP kernels are the only compute path" % dev)
        if self._arena is not None and self._arena.intact():
            return
        ar = ParamArena(self, dev, self.compute_dtype)
        geo = self.geo

        import os
        # transposed weight shadows: every dgrad GEMM then reads its weight K-contiguously (CCLIP_TRANSPOSED_SHADOWS=0: the
        # K-strided reads of the [out, in] weights, as in round 1)
        use_wt = os.environ.get("CCLIP_TRANSPOSED_SHADOWS", "1") == "1"

        def stack(prefix: str, width: int, heads: int, layers: int, tokens: int, causal: bool) -> BlockStack:
            blocks: List[BlockWeights] = []
            for i in range(layers):
                kw, grads = {}, {}
                for f, key in _BLOCK_KEYS.items():
                    name = f"{prefix}.resblocks.{i}.{key}"
                    kw[f] = ar.b[name] if f in _MATS else ar.params[name].data
                    grads[f] = ar.g[name]
                wt = None
                if use_wt:
                    ar.register_transposed([f"{prefix}.resblocks.{i}.{_BLOCK_KEYS[f]}" for f in _MATS])
                    wt = {f: ar.t[f"{prefix}.resblocks.{i}.{_BLOCK_KEYS[f]}"] for f in _MATS}
                blocks.append(BlockWeights(grads=grads, wt=wt, **kw))
            return BlockStack(StackGeometry(width, heads, tokens, True, ops.ACT_QUICKGELU, causal), blocks, Scratch(dev),
                              self.compute_dtype)

        self._arena = ar
        self._rt = dict(
            vis=stack("visual.transformer", geo.vision_width, geo.vision_heads, geo.vision_layers, geo.vision_tokens, False),
            txt=stack("transformer", geo.transformer_width, geo.transformer_heads, geo.transformer_layers,
                      geo.context_length, True),
            vis_names=[n for n in ar.names if n.startswith("visual.")],
            txt_names=[n for n in ar.names if not n.startswith("visual.") and n != "logit_scale"],
        )
        self._rt["vis"].fp8, self._rt["txt"].fp8 = getattr(self, "_fp8_projections", (False, False))
        self._rt["vis"].fp8_wide = self._rt["txt"].fp8_wide = getattr(self, "_fp8_wide", True)
        self._rt["vis"].grad_hook = self._rt["txt"].grad_hook = ar.notify_grads
        if use_wt:
            self._rt["vis"].refresh_transposed = self._rt["txt"].refresh_transposed = ar.refresh_transposed

    def _tail_rows(self) -> bool:
        """Each tower pools ONE row per sequence: the last block's out-proj / MLP run on those rows only (BlockStack tail_rows)."""
        import os
        return os.environ.get("CCLIP_TAIL_ROWS", "1") != "0"

    def _pack_text_rows(self) -> bool:
        import os
        v = getattr(self, "pack_text_rows", None)
        return (os.environ.get("CCLIP_PACK_TEXT", "1") != "0") if v is None else bool(v)

    def fp8_projections(self, enabled: bool = True, text: bool = False, wide: bool = True):
        """INFERENCE ONLY: run the block projections of the image tower - and, with text=True, of the text tower - with e4m3
        operands on the block-scaled fp8 MFMA (BASELINE.json configs[4] is encode_image).  The LayerNorm-fed projections
        (qkv, fc) take per-token / per-output-channel scales; with wide=True (default; needs width and hidden % 128 == 0)
        out-proj and c_proj run in e4m3 as well, their A operands quantised in 32-element blocks with E8M0 scales that the
        MFMA applies itself (the MLP hidden leaves the fc GEMM's epilogue in that format, the attention output takes one
        quantisation pass).  Weights are re-quantised from the current 16-bit shadows at this call; call again after changing
        parameters.  No reference fp8 behaviour exists: accuracy is bounded by test against the fp32 oracle (image features
        ~2.5e-2 relative, cosine > 0.999; the 12-layer causal text tower is ~7e-2 and therefore opt-in), not matched."""
        self._fp8_projections = (bool(enabled), bool(enabled and text))
        self._fp8_wide = bool(wide)
        if self._arena is not None and self._rt is not None:
            self._arena.refresh_shadows()
            for k, on in zip(("vis", "txt"), self._fp8_projections):
                self._rt[k].fp8 = on
                self._rt[k].fp8_wide = self._fp8_wide
                self._rt[k]._fp8_weights = None
        return self

    def initialize_parameters(self, seed: int = 567, finetuned_like: bool = True):
        from .weights import init_state_dict
        self.load_state_dict(init_state_dict(self.geo, seed, finetuned_like))
        return self

    # -- image tower ----------------------------------------------------------------------------
    def _image_forward(self, image: torch.Tensor, train: bool):
        self._ensure_runtime()
        ar, geo, st = self._arena, self.geo, self._rt["vis"]
        ar.refresh_shadows()
        dev = image.device
        B, T, D, P = image.shape[0], geo.vision_tokens, geo.vision_width, geo.vision_patch_size
        if tuple(image.shape[1:]) != (3, geo.image_resolution, geo.image_resolution):
            raise RuntimeError(f"encode_image: expected [N,3,{geo.image_resolution},{geo.image_resolution}], got {tuple(image.shape)}")
        if B == 0:                                   # empty batch (the reference returns an empty [0, embed] tensor)
            if train:
                raise RuntimeError("encode_image: empty batch in a training step")
            return torch.empty(0, geo.embed_dim, device=dev, dtype=torch.float32), None
        M = B * T
        img = image.detach().to(torch.float32).contiguous()
        KP = 3 * P * P
        KPAD = (KP + 7) // 8 * 8                   # P = 14: 588 -> 592, rows stay 16-byte aligned for the GEMM's DMA
        patches = torch.empty(M, KPAD, device=dev, dtype=self.compute_dtype)
        ops.patchify(img, patches, P)
        w16 = ar.b["visual.conv1.weight"].view(D, KP)
        if KPAD != KP:
            wp = torch.zeros(D, KPAD, device=dev, dtype=self.compute_dtype)
            wp[:, :KP].copy_(w16)
            w16 = wp
        patch_out = torch.empty(M, D, device=dev, dtype=torch.float32)
        ops.gemm_bf16(patches, w16, out_f32=patch_out)
        p = ar.params
        saved = st.alloc_saved(B, dev) if train else None
        x = saved["xs"][0, 0] if train else torch.empty(M, D, device=dev, dtype=torch.float32)
        x0 = torch.empty(M, D, device=dev, dtype=torch.float32) if train else None
        st0 = torch.empty(2, M, device=dev, dtype=torch.float32) if train else None
        ops.vit_embed_ln(patch_out, p["visual.class_embedding"].data, p["visual.positional_embedding"].data,
                         p["visual.ln_pre.weight"].data, p["visual.ln_pre.bias"].data, x, rows=M, T=T, x0=x0,
                         mean=st0[0] if train else None, rstd=st0[1] if train else None)
        rows = (torch.arange(B, device=dev, dtype=torch.int32) * T).contiguous()
        tail = self._tail_rows()
        xo = st.forward(x, B, saved=saved, tail_rows=rows.long() if tail else None,     # tail: [B, D], the class rows only
                        weights_version=ar._stamp)
        if tail:
            rows = None
        pooled = torch.empty(B, D, device=dev, dtype=torch.float32)
        stp = torch.empty(2, B, device=dev, dtype=torch.float32)
        ops.layernorm_fwd(xo, p["visual.ln_post.weight"].data, p["visual.ln_post.bias"].data, rows=B, row_index=rows,
                          out_f32=pooled, mean=stp[0], rstd=stp[1])
        feat = torch.empty(B, geo.embed_dim, device=dev, dtype=torch.float32)
        ops.gemm_f32(pooled, p["visual.proj"].data.t(), feat)
        ctx = dict(saved=saved, patches=patches, x0=x0, st0=st0, xo=xo, rows=rows, pooled=pooled, stp=stp, B=B) if train else None
        return feat, ctx

    def _image_backward(self, c: dict, dfeat: torch.Tensor):
        ar, geo, st = self._arena, self.geo, self._rt["vis"]
        p, g = ar.params, ar.g
        acc = ar.begin_backward()
        B, T, D = c["B"], geo.vision_tokens, geo.vision_width
        M = B * T
        dev = dfeat.device
        dfeat = dfeat.contiguous().float()
        S = ar.loss_scale()                       # fp16 operands: backward on S x dfeat, slots unscaled at the end
        if S != 1.0:
            dfeat = dfeat * S
            ar.scale_grads([n for n in self._rt["vis_names"] if acc[id(g[n])]], S)

        def A(name):
            return acc[id(g[name])]

        ops.gemm_f32(c["pooled"].t(), dfeat.t(), g["visual.proj"], beta=1.0 if A("visual.proj") else 0.0)
        dpooled = torch.empty(B, D, device=dev, dtype=torch.float32)
        ops.gemm_f32(dfeat, p["visual.proj"].data, dpooled)
        Mo = c["xo"].shape[0]                       # B when the last block ran on the class rows only (tail), else M
        dx = torch.zeros(Mo, D, device=dev, dtype=torch.float32)
        dxb = torch.zeros(Mo, D, device=dev, dtype=self.compute_dtype)
        sc = st.scratch
        ops.layernorm_bwd(dpooled, c["xo"], p["visual.ln_post.weight"].data, c["stp"][0], c["stp"][1], rows=B,
                          row_index=c["rows"], dx_out=dx, dx_out_bf16=dxb, dgamma=g["visual.ln_post.weight"],
                          dbeta=g["visual.ln_post.bias"], accumulate=A("visual.ln_post.weight"),
                          ws=sc.floats(ops.layernorm_bwd_ws_floats(B, D)))
        st.grad_hook_enabled = S == 1.0         # (under a loss scale the slots are final only after the unscale below)
        dxb = st.backward(dx, dxb, c["saved"], acc)
        dx = c["saved"]["dx_in"]
        ops.layernorm_bwd(dx, c["x0"], p["visual.ln_pre.weight"].data, c["st0"][0], c["st0"][1], rows=M, dx_out=dx,
                          dx_out_bf16=dxb, dgamma=g["visual.ln_pre.weight"], dbeta=g["visual.ln_pre.bias"],
                          accumulate=A("visual.ln_pre.weight"), ws=sc.floats(ops.layernorm_bwd_ws_floats(M, D)))
        # x0 = patch_out + positional (+ class on slot 0): batch sums of dx0 viewed [B, T*D]
        ops.colsum(dx, g["visual.positional_embedding"], sc.floats(ops.colsum_ws_floats(B, T * D)), R=B, C=T * D,
                   ld=T * D, accumulate=A("visual.positional_embedding"))
        ops.colsum(dx, g["visual.class_embedding"], sc.floats(ops.colsum_ws_floats(B, D)), R=B, C=D, ld=T * D,
                   accumulate=A("visual.class_embedding"))
        gw = g["visual.conv1.weight"].view(D, -1)
        if c["patches"].shape[1] == gw.shape[1]:
            st._wgrad(dxb, c["patches"], gw, M, A("visual.conv1.weight"))
        else:                                      # patch 14: im2col rows are zero-padded 588 -> 592; the pad columns' gradient is dropped
            gpad = torch.empty(D, c["patches"].shape[1], device=dev, dtype=torch.float32)
            st._wgrad(dxb, c["patches"], gpad, M, False)
            if A("visual.conv1.weight"):
                gw.add_(gpad[:, :gw.shape[1]])
            else:
                gw.copy_(gpad[:, :gw.shape[1]])
        ar.scale_grads(self._rt["vis_names"], 1.0 / S)
        ar.publish_grads(self._rt["vis_names"])

    # -- text tower -----------------------------------------------------------------------------
    def prefetch_text(self, text: torch.Tensor) -> None:
        """Tell the model about a token batch BEFORE the step that consumes it is launched (a training loop calls this on batch
        k+1, then launches step k).  The packed text tower sizes its launches by the batch's live-row count; this computes the
        count on a helper stream and mails it to pinned host memory, so the consuming call finds it there instead of draining
        the device queue for it (`.item()` in the middle of a step: the host cannot run ahead of the device across steps).
        Optional: a batch that was not announced is counted with one blocking read, as before.  Keyed by the tensor's storage,
        version and shape - pass the SAME tensor object to the model afterwards."""
        if text.device.type != "cuda" or text.dim() != 2 or text.shape[0] < 2:
            return
        hints = self.__dict__.setdefault("_text_hints", {})
        key = (text.data_ptr(), text._version, tuple(text.shape))
        if key in hints:
            return
        side = self.__dict__.get("_hint_stream")
        if side is None:
            side = self.__dict__["_hint_stream"] = torch.cuda.Stream(device=text.device)
        box = torch.empty(1, dtype=torch.int64, pin_memory=True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            total = (text.detach().argmax(dim=-1) + 1).sum().reshape(1)
            box.copy_(total, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        text.record_stream(side)
        hints[key] = (box, ev, total)
        while len(hints) > 8:                                   # a handful of batches ahead at most
            hints.pop(next(iter(hints)))

    def _live_text_rows(self, text: torch.Tensor, eot: torch.Tensor) -> int:
        """sum(eot + 1): from prefetch_text's mailbox when the batch was announced, else one blocking read."""
        hint = self.__dict__.get("_text_hints", {}).pop((text.data_ptr(), text._version, tuple(text.shape)), None)
        if hint is not None:
            box, ev, _ = hint
            ev.synchronize()                                    # recorded a step ago: normally long complete
            return int(box[0])
        total = (eot + 1).sum()
        duet.pause()                                            # (launching next to the image tower: it goes on while this thread waits)
        try:
            return int(total.item())
        finally:
            duet.resume()

    def _text_forward(self, text: torch.Tensor, train: bool):
        self._ensure_runtime()
        ar, geo, st = self._arena, self.geo, self._rt["txt"]
        ar.refresh_shadows()
        dev = text.device
        if text.dim() != 2 or text.shape[1] != geo.context_length:
            raise RuntimeError(f"encode_text: expected [N,{geo.context_length}] token ids, got {tuple(text.shape)}")
        B, D = text.shape[0], geo.transformer_width
        if B == 0:
            if train:
                raise RuntimeError("encode_text: empty batch in a training step")
            return torch.empty(0, geo.embed_dim, device=dev, dtype=torch.float32), None
        eot = text.detach().argmax(dim=-1)
        # The tower is causal and only the EOT row is pooled: positions after the LAST EOT of the batch influence nothing.
        # trim_text_padding runs the tower on [0, max EOT] only (real captions are ~10-30 tokens of the 77): identical
        # features, fewer rows.  Off by default: it costs one device->host sync per call to learn the length.
        L = geo.context_length
        if getattr(self, "trim_text_padding", False):
            L = max(2, int(eot.max().item()) + 1)
        M = B * L
        tok = text.detach()[:, :L].to(torch.int32).contiguous()
        p = ar.params
        # PACKED rows (round 2; CCLIP_PACK_TEXT=0 / model.pack_text_rows = False disables): caption b only needs positions
        # 0..eot_b - the tower is causal and only the EOT row is pooled, so later rows influence neither the features nor any
        # gradient (their upstream gradient is exactly zero).  The tower runs on the sum(eot_b + 1) live rows, sequences back
        # to back (cu = their row ranges), instead of on B*77: the same features and gradients (the weight gradients' sums lose
        # only exact-zero terms) at about half the rows for captions of uniformly distributed length.  The row count sizes
        # every launch, so the host must know it: from prefetch_text's mailbox (no queue drain) or one blocking read; the
        # embedding stays dense - its live rows are gathered in, and their gradients scattered back, by two index copies.
        rowmap = cu = None
        # (the packed attention kernels hold one sequence per work-group: head_dim 64, <= 128 positions; other geometries run dense)
        if (self._pack_text_rows() and B > 1 and st.geo.head_dim == 64 and L <= 128
                and not torch.cuda.is_current_stream_capturing()):
            Mp = self._live_text_rows(text, eot)                              # the one host-side number: it sizes every launch
            if Mp < M:
                cu = torch.zeros(B + 1, device=dev, dtype=torch.int32)
                cu[1:] = torch.cumsum(eot + 1, 0)
                # packed row j -> dense row b*L + t: b = the sequence whose range [cu[b], cu[b+1]) holds j (index math, no sync)
                j = torch.arange(Mp, device=dev, dtype=torch.int32)
                b = torch.searchsorted(cu[1:], j, right=True)
                rowmap = b * L + (j - cu[b]).long()
        xd = torch.empty(M, D, device=dev, dtype=torch.float32) if (cu is not None or not train) else None
        if cu is not None:
            saved = st.alloc_saved(B, dev, T=L, M=Mp) if train else None
            ops.text_embed(tok.view(-1), p["token_embedding.weight"].data, p["positional_embedding"].data, xd, rows=M, L=L)
            x = saved["xs"][0, 0] if train else torch.empty(Mp, D, device=dev, dtype=torch.float32)
            torch.index_select(xd, 0, rowmap, out=x)
            rows = (cu[1:] - 1).contiguous()                                  # each caption's EOT row = its last packed row
            xo = st.forward(x, B, saved=saved, T=L, cu=cu, tail_rows=rows.long() if self._tail_rows() else None)
        else:
            saved = st.alloc_saved(B, dev, T=L) if train else None
            x = saved["xs"][0, 0] if train else xd
            ops.text_embed(tok.view(-1), p["token_embedding.weight"].data, p["positional_embedding"].data, x, rows=M, L=L)
            # EOT = largest id in the row (openai/CLIP: x[arange, text.argmax(-1)]); integer index math only
            rows = (torch.arange(B, device=dev) * L + eot).to(torch.int32).contiguous()
            xo = st.forward(x, B, saved=saved, T=L, tail_rows=rows.long() if self._tail_rows() else None)
        if self._tail_rows():
            rows = None                                                       # xo is [B, D]: the EOT rows, in order
        rows_dense = (torch.arange(B, device=dev) * L + eot).to(torch.int32)
        pooled = torch.empty(B, D, device=dev, dtype=torch.float32)
        stp = torch.empty(2, B, device=dev, dtype=torch.float32)
        ops.layernorm_fwd(xo, p["ln_final.weight"].data, p["ln_final.bias"].data, rows=B, row_index=rows, out_f32=pooled,
                          mean=stp[0], rstd=stp[1])
        feat = torch.empty(B, geo.embed_dim, device=dev, dtype=torch.float32)
        ops.gemm_f32(pooled, p["text_projection"].data.t(), feat)
        ctx = dict(saved=saved, tok=tok, xo=xo, rows=rows, rows_dense=rows_dense, rowmap=rowmap if cu is not None else None,
                   Mp=x.shape[0], pooled=pooled, stp=stp, B=B, L=L) if train else None
        if train and ops.SCATTER_DETERMINISTIC and p["token_embedding.weight"].requires_grad:     # (a frozen table needs no gradient tables)
            # The index tables of the deterministic embedding-gradient sum depend on the token ids only: built NOW on a helper
            # stream (under the forward pass's GEMMs) instead of at the end of the backward pass, where their ~25 small
            # launches were 1.2 ms of the step's critical path.
            # positions after a row's EOT carry an exactly-zero gradient (causal tower, EOT pooling): dropped from the row list
            cur = torch.cuda.current_stream()
            aux = self._rt.get("aux_stream")
            if aux is None:
                aux = self._rt["aux_stream"] = torch.cuda.Stream(device=dev)
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                keep = (torch.arange(L, device=dev, dtype=torch.int32)[None, :] <= (rows_dense - torch.arange(B, device=dev, dtype=torch.int32) * L)[:, None]).reshape(-1)
                tables = ops.embed_scatter_tables(tok.view(-1), p["token_embedding.weight"].shape[0], rows=M, keep=keep)
                ev = torch.cuda.Event()
                ev.record(aux)
            ctx["scatter_tables"], ctx["scatter_ready"] = tables, ev
        return feat, ctx

    def _text_backward(self, c: dict, dfeat: torch.Tensor):
        ar, geo, st = self._arena, self.geo, self._rt["txt"]
        p, g = ar.params, ar.g
        acc = ar.begin_backward()
        B, L, D = c["B"], c["L"], geo.transformer_width
        M = B * L
        dev = dfeat.device
        dfeat = dfeat.contiguous().float()
        S = ar.loss_scale()
        if S != 1.0:
            dfeat = dfeat * S
            ar.scale_grads([n for n in self._rt["txt_names"] if acc[id(g[n])]], S)

        def A(name):
            return acc[id(g[name])]

        ops.gemm_f32(c["pooled"].t(), dfeat.t(), g["text_projection"], beta=1.0 if A("text_projection") else 0.0)
        dpooled = torch.empty(B, D, device=dev, dtype=torch.float32)
        ops.gemm_f32(dfeat, p["text_projection"].data, dpooled)
        Mp = c["Mp"]                                  # rows the tower ran on (= M unless the batch was packed)
        Mo = c["xo"].shape[0]                         # B when the last block ran on the EOT rows only (tail)
        dx = torch.zeros(Mo, D, device=dev, dtype=torch.float32)
        dxb = torch.zeros(Mo, D, device=dev, dtype=self.compute_dtype)
        sc = st.scratch
        ops.layernorm_bwd(dpooled, c["xo"], p["ln_final.weight"].data, c["stp"][0], c["stp"][1], rows=B,
                          row_index=c["rows"], dx_out=dx, dx_out_bf16=dxb, dgamma=g["ln_final.weight"],
                          dbeta=g["ln_final.bias"], accumulate=A("ln_final.weight"),
                          ws=sc.floats(ops.layernorm_bwd_ws_floats(B, D)))
        st.grad_hook_enabled = S == 1.0
        dxb = st.backward(dx, dxb, c["saved"], acc)
        dx = c["saved"]["dx_in"]
        if c["rowmap"] is not None:                   # back to dense rows for the (dense) embedding gradients; dead rows stay zero
            dxp, dx = dx, torch.zeros(M, D, device=dev, dtype=torch.float32)
            dx.index_copy_(0, c["rowmap"], dxp)
        gpos = g["positional_embedding"]
        if L < geo.context_length and not A("positional_embedding"):
            gpos[L:].zero_()                      # trimmed positions received no gradient
        ops.colsum(dx, gpos.view(-1)[:L * D], sc.floats(ops.colsum_ws_floats(B, L * D)), R=B, C=L * D, ld=L * D,
                   accumulate=A("positional_embedding"))
        if not A("token_embedding.weight"):
            g["token_embedding.weight"].zero_()
        if c.get("scatter_tables") is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(c["scatter_ready"])
            for t in c["scatter_tables"]:
                t.record_stream(cur)               # allocated on the helper stream, consumed here
            ops.embed_scatter_add(c["tok"].view(-1), dx, g["token_embedding.weight"], rows=M, tables=c["scatter_tables"], scratch=sc.floats)
        else:
            # positions after a row's EOT carry an exactly-zero gradient (causal tower, EOT pooling): drop them from the row list
            keep = (torch.arange(L, device=dev, dtype=torch.int32)[None, :] <= (c["rows_dense"] - torch.arange(B, device=dev, dtype=torch.int32) * L)[:, None]).reshape(-1)
            ops.embed_scatter_add(c["tok"].view(-1), dx, g["token_embedding.weight"], rows=M, keep=keep, scratch=sc.floats)
        ar.scale_grads(self._rt["txt_names"], 1.0 / S)
        ar.publish_grads(self._rt["txt_names"])

    # -- public API (same names / argument meaning as openai/CLIP) --------------------------------
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        _require_cuda(image, "encode_image")
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.visual.parameters()):
            self._ensure_runtime()
            return _ImageTower.apply(self, image, *[self._arena.params[n] for n in self._rt["vis_names"]])
        return self._image_forward_lanes(image)

    def _image_forward_lanes(self, image: torch.Tensor, streams=None) -> torch.Tensor:
        """Inference: a large batch runs as TWO half batches side by side on two HIP streams - independent kernels fill each
        other's tile tails and epilogue phases (measured on ViT-B/32, 1024 images: 13.40 -> 12.47 ms, bit-identical features;
        four quarters: no further gain; tools/micro/half_batch_overlap.py).  Below ~32 k token rows, under stream capture, or with
        CCLIP_IMAGE_LANES=1 the batch runs whole.  `streams`: two streams to use (default: the model's lane streams)."""
        import os
        B = image.shape[0]
        rows = B * self.geo.vision_tokens
        if (B < 2 or rows < 32768 or os.environ.get("CCLIP_IMAGE_LANES", "2") == "1" or torch.cuda.is_current_stream_capturing()):
            return self._image_forward(image, train=False)[0]
        self._ensure_runtime()
        self._arena.refresh_shadows()                  # once, on the caller's stream, before the fork
        vis = self._rt["vis"]
        if vis.fold_enabled():
            vis._fold_weights(self._arena._stamp)      # (the folded-LayerNorm weight copies too: built here, not inside a lane)
        if vis.fp8 and vis._fp8_weights is None:       # lazily built e4m3 weights: build them HERE, before the lanes fork - lane 1 would
            vis.quantise_weights_fp8()                 # otherwise read weights lane 0 is still quantising on the other stream
        if streams is None:
            if self._rt.get("lane_streams") is None:
                self._rt["lane_streams"] = (torch.cuda.Stream(device=image.device), torch.cuda.Stream(device=image.device))
            streams = self._rt["lane_streams"]
        cur = torch.cuda.current_stream()
        h = (B + 1) // 2
        outs = []
        for s_, part in zip(streams, (image[:h], image[h:])):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs.append(self._image_forward(part, train=False)[0])
        for s_, o in zip(streams, outs):
            cur.wait_stream(s_)
            o.record_stream(cur)
        return torch.cat(outs)

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        _require_cuda(text, "encode_text")
        self._ensure_runtime()
        names = self._rt["txt_names"]
        if torch.is_grad_enabled() and any(self._arena.params[n].requires_grad for n in names):
            return _TextTower.apply(self, text, *[self._arena.params[n] for n in names])
        return self._text_forward(text, train=False)[0]

    def _tower_streams(self, device):
        if self._rt.get("streams") is None:
            self._rt["streams"] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return self._rt["streams"]

    def encode_image_text(self, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Both towers, on two HIP streams (CCLIP_TOWER_STREAMS=1 disables): they are independent until the logits,
        and their kernels' store phases and MFMA phases interleave on the CUs (~5 % on the bs=1024 step).  In a
        training step the pair is ONE autograd node (_BothTowers) that forks to the two streams and joins back onto the
        caller's stream in forward and in backward, so every gradient slot of the arena is complete, in stream order,
        before anything the caller enqueues after `backward()` (optimizer, gradient all-reduce, a .grad reader)."""
        import os
        _require_cuda(image, "encode_image_text")
        if os.environ.get("CCLIP_TOWER_STREAMS", "2") != "2":
            return self.encode_image(image), self.encode_text(text)
        self._ensure_runtime()
        self._arena.refresh_shadows()                # on the current stream, before the fork
        names = self._rt["vis_names"] + self._rt["txt_names"]
        train = torch.is_grad_enabled()
        if train and all(self._arena.params[n].requires_grad for n in names):
            return _BothTowers.apply(self, image, text, *[self._arena.params[n] for n in names])
        if train and any(self._arena.params[n].requires_grad for n in names):
            return self.encode_image(image), self.encode_text(text)      # partly frozen model: one stream, plain nodes
        s0, s1 = self._tower_streams(image.device)
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur); s1.wait_stream(cur)
        # (the image batch whole: the text tower is the partner here - image lanes on top measured 18.3 -> 19.4 ms)
        (fi, _), (ft, _) = _side_by_side(image.device, s0, lambda: self._image_forward(image, train=False),
                                         s1, lambda: self._text_forward(text, train=False))
        cur.wait_stream(s0); cur.wait_stream(s1)
        fi.record_stream(cur); ft.record_stream(cur)
        return fi, ft

    def forward(self, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if image.shape[0] == 0 or text.shape[0] == 0:      # nothing to score: empty logits of the right shape
            _require_cuda(image, "forward")
            li = torch.empty(image.shape[0], text.shape[0], device=image.device, dtype=torch.float32)
            return li, li.t()
        fi, ft = self.encode_image_text(image, text)
        logits_per_image = _Logits.apply(fi, ft, self.logit_scale)
        return logits_per_image, logits_per_image.t()


# ------------------------------------------------------------------------------------------------
# autograd nodes: forward = kernel sequence; backward = hand-written kernel sequence that writes the
# arena's gradient slots directly (parameter inputs get None back; .grad is pointed at the slots).
# ------------------------------------------------------------------------------------------------
def _side_by_side(device, s0, f0, s1, f1):
    """f0's launches on stream s0 and f1's on s1: f0 entirely, then f1 (default), or - CCLIP_TOWER_INTERLEAVE=1 - by two host
    threads taking strict turns block by block (cclip_hip/duet.py), so that both streams are fed from the start.  The turns
    matter when the host is slow next to the device (under rocprofv3 the text tower's backward started 14 ms after the image
    tower's); on an unencumbered host the launches are far enough ahead either way: 50.10-50.14 ms in turns, 49.7-50.0 in sequence
    (ViT-B/32 bs 1024) - hence not the default."""
    import os
    if os.environ.get("CCLIP_TOWER_INTERLEAVE", "0") != "1":
        with torch.cuda.stream(s0):
            r0 = f0()
        with torch.cuda.stream(s1):
            r1 = f1()
        return r0, r1

    def first():
        with torch.no_grad(), torch.cuda.stream(s0):
            return f0()

    def second():
        with torch.no_grad(), torch.cuda.stream(s1):
            return f1()

    return duet.run(first, second, setup_second=lambda: torch.cuda.set_device(device))


class _ImageTower(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: CLIP, image, *params):
        feat, c = model._image_forward(image, train=True)
        ctx.model, ctx.c, ctx.n = model, c, len(params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        ctx.model._image_backward(ctx.c, dfeat)
        ctx.c = None
        return (None, None) + (None,) * ctx.n


class _TextTower(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: CLIP, text, *params):
        feat, c = model._text_forward(text, train=True)
        ctx.model, ctx.c, ctx.n = model, c, len(params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        ctx.model._text_backward(ctx.c, dfeat)
        ctx.c = None
        return (None, None) + (None,) * ctx.n


class _BothTowers(torch.autograd.Function):
    """encode_image + encode_text of one training step on the two tower streams, with an EXPLICIT fork / join around both
    the forward and the backward kernel sequences.  The backward writes parameter gradients into the arena as a side effect
    (parameter inputs get None back), so nothing in autograd would order a consumer of those slots behind the tower
    streams; the join at the end of backward() does, on the stream autograd runs this node on (= the stream of forward)."""

    @staticmethod
    def forward(ctx, model: CLIP, image, text, *params):
        s0, s1 = model._tower_streams(image.device)
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur); s1.wait_stream(cur)
        (fi, ci), (ft, ct) = _side_by_side(image.device, s0, lambda: model._image_forward(image, train=True),
                                           s1, lambda: model._text_forward(text, train=True))
        cur.wait_stream(s0); cur.wait_stream(s1)
        fi.record_stream(cur); ft.record_stream(cur)
        ctx.model, ctx.ci, ctx.ct, ctx.n = model, ci, ct, len(params)
        ctx.set_materialize_grads(False)
        return fi, ft

    @staticmethod
    def backward(ctx, dfi, dft):
        model = ctx.model
        dev = (dfi if dfi is not None else dft).device
        s0, s1 = model._tower_streams(dev)
        cur = torch.cuda.current_stream()
        model._arena.refresh_transposed()          # before the fork: both towers' dgrad GEMMs read the transposed shadows
        s0.wait_stream(cur); s1.wait_stream(cur)
        if dfi is not None and dft is not None:
            dfi.record_stream(s0); dft.record_stream(s1)
            _side_by_side(dev, s0, lambda: model._image_backward(ctx.ci, dfi), s1, lambda: model._text_backward(ctx.ct, dft))
        elif dfi is not None:
            with torch.cuda.stream(s0):
                dfi.record_stream(s0)
                model._image_backward(ctx.ci, dfi)
        elif dft is not None:
            with torch.cuda.stream(s1):
                dft.record_stream(s1)
                model._text_backward(ctx.ct, dft)
        cur.wait_stream(s0); cur.wait_stream(s1)      # the join: every gradient slot is final on the caller's stream
        ctx.ci = ctx.ct = None
        return (None, None, None) + (None,) * ctx.n


def normalized_logits(fi: torch.Tensor, ft: torch.Tensor, logit_scale: torch.Tensor):
    """L2-normalise both feature sets and form exp(logit_scale) * I @ T^T in exact fp32 (kernels only)."""
    dev = fi.device
    Ni, E = fi.shape
    Nt = ft.shape[0]
    i_n, t_n = torch.empty_like(fi), torch.empty_like(ft)
    inv_i = torch.empty(Ni, device=dev, dtype=torch.float32)
    inv_t = torch.empty(Nt, device=dev, dtype=torch.float32)
    ops.l2norm_fwd(fi, i_n, inv_i)
    ops.l2norm_fwd(ft, t_n, inv_t)
    logits = torch.empty(Ni, Nt, device=dev, dtype=torch.float32)
    ops.gemm_f32(i_n, t_n, logits, alpha_log_dev=logit_scale)
    return logits, i_n, t_n, inv_i, inv_t


class _Logits(torch.autograd.Function):
    """CLIP.forward's tail (openai/CLIP; call sites CLIP/train.py:161, CLIP/predict.py:46)."""

    @staticmethod
    def forward(ctx, fi, ft, logit_scale):
        fi, ft = fi.contiguous().float(), ft.contiguous().float()
        ls = logit_scale.detach().float().reshape(1).contiguous()
        logits, i_n, t_n, inv_i, inv_t = normalized_logits(fi, ft, ls)
        ctx.saved = (i_n, t_n, inv_i, inv_t, ls, logits)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        i_n, t_n, inv_i, inv_t, ls, logits = ctx.saved
        dl = dlogits.contiguous().float()
        d_in, d_tn = torch.empty_like(i_n), torch.empty_like(t_n)
        ops.gemm_f32(dl, t_n.t(), d_in, alpha_log_dev=ls)          # s * dL @ Tn
        ops.gemm_f32(dl.t(), i_n.t(), d_tn, alpha_log_dev=ls)      # s * dL^T @ In
        dfi, dft = torch.empty_like(i_n), torch.empty_like(t_n)
        ops.l2norm_bwd(d_in, i_n, inv_i, dfi)
        ops.l2norm_bwd(d_tn, t_n, inv_t, dft)
        dscale = torch.empty(1, device=dl.device, dtype=torch.float32)
        ops.reduce_dot(dl, logits, dscale)                          # d/d(log s) of s*C = logits
        ctx.saved = None
        return dfi, dft, dscale.reshape(())


def build_model(state_dict: Dict[str, torch.Tensor], compute_dtype: torch.dtype = torch.bfloat16) -> CLIP:
    """openai/CLIP's build_model(): geometry from tensor shapes, then load (CLIP/train.py:111 round trip)."""
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model = CLIP(geometry_from_state_dict(sd), compute_dtype)
    model.load_state_dict(sd)
    return model.eval()
