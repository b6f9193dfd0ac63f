"""Prefix-caption model of /root/reference/CLIP_prefix_caption/train.py, rebuilt on the HIP kernels:

  MLP                 train.py:110-123   Linear(512, 768P/2) -> Tanh -> Linear(768P/2, 768P)
  ClipCaptionModel    train.py:251-283   cat(clip_project(prefix), wte(cat(attribute, tokens))) -> GPT-2 -> logits
  ClipCaptionPrefix   train.py:286-294   mapper-only training (GPT-2 frozen); the reference's version raises
                                         (`self.gpt` vs `self.model`) - fixed here, not copied.
  GPT2LMHeadModel     the piece `transformers` contributes at train.py:275; same state_dict keys, so the
                      files written by torch.save(model.state_dict()) (train.py:371-381) load unchanged.

Same call surface as the reference (`model(tokens, prefix, attribute, mask)` -> object with `.logits`;
`model.clip_project(prefix)`; `model.model.transformer.wte(ids)`; `model.model(inputs_embeds=..., attention_mask=...)`),
plus `caption_loss()` - the fused form of train.py:354-357 that only ever materialises the sliced logits rows.

Modules are parameter holders; arithmetic = cclip_hip launches (BlockStack with Conv1D layout + gelu_new +
causal & key-padding attention).  fp32 masters in a flat arena, bf16 MFMA operands, fp32 residual stream.
`TransformerMapper` (--mapping_type transformer, train.py:126-248) runs on the same BlockStack with head_dim 96
(generic small-attention kernel), ReLU and MLP ratio 2.
"""
from __future__ import annotations

import os
import warnings
from enum import Enum
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from cclip_hip import ops
from cclip_hip.arena import ParamArena
from cclip_hip.stack import BlockStack, BlockWeights, Scratch, StackGeometry

from .weights import GPT2_MODELS, CaptionGeometry, init_caption_state_dict, init_transformer_mapper_state_dict


class MappingType(Enum):
    MLP = "mlp"
    Transformer = "transformer"


class _Holder(nn.Module):
    pass


class _Affine(_Holder):       # LayerNorm / Linear / Conv1D parameter pair
    def __init__(self, w_shape, b_shape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*w_shape))
        self.bias = nn.Parameter(torch.zeros(*b_shape))


def _reset_linear(weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> None:
    """nn.Linear.reset_parameters (what the reference's mappers get at construction): U(+-1/sqrt(fan_in)) for both."""
    bound = weight.shape[1] ** -0.5
    with torch.no_grad():
        weight.uniform_(-bound, bound)
        if bias is not None:
            bias.uniform_(-bound, bound)


class _Sequential(_Holder):
    """Key layout of nn.Sequential(Linear, Tanh, Linear): parameters live at .0 and .2"""

    def __init__(self, sizes):
        super().__init__()
        for i in range(len(sizes) - 1):
            self.add_module(str(2 * i), _Affine((sizes[i + 1], sizes[i]), (sizes[i + 1],)))
            _reset_linear(getattr(self, str(2 * i)).weight, getattr(self, str(2 * i)).bias)


class MLP(_Holder):
    """train.py:110-123.  Only the default 3-size / Tanh form is on the hot path."""

    def __init__(self, sizes: Tuple[int, ...], bias=True, act=nn.Tanh):
        super().__init__()
        if len(sizes) != 3 or not bias or act is not nn.Tanh:
            raise NotImplementedError("HIP mapper implements the reference's default MLP((in, hidden, out), Tanh)")
        self.sizes = tuple(sizes)
        self.model = _Sequential(sizes)
        self._owner = None        # set by ClipCaptionModel: the arena lives on the top-level module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._owner is None:
            raise RuntimeError("MLP must be used as ClipCaptionModel.clip_project")
        return self._owner()._mapper_only(x)


class _Weight(_Holder):      # bias-free Linear (TransformerMapper q / kv projections, train.py:188,192)
    def __init__(self, n_out, n_in):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        _reset_linear(self.weight)


class _MapperAttention(_Holder):
    def __init__(self, d):
        super().__init__()
        # registration order matters: to_queries.weight and to_keys_values.weight are adjacent in the flat arena, so
        # one packed [3D, D] view serves the fused q|k|v projection GEMM, its wgrad and its 16-bit shadow
        self.to_queries = _Weight(d, d)
        self.to_keys_values = _Weight(2 * d, d)
        self.project = _Affine((d, d), (d,))
        _reset_linear(self.project.weight, self.project.bias)


class _MapperMLP(_Holder):
    def __init__(self, d, h):
        super().__init__()
        self.fc1 = _Affine((h, d), (h,))
        self.fc2 = _Affine((d, h), (d,))
        _reset_linear(self.fc1.weight, self.fc1.bias)
        _reset_linear(self.fc2.weight, self.fc2.bias)


class _MapperLayer(_Holder):
    def __init__(self, d, mlp_ratio=2.0):
        super().__init__()
        self.norm1 = _Affine((d,), (d,))
        self.attn = _MapperAttention(d)
        self.norm2 = _Affine((d,), (d,))
        self.mlp = _MapperMLP(d, int(d * mlp_ratio))
        with torch.no_grad():
            self.norm1.weight.fill_(1.0)
            self.norm2.weight.fill_(1.0)


class _MapperTransformer(_Holder):
    def __init__(self, d, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([_MapperLayer(d) for _ in range(num_layers)])


class TransformerMapper(_Holder):
    """train.py:233-248: Linear(dim_clip, clip_length*D) -> cat(learned prefix_const) -> 8-head pre-LN Transformer
    (mlp_ratio 2, ReLU, bias-free q/kv) -> the last prefix_length tokens."""

    def __init__(self, dim_clip: int, dim_embedding: int, prefix_length: int, clip_length: int, num_layers: int = 8):
        super().__init__()
        self.clip_length, self.prefix_length, self.dim, self.num_heads = clip_length, prefix_length, dim_embedding, 8
        self.sizes = (dim_clip,)
        self.transformer = _MapperTransformer(dim_embedding, num_layers)
        self.linear = _Affine((clip_length * dim_embedding, dim_clip), (clip_length * dim_embedding,))
        _reset_linear(self.linear.weight, self.linear.bias)
        self.prefix_const = nn.Parameter(torch.randn(prefix_length, dim_embedding))
        self._owner = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._owner is None:
            raise RuntimeError("TransformerMapper must be used as ClipCaptionModel.clip_project")
        return self._owner()._mapper_only(x)


class _Embedding(_Holder):
    def __init__(self, n, d, owner_ref=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d))

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """wte(ids): fp32 embedding rows through the gather kernel (test.py:540, application.py:105 call this)."""
        if not ids.is_cuda:
            raise RuntimeError("wte: HIP path only (no CPU fallback)")
        flat = ids.reshape(-1).to(torch.int32).contiguous()
        out = torch.empty(flat.numel(), self.weight.shape[1], device=ids.device, dtype=torch.float32)
        ops.text_embed(flat, self.weight.data, None, out, rows=flat.numel(), L=flat.numel())
        return out.view(*ids.shape, -1)


class _GPT2Attention(_Holder):
    def __init__(self, d):
        super().__init__()
        self.c_attn = _Affine((d, 3 * d), (3 * d,))      # Conv1D: [in, out]
        self.c_proj = _Affine((d, d), (d,))


class _GPT2MLP(_Holder):
    def __init__(self, d):
        super().__init__()
        self.c_fc = _Affine((d, 4 * d), (4 * d,))
        self.c_proj = _Affine((4 * d, d), (d,))


class _GPT2Block(_Holder):
    def __init__(self, d):
        super().__init__()
        self.ln_1 = _Affine((d,), (d,))
        self.attn = _GPT2Attention(d)
        self.ln_2 = _Affine((d,), (d,))
        self.mlp = _GPT2MLP(d)


class _GPT2Transformer(_Holder):
    def __init__(self, geo: CaptionGeometry):
        super().__init__()
        self.wte = _Embedding(geo.vocab_size, geo.n_embd)
        self.wpe = _Embedding(geo.n_positions, geo.n_embd)
        self.h = nn.ModuleList([_GPT2Block(geo.n_embd) for _ in range(geo.n_layer)])
        self.ln_f = _Affine((geo.n_embd,), (geo.n_embd,))


class _LMHead(_Holder):
    def __init__(self, wte: _Embedding):
        super().__init__()
        self.weight = wte.weight          # tied, as GPT2LMHeadModel ties lm_head to wte


class GPT2LMHeadModel(_Holder):
    def __init__(self, geo: CaptionGeometry):
        super().__init__()
        self.config = SimpleNamespace(vocab_size=geo.vocab_size, n_embd=geo.n_embd, n_layer=geo.n_layer, n_head=geo.n_head,
                                      n_positions=geo.n_positions)
        self.geo = geo
        self.transformer = _GPT2Transformer(geo)
        self.lm_head = _LMHead(self.transformer.wte)
        self._owner = None
        # Checkpoint wire format: every reference script strict-loads a full ClipCaptionModel state_dict
        # (train.py:320, test.py:337/598, predict.py:61).  transformers 4.x releases that still ship `AdamW` (train.py:6)
        # register GPT2Attention's causal-mask buffers as persistent, so such files carry
        # `<prefix>transformer.h.N.attn.bias` / `.attn.masked_bias`; newer ones omit them and may omit the tied
        # `lm_head.weight`.  Both load unchanged: the buffers are dropped (the mask is applied inside the attention kernel),
        # the tied head is filled from wte.
        self._register_load_state_dict_pre_hook(self._normalise_hf_gpt2_keys)

    @staticmethod
    def _normalise_hf_gpt2_keys(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in [k for k in state_dict if k.startswith(prefix) and k.endswith((".attn.bias", ".attn.masked_bias"))]:
            del state_dict[k]
        wte = prefix + "transformer.wte.weight"
        if wte in state_dict:
            state_dict.setdefault(prefix + "lm_head.weight", state_dict[wte])

    @classmethod
    def from_pretrained(cls, name: str, **kw) -> "GPT2LMHeadModel":
        """The reference fetches weights here (train.py:275).  Offline: `name` may be a directory / file holding a
        state_dict (`pytorch_model.bin`, read with weights_only=True); a known name without local files gets seeded
        synthetic weights and a warning."""
        geo = GPT2_MODELS.get(name, GPT2_MODELS["ckiplab/gpt2-base-chinese"])
        m = cls(geo)
        path = name if os.path.isfile(name) else os.path.join(name, "pytorch_model.bin")
        if os.path.isfile(path):
            m.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))   # (HF buffer keys: pre-hook above)
        else:
            warnings.warn(f"GPT2LMHeadModel.from_pretrained({name!r}): no local weights and no network - seeded synthetic init")
            full = init_caption_state_dict(geo, 567)
            m.load_state_dict({k[len("model."):]: v for k, v in full.items() if k.startswith("model.")})
        return m

    def forward(self, inputs_embeds=None, attention_mask=None, labels=None, input_ids=None, past_key_values=None,
                use_cache: bool = False, **kw):
        """`.logits` of GPT2LMHeadModel(inputs_embeds=..., attention_mask=...).  With use_cache / past_key_values (a
        KVCache) only the NEW positions are computed: prefill of a whole prefix when past_key_values is None, one
        token per sequence afterwards (what HF's own use_cache does; the reference's generate loops never pass it and
        recompute the full sequence every step, test.py:381)."""
        if self._owner is None:
            raise RuntimeError("GPT2LMHeadModel must be used as ClipCaptionModel.model")
        if labels is not None:
            raise NotImplementedError("labels= is never used by the reference's live code path (train.py:354 passes None)")
        if inputs_embeds is None:
            inputs_embeds = self.transformer.wte(input_ids)
        if use_cache or past_key_values is not None:
            if attention_mask is not None:
                raise NotImplementedError("KV-cached decode: no key padding (the generate loops pass none)")
            logits, cache = self._owner()._cached_logits(inputs_embeds, past_key_values)
            return SimpleNamespace(logits=logits, past_key_values=cache)
        return SimpleNamespace(logits=self._owner()._logits_from_embeds(inputs_embeds, attention_mask))


class KVCache:
    """Per-layer keys / values of the positions processed so far: k, v [n_layer, n_seq, max_len, n_embd] in the
    compute dtype, `length` positions valid.  reorder(idx) is beam search's `generated = generated[next_tokens_source]`
    (test.py:416) applied to the cache."""

    def __init__(self, n_layer: int, n_seq: int, max_len: int, width: int, device, dtype):
        self.k = torch.empty(n_layer, n_seq, max_len, width, device=device, dtype=dtype)
        self.v = torch.empty_like(self.k)
        self.length = 0

    @property
    def n_seq(self) -> int:
        return self.k.shape[1]

    @property
    def max_len(self) -> int:
        return self.k.shape[2]

    def reorder(self, idx: torch.Tensor) -> "KVCache":
        idx = idx.to(self.k.device).long()
        L = self.length
        out = KVCache.__new__(KVCache)
        out.k = torch.empty(self.k.shape[0], idx.numel(), *self.k.shape[2:], device=self.k.device, dtype=self.k.dtype)
        out.v = torch.empty_like(out.k)
        out.k[:, :, :L] = self.k[:, idx, :L]
        out.v[:, :, :L] = self.v[:, idx, :L]
        out.length = L
        return out

    def expand(self, n: int) -> "KVCache":
        """one sequence -> n identical ones (`generated.expand(beam_size, ...)`, test.py:398)"""
        assert self.n_seq == 1
        return self.reorder(torch.zeros(n, dtype=torch.long))


_GPT_KEYS = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "w_qkv": "attn.c_attn.weight", "b_qkv": "attn.c_attn.bias",
             "w_o": "attn.c_proj.weight", "b_o": "attn.c_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
             "w_fc": "mlp.c_fc.weight", "b_fc": "mlp.c_fc.bias", "w_proj": "mlp.c_proj.weight", "b_proj": "mlp.c_proj.bias"}
_MATS = ("w_qkv", "w_o", "w_fc", "w_proj")


class ClipCaptionModel(nn.Module):
    def __init__(self, prefix_length: int, clip_length: Optional[int] = None, prefix_size: int = 512, num_layers: int = 8,
                 mapping_type: MappingType = MappingType.MLP, gpt2_type: str = ""):
        super().__init__()
        import weakref
        self.prefix_length = prefix_length
        self.model = GPT2LMHeadModel.from_pretrained(gpt2_type) if isinstance(gpt2_type, str) else GPT2LMHeadModel(gpt2_type)
        self.model_embedding_size = self.model.transformer.wte.weight.shape[1]
        d = self.model_embedding_size
        if mapping_type == MappingType.MLP:
            self.clip_project = MLP((prefix_size, (d * prefix_length) // 2, d * prefix_length))
        else:
            self.clip_project = TransformerMapper(prefix_size, d, prefix_length, clip_length or prefix_length, num_layers)
        ref = weakref.ref(self)
        self.model._owner = ref
        self.clip_project._owner = ref
        self._arena: Optional[ParamArena] = None
        self._stack: Optional[BlockStack] = None
        self.compute_dtype = torch.bfloat16

    def half(self):
        return self.set_compute_dtype(torch.float16)

    def bfloat16(self):
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        return self

    def set_compute_dtype(self, dtype: torch.dtype):
        assert dtype in (torch.bfloat16, torch.float16)
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._arena = self._stack = None
        return self

    @property
    def gpt(self):            # test.py / application.py name the same submodule `gpt`
        return self.model

    def get_dummy_token(self, batch_size: int, device) -> torch.Tensor:
        return torch.zeros(batch_size, self.prefix_length, dtype=torch.int64, device=device)

    def initialize_parameters(self, seed: int = 567):
        geo = self.model.geo
        geo = CaptionGeometry(vocab_size=geo.vocab_size, n_embd=geo.n_embd, n_layer=geo.n_layer, n_head=geo.n_head,
                              n_positions=geo.n_positions, prefix_length=self.prefix_length, attribute_length=0,
                              prefix_size=self.clip_project.sizes[0])
        sd = init_caption_state_dict(geo, seed)
        if isinstance(self.clip_project, TransformerMapper):
            sd = {k: v for k, v in sd.items() if not k.startswith("clip_project.")}
            sd.update(init_transformer_mapper_state_dict(geo, self.clip_project.clip_length, len(self.clip_project.transformer.layers), seed))
        self.load_state_dict(sd)
        return self

    # ---- runtime ----
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._arena = self._stack = None
        return out

    @property
    def arena(self) -> ParamArena:
        self._ensure_runtime()
        return self._arena

    def _ensure_runtime(self):
        dev = self.model.transformer.wte.weight.device
        if dev.type != "cuda":
            raise RuntimeError(f"ClipCaptionModel parameters are on {dev}: the HIP kernels are the only compute path")
        if self._arena is not None and self._arena.intact():
            return
        ar = ParamArena(self, dev, self.compute_dtype)
        g = self.model.geo
        blocks = []
        for i in range(g.n_layer):
            kw, grads = {}, {}
            for f, key in _GPT_KEYS.items():
                name = f"model.transformer.h.{i}.{key}"
                kw[f] = ar.b[name] if f in _MATS else ar.params[name].data
                grads[f] = ar.g[name] if ar.params[name].requires_grad else None
            blocks.append(BlockWeights(grads=grads if all(v is not None for v in grads.values()) else None, **kw))
        self._arena = ar
        self._stack = BlockStack(StackGeometry(g.n_embd, g.n_head, 0, False, ops.ACT_GELU_NEW, True), blocks, Scratch(dev),
                                 self.compute_dtype)
        self._mstack = None
        if isinstance(self.clip_project, TransformerMapper):
            mp = self.clip_project
            D = mp.dim
            mblocks = []
            for i in range(len(mp.transformer.layers)):
                q = f"clip_project.transformer.layers.{i}."
                oq, okv = ar.offsets[q + "attn.to_queries.weight"], ar.offsets[q + "attn.to_keys_values.weight"]
                assert okv == oq + D * D, "q and kv projection weights must be adjacent in the arena"
                packed = lambda flat: flat[oq:oq + 3 * D * D].view(3 * D, D)
                trainable = ar.params[q + "attn.to_queries.weight"].requires_grad
                grads = None
                if trainable:
                    grads = dict(ln1_w=ar.g[q + "norm1.weight"], ln1_b=ar.g[q + "norm1.bias"], w_qkv=packed(ar.gflat), b_qkv=None,
                                 w_o=ar.g[q + "attn.project.weight"], b_o=ar.g[q + "attn.project.bias"],
                                 ln2_w=ar.g[q + "norm2.weight"], ln2_b=ar.g[q + "norm2.bias"],
                                 w_fc=ar.g[q + "mlp.fc1.weight"], b_fc=ar.g[q + "mlp.fc1.bias"],
                                 w_proj=ar.g[q + "mlp.fc2.weight"], b_proj=ar.g[q + "mlp.fc2.bias"])
                P_ = ar.params
                mblocks.append(BlockWeights(
                    ln1_w=P_[q + "norm1.weight"].data, ln1_b=P_[q + "norm1.bias"].data, w_qkv=packed(ar.bflat), b_qkv=None,
                    w_o=ar.b[q + "attn.project.weight"], b_o=P_[q + "attn.project.bias"].data,
                    ln2_w=P_[q + "norm2.weight"].data, ln2_b=P_[q + "norm2.bias"].data,
                    w_fc=ar.b[q + "mlp.fc1.weight"], b_fc=P_[q + "mlp.fc1.bias"].data,
                    w_proj=ar.b[q + "mlp.fc2.weight"], b_proj=P_[q + "mlp.fc2.bias"].data, grads=grads))
            hidden = mp.transformer.layers[0].mlp.fc1.weight.shape[0]
            self._mstack = BlockStack(StackGeometry(D, mp.num_heads, mp.clip_length + mp.prefix_length, True, ops.ACT_RELU,
                                                    False, head_dim=D // mp.num_heads, hidden=hidden), mblocks, Scratch(dev),
                                      self.compute_dtype)

    # ---- mapper ----
    def _tmapper_forward(self, prefix: torch.Tensor, train: bool):
        """TransformerMapper.forward (train.py:235-240) on the generalised BlockStack (head_dim 96, ReLU, ratio 2)."""
        ar, mp, st = self._arena, self.clip_project, self._mstack
        B, dev = prefix.shape[0], prefix.device
        CL, P, D = mp.clip_length, mp.prefix_length, mp.dim
        T = CL + P
        pb = torch.empty(B, mp.sizes[0], device=dev, dtype=self.compute_dtype)
        ops.cast_f32_to_bf16(prefix.detach().float().contiguous(), pb)
        lin = torch.empty(B, CL * D, device=dev, dtype=torch.float32)
        ops.gemm_bf16(pb, ar.b["clip_project.linear.weight"], bias=ar.params["clip_project.linear.bias"].data, out_f32=lin)
        saved = st.alloc_saved(B, dev, T=T) if train else None
        x = saved["xs"][0, 0] if train else torch.empty(B * T, D, device=dev, dtype=torch.float32)
        xv = x.view(B, T, D)
        xv[:, :CL].copy_(lin.view(B, CL, D))                       # torch.cat((x, prefix), dim=1): data movement only
        xv[:, CL:].copy_(ar.params["clip_project.prefix_const"].data)
        xo = st.forward(x, B, saved=saved, T=T)
        out = xo.view(B, T, D)[:, CL:].contiguous().view(B, P * D)  # [:, clip_length:] (train.py:239), dense for caption_embed
        return out, (pb, saved) if train else None

    def _tmapper_backward(self, msave, dx_gpt: torch.Tensor, dxb_gpt: torch.Tensor, B: int, S: int, acc, A):
        """dx_gpt / dxb_gpt: fp32 / 16-bit gradient of the GPT-2 input rows [B*S, D]; the first P rows of every
        sequence are the mapper's output."""
        ar, mp, st = self._arena, self.clip_project, self._mstack
        g = ar.g
        pb, saved = msave
        dev = dx_gpt.device
        CL, P, D = mp.clip_length, mp.prefix_length, mp.dim
        T = CL + P
        dxm = torch.zeros(B * T, D, device=dev, dtype=torch.float32)
        dxbm = torch.zeros(B * T, D, device=dev, dtype=self.compute_dtype)
        dxm.view(B, T, D)[:, CL:].copy_(dx_gpt.view(B, S, D)[:, :P])
        dxbm.view(B, T, D)[:, CL:].copy_(dxb_gpt.view(B, S, D)[:, :P])
        for i, blk in enumerate(st.blocks):                          # packed q|kv grad view shares the q slot's state
            if blk.grads is not None:
                acc[id(blk.grads["w_qkv"])] = A(f"clip_project.transformer.layers.{i}.attn.to_queries.weight")
        dxbm = st.backward(dxm, dxbm, saved, acc)
        sc = st.scratch
        pc, lw, lb = "clip_project.prefix_const", "clip_project.linear.weight", "clip_project.linear.bias"
        ops.colsum(dxm.view(B, T * D)[:, CL * D:], g[pc].view(-1), sc.floats(ops.colsum_ws_floats(B, P * D)), R=B, C=P * D,
                   ld=T * D, accumulate=A(pc))
        dlin = dxbm.view(B, T * D)[:, :CL * D]
        ops.gemm_bf16(dlin, pb, a_kcontig=False, b_kcontig=False, residual=g[lw] if A(lw) else None, out_f32=g[lw])
        ops.colsum(dlin, g[lb], sc.floats(ops.colsum_ws_floats(B, CL * D)), R=B, C=CL * D, ld=T * D, accumulate=A(lb))

    def _mapper_forward(self, prefix: torch.Tensor, train: bool):
        if self._mstack is not None:
            return self._tmapper_forward(prefix, train)
        ar = self._arena
        B = prefix.shape[0]
        dev = prefix.device
        n_in, n_hid, n_out = self.clip_project.sizes
        pb = torch.empty(B, n_in, device=dev, dtype=self.compute_dtype)
        ops.cast_f32_to_bf16(prefix.detach().float().contiguous(), pb)
        h1 = torch.empty(B, n_hid, device=dev, dtype=self.compute_dtype)
        ops.gemm_bf16(pb, ar.b["clip_project.model.0.weight"], bias=ar.params["clip_project.model.0.bias"].data,
                      act=ops.ACT_TANH, out_bf16=h1)
        out = torch.empty(B, n_out, device=dev, dtype=torch.float32)
        ops.gemm_bf16(h1, ar.b["clip_project.model.2.weight"], bias=ar.params["clip_project.model.2.bias"].data, out_f32=out)
        return out, (pb, h1) if train else None

    def _mlp_mapper_backward(self, msave, dproj: torch.Tensor, ld: int, A):
        """MLP mapper backward (Linear - Tanh - Linear, train.py:110-123).  dproj: 16-bit [B, n_out] gradient of the mapper
        output with row stride `ld` (a column slice of the GPT-2 input gradient, or a dense matrix)."""
        ar = self._arena
        g, sc = ar.g, self._stack.scratch
        pb, h1 = msave
        B = pb.shape[0]
        n_in, n_hid, n_out = self.clip_project.sizes
        w0, b0, w2, b2 = ("clip_project.model.0.weight", "clip_project.model.0.bias", "clip_project.model.2.weight",
                          "clip_project.model.2.bias")
        ops.gemm_bf16(dproj, h1, a_kcontig=False, b_kcontig=False, residual=g[w2] if A(w2) else None, out_f32=g[w2])
        ops.colsum(dproj, g[b2], sc.floats(ops.colsum_ws_floats(B, n_out)), R=B, C=n_out, ld=ld, accumulate=A(b2))
        dh1 = torch.empty(B, n_hid, device=dproj.device, dtype=self.compute_dtype)
        ops.gemm_bf16(dproj, ar.b[w2], b_kcontig=False, act=ops.ACT_DTANH, aux=h1, out_bf16=dh1)
        ops.gemm_bf16(dh1, pb, a_kcontig=False, b_kcontig=False, residual=g[w0] if A(w0) else None, out_f32=g[w0])
        ops.colsum(dh1, g[b0], sc.floats(ops.colsum_ws_floats(B, n_hid)), R=B, C=n_hid, ld=n_hid, accumulate=A(b0))

    def _mapper_only_backward(self, msave, dout: torch.Tensor):
        """gradient of `clip_project(prefix)` alone: dout fp32 [B, P*D] (MLP) / [B, P, D] (TransformerMapper)"""
        ar = self._arena
        acc = ar.begin_backward()
        g = ar.g

        def A(name):
            return acc[id(g[name])]

        B, P, D = dout.shape[0], self.prefix_length, self.model_embedding_size
        LS = ar.loss_scale()
        names = [n for n in ar.names if n.startswith("clip_project.") and ar.params[n].requires_grad]
        d32 = (dout.detach().float() * LS).contiguous().view(B, P * D)
        if LS != 1.0:
            ar.scale_grads([n for n in names if A(n)], LS)
        d16 = torch.empty(B, P * D, device=dout.device, dtype=self.compute_dtype)
        ops.cast_f32_to_bf16(d32, d16)
        if self._mstack is not None:
            self._tmapper_backward(msave, d32.view(B * P, D), d16.view(B * P, D), B, P, acc, A)
        else:
            self._mlp_mapper_backward(msave, d16, P * D, A)
        ar.scale_grads(names, 1.0 / LS)
        ar.publish_grads(names)

    def _mapper_only(self, prefix: torch.Tensor) -> torch.Tensor:
        """`model.clip_project(prefix)` as the inference scripts call it (test.py:540, application.py:104); differentiable
        with respect to the mapper's parameters (the prefix embedding is data)."""
        self._ensure_runtime()
        self._arena.refresh_shadows()
        names = [n for n in self._arena.names if n.startswith("clip_project.")]
        if torch.is_grad_enabled() and any(self._arena.params[n].requires_grad for n in names):
            out = _MapperOnly.apply(self, prefix, *[self._arena.params[n] for n in names])
        else:
            out = self._mapper_forward(prefix, False)[0]
        if isinstance(self.clip_project, TransformerMapper):          # train.py:247 returns [B, prefix_length, D]
            out = out.view(-1, self.prefix_length, self.model_embedding_size)
        return out

    # ---- shared forward to the final hidden states ----
    def _hidden_forward(self, x: torch.Tensor, B: int, S: int, mask: Optional[torch.Tensor], saved: Optional[dict],
                        cu: Optional[torch.Tensor] = None, rowmap: Optional[torch.Tensor] = None,
                        tail_rows: Optional[torch.Tensor] = None):
        keep = None
        if mask is not None:
            keep = mask.detach().to(torch.float32).contiguous()
            assert keep.shape == (B, S), f"attention_mask {tuple(mask.shape)} vs sequence {(B, S)}"
            if rowmap is not None:
                keep = keep.view(-1)[rowmap].contiguous()              # packed rows: the mask entries of the live positions
        return self._stack.forward(x, B, saved=saved, key_keep=keep, T=S, cu=cu, tail_rows=tail_rows)

    def _pack_rows(self) -> bool:
        import os
        v = getattr(self, "pack_rows", None)
        return (os.environ.get("CCLIP_PACK_TEXT", "1") != "0") if v is None else bool(v)

    def _lm_rows(self, xo: torch.Tensor, rows: torch.Tensor, train: bool):
        """ln_f on the selected rows + lm_head (tied wte): logits fp32 [len(rows), V]."""
        ar = self._arena
        p = ar.params
        R, D = (rows.numel() if rows is not None else xo.shape[0]), self.model_embedding_size
        dev = xo.device
        xf = torch.empty(R, D, device=dev, dtype=self.compute_dtype)
        st = torch.empty(2, R, device=dev, dtype=torch.float32)
        ops.layernorm_fwd(xo, p["model.transformer.ln_f.weight"].data, p["model.transformer.ln_f.bias"].data, rows=R,
                          row_index=rows, out_bf16=xf, mean=st[0], rstd=st[1])
        V = self.model.geo.vocab_size
        logits = torch.empty(R, (V + 7) // 8 * 8, device=dev, dtype=torch.float32)[:, :V]     # row stride padded to 8
        ops.gemm_bf16(xf, ar.b["model.transformer.wte.weight"], out_f32=logits)
        return logits, (xf, st) if train else None

    def _logits_from_embeds(self, inputs_embeds: torch.Tensor, attention_mask) -> torch.Tensor:
        """GPT2LMHeadModel(inputs_embeds=..., attention_mask=...).logits, inference only (generate loops, test.py:381)."""
        if torch.is_grad_enabled() and inputs_embeds.requires_grad:
            raise NotImplementedError("train through ClipCaptionModel.forward / caption_loss")
        self._ensure_runtime()
        self._arena.refresh_shadows()
        B, S, D = inputs_embeds.shape
        x = torch.empty(B * S, D, device=inputs_embeds.device, dtype=torch.float32)
        ops.add_positional(inputs_embeds.detach().float().contiguous().view(B * S, D),
                           self._arena.params["model.transformer.wpe.weight"].data, x, rows=B * S, S=S)
        xo = self._hidden_forward(x, B, S, attention_mask, None)
        rows = torch.arange(B * S, device=x.device, dtype=torch.int32)
        return self._lm_rows(xo, rows, False)[0].unflatten(0, (B, S))

    def _cached_logits(self, inputs_embeds: torch.Tensor, cache: Optional[KVCache]):
        """(logits [B, S_new, V], cache) for the new positions `inputs_embeds` [B, S_new, D] appended after cache.length."""
        if torch.is_grad_enabled() and inputs_embeds.requires_grad:
            raise NotImplementedError("KV-cached decode is inference only")
        self._ensure_runtime()
        self._arena.refresh_shadows()
        g = self.model.geo
        B, S, D = inputs_embeds.shape
        dev = inputs_embeds.device
        wpe = self._arena.params["model.transformer.wpe.weight"].data
        emb = inputs_embeds.detach().float().contiguous()
        if cache is None:                                     # prefill: the ordinary causal forward, keys / values kept
            cache = KVCache(g.n_layer, B, g.n_positions, D, dev, self.compute_dtype)
            x = torch.empty(B * S, D, device=dev, dtype=torch.float32)
            ops.add_positional(emb.view(B * S, D), wpe, x, rows=B * S, S=S)
            xo = self._stack.forward(x, B, T=S, kv_out=(cache.k, cache.v))
            cache.length = S
            rows = torch.arange(B * S, device=dev, dtype=torch.int32)
            return self._lm_rows(xo, rows, False)[0].unflatten(0, (B, S)), cache
        if S != 1 or B != cache.n_seq:
            raise NotImplementedError(f"decode step takes one new token per cached sequence, got {tuple(inputs_embeds.shape)} for {cache.n_seq} sequences")
        pos = cache.length
        if pos >= cache.max_len:
            raise RuntimeError(f"sequence longer than n_positions = {cache.max_len}")
        x = emb.view(B, D) + wpe[pos]
        if os.environ.get("CCLIP_DECODE_DRIVER", "native") != "native":       # launch by launch from Python (reference form)
            xo = self._stack.decode_step(x, cache.k, cache.v, pos)
            cache.length = pos + 1
            rows = torch.arange(B, device=dev, dtype=torch.int32)
            return self._lm_rows(xo, rows, False)[0].view(B, 1, -1), cache
        # the whole per-token launch sequence (and ln_f + lm_head) from one native call
        st = self._stack
        if getattr(self, "_decode_ptrs", None) is None or self._decode_ptrs[0] is not self._arena:
            self._decode_ptrs = (self._arena, ops.block_ptr_array(st.blocks))
        V = g.vocab_size
        p = self._arena.params
        hidden = st.geo.hidden or 4 * D
        scratch = torch.empty(B * (5 * D + hidden), device=dev, dtype=self.compute_dtype)
        logits = torch.empty(B, (V + 7) // 8 * 8, device=dev, dtype=torch.float32)[:, :V]
        ops.gpt2_decode_step(self._decode_ptrs[1], g.n_layer, x, cache.k, cache.v, pos, scratch, heads=st.geo.heads, hidden=hidden,
                             act=st.geo.act, linear_layout=st.geo.linear_layout, lnf_w=p["model.transformer.ln_f.weight"].data,
                             lnf_b=p["model.transformer.ln_f.bias"].data, wte16=self._arena.b["model.transformer.wte.weight"], logits=logits)
        cache.length = pos + 1
        return logits.view(B, 1, -1) if logits.is_contiguous() else logits.unsqueeze(1), cache

    def beam_native_ok(self, beam_size: int) -> bool:
        """the persistent beam-search kernel covers GPT-2 geometry (Conv1D weights, head_dim 64, <= 24 layers) and <= 8 beams"""
        if os.environ.get("CCLIP_BEAM_NATIVE", "1") == "0" or not 1 <= beam_size <= ops.BEAM_MAX_BEAMS:
            return False
        self._ensure_runtime()
        g, sg = self.model.geo, self._stack.geo
        V = g.vocab_size
        return (not sg.linear_layout and g.n_layer <= ops.BEAM_MAX_LAYERS and sg.head_dim == 64 and sg.width % 64 == 0
                and (sg.hidden or 4 * sg.width) % 32 == 0 and sg.act in (ops.ACT_NONE, ops.ACT_GELU_NEW) and g.n_positions <= 2048
                and -(-V // 256) <= 256 and next(self.parameters()).is_cuda)

    @torch.no_grad()
    def beam_search_native(self, inputs_embeds: torch.Tensor, beam_size: int, entry_length: int, temperature: float,
                           stop_token: int, prompt_tokens: Optional[torch.Tensor] = None, grid_cap: int = 0):
        """The reference's generate_beam loop (test.py:380-434) after the prefix: prefill here, then every decode step and
        every selection inside ONE persistent kernel launch (ops.gpt2_beam_search).  Returns (tokens [beams, n] int64,
        seq_lengths [beams], scores [beams]) exactly where the reference's loop stops (`is_stopped.all()` or entry_length)."""
        self._ensure_runtime()
        self._arena.refresh_shadows()
        g, st = self.model.geo, self._stack
        B, S, D = inputs_embeds.shape
        assert B == 1, "beam search starts from one sequence"
        dev = inputs_embeds.device
        p = self._arena.params
        wpe = p["model.transformer.wpe.weight"].data
        max_len = g.n_positions
        if S + entry_length - 1 > max_len:
            raise RuntimeError(f"sequence longer than n_positions = {max_len}")
        cache = KVCache(g.n_layer, beam_size, max_len, D, dev, self.compute_dtype)
        x = torch.empty(S, D, device=dev, dtype=torch.float32)
        ops.add_positional(inputs_embeds.detach().float().contiguous().view(S, D), wpe, x, rows=S, S=S)
        xo = st.forward(x, 1, T=S, kv_out=(cache.k, cache.v))          # prefix keys / values -> cache slot 0
        first_logits = self._lm_rows(xo, torch.tensor([S - 1], device=dev, dtype=torch.int32), False)[0].view(-1).float().contiguous()
        n_prompt = 0 if prompt_tokens is None else int(prompt_tokens.numel())
        bs = ops.BeamState(beam_size, max_len, n_prompt + entry_length, dev)
        if n_prompt:
            bs.tokens[0, :n_prompt] = prompt_tokens.view(-1).to(dev, torch.int32)
            bs.state[4] = n_prompt
        if getattr(self, "_decode_ptrs", None) is None or self._decode_ptrs[0] is not self._arena:
            self._decode_ptrs = (self._arena, ops.block_ptr_array(st.blocks))
        hidden = st.geo.hidden or 4 * D
        scratch = torch.empty(beam_size * (5 * D + hidden), device=dev, dtype=self.compute_dtype)
        ops.gpt2_beam_search(self._decode_ptrs[1], g.n_layer, bs, cache.k, cache.v, S, scratch, entry_length - 1, heads=st.geo.heads,
                             hidden=hidden, act=st.geo.act, lnf_w=p["model.transformer.ln_f.weight"].data,
                             lnf_b=p["model.transformer.ln_f.bias"].data, wte16=self._arena.b["model.transformer.wte.weight"],
                             wte_f32=p["model.transformer.wte.weight"].data, wpe_f32=wpe, temperature=float(temperature),
                             stop_token=int(stop_token), first_logits=first_logits,
                             grid_cap=grid_cap or int(os.environ.get("CCLIP_BEAM_GRID", "0")))
        state = bs.state.tolist()                                       # the one host sync of the caption
        if state[1]:
            raise RuntimeError("cclip_gpt2_beam_search: a grid barrier timed out (workgroups not co-resident?)")
        n_sel = state[3] if state[2] else entry_length
        return bs.tokens[:, :n_prompt + n_sel].long(), bs.seq_lengths, bs.scores

    def _embed_and_run(self, tokens, prefix, attribute, mask, train: bool, pack: bool = False):
        self._ensure_runtime()
        ar = self._arena
        ar.refresh_shadows()
        dev = tokens.device
        B, P, D = tokens.shape[0], self.prefix_length, self.model_embedding_size
        ids = torch.cat((attribute, tokens), dim=1).to(torch.int32).contiguous()       # train.py:257
        Lt = ids.shape[1]
        S = P + Lt
        proj, msave = self._mapper_forward(prefix, train)                              # train.py:262
        p = ar.params
        # PACKED rows (caption_loss only; CCLIP_PACK_TEXT=0 / model.pack_rows = False disables): GPT-2 is causal and the loss
        # ignores targets equal to 0 (train.py:357), so of sequence b only positions 0 .. P+A+len_b-2 matter (len_b = the
        # caption's length up to its last non-zero token): the row that predicts token j is position P+A-1+j, and later rows
        # are neither targets nor keys of a needed row.  The stack runs on those rows, sequences back to back (cu); the
        # embedding is formed densely and its live rows gathered in, the gradient rows scattered back (as the CLIP text tower).
        rowmap = cu = lens = tsel = None
        if pack and self._pack_rows() and B > 1 and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            Lc = tokens.shape[1]
            lens = ((tokens != 0) * torch.arange(1, Lc + 1, device=dev)[None, :]).amax(dim=1)        # last non-zero index + 1
            need = (S - Lc + lens - 1).clamp(min=1)
            live = torch.arange(S, device=dev)[None, :] < need[:, None]
            tlive = torch.arange(Lc, device=dev)[None, :] < lens[:, None]                         # targets that can be non-zero
            nzi = torch.cat((live.reshape(-1), tlive.reshape(-1))).nonzero().squeeze(1)          # ONE sync for both row lists
            k = int((nzi < B * S).sum().item())
            rowmap, tsel = nzi[:k], nzi[k:] - B * S
            if k < B * S:
                cu = torch.zeros(B + 1, device=dev, dtype=torch.int32)
                cu[1:] = torch.cumsum(need, 0)
            else:
                rowmap = tsel = lens = None
        Mp = int(rowmap.numel()) if cu is not None else B * S
        saved = self._stack.alloc_saved(B, dev, T=S, M=Mp) if train else None
        if cu is not None:
            xd = torch.empty(B * S, D, device=dev, dtype=torch.float32)
            ops.caption_embed(proj, ids, p["model.transformer.wte.weight"].data, p["model.transformer.wpe.weight"].data, xd,
                              B=B, P=P, Lt=Lt)
            x = saved["xs"][0, 0] if train else torch.empty(Mp, D, device=dev, dtype=torch.float32)
            torch.index_select(xd, 0, rowmap, out=x)
        else:
            x = saved["xs"][0, 0] if train else torch.empty(B * S, D, device=dev, dtype=torch.float32)
            ops.caption_embed(proj, ids, p["model.transformer.wte.weight"].data, p["model.transformer.wpe.weight"].data, x,
                              B=B, P=P, Lt=Lt)
        # The fused loss reads only the rows that predict a target: the LAST block's out-proj / LayerNorm / MLP run on those
        # rows alone (BlockStack tail_rows; CCLIP_TAIL_ROWS=0 disables) and xo comes back compact, in `lm_rows` order.
        lm_rows = None
        if pack and os.environ.get("CCLIP_TAIL_ROWS", "1") != "0":
            Lc = tokens.shape[1]
            first = S - Lc - 1                                                    # = P + A - 1 (train.py:356)
            if cu is not None:
                lm_rows = cu[:-1].long()[tsel // Lc] + first + tsel % Lc
            else:
                lm_rows = (torch.arange(B, device=dev)[:, None] * S + first + torch.arange(Lc, device=dev)[None, :]).reshape(-1)
        xo = self._hidden_forward(x, B, S, mask, saved, cu=cu, rowmap=rowmap if cu is not None else None, tail_rows=lm_rows)
        ctx = dict(saved=saved, msave=msave, ids=ids, B=B, S=S, Lt=Lt, xo=xo, cu=cu, rowmap=rowmap if cu is not None else None,
                   tsel=tsel, Mp=Mp, compact=lm_rows is not None)
        if train and ops.SCATTER_DETERMINISTIC and dev.type == "cuda" and p["model.transformer.wte.weight"].requires_grad:
            # index tables of the deterministic wte-gradient sum: token ids only, so they are built now on a helper stream, under
            # the forward pass, instead of inside the backward pass (clip/model.py does the same for the text tower)
            cur = torch.cuda.current_stream()
            if getattr(self, "_aux_stream", None) is None:
                self._aux_stream = torch.cuda.Stream(device=dev)
            self._aux_stream.wait_stream(cur)
            with torch.cuda.stream(self._aux_stream):
                ctx["scatter_tables"] = ops.embed_scatter_tables(ids.view(-1), p["model.transformer.wte.weight"].shape[0], rows=B * Lt)
                ctx["scatter_ready"] = torch.cuda.Event()
                ctx["scatter_ready"].record(self._aux_stream)
        return xo, ctx

    # ---- public: reference call signature ----
    def forward(self, tokens: torch.Tensor, prefix: torch.Tensor, attribute: torch.Tensor,
                mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None):
        """train.py:256-269.  Returns an object with `.logits` [B, P+A+L, V] (fp32)."""
        if labels is not None:
            raise NotImplementedError("labels= is dead code in the reference (train.py:354 never passes it)")
        if not tokens.is_cuda:
            raise RuntimeError("ClipCaptionModel: HIP path only (no CPU fallback)")
        B, S = tokens.shape[0], self.prefix_length + attribute.shape[1] + tokens.shape[1]
        rows = torch.arange(B * S, device=tokens.device, dtype=torch.int32)
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            self._ensure_runtime()
            logits = _CaptionLogits.apply(self, tokens, prefix, attribute, mask, rows, *self._arena.params.values())
        else:
            xo, _ = self._embed_and_run(tokens, prefix, attribute, mask, False)
            logits = self._lm_rows(xo, rows, False)[0]
        return SimpleNamespace(logits=logits.reshape(B, S, -1) if logits.stride(0) == logits.shape[1] else logits.unflatten(0, (B, S)))

    def caption_loss(self, tokens, prefix, attribute, mask=None, attribute_length: Optional[int] = None):
        """Fused train.py:354-357: logits[:, P+A-1:-1] vs tokens, CE(ignore_index=0, mean over kept targets).
        Only the B*L needed rows go through ln_f / lm_head / softmax."""
        self._ensure_runtime()
        return _CaptionLoss.apply(self, tokens, prefix, attribute, mask, *self._arena.params.values())

    # ---- backward shared by both autograd nodes ----
    def _backward_from_dlogits(self, c: dict, dlog_b: torch.Tensor, rows: torch.Tensor, lm):
        """dlog_b: bf16 [R, V] gradient of the selected logits rows."""
        ar, stack = self._arena, self._stack
        p, g = ar.params, ar.g
        acc = ar.begin_backward()
        B, S, Lt, P, D = c["B"], c["S"], c["Lt"], self.prefix_length, self.model_embedding_size
        M = c.get("Mp") or B * S                       # rows the stack ran on (packed: the live rows only)
        dev = dlog_b.device
        xf, st = lm
        R = dlog_b.shape[0]
        sc = stack.scratch
        frozen = not p["model.transformer.wte.weight"].requires_grad

        def A(name):
            return acc[id(g[name])]

        LS = ar.loss_scale()         # fp16 operands: dlog_b arrives LS x too large (callers), and so is every slot written here
        trainable_names = [n for n in ar.names if ar.params[n].requires_grad]
        if LS != 1.0:
            ar.scale_grads([n for n in trainable_names if A(n)], LS)
        wte_name = "model.transformer.wte.weight"
        wrote_wte = False
        if not frozen:
            # tied lm_head: gwte (+)= dlogits^T xf      [V, D]
            n_out, k_in = g[wte_name].shape
            from cclip_hip.stack import wgrad_candidates
            ops.gemm_bf16(dlog_b, xf, a_kcontig=False, b_kcontig=False, residual=g[wte_name] if A(wte_name) else None,
                          out_f32=g[wte_name], split_candidates=wgrad_candidates(n_out, k_in, R), scratch=sc.floats)
            wrote_wte = True
        dxf = torch.empty(R, D, device=dev, dtype=self.compute_dtype)
        ops.gemm_bf16(dlog_b, ar.b[wte_name], b_kcontig=False, out_bf16=dxf)           # dlogits @ wte
        Mo = c["xo"].shape[0]                          # R when the last block ran on the target rows only (compact), else M
        dx = torch.zeros(Mo, D, device=dev, dtype=torch.float32)
        dxb = torch.zeros(Mo, D, device=dev, dtype=self.compute_dtype)
        lnf_w, lnf_b = "model.transformer.ln_f.weight", "model.transformer.ln_f.bias"
        ops.layernorm_bwd(dxf, c["xo"], p[lnf_w].data, st[0], st[1], rows=R, row_index=rows, dx_out=dx, dx_out_bf16=dxb,
                          dgamma=None if frozen else g[lnf_w], dbeta=None if frozen else g[lnf_b],
                          accumulate=False if frozen else A(lnf_w),
                          ws=None if frozen else sc.floats(ops.layernorm_bwd_ws_floats(R, D)))
        dxb = stack.backward(dx, dxb, c["saved"], acc)
        dx = c["saved"]["dx_in"]
        if c.get("rowmap") is not None:                # back to dense [B*S, D] rows for the embedding / mapper gradients
            dxp, dxbp = dx, dxb
            dx = torch.zeros(B * S, D, device=dev, dtype=torch.float32)
            dxb = torch.zeros(B * S, D, device=dev, dtype=self.compute_dtype)
            dx.index_copy_(0, c["rowmap"], dxp)
            dxb.index_copy_(0, c["rowmap"], dxbp)
        if not frozen:
            # x = [prefix_proj | wte[ids]] + wpe[s]
            wpe = "model.transformer.wpe.weight"
            if not A(wpe):
                g[wpe].zero_()
            ops.colsum(dx, g[wpe][:S].view(-1), sc.floats(ops.colsum_ws_floats(B, S * D)), R=B, C=S * D, ld=S * D, accumulate=True)
            tables = c.get("scatter_tables")
            if tables is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(c["scatter_ready"])
                for t in tables:
                    t.record_stream(cur)
            ops.embed_scatter_add(c["ids"].view(-1), dx, g[wte_name], rows=B * Lt, L=Lt, seq_stride=S, seq_off=P, tables=tables, scratch=sc.floats)
        # mapper: d prefix_proj = dx[b, :P]  -> a [B, P*D] matrix with row stride S*D inside dx / dxb
        if self._mstack is not None:
            if p["clip_project.linear.weight"].requires_grad:
                self._tmapper_backward(c["msave"], dx, dxb, B, S, acc, A)
        elif p["clip_project.model.2.weight"].requires_grad:
            self._mlp_mapper_backward(c["msave"], dxb.view(B, S * D)[:, :P * D], S * D, A)
        ar.scale_grads(trainable_names, 1.0 / LS)
        ar.publish_grads(trainable_names)


class ClipCaptionPrefix(ClipCaptionModel):
    """Mapper-only training (train.py:286-294): GPT-2 frozen."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for q in self.model.parameters():
            q.requires_grad_(False)

    def parameters(self, recurse: bool = True):
        return self.clip_project.parameters()

    def train(self, mode: bool = True):
        super().train(mode)
        self.model.eval()
        return self


class _MapperOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "ClipCaptionModel", prefix, *params):
        out, msave = model._mapper_forward(prefix, True)
        ctx.model, ctx.msave, ctx.n = model, msave, len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.model._mapper_only_backward(ctx.msave, dout)
        ctx.msave = None
        return (None, None) + (None,) * ctx.n


class _CaptionLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: ClipCaptionModel, tokens, prefix, attribute, mask, rows, *params):
        xo, c = model._embed_and_run(tokens, prefix, attribute, mask, True)
        logits, lm = model._lm_rows(xo, rows, True)
        ctx.model, ctx.c, ctx.lm, ctx.rows, ctx.n = model, c, lm, rows, len(params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        R, V = dlogits.shape
        Vp = (V + 7) // 8 * 8
        d = torch.zeros(R, Vp, device=dlogits.device, dtype=torch.float32)
        d[:, :V].copy_(dlogits)                                             # re-stride onto the 8-padded layout (plumbing)
        LS = ctx.model._arena.loss_scale()
        if LS != 1.0:
            ops.scale_f32(d.view(-1), LS)                                   # before the 16-bit cast (fp16 loss scale)
        db = torch.empty(R, Vp, device=d.device, dtype=ctx.model.compute_dtype)
        ops.cast_f32_to_bf16(d, db)
        ctx.model._backward_from_dlogits(ctx.c, db[:, :V], ctx.rows, ctx.lm)
        ctx.c = ctx.lm = None
        return (None,) * (6 + ctx.n)


class _CaptionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: ClipCaptionModel, tokens, prefix, attribute, mask, *params):
        need_grad = any(ctx.needs_input_grad)
        xo, c = model._embed_and_run(tokens, prefix, attribute, mask, need_grad, pack=True)
        B, S, Lc = c["B"], c["S"], tokens.shape[1]
        dev = tokens.device
        first = S - Lc - 1                                                    # = P + A - 1 (train.py:356)
        if c["cu"] is not None:
            # packed: only the targets up to each caption's last non-zero token (the rest are ignored by the loss anyway)
            tsel = c["tsel"]
            b_of, j_of = tsel // Lc, tsel % Lc
            rows = (c["cu"][:-1].long()[b_of] + first + j_of).to(torch.int32).contiguous()
            labels = tokens.reshape(-1)[tsel].to(torch.int32).contiguous()
        else:
            rows = (torch.arange(B, device=dev)[:, None] * S + first + torch.arange(Lc, device=dev)[None, :]).reshape(-1).to(torch.int32)
            labels = tokens.reshape(-1).to(torch.int32).contiguous()
        if c["compact"]:
            rows = None                                                           # xo holds exactly these rows, in this order
        logits, lm = model._lm_rows(xo, rows, need_grad)
        kept = int((labels != 0).sum().item())                               # mean over non-ignored targets
        R = logits.shape[0]
        loss_rows = torch.empty(R, device=dev, dtype=torch.float32)
        V = logits.shape[1]
        dlog = torch.zeros(R, (V + 7) // 8 * 8, device=dev, dtype=model.compute_dtype)[:, :V] if need_grad else None   # finite pads
        ops.xent_rows(logits, labels, loss_row=loss_rows, dlogits=dlog, grad_scale=model._arena.loss_scale() / max(kept, 1),
                      ignore_index=0)                                        # (fp16 operands: 16-bit dlogits under the loss scale)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        ops.reduce_dot(loss_rows, None, out, alpha=1.0 / max(kept, 1))
        if need_grad:
            ctx.model, ctx.c, ctx.lm, ctx.rows, ctx.dlog, ctx.n = model, c, lm, rows, dlog, len(params)
        return out.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        # dlogits were formed for dloss = 1; a different upstream scalar would need a rescale of the bf16 rows
        if float(dloss) != 1.0:
            raise NotImplementedError("caption_loss() must be the root of backward (loss.backward(), as train.py:358 does); "
                                      "for a scaled loss use ClipCaptionModel.forward + torch cross_entropy")
        ctx.model._backward_from_dlogits(ctx.c, ctx.dlog, ctx.rows, ctx.lm)
        ctx.c = ctx.lm = ctx.dlog = None
        return (None,) * (5 + ctx.n)
