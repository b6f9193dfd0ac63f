"""Geometry + seeded weights for the prefix-caption model, in the key layout that
`torch.save(model.state_dict())` produces at /root/reference/CLIP_prefix_caption/train.py:371-381:
`clip_project.model.{0,2}.{weight,bias}` (MLP mapper, train.py:110-123) and `model.*`
(= GPT2LMHeadModel, train.py:275).  `GPT2LMHeadModel.from_pretrained(name)` needs the network
(SURVEY.md 8c), so weights are seeded synthetic with GPT-2's published init scales
(N(0, 0.02); residual projections scaled by (2*n_layer)^-0.5)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch


@dataclass(frozen=True)
class CaptionGeometry:
    vocab_size: int = 21128          # ckiplab/gpt2-base-chinese (train.py:401 default tokenizer)
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    n_positions: int = 1024
    prefix_length: int = 20          # train.py:391
    attribute_length: int = 20       # train.py:392
    prefix_size: int = 512           # train.py:407


GPT2_MODELS: Dict[str, CaptionGeometry] = {
    "ckiplab/gpt2-base-chinese": CaptionGeometry(),
    "gpt2": CaptionGeometry(vocab_size=50257),
    "test-tiny": CaptionGeometry(vocab_size=300, n_embd=128, n_layer=2, n_head=2, n_positions=64,
                                 prefix_length=4, attribute_length=4, prefix_size=64),
}


def init_caption_state_dict(geo: CaptionGeometry, seed: int = 567, finetuned_like: bool = True) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def un(*shape, bound=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    D, L = geo.n_embd, geo.n_layer
    b_std = 0.02 if finetuned_like else 0.0
    ln_std = 0.1 if finetuned_like else 0.0
    sd: Dict[str, torch.Tensor] = {}
    # MLP mapper: nn.Linear default init, U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    hidden, out = (D * geo.prefix_length) // 2, D * geo.prefix_length
    sd["clip_project.model.0.weight"] = un(hidden, geo.prefix_size, bound=geo.prefix_size ** -0.5)
    sd["clip_project.model.0.bias"] = un(hidden, bound=geo.prefix_size ** -0.5)
    sd["clip_project.model.2.weight"] = un(out, hidden, bound=hidden ** -0.5)
    sd["clip_project.model.2.bias"] = un(out, bound=hidden ** -0.5)
    p = "model.transformer."
    sd[p + "wte.weight"] = rn(geo.vocab_size, D, std=0.02)
    sd[p + "wpe.weight"] = rn(geo.n_positions, D, std=0.02)
    for i in range(L):
        q = f"{p}h.{i}."
        sd[q + "ln_1.weight"] = 1.0 + rn(D, std=ln_std)
        sd[q + "ln_1.bias"] = rn(D, std=ln_std)
        sd[q + "attn.c_attn.weight"] = rn(D, 3 * D, std=0.02)        # Conv1D: [in, out]
        sd[q + "attn.c_attn.bias"] = rn(3 * D, std=b_std)
        sd[q + "attn.c_proj.weight"] = rn(D, D, std=0.02 / math.sqrt(2 * L))
        sd[q + "attn.c_proj.bias"] = rn(D, std=b_std)
        sd[q + "ln_2.weight"] = 1.0 + rn(D, std=ln_std)
        sd[q + "ln_2.bias"] = rn(D, std=ln_std)
        sd[q + "mlp.c_fc.weight"] = rn(D, 4 * D, std=0.02)
        sd[q + "mlp.c_fc.bias"] = rn(4 * D, std=b_std)
        sd[q + "mlp.c_proj.weight"] = rn(4 * D, D, std=0.02 / math.sqrt(2 * L))
        sd[q + "mlp.c_proj.bias"] = rn(D, std=b_std)
    sd[p + "ln_f.weight"] = 1.0 + rn(D, std=ln_std)
    sd[p + "ln_f.bias"] = rn(D, std=ln_std)
    sd["model.lm_head.weight"] = sd[p + "wte.weight"]   # tied, as GPT2LMHeadModel.state_dict() reports it
    return sd


def synthetic_caption_batch(b: int, geo: CaptionGeometry, caption_len: int = 40, seed: int = 567):
    """(tokens [B,Lc] int64 with trailing zeros, mask [B,P+A+Lc] float ones, prefix [B,512] fp32,
    attribute [B,A] int64) shaped like ClipCocoDataset.__getitem__ (train.py:30-63)."""
    g = torch.Generator().manual_seed(seed)
    V = geo.vocab_size
    tokens = torch.randint(1, V, (b, caption_len), generator=g)
    lens = torch.randint(max(2, caption_len // 3), caption_len + 1, (b,), generator=g)
    tokens = torch.where(torch.arange(caption_len)[None, :] < lens[:, None], tokens, torch.zeros_like(tokens))
    attribute = torch.randint(1, V, (b, geo.attribute_length), generator=g)
    prefix = torch.randn(b, geo.prefix_size, generator=g)
    mask = torch.ones(b, geo.prefix_length + geo.attribute_length + caption_len)
    return tokens, mask, prefix, attribute


def init_transformer_mapper_state_dict(geo: CaptionGeometry, clip_length: int, num_layers: int = 8, seed: int = 567,
                                       mlp_ratio: float = 2.0) -> Dict[str, torch.Tensor]:
    """`clip_project.*` tensors of the reference's TransformerMapper (CLIP_prefix_caption/train.py:233-248), nn.Linear /
    nn.LayerNorm default inits with perturbed LayerNorm affines (so every term is exercised)."""
    g = torch.Generator().manual_seed(seed)

    def un(*shape, bound=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    D, h = geo.n_embd, int(geo.n_embd * mlp_ratio)
    sd: Dict[str, torch.Tensor] = {}
    p = "clip_project."
    sd[p + "prefix_const"] = torch.randn(geo.prefix_length, D, generator=g)
    for i in range(num_layers):
        q = f"{p}transformer.layers.{i}."
        sd[q + "norm1.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
        sd[q + "norm1.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[q + "attn.to_queries.weight"] = un(D, D, bound=D ** -0.5)
        sd[q + "attn.to_keys_values.weight"] = un(2 * D, D, bound=D ** -0.5)
        sd[q + "attn.project.weight"] = un(D, D, bound=D ** -0.5)
        sd[q + "attn.project.bias"] = un(D, bound=D ** -0.5)
        sd[q + "norm2.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
        sd[q + "norm2.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[q + "mlp.fc1.weight"] = un(h, D, bound=D ** -0.5)
        sd[q + "mlp.fc1.bias"] = un(h, bound=D ** -0.5)
        sd[q + "mlp.fc2.weight"] = un(D, h, bound=h ** -0.5)
        sd[q + "mlp.fc2.bias"] = un(D, bound=h ** -0.5)
    sd[p + "linear.weight"] = un(clip_length * D, geo.prefix_size, bound=geo.prefix_size ** -0.5)
    sd[p + "linear.bias"] = un(clip_length * D, bound=geo.prefix_size ** -0.5)
    return sd
