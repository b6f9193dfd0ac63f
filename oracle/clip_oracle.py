"""CPU fp32 ORACLE for the CLIP hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (construction-clip_amd/) never does.

What it restates
----------------
The reference's hot path is `model(image, text)` / `encode_image` / `encode_text`
of the third-party `clip` package (openai/CLIP, un-pinned, NOT vendored under
/root/reference and not installed here - SURVEY.md section 8c), called from
  /root/reference/CLIP/train.py:161,196      (contrastive step)
  /root/reference/CLIP/train_caption.py:124  (per-caption variant)
  /root/reference/CLIP/predict.py:46         (zero-shot)
  /root/reference/CLIP_prefix_caption/parse_coco.py:43,45,50 (encode_image + 2 zero-shots)
This file restates that package's published algorithm with plain torch fp32
CPU primitives, over a state_dict in the OpenAI key layout.

Pinning
-------
The reference holds no tests, fixtures or golden vectors for this path
(SURVEY.md section 4), and `clip` cannot be imported here.  The restatement is
therefore pinned against the only independent CLIP arithmetic present in the
container: `transformers.CLIPModel` (config-only construction, random init, no
fetch) through the key mapping in oracle/hf_crosscheck.py.  Cited below as
HF:<file>:<line> (transformers 5.x, models/clip/modeling_clip.py).
Against the *reference's own* outputs: PARITY UNPINNED.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# shape bookkeeping (mirrors openai/CLIP build_model(): everything is inferred
# from the state_dict, which is how CLIP/train.py:111 round-trips checkpoints)
# ----------------------------------------------------------------------------
def infer_config(sd: SD) -> dict:
    vw = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    v_layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    tw = sd["ln_final.weight"].shape[0]
    t_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    return dict(
        embed_dim=sd["text_projection"].shape[1],
        image_resolution=patch * grid,
        vision_layers=v_layers,
        vision_width=vw,
        vision_patch_size=patch,
        vision_heads=vw // 64,
        context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0],
        transformer_width=tw,
        transformer_heads=tw // 64,
        transformer_layers=t_layers,
    )


def quick_gelu(x: Tensor) -> Tensor:
    # openai/CLIP QuickGELU; HF:activations.py QuickGELUActivation: x * sigmoid(1.702 x)
    return x * torch.sigmoid(1.702 * x)


def _layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    # fp32 statistics, eps 1e-5 (HF:modeling_clip.py:358-361)
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), 1e-5)


def _attention(x: Tensor, p: str, sd: SD, heads: int, mask: Tensor | None) -> Tensor:
    """nn.MultiheadAttention restated: packed in_proj [3D, D] (q rows, k rows, v rows),
    scale dh**-0.5, softmax over keys, out_proj.  x: [N, T, D]."""
    n, t, d = x.shape
    dh = d // heads
    qkv = x @ sd[p + "attn.in_proj_weight"].float().t() + sd[p + "attn.in_proj_bias"].float()
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(n, t, heads, dh).transpose(1, 2)
    k = k.view(n, t, heads, dh).transpose(1, 2)
    v = v.view(n, t, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(n, t, d)
    return a @ sd[p + "attn.out_proj.weight"].float().t() + sd[p + "attn.out_proj.bias"].float()


def residual_block(x: Tensor, p: str, sd: SD, heads: int, mask: Tensor | None) -> Tensor:
    """ResidualAttentionBlock (HF:modeling_clip.py:353-384 CLIPEncoderLayer):
    x + attn(ln_1(x)); x + c_proj(QuickGELU(c_fc(ln_2(x))))."""
    x = x + _attention(_layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), p, sd, heads, mask)
    h = _layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = h @ sd[p + "mlp.c_fc.weight"].float().t() + sd[p + "mlp.c_fc.bias"].float()
    h = quick_gelu(h)
    h = h @ sd[p + "mlp.c_proj.weight"].float().t() + sd[p + "mlp.c_proj.bias"].float()
    return x + h


def encode_image(sd: SD, image: Tensor) -> Tensor:
    """VisionTransformer.forward (call sites: parse_coco.py:43, via CLIP/train.py:161).
    conv1 k=s=patch no bias -> [N, grid*grid, W]; prepend class_embedding; + positional;
    ln_pre; blocks; ln_post(x[:,0]) @ proj.  HF:modeling_clip.py:138-218, 594-656."""
    cfg = infer_config(sd)
    x = F.conv2d(image.float(), sd["visual.conv1.weight"].float(), stride=cfg["vision_patch_size"])
    n, w = x.shape[0], x.shape[1]
    x = x.reshape(n, w, -1).permute(0, 2, 1)  # [N, grid**2, W]
    cls = sd["visual.class_embedding"].float().expand(n, 1, w)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].float()
    x = _layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    for i in range(cfg["vision_layers"]):
        x = residual_block(x, f"visual.transformer.resblocks.{i}.", sd, cfg["vision_heads"], None)
    x = _layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"].float()


def causal_mask(t: int) -> Tensor:
    # additive mask: -inf strictly above the diagonal (openai/CLIP build_attention_mask)
    return torch.full((t, t), float("-inf")).triu_(1)


def encode_text(sd: SD, text: Tensor) -> Tensor:
    """CLIP.encode_text: token_embedding[text] + positional; causal blocks; ln_final;
    row at argmax(text) (EOT = largest id) @ text_projection.  HF:modeling_clip.py:221-256, 494-591."""
    cfg = infer_config(sd)
    x = sd["token_embedding.weight"].float()[text.long()] + sd["positional_embedding"].float()[: text.shape[1]]
    mask = causal_mask(text.shape[1])
    for i in range(cfg["transformer_layers"]):
        x = residual_block(x, f"transformer.resblocks.{i}.", sd, cfg["transformer_heads"], mask)
    x = _layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    x = x[torch.arange(x.shape[0]), text.long().argmax(dim=-1)]
    return x @ sd["text_projection"].float()


def clip_forward(sd: SD, image: Tensor, text: Tensor) -> Tuple[Tensor, Tensor]:
    """CLIP.forward (CLIP/train.py:161; CLIP/predict.py:46): L2-normalise both feature
    sets, logits_per_image = exp(logit_scale) * I @ T.t(), logits_per_text = transpose.
    HF:modeling_clip.py:809-817."""
    i = encode_image(sd, image)
    t = encode_text(sd, text)
    i = i / i.norm(dim=1, keepdim=True)
    t = t / t.norm(dim=1, keepdim=True)
    li = sd["logit_scale"].float().exp() * i @ t.t()
    return li, li.t()


def contrastive_loss(logits_per_image: Tensor, logits_per_text: Tensor) -> Tuple[Tensor, Tensor]:
    """CLIP/train.py:162-173: label = arange(N); (CE_i + CE_t)/2; accuracy on image rows."""
    n = logits_per_image.shape[0]
    label = torch.arange(n)
    loss = (F.cross_entropy(logits_per_image, label) + F.cross_entropy(logits_per_text, label)) / 2
    acc = (logits_per_image.argmax(dim=1) == label).float().mean()
    return loss, acc


def zero_shot(sd: SD, image: Tensor, text: Tensor) -> Tuple[Tensor, Tensor]:
    """CLIP/predict.py:46-54 and parse_coco.py:45-53: softmax over prompts, argmax index."""
    li, _ = clip_forward(sd, image, text)
    sim = li.softmax(dim=-1)
    return sim, sim.argmax(dim=1)
