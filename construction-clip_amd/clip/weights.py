"""Model geometry and seeded weight construction in the OpenAI-CLIP state_dict layout.

The reference never builds weights itself: `clip.load("ViT-B/32")` downloads them
(/root/reference/CLIP/train.py:105) and checkpoints round-trip as
`torch.save(model.state_dict())` (/root/reference/CLIP/train.py:213-217).  There is no
network here, so every run uses seeded synthetic weights with the init scales the
openai/CLIP package publishes (CLIP.initialize_parameters): token emb sigma 0.02,
positional sigma 0.01, attn sigma W^-0.5, proj sigma W^-0.5 (2L)^-0.5, fc sigma (2W)^-0.5,
visual class/pos/proj scale W^-0.5, logit_scale ln(1/0.07).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict

import torch


@dataclass(frozen=True)
class CLIPGeometry:
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def vision_tokens(self) -> int:
        return self.grid * self.grid + 1

    def as_dict(self) -> dict:
        return asdict(self)


# name -> geometry, as clip.available_models() would list them.  Only the ViT family is on the
# hot path (the RN* names the reference's argparse mentions at parse_coco.py:80 are a different
# architecture and are not provided).
MODELS: Dict[str, CLIPGeometry] = {
    "ViT-B/32": CLIPGeometry(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-B/16": CLIPGeometry(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-L/14": CLIPGeometry(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": CLIPGeometry(768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # small geometries for tests (same code path, seconds on CPU)
    "test-tiny": CLIPGeometry(64, 64, 2, 128, 32, 16, 512, 128, 2, 2),
    "test-small": CLIPGeometry(128, 96, 3, 256, 32, 24, 1024, 192, 3, 2),
    # patch 14 (3*14*14 = 588 is not a multiple of 8) and 145 image tokens (> 128): the ViT-L/14 code path at toy size
    "test-long": CLIPGeometry(64, 168, 2, 128, 14, 16, 512, 128, 2, 2),
}


def geometry_from_state_dict(sd: Dict[str, torch.Tensor]) -> CLIPGeometry:
    """Same inference openai/CLIP's build_model() does: every dimension comes from tensor shapes."""
    vw = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    v_layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    tw = sd["ln_final.weight"].shape[0]
    t_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    return CLIPGeometry(
        embed_dim=sd["text_projection"].shape[1], image_resolution=patch * grid, vision_layers=v_layers,
        vision_width=vw, vision_patch_size=patch, context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=tw,
        transformer_heads=tw // 64, transformer_layers=t_layers)


def init_state_dict(geo: CLIPGeometry, seed: int = 567, finetuned_like: bool = True) -> Dict[str, torch.Tensor]:
    """fp32 CPU state_dict in the OpenAI key layout, deterministic in `seed`
    (567 is the reference's seed, CLIP/train.py:28).

    finetuned_like=True perturbs biases and LayerNorm affines away from 0/1 the way a trained
    checkpoint has them, so that parity tests exercise every term of every kernel."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd: Dict[str, torch.Tensor] = {}
    vw, tw = geo.vision_width, geo.transformer_width
    b_std = 0.02 if finetuned_like else 0.0
    ln_std = 0.1 if finetuned_like else 0.0

    def ln(prefix, w):
        sd[prefix + ".weight"] = 1.0 + rn(w, std=ln_std)
        sd[prefix + ".bias"] = rn(w, std=ln_std)

    def blocks(prefix, width, layers):
        attn_std = width ** -0.5
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        fc_std = (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}."
            ln(p + "ln_1", width)
            sd[p + "attn.in_proj_weight"] = rn(3 * width, width, std=attn_std)
            sd[p + "attn.in_proj_bias"] = rn(3 * width, std=b_std)
            sd[p + "attn.out_proj.weight"] = rn(width, width, std=proj_std)
            sd[p + "attn.out_proj.bias"] = rn(width, std=b_std)
            ln(p + "ln_2", width)
            sd[p + "mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std)
            sd[p + "mlp.c_fc.bias"] = rn(4 * width, std=b_std)
            sd[p + "mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
            sd[p + "mlp.c_proj.bias"] = rn(width, std=b_std)

    scale = vw ** -0.5
    fan_in = 3 * geo.vision_patch_size ** 2
    sd["visual.conv1.weight"] = rn(vw, 3, geo.vision_patch_size, geo.vision_patch_size, std=fan_in ** -0.5)
    sd["visual.class_embedding"] = rn(vw, std=scale)
    sd["visual.positional_embedding"] = rn(geo.vision_tokens, vw, std=scale)
    ln("visual.ln_pre", vw)
    blocks("visual.transformer", vw, geo.vision_layers)
    ln("visual.ln_post", vw)
    sd["visual.proj"] = rn(vw, geo.embed_dim, std=scale)

    sd["token_embedding.weight"] = rn(geo.vocab_size, tw, std=0.02)
    sd["positional_embedding"] = rn(geo.context_length, tw, std=0.01)
    blocks("transformer", tw, geo.transformer_layers)
    ln("ln_final", tw)
    sd["text_projection"] = rn(tw, geo.embed_dim, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(1.0 / 0.07))
    return sd


def synthetic_text(n: int, geo: CLIPGeometry, seed: int = 567) -> torch.Tensor:
    """Token rows shaped like clip.tokenize output: SOT, body ids, ONE EOT (= largest id)
    at a per-row position in [2, L-1], zeros after.  int32, CPU."""
    g = torch.Generator().manual_seed(seed)
    L, V = geo.context_length, geo.vocab_size
    sot, eot = V - 2, V - 1
    t = torch.randint(1, V - 2, (n, L), generator=g, dtype=torch.int64)
    pos = torch.randint(2, L, (n,), generator=g)
    ar = torch.arange(L)[None, :]
    t = torch.where(ar < pos[:, None], t, torch.zeros_like(t))
    t[:, 0] = sot
    t[torch.arange(n), pos] = eot
    return t.to(torch.int32)


def synthetic_images(n: int, geo: CLIPGeometry, seed: int = 567) -> torch.Tensor:
    """N(0,1) fp32 NCHW = the post-Normalize distribution preprocess() produces."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, geo.image_resolution, geo.image_resolution, generator=g)
