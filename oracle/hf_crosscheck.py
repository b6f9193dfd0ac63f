"""Pin the oracle against the only independent CLIP / GPT-2 arithmetic present in the build
container: `transformers` (config-only construction, random init, nothing fetched).
TEST INFRASTRUCTURE ONLY - used by tests/test_oracle_pinning.py (CPU, `-m "not gpu"`) and by
tests/golden/make_golden.py.  Nothing here travels into the product path.

Key mapping OpenAI layout -> HF layout:
  in_proj_weight rows [0:D]/[D:2D]/[2D:3D] -> q_proj/k_proj/v_proj; ln_pre -> pre_layrnorm;
  ln_post -> post_layernorm; ln_1/ln_2 -> layer_norm1/2; mlp.c_fc/c_proj -> mlp.fc1/fc2;
  visual.proj[W,E] -> visual_projection.weight[E,W] (transposed); text_projection likewise;
  positional_embedding -> position_embedding.weight; ln_final -> final_layer_norm.
"""
from __future__ import annotations

from typing import Dict

import torch


def openai_to_hf_clip(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}

    def blocks(src: str, dst: str):
        idx = sorted({int(k[len(src):].split(".")[0]) for k in sd if k.startswith(src)})
        for i in idx:
            s, d = f"{src}{i}.", f"{dst}{i}."
            w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
            D = w.shape[1]
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[d + f"self_attn.{n}.weight"] = w[j * D:(j + 1) * D].clone()
                out[d + f"self_attn.{n}.bias"] = b[j * D:(j + 1) * D].clone()
            out[d + "self_attn.out_proj.weight"] = sd[s + "attn.out_proj.weight"]
            out[d + "self_attn.out_proj.bias"] = sd[s + "attn.out_proj.bias"]
            for a, bb in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
                out[d + bb + ".weight"] = sd[s + a + ".weight"]
                out[d + bb + ".bias"] = sd[s + a + ".bias"]

    blocks("visual.transformer.resblocks.", "vision_model.encoder.layers.")
    blocks("transformer.resblocks.", "text_model.encoder.layers.")
    out["logit_scale"] = sd["logit_scale"]
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["text_model.final_layer_norm.weight"] = sd["ln_final.weight"]
    out["text_model.final_layer_norm.bias"] = sd["ln_final.bias"]
    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    out["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    out["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    out["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"]
    out["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    out["text_projection.weight"] = sd["text_projection"].t().contiguous()
    return out


def build_hf_clip(sd: Dict[str, torch.Tensor]):
    """Config-only HF CLIPModel with the oracle's weights loaded (no network)."""
    from transformers import CLIPConfig, CLIPModel
    from oracle.clip_oracle import infer_config

    c = infer_config(sd)
    cfg = CLIPConfig(
        text_config=dict(hidden_size=c["transformer_width"], intermediate_size=4 * c["transformer_width"],
                         num_hidden_layers=c["transformer_layers"], num_attention_heads=c["transformer_heads"],
                         max_position_embeddings=c["context_length"], vocab_size=c["vocab_size"],
                         projection_dim=c["embed_dim"], eos_token_id=c["vocab_size"] - 1,
                         bos_token_id=c["vocab_size"] - 2, pad_token_id=0),
        vision_config=dict(hidden_size=c["vision_width"], intermediate_size=4 * c["vision_width"],
                           num_hidden_layers=c["vision_layers"], num_attention_heads=c["vision_heads"],
                           image_size=c["image_resolution"], patch_size=c["vision_patch_size"],
                           projection_dim=c["embed_dim"]),
        projection_dim=c["embed_dim"])
    cfg._attn_implementation = "eager"
    model = CLIPModel(cfg).eval()
    missing, unexpected = model.load_state_dict(openai_to_hf_clip(sd), strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return model


def build_hf_gpt2(sd: Dict[str, torch.Tensor], n_head: int, prefix: str = "model."):
    """Config-only HF GPT2LMHeadModel with the caption oracle's `model.*` weights loaded."""
    from transformers import GPT2Config, GPT2LMHeadModel

    wte = sd[prefix + "transformer.wte.weight"]
    n_layer = len({k.split(".")[3] for k in sd if k.startswith(prefix + "transformer.h.")})
    cfg = GPT2Config(vocab_size=wte.shape[0], n_embd=wte.shape[1], n_layer=n_layer, n_head=n_head,
                     n_positions=sd[prefix + "transformer.wpe.weight"].shape[0],
                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    cfg._attn_implementation = "eager"
    model = GPT2LMHeadModel(cfg).eval()
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    sub.setdefault("lm_head.weight", sub["transformer.wte.weight"])
    missing, unexpected = model.load_state_dict(sub, strict=False)
    missing = [k for k in missing if not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"))]
    assert not missing and not unexpected, (missing, unexpected)
    return model
