"""Generate the committed golden vectors from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference holds no fixtures for this path and its `clip` dependency is absent,
so the vectors come from oracle/clip_oracle.py + oracle/caption_oracle.py, which are themselves pinned
against the local `transformers` CLIP / GPT-2 implementations (tests/test_oracle_pinning.py).
Weights are NOT stored: they are regenerated from the seed by clip.weights.init_state_dict /
clip_caption.weights.init_caption_state_dict (deterministic torch CPU RNG).  Stored: seeds, inputs that
are not seed-derivable, and expected outputs - a few hundred KB in total.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]

from clip.weights import MODELS, init_state_dict, synthetic_images, synthetic_text  # noqa: E402
from clip_caption.weights import (GPT2_MODELS, init_caption_state_dict, init_transformer_mapper_state_dict,  # noqa: E402
                                  synthetic_caption_batch)
from oracle import caption_oracle as CO  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
GRAD_KEYS = ["logit_scale", "visual.proj", "text_projection", "visual.conv1.weight", "visual.class_embedding",
             "visual.positional_embedding", "positional_embedding", "visual.ln_pre.weight", "ln_final.bias",
             "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.1.mlp.c_fc.bias",
             "transformer.resblocks.0.attn.out_proj.weight", "transformer.resblocks.1.mlp.c_proj.weight",
             "transformer.resblocks.0.ln_2.weight"]


def sample(t: torch.Tensor, keep: int = 4096) -> torch.Tensor:
    """Deterministic strided subsample of a gradient (keeps fixtures small); tests apply the same rule."""
    f = t.detach().flatten()
    k = max(1, -(-f.numel() // keep))
    return f[::k].clone()


def clip_case(name: str, n: int, seed: int, with_grads: bool):
    geo = MODELS[name]
    sd = init_state_dict(geo, seed)
    img, txt = synthetic_images(n, geo, seed + 1), synthetic_text(n, geo, seed + 2)
    # text rows with EOT at the first / last allowed position, and junk after EOT that must not matter
    txt[0] = 0; txt[0, 0] = geo.vocab_size - 2; txt[0, 1] = geo.vocab_size - 1
    txt[1, 1:] = torch.randint(1, geo.vocab_size - 2, (geo.context_length - 1,), generator=torch.Generator().manual_seed(9))
    txt[1, -1] = geo.vocab_size - 1
    out = dict(model=name, seed=seed, n=n, text=txt)
    with torch.no_grad():
        out["image_features"] = O.encode_image(sd, img)
        out["text_features"] = O.encode_text(sd, txt)
        li, lt = O.clip_forward(sd, img, txt)
        out["logits_per_image"] = li
        loss, acc = O.contrastive_loss(li, lt)
        out["loss"], out["acc"] = loss, acc
        # zero-shot shapes of CLIP/predict.py (1..n images x 2 prompts) and parse_coco.py (1 x 9)
        sim2, idx2 = O.zero_shot(sd, img, txt[:2])
        out["zs2_sim"], out["zs2_idx"] = sim2, idx2
        if n >= 9:
            sim9, idx9 = O.zero_shot(sd, img[:1], txt[:9])
            out["zs9_sim"], out["zs9_idx"] = sim9, idx9
    if with_grads:
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        li, lt = O.clip_forward(sdg, img, txt)
        loss, _ = O.contrastive_loss(li, lt)
        loss.backward()
        out["grads"] = {k: sample(sdg[k].grad) for k in GRAD_KEYS}
        out["grad_norms"] = {k: sdg[k].grad.double().norm().float() for k in sdg}
    return out


def image_only_case(name: str, n: int, seed: int):
    """encode_image alone (BASELINE.json configs[4]: ViT-L/14@336px; 577 tokens, patch 14)."""
    geo = MODELS[name]
    sd = init_state_dict(geo, seed)
    img = synthetic_images(n, geo, seed + 1)
    with torch.no_grad():
        feat = O.encode_image(sd, img)
    return dict(model=name, seed=seed, n=n, image_features=feat)


def caption_case(name: str, b: int, lc: int, seed: int):
    geo = GPT2_MODELS[name]
    sd = init_caption_state_dict(geo, seed)
    tokens, mask, prefix, attribute = synthetic_caption_batch(b, geo, lc, seed + 1)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "model.lm_head.weight"}
    sdg["model.lm_head.weight"] = sdg["model.transformer.wte.weight"]
    logits = CO.caption_forward(sdg, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
    loss = CO.caption_loss(logits, tokens, geo.prefix_length, geo.attribute_length)
    loss.backward()
    keys = ["clip_project.model.0.weight", "clip_project.model.2.bias", "model.transformer.wte.weight",
            "model.transformer.h.0.attn.c_attn.weight", "model.transformer.h.1.mlp.c_proj.weight",
            "model.transformer.ln_f.weight", "model.transformer.wpe.weight"]
    with torch.no_grad():
        mapped = CO.mlp_mapper(sd, prefix)
    return dict(model=name, seed=seed, b=b, lc=lc, mapper_out=sample(mapped, 8192),
                logits_slice=sample(logits.detach()[:, geo.prefix_length + geo.attribute_length - 1:-1], 8192),
                loss=loss.detach(), grads={k: sample(sdg[k].grad) for k in keys},
                grad_norms={k: v.grad.double().norm().float() for k, v in sdg.items() if v.grad is not None})


def caption_tmapper_case(name: str, b: int, lc: int, seed: int, clip_length: int, num_layers: int):
    """ClipCaptionModel with --mapping_type transformer (train.py:397): 8 heads, ReLU, mlp_ratio 2."""
    geo = GPT2_MODELS[name]
    sd = {k: v for k, v in init_caption_state_dict(geo, seed).items() if not k.startswith("clip_project.")}
    sd.update(init_transformer_mapper_state_dict(geo, clip_length, num_layers, seed + 7))
    tokens, mask, prefix, attribute = synthetic_caption_batch(b, geo, lc, seed + 1)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "model.lm_head.weight"}
    sdg["model.lm_head.weight"] = sdg["model.transformer.wte.weight"]
    logits = CO.caption_forward(sdg, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head, clip_length=clip_length)
    loss = CO.caption_loss(logits, tokens, geo.prefix_length, geo.attribute_length)
    loss.backward()
    with torch.no_grad():
        mapped = CO.transformer_mapper(sd, prefix, clip_length)
    keys = [k for k in sdg if k.startswith("clip_project.") and (".layers.0." in k or ".layers.1.attn" in k or "linear" in k or "prefix_const" in k)]
    return dict(model=name, seed=seed, b=b, lc=lc, clip_length=clip_length, num_layers=num_layers, mapper_out=sample(mapped, 8192),
                loss=loss.detach(), grads={k: sample(sdg[k].grad) for k in keys},
                grad_norms={k: v.grad.double().norm().float() for k, v in sdg.items() if v.grad is not None and k.startswith("clip_project.")})


CASES = {
    "clip_test_tiny.pt": lambda: clip_case("test-tiny", 9, 11, True),
    "clip_test_small.pt": lambda: clip_case("test-small", 9, 12, True),
    # BASELINE configs[1] geometry WITH gradients (round 2: the benched workload's backward against the oracle)
    "clip_vit_b32.pt": lambda: clip_case("ViT-B/32", 9, 567, True),
    "clip_test_long.pt": lambda: clip_case("test-long", 9, 13, True),
    "clip_vit_l14_336.pt": lambda: image_only_case("ViT-L/14@336px", 2, 567),
    "caption_test_tiny.pt": lambda: caption_case("test-tiny", 3, 12, 21),
    "caption_tmapper_tiny.pt": lambda: caption_tmapper_case("test-tiny", 3, 12, 23, clip_length=6, num_layers=2),
    # BASELINE configs[3] at its real geometry: GPT-2-small (V = 21128, 12 layers), mapper 512 -> 7680 -> 15360, P = A = 20,
    # Lc = 40 -> S = 80 (/root/reference/CLIP_prefix_caption/train.py:277-279, 354-357)
    "caption_gpt2_base_chinese.pt": lambda: caption_case("ckiplab/gpt2-base-chinese", 2, 40, 31),
}


def main():
    """python tests/golden/make_golden.py [file.pt ...]   (no arguments: every case)"""
    torch.manual_seed(0)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        torch.save(CASES[n](), os.path.join(OUT, n))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
