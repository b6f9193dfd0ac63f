#!/usr/bin/env python3
"""Prefix-caption trainer - the loop of /root/reference/CLIP_prefix_caption/train.py:326-421 on the MI355X `clip_caption`
package: ClipCocoDataset (pickle of CLIP embeddings + captions), ClipCaptionModel (MLP or transformer mapper + GPT-2),
cross-entropy on the shifted logits with ignore_index 0, HF-AdamW lr 2e-5 with 5000 warm-up steps, periodic checkpoints.
Flags are the reference's (train.py:386-402).

    python scripts/train_caption.py --data ./data/embedding.pkl --out_dir ./checkpoints --prefix caption --tokenizer ckiplab/gpt2-base-chinese
    python scripts/train_caption.py --synthetic --gpt2 test-tiny --epochs 1 --max-steps 3 --bs 8      # offline smoke run

`--fused-loss` (default) uses model.caption_loss - the same loss with only the needed rows through ln_f / lm_head; `--no-fused-loss`
is the reference's literal `model(tokens, prefix, attribute, mask)` -> logits slice -> F.cross_entropy."""
from __future__ import annotations

import argparse
import os
import tempfile

import _common as C
import torch
from torch.nn import functional as nnf
from torch.utils.data import DataLoader


def make_synthetic_pickle(path: str, geo, n: int = 64, seed: int = 567):
    """an embedding pickle in parse_coco.py's layout: {"clip_embedding": [N, prefix_size], "captions": [annotation dicts]}"""
    from clip_caption.data import save_embeddings
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(n, geo.prefix_size, generator=g)
    caps = [dict(clip_embedding=i, caption=f"worker {i} near zone {i % 7} edge", violation_list=f"hazard {i % 5}",
                 attribute=f"{'violation' if i % 2 else 'status'} {C.CLASSES[i % 9]} ") for i in range(n)]
    save_embeddings(path, emb, caps)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="./data/embedding.pkl")
    ap.add_argument("--out_dir", default="./checkpoints")
    ap.add_argument("--prefix", default="caption", help="prefix for saved filenames")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--save_every", type=int, default=1)
    ap.add_argument("--prefix_length", type=int, default=None)
    ap.add_argument("--attribute_length", type=int, default=None)
    ap.add_argument("--prefix_length_clip", type=int, default=10)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--only_prefix", action="store_true")
    ap.add_argument("--mapping_type", default="mlp", help="mlp/transformer")
    ap.add_argument("--num_layers", type=int, default=8)
    ap.add_argument("--normalize_prefix", action="store_true")
    ap.add_argument("--tokenizer", default="ckiplab/gpt2-base-chinese")
    ap.add_argument("--gpt2", default=None, help="geometry name in clip_caption.GPT2_MODELS (default: the tokenizer's)")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--warmup_steps", type=int, default=5000)
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--no-fused-loss", dest="fused", action="store_false")
    ap.add_argument("--synthetic", action="store_true")
    args = ap.parse_args(argv)

    from clip import optim as coptim
    from clip_caption import (ClipCaptionModel, ClipCaptionPrefix, GPT2_MODELS, MappingType, init_caption_state_dict,
                              init_transformer_mapper_state_dict)
    from clip_caption.data import ClipCocoDataset
    device = torch.device("cuda:0")
    geo = GPT2_MODELS[args.gpt2 or args.tokenizer]
    P = args.prefix_length or geo.prefix_length
    A = args.attribute_length or geo.attribute_length
    tmp, tokenizer = None, None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        args.data = os.path.join(tmp.name, "embedding.pkl")
        make_synthetic_pickle(args.data, geo)
        tokenizer = C.ByteCaptionTokenizer(geo.vocab_size)
        args.out_dir = os.path.join(tmp.name, "checkpoints")
    dataset = ClipCocoDataset(args.data, P, A, gpt2_type=args.tokenizer, normalize_prefix=args.normalize_prefix,
                              tokenizer=tokenizer)                                        # train.py:406
    mt = {"mlp": MappingType.MLP, "transformer": MappingType.Transformer}[args.mapping_type]
    cls = ClipCaptionPrefix if args.only_prefix else ClipCaptionModel                     # train.py:409-416
    model = cls(P, clip_length=args.prefix_length_clip, prefix_size=geo.prefix_size, num_layers=args.num_layers,
                mapping_type=mt, gpt2_type=geo)
    if args.synthetic or not os.environ.get("CCLIP_GPT2_CHECKPOINT"):
        sd = init_caption_state_dict(geo, 567)                                            # no network: seeded GPT-2 + mapper
        if mt == MappingType.Transformer:
            sd = {k: v for k, v in sd.items() if not k.startswith("clip_project.")}
            sd.update(init_transformer_mapper_state_dict(geo, args.prefix_length_clip, args.num_layers, 567))
        model.load_state_dict(sd)
    else:
        model.load_state_dict(torch.load(os.environ["CCLIP_GPT2_CHECKPOINT"], map_location="cpu", weights_only=True))
    model = model.to(device)
    model.train()
    os.makedirs(args.out_dir, exist_ok=True)
    opt = coptim.AdamW(model, lr=args.lr)                                                  # train.py:336
    loader = DataLoader(dataset, batch_size=args.bs, shuffle=True, drop_last=True)        # train.py:337
    sched = coptim.get_linear_schedule_with_warmup(opt, args.warmup_steps, args.epochs * len(loader))   # train.py:338-340
    step = 0
    for epoch in range(args.epochs):
        for tokens, mask, prefix, attribute in loader:                                    # train.py:348
            opt.zero_grad()
            tokens, mask, attribute = tokens.to(device), mask.to(device), attribute.to(device)
            prefix = prefix.to(device, dtype=torch.float32)                               # train.py:353
            if args.fused:
                loss = model.caption_loss(tokens, prefix, attribute, mask)
            else:
                outputs = model(tokens, prefix, attribute, mask)                          # train.py:354
                logits = outputs.logits[:, P + A - 1: -1]                                 # train.py:356
                loss = nnf.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)   # :357
            loss.backward()
            opt.step()
            sched.step()
            step += 1
            C.log_line(epoch=epoch, step=step, loss=round(float(loss.detach()), 6))
            if args.max_steps and step >= args.max_steps:
                break
        if epoch % args.save_every == 0 or epoch == args.epochs - 1:                      # train.py:377-381
            path = os.path.join(args.out_dir, f"{args.prefix}-{epoch:03d}.pt")
            torch.save(model.state_dict(), path)
            C.log_line(saved=path)
        if args.max_steps and step >= args.max_steps:
            break
    if tmp is not None:
        tmp.cleanup()
    return step


if __name__ == "__main__":
    main()
