#!/usr/bin/env python3
"""Contrastive fine-tune of CLIP on (image, label) groups - the loop of /root/reference/CLIP/train.py:101-217 on the MI355X
`clip` package: class-balanced K-way groups (ClipPairDataset), `model(image, text)`, symmetric cross-entropy, HF-AdamW with
linear warm-up, per-epoch evaluation, periodic state_dict checkpoints in the OpenAI key layout.

    python scripts/train_clip.py --json all.json --image-path data/ --key violation_type --combination-num 9
    python scripts/train_clip.py --synthetic --model test-small --epochs 1 --max-steps 3        # offline smoke run

Differences from the reference script, on purpose: constants became flags; TensorBoard (not installed) became one JSON line per
step on stdout; `--fused-loss` uses clip.contrastive_loss (fused logits + CE, the data-parallel form under torchrun)."""
from __future__ import annotations

import argparse
import os
import tempfile

import _common as C  # noqa: F401  (path setup)
import torch
from torch.utils.data import DataLoader


def build_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--checkpoint", default=None, help="state_dict to start from (train.py:108-111)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--image-path", default="")
    ap.add_argument("--key", default="violation_type")
    ap.add_argument("--combination-num", type=int, default=9)
    ap.add_argument("--train-ratio", type=float, default=0.8)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--batch-size", type=int, default=1, help="groups per step (train.py:138: 1 group of K pairs)")
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--warmup-steps", type=int, default=5000)
    ap.add_argument("--save-every", type=int, default=100)
    ap.add_argument("--out-dir", default="models")
    ap.add_argument("--name", default="clip_balance")
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many optimiser steps (0: run all epochs)")
    ap.add_argument("--fused-loss", action="store_true")
    ap.add_argument("--synthetic", action="store_true", help="generated images + labels, seeded weights, byte tokenizer")
    ap.add_argument("--seed", type=int, default=567)
    return ap


def evaluate(model, loader, device):
    """train.py:191-207: accuracy of arg-max(logits_per_image) against arange over the test groups"""
    model.eval()
    hit = tot = 0
    with torch.no_grad():
        for image, text in loader:
            image, text = image.to(device).flatten(0, 1), text.to(device).flatten(0, 1)
            logits_i, _ = model(image, text)
            label = torch.arange(image.shape[0], device=device)
            hit += int((logits_i.argmax(1) == label).sum())
            tot += image.shape[0]
    model.train()
    return hit / max(tot, 1)


def main(argv=None):
    args = build_args().parse_args(argv)
    import clip
    from clip import optim as coptim
    from clip.data import ClipPairDataset
    torch.manual_seed(args.seed)
    device = torch.device("cuda:0")
    tmp = None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        args.json, args.image_path = C.make_synthetic_annotations(tmp.name), tmp.name
    model, preprocess = clip.load(args.model, device=device, jit=False)
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=True))
    tokenize = C.get_tokenize(model)
    mk = lambda split: ClipPairDataset(preprocess, args.json, args.image_path, args.train_ratio, args.key, split,   # noqa: E731
                                       args.combination_num, tokenize=tokenize)
    train_ds, test_ds = mk("train"), mk("test")
    train_dl = DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, num_workers=args.workers)
    test_dl = DataLoader(test_ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers)
    model.train()
    opt = coptim.AdamW(model, lr=args.lr)                                                   # train.py:143
    sched = coptim.get_linear_schedule_with_warmup(opt, args.warmup_steps, args.epochs * len(train_dl))   # train.py:145-147
    ce = torch.nn.CrossEntropyLoss()
    os.makedirs(args.out_dir, exist_ok=True)
    step = 0
    for epoch in range(1, args.epochs + 1):
        for image, text in train_dl:                                                        # train.py:157
            image, text = image.to(device).flatten(0, 1), text.to(device).flatten(0, 1)     # [G, K, ...] -> [G*K, ...]
            opt.zero_grad()
            label = torch.arange(image.shape[0], device=device)
            if args.fused_loss:
                fi, ft = model.encode_image_text(image, text)
                loss, stats = clip.contrastive_loss(fi, ft, model.logit_scale, None)
                acc = float(stats[1]) / image.shape[0] if len(stats) > 1 else float("nan")
            else:
                logits_i, logits_t = model(image, text)                                     # train.py:161
                loss = (ce(logits_i, label) + ce(logits_t, label)) / 2                      # train.py:164-166
                acc = float((logits_i.argmax(1) == label).float().mean())                   # train.py:173
            loss.backward()
            opt.step()
            sched.step()
            step += 1
            C.log_line(epoch=epoch, step=step, loss=round(float(loss.detach()), 6), accuracy=round(acc, 4), lr=sched.get_last_lr()[0])
            if args.max_steps and step >= args.max_steps:
                break
        test_acc = evaluate(model, test_dl, device)
        C.log_line(epoch=epoch, testing_accuracy=round(test_acc, 4))
        if epoch % args.save_every == 0 or (args.max_steps and step >= args.max_steps) or epoch == args.epochs:
            path = os.path.join(args.out_dir, f"{args.name}_comb{args.combination_num}_{epoch}.pt")
            torch.save(model.state_dict(), path)                                            # train.py:211-217
            C.log_line(saved=path)
        if args.max_steps and step >= args.max_steps:
            break
    if tmp is not None:
        tmp.cleanup()
    return step


if __name__ == "__main__":
    main()
