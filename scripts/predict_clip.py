#!/usr/bin/env python3
"""Zero-shot prediction - /root/reference/CLIP/predict.py:12-55 on the MI355X `clip` package: load the model (+ a fine-tuned
state_dict), preprocess the images, tokenize the prompts, `model(image, text)`, softmax over the prompts, arg-max label per image.
The reference's matplotlib figure (predict.py:57-75) is out of scope; the similarities are printed as JSON instead.

    python scripts/predict_clip.py --checkpoint models/clip_latest.pt --prompts violation status img1.jpg img2.jpg
    python scripts/predict_clip.py --synthetic --model test-small                                  # offline smoke run
"""
from __future__ import annotations

import argparse
import os
import tempfile

import _common as C
import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*")
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--prompts", nargs="+", default=["violation", "status"])              # predict.py:39
    ap.add_argument("--synthetic", action="store_true")
    args = ap.parse_args(argv)
    import clip
    from PIL import Image
    device = torch.device("cuda:0")
    tmp = None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        C.make_synthetic_annotations(tmp.name, per_class=2)
        d = os.path.join(tmp.name, "images")
        args.images = sorted(os.path.join(d, f) for f in os.listdir(d))[:16]
    model, preprocess = clip.load(args.model, device=device, jit=False)                   # predict.py:12
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=True))   # predict.py:14-16
    model.eval()
    image = torch.stack([preprocess(Image.open(p)) for p in args.images]).to(device)     # predict.py:28-34
    text = C.get_tokenize(model)(args.prompts).to(device)                                      # predict.py:39-40
    with torch.no_grad():
        logits_per_image, _ = model(image, text)                                          # predict.py:46
        similarity = logits_per_image.softmax(dim=-1)                                     # predict.py:47
    index = similarity.argmax(dim=1)                                                      # predict.py:54
    out = []
    for p, i, s in zip(args.images, index.tolist(), similarity.tolist()):
        out.append(dict(image=os.path.basename(p), label=args.prompts[i], similarity=[round(v, 5) for v in s]))
        C.log_line(**out[-1])
    if tmp is not None:
        tmp.cleanup()
    return out


if __name__ == "__main__":
    main()
