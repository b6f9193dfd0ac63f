"""CPU: the host-side datasets on either side of the hot path, pinned by the structure of the reference's
own annotation file through tests/golden/all_json_summary.json (class order + counts derived from /root/reference/all.json
by tests/golden/make_dataset_fixture.py; no images exist offline, so a stub loader stands in)."""
import json
import os

import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _synthetic_json(tmp_path):
    s = json.load(open(os.path.join(GOLD, "all_json_summary.json")))
    info = s["keys"]["violation_type"]
    ann, i = [], 0
    # interleave classes so that "first seen" order equals the fixture's order
    for lab in info["labels"]:
        ann.append({"id": i, "violation_type": lab, "violation_list": f"v{i}", "caption": "", "file_name": f"{i}.jpg"}); i += 1
    for lab, cnt in zip(info["labels"], info["counts"]):
        for _ in range(cnt - 1):
            ann.append({"id": i, "violation_type": lab, "violation_list": f"v{i}", "caption": "", "file_name": f"{i}.jpg"}); i += 1
    for _ in range(s["n_annotations"] - info["n_nonempty"]):
        ann.append({"id": i, "violation_type": "", "violation_list": "", "caption": "", "file_name": f"{i}.jpg"}); i += 1
    p = tmp_path / "all.json"
    json.dump({"type": "captions", "annotations": ann}, open(p, "w"), ensure_ascii=False)
    return str(p), s


def _stubs():
    pre = lambda img: torch.full((3, 4, 4), float(img))                  # "image" = its id
    loader = lambda path: int(os.path.basename(path).split(".")[0])
    tok = lambda texts: torch.tensor([[hash(t) % 1000] for t in ([texts] if isinstance(texts, str) else texts)])
    return pre, loader, tok


def test_clip_pair_dataset_matches_reference_rules(tmp_path):
    from clip.data import ClipPairDataset
    path, s = _synthetic_json(tmp_path)
    pre, loader, tok = _stubs()
    ds9 = ClipPairDataset(pre, path, "", 0.8, "violation_type", "train", 9, image_loader=loader, tokenize=tok)
    assert list(ds9.train_count.values()) == [349, 17, 24, 39, 103, 20, 24, 39, 11]
    assert len(ds9.combination) == 1 and len(ds9) == 50
    img, text = ds9[7]
    assert img.shape == (9, 3, 4, 4) and text.shape[0] == 9
    anns = ds9.annotations_for(7)
    assert [a["violation_type"] for a in anns] == s["keys"]["violation_type"]["labels"]
    # class with 11 train items cycles: item 7 and item 18 pick the same annotation
    assert ds9.annotations_for(7)[-1]["id"] == ds9.annotations_for(18)[-1]["id"]
    ds2 = ClipPairDataset(pre, path, "", 0.8, "violation_type", "train", 2, image_loader=loader, tokenize=tok)
    assert len(ds2.combination) == 36 and len(ds2) == 36 * 50
    assert ds2.locate(0) == (0, 0) and ds2.locate(50) == (1, 0) and ds2.locate(1799) == (35, 49)
    test9 = ClipPairDataset(pre, path, "", 0.8, "violation_type", "test", 9, image_loader=loader, tokenize=tok)
    assert [len(v) for v in test9.pair_list[0].values()] == [437 - 349, 22 - 17, 31 - 24, 49 - 39, 129 - 103, 26 - 20, 30 - 24, 49 - 39, 14 - 11]


def test_caption_pair_dataset_split(tmp_path):
    from clip.data import ClipCaptionPairDataset
    path, s = _synthetic_json(tmp_path)
    pre, loader, tok = _stubs()
    n = s["keys"]["violation_type"]["n_nonempty"]
    tr = ClipCaptionPairDataset(pre, path, "", 0.8, "violation_list", "train", image_loader=loader, tokenize=tok)
    te = ClipCaptionPairDataset(pre, path, "", 0.8, "violation_list", "test", image_loader=loader, tokenize=tok)
    assert len(tr) == int(n * 0.8) and len(tr) + len(te) == n
    image, text = tr[3]
    assert image.shape == (3, 4, 4) and text.dim() == 1


class _CharTok:
    def encode(self, s):
        return [1 + (ord(c) % 250) for c in s]


def test_clip_coco_dataset_padding_mask_and_roundtrip(tmp_path):
    from clip_caption.data import ClipCocoDataset, load_embeddings, save_embeddings
    caps = [{"caption": "abc", "violation_list": "x", "attribute": "現況 墜落 ", "clip_embedding": 0},
            {"caption": "", "violation_list": "fallback text", "attribute": "缺失 感電 ", "clip_embedding": 1},
            {"caption": "a much longer caption than the others", "violation_list": "", "attribute": "缺失 物料 ", "clip_embedding": 2}]
    emb = torch.randn(3, 512)
    p = str(tmp_path / "emb.pkl")
    save_embeddings(p, emb, caps)
    assert torch.equal(load_embeddings(p)["clip_embedding"], emb)
    ds = ClipCocoDataset(p, prefix_length=20, attribute_length=8, tokenizer=_CharTok())
    lens = torch.tensor([3.0, 13.0, 37.0])
    assert ds.max_seq_len == min(int(lens.mean() + lens.std() * 10), 37)
    assert os.path.exists(str(tmp_path / "emb_tokens.pkl"))                       # train.py:103 side effect
    tokens, mask, prefix, attribute = ds[1]
    assert ds.captions[1] == "fallback text"                                      # empty caption -> violation_list (train.py:85)
    assert tokens.shape == (ds.max_seq_len,) and attribute.shape == (8,) and mask.shape == (20 + 8 + ds.max_seq_len,)
    assert mask.sum() == mask.numel() and tokens[13:].sum() == 0                  # pads are id 0, mask stays all ones
    assert torch.equal(prefix, emb[1])
    ds_n = ClipCocoDataset(p, 20, 8, tokenizer=_CharTok(), normalize_prefix=True, write_tokens_cache=False)
    assert abs(ds_n[0][2].norm().item() - 1.0) < 1e-6


def test_clip_coco_dataset_matches_reference_pad_tokens_fixture():
    """tests/golden/ref_pad_tokens.pt was produced by running the reference's own ClipCocoDataset.pad_tokens / __getitem__
    (CLIP_prefix_caption/train.py:32-63) on hand-made token lists (tests/golden/make_reference_fixtures.py): captions shorter
    than / equal to / longer than max_seq_len, attributes on both sides of attribute_length, one negative id."""
    from clip_caption.data import ClipCocoDataset
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pad_tokens.pt"), weights_only=True)
    for key, normalize in (("plain", False), ("normalized", True)):
        ds = ClipCocoDataset.from_token_lists(fx["captions_tokens"], fx["attributes_tokens"], fx["caption2embedding"], fx["prefixes"],
                                              fx["prefix_length"], fx["attribute_length"], normalize, max_seq_len=fx["max_seq_len"])
        assert len(ds) == len(fx["captions_tokens"])
        for epoch in range(2):                              # the dense layout answers every epoch like the reference's FIRST visit
            for i, (tokens, mask, prefix, attribute) in enumerate(fx["items"][key]["first"]):
                t, m, p, a = ds[i]
                assert torch.equal(t, tokens) and torch.equal(m, mask) and torch.equal(a, attribute), i
                assert torch.allclose(p, prefix, rtol=0, atol=0 if not normalize else 1e-7), i
                pt, pa, pm = ds.pad_tokens(i)
                assert torch.equal(pt, tokens) and torch.equal(pa, attribute) and torch.equal(pm, mask)
        # the one documented difference: the reference zeroes a negative id inside its cache, so its second visit of item 2
        # reports an all-ones mask; every other item is identical on both visits
        second = fx["items"][key]["second"]
        diff = [i for i in range(len(ds)) if not torch.equal(second[i][1], ds[i][1])]
        assert diff == [2] and second[2][1].min() == 1 and ds[2][1].min() == 0
        tk, mk, pf, at = ds.tensors()
        assert tk.shape == (len(ds), fx["max_seq_len"]) and mk.shape[1] == fx["prefix_length"] + fx["attribute_length"] + fx["max_seq_len"]
        assert torch.allclose(pf[3], ds[3][2])
