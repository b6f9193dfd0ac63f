"""The reference's inference pipeline end to end on the device (application.py:80-229 / parse_coco.py:24-56 + test.py:517-545):
decoded image -> device preprocess -> encode_image -> zero-shot attribute (fixed prompts, encoded once) -> prefix mapper ->
KV-cached beam search.  Synthetic weights: the check is that every stage hands the next one what it expects, results are
deterministic, and each stage equals its separately tested form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Tok:
    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def test_image_to_caption_pipeline():
    from PIL import Image
    import clip
    from clip.data import ZeroShotClassifier
    from clip.weights import MODELS, init_state_dict, synthetic_text
    from clip_caption import ClipCaptionModel, CaptionGeometry, generate_beam, init_caption_state_dict
    cgeo = MODELS["test-small"]                                   # embed_dim 128 -> prefix_size of the caption model
    model = clip.build_model(init_state_dict(cgeo, 3), torch.float16).cuda().eval()
    geo = CaptionGeometry(vocab_size=300, n_embd=128, n_layer=2, n_head=2, n_positions=64, prefix_length=4, attribute_length=4,
                          prefix_size=cgeo.embed_dim)
    cap = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    cap.load_state_dict(init_caption_state_dict(geo, 9))
    cap = cap.cuda().eval().half()
    rng = np.random.default_rng(1)
    imgs = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB") for h, w in ((200, 300), (310, 170))]
    pre = clip.DevicePreprocess(cgeo.image_resolution)
    prompts = synthetic_text(9, cgeo, 4)
    zs = ZeroShotClassifier(model, prompts, [f"class{i}" for i in range(9)])

    def run():
        x = pre.batch(imgs)
        with torch.no_grad():
            feat = model.encode_image(x)
            sim, idx, labels = zs(image_features=feat)
            outs = []
            for i in range(len(imgs)):
                attribute = torch.full((1, geo.attribute_length), 1 + int(idx[i]), device="cuda", dtype=torch.int64)   # stands in for the tokenised label
                prefix_embed = cap.clip_project(feat[i:i + 1].float()).reshape(1, geo.prefix_length, -1)
                embed = torch.cat((prefix_embed, cap.gpt.transformer.wte(attribute)), dim=1)
                outs.append(generate_beam(cap, _Tok(), embed=embed, beam_size=3, entry_length=8, stop_token=5))
        return x, feat, sim, idx, outs

    x, feat, sim, idx, outs = run()
    assert x.shape == (2, 3, cgeo.image_resolution, cgeo.image_resolution) and feat.shape == (2, cgeo.embed_dim)
    assert sim.shape == (2, 9) and torch.allclose(sim.sum(1), torch.ones(2, device="cuda"), atol=1e-5)
    assert all(len(o) == 3 and all(isinstance(t, str) and t for t in o) for o in outs)
    x2, feat2, sim2, idx2, outs2 = run()
    assert torch.equal(x, x2) and torch.equal(feat, feat2) and torch.equal(idx, idx2) and outs == outs2
