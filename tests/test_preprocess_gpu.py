"""Device-side preprocess (csrc/preprocess.hip) against the host pipeline clip._transform (PIL resize + crop + numpy normalise):
bit-identical fp32 tensors, for down- and up-sampling, portrait / landscape / exact-size inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(640, 480), (480, 640), (224, 224), (1000, 333), (97, 61), (225, 224), (224, 500), (1920, 1080)])
@pytest.mark.parametrize("n", [224, 336])
def test_device_preprocess_is_bit_identical_to_pil_pipeline(w, h, n):
    from PIL import Image
    import clip
    rng = np.random.default_rng(w + 7 * h + n)
    img = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB")
    want = clip._transform(n)(img)
    got = clip.DevicePreprocess(n)(img)
    assert got.shape == (3, n, n) and got.dtype == torch.float32 and got.is_cuda
    assert torch.equal(got.cpu(), want)


def test_device_preprocess_batch_feeds_encode_image():
    from PIL import Image
    import clip
    from clip.weights import MODELS, init_state_dict
    rng = np.random.default_rng(5)
    imgs = [Image.fromarray(rng.integers(0, 256, size=(s, t, 3), dtype=np.uint8), "RGB") for s, t in ((300, 400), (500, 250), (224, 224))]
    pre = clip.DevicePreprocess(96)
    x = pre.batch(imgs)
    want = torch.stack([clip._transform(96)(im) for im in imgs])
    assert torch.equal(x.cpu(), want)
    model = clip.build_model(init_state_dict(MODELS["test-small"], 3)).cuda().eval()
    with torch.no_grad():
        assert torch.equal(model.encode_image(x), model.encode_image(want.cuda()))
    gray = Image.fromarray(rng.integers(0, 256, size=(120, 90), dtype=np.uint8), "L")      # non-RGB: host pipeline (RGB conversion happens after the resize)
    assert torch.equal(pre(gray).cpu(), clip._transform(96)(gray))
