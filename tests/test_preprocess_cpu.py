"""Host side of the device preprocess: PIL's BICUBIC resampling windows / integer coefficients restated in
clip/preprocess_device.py, checked BIT-EXACT against PIL itself (the library the reference's `preprocess` runs on)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "construction-clip_amd"))


@pytest.mark.parametrize("w,h", [(640, 480), (480, 640), (224, 224), (1000, 333), (97, 61), (225, 224), (224, 500), (231, 229)])
def test_integer_resampler_restatement_equals_pil(w, h):
    from PIL import Image
    from clip.preprocess_device import reference_numpy, resized_size
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    n = 224
    nw, nh = resized_size(w, h, n)
    pil = Image.fromarray(img, "RGB")
    r = pil.resize((nw, nh), Image.BICUBIC) if (nw, nh) != (w, h) else pil
    left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
    want = np.asarray(r.crop((left, top, left + n, top + n)))
    assert np.array_equal(reference_numpy(img, n), want)


def test_coefficients_sum_to_one_and_windows_stay_inside():
    from clip.preprocess_device import PRECISION_BITS, resample_coeffs
    for a, b in [(640, 298), (61, 224), (224, 224), (1080, 224)]:
        bounds, kk, ksize = resample_coeffs(a, b)
        assert kk.shape == (b, ksize) and (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= a).all()
        assert np.abs(kk.sum(1) - (1 << PRECISION_BITS)).max() <= ksize          # rounding of each tap only
