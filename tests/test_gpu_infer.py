"""GPU: the on-device cubic resize of `umi.infer.preprocess` (reference test_mc3serousv5.py:100-127) against SciPy's outputs
(fixtures) and the whole preprocess against the reference's formula on the oracle's resize."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_resize
from tools import gen_golden_resize as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(len(G.CASES)))
def test_zoom_cubic_on_device_matches_scipy(golden_dir, i):
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    from umi import infer
    g = np.load(os.path.join(golden_dir, "zoom_cubic.npz"))
    seed, shape, ohw, dtype = G.CASES[i]
    img = G.make(seed, shape, dtype)
    got = infer.zoom_cubic(img, ohw).cpu().numpy()
    want = g[f"case{i}"]
    assert got.shape == want.shape and got.dtype == want.dtype
    if dtype == "uint8":
        np.testing.assert_array_equal(got, want)          # bit-exact for byte images (what cv2.imread gives the reference)
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("shape", [(100, 80, 3), (70, 90)])
def test_preprocess_with_resize_matches_reference_formula(shape):
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    from umi import infer
    rng = np.random.default_rng(5)
    img = (rng.random(shape) * 255).astype(np.uint8)
    size = (64, 96)
    x = infer.preprocess(img, input_size=size).cpu().numpy()
    z = ref_resize.zoom_cubic(img, size)                     # == scipy.ndimage.zoom(order=3) (tests/test_oracle_resize.py)
    z = (z - np.mean(z, axis=(0, 1))) / np.std(z, axis=(0, 1))
    want = z.astype(np.float32)[None, None] if len(shape) == 2 else z.transpose((2, 0, 1))[::-1].astype(np.float32)[None]
    assert x.shape == want.shape
    np.testing.assert_allclose(x, want, rtol=0, atol=2e-6)
