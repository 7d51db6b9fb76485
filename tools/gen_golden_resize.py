"""Fixtures for the cubic resize of the reference's `preprocess` (test_mc3serousv5.py:100-113): outputs of SciPy itself,
`scipy.ndimage.zoom(img, (oh / H, ow / W[, 1]), order=3)`, on seeded images -> tests/golden/zoom_cubic.npz.  The inputs are
regenerated from the seeds by the tests (numpy default_rng), only SciPy's outputs are stored."""
import os
import sys

import numpy as np
from scipy.ndimage import zoom

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # seed, input shape, (out_h, out_w), dtype
    (1, (37, 53), (64, 64), "uint8"), (2, (100, 80, 3), (64, 96), "uint8"), (3, (64, 48), (128, 96), "float32"),
    (4, (150, 120, 3), (128, 128), "float32"), (5, (17, 19), (33, 7), "uint8"), (6, (90, 70, 1), (64, 64), "uint8"),
]


def make(seed, shape, dtype):
    rng = np.random.default_rng(seed)
    img = rng.random(shape) * 255.0
    return img.astype(np.uint8) if dtype == "uint8" else (img / 255.0 - 0.3).astype(np.float32)


if __name__ == "__main__":
    import scipy
    out = {"scipy_version": np.array(scipy.__version__)}
    for i, (seed, shape, ohw, dtype) in enumerate(CASES):
        img = make(seed, shape, dtype)
        zf = (ohw[0] / shape[0], ohw[1] / shape[1]) + ((1,) if len(shape) == 3 else ())
        out[f"case{i}"] = zoom(img, zf, order=3)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "zoom_cubic.npz"), **out)
    print("wrote zoom_cubic.npz", {k: v.shape for k, v in out.items()})
