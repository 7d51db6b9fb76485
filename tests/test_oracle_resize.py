"""CPU: the resize oracle (oracle/ref_resize.py, SciPy's cubic-spline zoom restated) against outputs of SciPy itself
(tests/golden/zoom_cubic.npz, tools/gen_golden_resize.py) and, where SciPy is importable, against SciPy directly."""
import os

import numpy as np
import pytest

from oracle import ref_resize
from tools import gen_golden_resize as G


@pytest.mark.parametrize("i", range(len(G.CASES)))
def test_zoom_oracle_matches_scipy_fixture(golden_dir, i):
    g = np.load(os.path.join(golden_dir, "zoom_cubic.npz"))
    seed, shape, ohw, dtype = G.CASES[i]
    img = G.make(seed, shape, dtype)
    got = ref_resize.zoom_cubic(img, ohw)
    want = g[f"case{i}"]
    assert got.shape == want.shape and got.dtype == want.dtype
    if dtype == "uint8":
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


def test_zoom_oracle_matches_scipy_live():
    zoom = pytest.importorskip("scipy.ndimage").zoom
    rng = np.random.default_rng(11)
    img = (rng.random((45, 61, 3)) * 255).astype(np.uint8)
    np.testing.assert_array_equal(ref_resize.zoom_cubic(img, (96, 32)), zoom(img, (96 / 45, 32 / 61, 1), order=3))
