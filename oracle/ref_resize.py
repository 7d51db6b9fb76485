"""CPU restatement of `scipy.ndimage.zoom(img, zoom, order=3)` as the reference's `preprocess` calls it
(/root/reference/test_mc3serousv5.py:100-113: defaults mode='constant', cval=0, prefilter=True, grid_mode=False).

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The algorithm lives in a third-party dependency of the reference (SciPy's ndimage: _interpolation.py `zoom`, C sources
ni_splines.c / ni_interpolation.c NI_ZoomShift); SciPy is importable in the build container, so this restatement is pinned
against SciPy itself (tests/test_oracle_resize.py, fixtures tests/golden/zoom_cubic.npz made by tools/gen_golden_resize.py):

  1. output shape = round(in * zoom) per axis; the sampling step is (in - 1) / (out - 1) (corner aligned), not 1 / zoom;
  2. B-spline prefilter in float64 along every zoomed axis: gain (1 - z)(1 - 1/z) = 6, pole z = sqrt(3) - 2, MIRROR boundary
     initialisation of the causal / anticausal recursions (mode 'constant' filters as 'mirror');
  3. output[i] = sum of 4 x 4 coefficients around floor(x) - 1 with the cubic B-spline weights, support indices mirrored;
  4. integer inputs: the float64 value is rounded to nearest (floor(v + 0.5)) and clipped to the input type's range
     (cv2.imread gives uint8); float inputs keep their type."""
import numpy as np

_POLE = np.sqrt(3.0) - 2.0


def _prefilter_axis(c, axis):
    """In-place cubic B-spline prefilter of float64 array `c` along `axis` (ni_splines.c apply_filter, mirror)."""
    c = np.moveaxis(c, axis, 0)
    n = c.shape[0]
    if n < 2:
        return
    z = _POLE
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    # causal initialisation (_init_causal_mirror)
    z_n_1 = z ** (n - 1)
    c0 = c[0] + z_n_1 * c[n - 1]
    z_i = z
    for i in range(1, n - 1):
        c0 = c0 + z_i * (c[i] + z_n_1 * c[n - 1 - i])
        z_i *= z
    c[0] = c0 / (1.0 - z_n_1 * z_n_1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    # anticausal (_init_anticausal_mirror)
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])


def _weights(t):
    """Cubic B-spline weights of ni_interpolation.c get_spline_interpolation_weights (order 3) for offsets t in [0, 1)."""
    z = 1.0 - t
    w1 = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0
    w2 = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0
    w0 = z * z * z / 6.0
    w3 = 1.0 - w0 - w1 - w2
    return np.stack([w0, w1, w2, w3], axis=-1)


def _mirror(idx, n):
    if n <= 1:
        return np.zeros_like(idx)
    s2 = 2 * n - 2
    idx = np.abs(idx) % s2
    return np.where(idx >= n, s2 - idx, idx)


def _axis_plan(n_in, n_out):
    x = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0)
    f = np.floor(x)
    idx = _mirror(f.astype(np.int64)[:, None] - 1 + np.arange(4)[None, :], n_in)
    return idx, _weights(x - f)


def zoom_cubic(img, out_hw):
    """img: [H, W] or [H, W, C] (uint8 or float); returns the array `zoom(img, (oh / H, ow / W[, 1]), order=3)` gives."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    oh, ow = int(round(H * (out_hw[0] / H))), int(round(W * (out_hw[1] / W)))
    c = img.astype(np.float64, copy=True)
    _prefilter_axis(c, 0)
    _prefilter_axis(c, 1)
    iy, wy = _axis_plan(H, oh)
    ix, wx = _axis_plan(W, ow)
    # rows first, in SciPy's accumulation order: sum over ky of wy * (sum over kx of wx * c)
    out = np.zeros((oh, ow) + img.shape[2:], dtype=np.float64)
    for ky in range(4):
        rows = c[iy[:, ky]]                                       # [oh, W, ...]
        acc = np.zeros((oh, ow) + img.shape[2:], dtype=np.float64)
        for kx in range(4):
            wgt = wx[:, kx].reshape((1, ow) + (1,) * (img.ndim - 2))
            acc += wgt * rows[:, ix[:, kx]]
        out += wy[:, ky].reshape((oh, 1) + (1,) * (img.ndim - 2)) * acc
    if np.issubdtype(img.dtype, np.integer):
        info = np.iinfo(img.dtype)
        out = np.clip(np.floor(out + 0.5), info.min, info.max)
    return out.astype(img.dtype)
