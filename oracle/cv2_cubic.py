"""TEST INFRASTRUCTURE ONLY.  Restatement of OpenCV's `cv2.resize(..., INTER_CUBIC)`.

The reference derives its bicubic taps from OpenCV
(`/root/reference/codes/CEM/imresize_CEM.py:104-110`: `cv2.resize(delta_11x11,
(11*sf, 11*sf), interpolation=cv2.INTER_CUBIC)`).  `opencv-python` is an
un-pinned, un-vendored dependency that is absent from this image, so the
published algorithm is restated here (OpenCV `modules/imgproc/src/resize.cpp`,
`resizeGeneric_` + `interpolateCubic`, any 4.x):

  * destination pixel dx samples the source at fx = (dx + 0.5) * (src/dst) - 0.5,
    evaluated in double and rounded to float32; sx = floor(fx); t = fx - sx (float32)
  * four taps at sx-1 .. sx+2, indices clamped to the image (replicate border)
  * coefficients = Keys cubic with A = -0.75, evaluated in float32 exactly in
    OpenCV's operation order; for CV_64F images the coefficient type is still
    float32 (`HResizeCubic<double,double,float>`), accumulation is double
  * separable: horizontal pass, then vertical pass.

PARITY UNPINNED at this boundary: no OpenCV binary is available to check against
and the reference holds no golden taps.  What IS pinned: everything the reference
derives from these taps (tests/golden/cem_taps.npz) with this restatement plugged
in as `cv2`, and the structural known answers of SURVEY.md §4 (16x16 support for
x4, sum = sf**2, separability, margins 2/6/10).
"""
import numpy as np

INTER_CUBIC = 2


def _cubic_coeffs_f32(t):
    """OpenCV interpolateCubic(), float32 arithmetic in OpenCV's order."""
    f = np.float32
    A = f(-0.75)
    x = f(t)
    one = f(1.0)
    xp1 = f(x + one)
    c0 = f(f(f(f(f(A * xp1) - f(f(5) * A)) * xp1) + f(f(8) * A)) * xp1) - f(f(4) * A)
    c0 = f(c0)
    c1 = f(f(f(f(f(f(A + f(2)) * x) - f(A + f(3))) * x) * x) + one)
    omx = f(one - x)
    c2 = f(f(f(f(f(f(A + f(2)) * omx) - f(A + f(3))) * omx) * omx) + one)
    c3 = f(f(f(one - c0) - c1) - c2)
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def _axis_table(src_len, dst_len):
    scale = float(src_len) / float(dst_len)
    idx = np.zeros((dst_len, 4), dtype=np.int64)
    coef = np.zeros((dst_len, 4), dtype=np.float32)
    for d in range(dst_len):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        t = np.float32(fx - np.float32(sx))
        coef[d] = _cubic_coeffs_f32(t)
        idx[d] = np.clip(np.arange(sx - 1, sx + 3), 0, src_len - 1)
    return idx, coef


def resize(src, dsize, interpolation=INTER_CUBIC, **_):
    """cv2.resize for a single-channel 2-D float array; dsize = (width, height)."""
    assert interpolation == INTER_CUBIC, 'only INTER_CUBIC is restated'
    src = np.asarray(src, dtype=np.float64)
    assert src.ndim == 2
    dw, dh = int(dsize[0]), int(dsize[1])
    xi, xc = _axis_table(src.shape[1], dw)
    yi, yc = _axis_table(src.shape[0], dh)
    # horizontal pass (double accumulation of float32 coefficients)
    tmp = np.zeros((src.shape[0], dw), dtype=np.float64)
    for k in range(4):
        tmp += src[:, xi[:, k]] * xc[:, k].astype(np.float64)[None, :]
    out = np.zeros((dh, dw), dtype=np.float64)
    for k in range(4):
        out += tmp[yi[:, k], :] * yc[:, k].astype(np.float64)[:, None]
    return out
