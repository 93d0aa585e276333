"""Known-answer pins for the two OpenCV calls of the evaluation path that cannot be run here (opencv-python 4.5.5.62 is a
third-party dependency absent from /root/reference and from this image):

  * ``cv2.remap(depth_src, x_src, y_src, interpolation=cv2.INTER_LINEAR)``  (eval_rcmvsnet_dtu.py:301) -- restated as
    oracle.fusion.remap_linear, implemented on the GPU by csrc/fusion.hip;
  * ``cv2.resize(img, (new_w, new_h))``  (datasets/dtu_test.py:107-112,127-137) -- restated as oracle.dataset.resize_linear,
    implemented on the GPU by csrc/image_prep.hip.

The expected values below are NOT produced by the restatements: they are worked out by hand / in exact rational arithmetic
from OpenCV's published algorithm, in cases where that algorithm's result is known in closed form:

  remap, float32 single channel (modules/imgproc/src/imgwarp.cpp, cv::remap -> remapBilinear with the tables of
  initInterTab2D): the map is converted to fixed point with INTER_BITS = 5 (sx = cvRound(x * 32), round-half-to-even;
  integer part sx >> 5, fraction sx & 31), the four weights come from the 32 x 32 float table tab[fy][fx] =
  {(1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx} (products of the exactly representable 1-D taps i/32), and taps outside the image
  read the border value 0 (BORDER_CONSTANT, the default of cv2.remap).  Hence: integer coordinates return the source pixel;
  coordinates on the 1/32 grid return the exact bilinear value; any other coordinate returns the value of the NEAREST 1/32 grid
  point (ties to the even grid index); half-outside footprints lose exactly the outside taps.

  resize, float32 (modules/imgproc/src/resize.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear): source coordinate
  fx = (dx + 0.5) * scale - 0.5 in float, sx = cvFloor(fx), fx -= sx; sx < 0 -> (0, fx = 0); sx >= width-1 -> (width-1, fx = 0);
  taps {1-fx, fx}; horizontal pass over the two source rows first, vertical blend second, all float32.  Hence: an exact 2x
  shrink averages 2 x 2 blocks (0.5-weights: exact), the first / last columns of an enlargement replicate the border pixel,
  and a linear ramp is reproduced at the clamped source coordinate.
"""
from fractions import Fraction

import numpy as np

from oracle import dataset as ods
from oracle import fusion as ofu


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 512, size=(h, w)).astype(np.float32)       # small integers: every product below is exact


def _exact_bilinear(img, X, Y):
    """Exact rational bilinear value at rational coordinates (X, Y); taps outside the image read 0."""
    H, W = img.shape
    ix, iy = X.numerator // X.denominator, Y.numerator // Y.denominator
    ax, ay = X - ix, Y - iy
    acc = Fraction(0)
    for dy, wy in ((0, 1 - ay), (1, ay)):
        for dx, wx in ((0, 1 - ax), (1, ax)):
            x, y = ix + dx, iy + dy
            if 0 <= x < W and 0 <= y < H:
                acc += Fraction(int(img[y, x])) * wx * wy
    return float(acc)


def test_remap_integer_coordinates_and_outside():
    img = _img(6, 9, 0)
    xs, ys = np.meshgrid(np.arange(-2, 11, dtype=np.float32), np.arange(-2, 8, dtype=np.float32))
    out = ofu.remap_linear(img, xs, ys)
    for j in range(ys.shape[0]):
        for i in range(xs.shape[1]):
            x, y = int(xs[j, i]), int(ys[j, i])
            want = img[y, x] if (0 <= x < 9 and 0 <= y < 6) else 0.0
            assert out[j, i] == want, (x, y)


def test_remap_on_the_32nd_grid_is_exact_bilinear():
    img = _img(7, 8, 1)
    rng = np.random.default_rng(2)
    kx, ky = rng.integers(-40, 8 * 32 + 40, size=400), rng.integers(-40, 7 * 32 + 40, size=400)
    mx, my = (kx / 32.0).astype(np.float32), (ky / 32.0).astype(np.float32)                   # exactly representable
    out = ofu.remap_linear(img, mx.reshape(20, 20), my.reshape(20, 20)).reshape(-1)
    for n in range(400):
        assert out[n] == np.float32(_exact_bilinear(img, Fraction(int(kx[n]), 32), Fraction(int(ky[n]), 32))), (kx[n], ky[n])


def test_remap_quantises_to_the_nearest_32nd_ties_to_even():
    img = _img(5, 6, 3)
    base = np.array([[37, 70], [64, 33], [100, 99]], dtype=np.int64)                          # grid indices (x, y)
    for kx, ky in base:
        ref = np.float32(_exact_bilinear(img, Fraction(int(kx), 32), Fraction(int(ky), 32)))
        for dx, dy in ((0.3, -0.45), (-0.49, 0.2), (0.0, 0.49)):                              # within half a grid step: same grid point
            mx = np.array([[(kx + dx) / 32.0]], np.float32)
            my = np.array([[(ky + dy) / 32.0]], np.float32)
            assert ofu.remap_linear(img, mx, my)[0, 0] == ref, (kx, ky, dx, dy)
    # exact ties x * 32 = k + 0.5 go to the EVEN grid index (cvRound = round-half-to-even)
    for k in (40, 41, 66, 67):
        mx = np.array([[(k + 0.5) / 32.0]], np.float32)                                       # k + 0.5 over 32: exact in binary
        even = k if k % 2 == 0 else k + 1
        want = np.float32(_exact_bilinear(img, Fraction(even, 32), Fraction(2)))
        assert ofu.remap_linear(img, mx, np.array([[2.0]], np.float32))[0, 0] == want, k


def test_remap_half_outside_footprint_and_far_coordinates():
    img = _img(4, 5, 4)
    # x = -0.5: taps x = -1 (border 0) and x = 0 at weight 1/2 each; y integer
    assert ofu.remap_linear(img, np.array([[-0.5]], np.float32), np.array([[1.0]], np.float32))[0, 0] == np.float32(0.5) * img[1, 0]
    # x = W - 0.75: taps x = W-1 (weight 3/4) and x = W (border 0)
    assert ofu.remap_linear(img, np.array([[4.25]], np.float32), np.array([[2.0]], np.float32))[0, 0] == np.float32(0.75) * img[2, 4]
    # corner: x = -0.25, y = H - 0.5: only tap (0, H-1) is inside: weight 3/4 * 1/2
    assert ofu.remap_linear(img, np.array([[-0.25]], np.float32), np.array([[3.5]], np.float32))[0, 0] == np.float32(0.375) * img[3, 0]
    far = ofu.remap_linear(img, np.array([[1e12, -1e12, np.inf, np.nan]], np.float32), np.array([[1.0, 1.0, 1.0, 1.0]], np.float32))
    assert np.all(far == 0.0)


def test_resize_exact_halving_and_identity():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(8, 12, 3)).astype(np.float32)
    assert np.array_equal(ods.resize_linear(img, (12, 8)), img)
    out = ods.resize_linear(img, (6, 4))                                                      # (new_w, new_h)
    want = 0.25 * (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2])    # 0.5-weights: exact in fp32
    assert out.shape == (4, 6, 3) and np.array_equal(out, want.astype(np.float32))


def test_resize_enlargement_replicates_the_borders():
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(5, 7, 1)).astype(np.float32)
    out = ods.resize_linear(img, (14, 10))
    # first / last destination column and row map to fx = -0.25 -> (0, 0) and fx = n - 0.75 -> (n-1, 0): pure copies
    assert out[0, 0, 0] == img[0, 0, 0] and out[-1, -1, 0] == img[-1, -1, 0]
    assert out[0, -1, 0] == img[0, -1, 0] and out[-1, 0, 0] == img[-1, 0, 0]
    # interior destination (3, 4): fx = 3.5*0.5-0.5 = 1.25 -> taps 1, 2 at (0.75, 0.25); fy = 4.5*0.5-0.5 = 1.75 -> rows 1, 2 at (0.25, 0.75)
    r1 = img[1, 1, 0] * np.float32(0.75) + img[1, 2, 0] * np.float32(0.25)
    r2 = img[2, 1, 0] * np.float32(0.75) + img[2, 2, 0] * np.float32(0.25)
    assert out[4, 3, 0] == r1 * np.float32(0.25) + r2 * np.float32(0.75)


def test_resize_reproduces_a_linear_ramp_at_the_clamped_source_coordinate():
    H, W, nh, nw = 9, 13, 6, 8                                                                 # non-integer scales 1.5 and 1.625
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    img = (3.0 * xx + 5.0 * yy + 7.0)[..., None].astype(np.float32)
    out = ods.resize_linear(img, (nw, nh))[..., 0]
    fx = np.clip((np.arange(nw) + 0.5) * (W / nw) - 0.5, 0, W - 1)
    fy = np.clip((np.arange(nh) + 0.5) * (H / nh) - 0.5, 0, H - 1)
    want = 3.0 * fx[None, :] + 5.0 * fy[:, None] + 7.0
    assert np.abs(out - want).max() < 2e-5
