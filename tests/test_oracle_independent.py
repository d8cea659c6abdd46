"""The oracle's image primitives against an implementation its author did not write: scipy.ndimage.

The oracle restates OpenCV's algorithms from recall (parity unpinned -- no OpenCV in the image); every stage is
also restated in numpy in tests/test_oracle.py, but by the same hand.  scipy is not OpenCV, yet it is independent: it takes the
border modes (replicate / reflect-101), window placements and tap alignments of the standard primitives off single-author recall.
What it cannot check is what only OpenCV defines: the 8-bit Gaussian's fixed-point rounding, fastAtan2's polynomial, NORMAL_LUT.
CPU only; runs where scipy is installed (the build container)."""
import numpy as np
import pytest

ndi = pytest.importorskip("scipy.ndimage")

SHAPES = [(48, 64), (23, 91), (12, 12), (17, 16), (9, 33), (5, 7)]


def _images(h, w, c=None, seed=0):
    rng = np.random.default_rng(seed + 31 * h + w)
    shape = (h, w) if c is None else (h, w, c)
    yield rng.integers(0, 256, shape).astype(np.uint8)
    yield (rng.integers(0, 2, shape) * 255).astype(np.uint8)
    ramp = (np.add.outer(np.arange(h) * 7, np.arange(w) * 3) % 256).astype(np.uint8)
    yield ramp if c is None else np.repeat(ramp[:, :, None], c, 2)


@pytest.mark.parametrize("shape", SHAPES)
def test_median5_is_scipys_median_filter_with_replicated_border(orc, shape):
    """medianBlur(.., 5) = the 13th of the 25 values of the 5 x 5 window, border pixels replicated: ndimage.median_filter(size 5,
    mode 'nearest')."""
    for img in _images(*shape):
        assert np.array_equal(orc.median5(img), ndi.median_filter(img, size=5, mode="nearest"))
    # one-hot labels, as the depth modality feeds it
    rng = np.random.default_rng(shape[0])
    lab = ((1 << rng.integers(0, 8, shape)) * (rng.random(shape) < 0.7)).astype(np.uint8)
    assert np.array_equal(orc.median5(lab), ndi.median_filter(lab, size=5, mode="nearest"))


@pytest.mark.parametrize("shape", SHAPES + [(480, 640)])
def test_pyrdown_is_a_mirrored_binomial_correlation_then_one_rounding(orc, shape):
    """cv::pyrDown = [1 4 6 4 1]^2 / 256 with BORDER_REFLECT_101 (scipy 'mirror': d c b | a b c d | c b a), every second pixel, one
    (s + 128) >> 8 at the end."""
    K = np.array([1, 4, 6, 4, 1], np.int64)
    for img in _images(*shape, c=3):
        s = ndi.correlate1d(ndi.correlate1d(img.astype(np.int64), K, axis=1, mode="mirror"), K, axis=0, mode="mirror")
        want = ((s[0:2 * (shape[0] // 2):2, 0:2 * (shape[1] // 2):2] + 128) >> 8).astype(np.uint8)
        assert np.array_equal(orc.pyrdown(img), want)


@pytest.mark.parametrize("shape", SHAPES)
def test_sobel3_is_scipys_correlation_with_replicated_border(orc, shape):
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    for img in _images(*shape, c=3):
        dx, dy = orc.sobel3(img)
        for ch in range(3):
            a = img[:, :, ch].astype(np.int64)
            assert np.array_equal(dx[:, :, ch], ndi.correlate(a, kx, mode="nearest"))
            assert np.array_equal(dy[:, :, ch], ndi.correlate(a, kx.T, mode="nearest"))
            # and scipy's own Sobel operator (smoothing [1 2 1] across, derivative [-1 0 1] along the axis)
            assert np.array_equal(dx[:, :, ch], ndi.sobel(a, axis=1, mode="nearest"))
            assert np.array_equal(dy[:, :, ch], ndi.sobel(a, axis=0, mode="nearest"))


@pytest.mark.parametrize("shape", SHAPES)
def test_gaussian7_taps_and_border(orc, shape):
    """The 7-tap kernel {8, 28, 56, 72, 56, 28, 8} / 256 per axis with BORDER_REPLICATE, in exact integers, one rounding
    (s + 2^15) >> 16: scipy's separable correlation gives the same integers before the rounding.  (WHICH rounding OpenCV's 8-bit path
    applies is not something scipy can say: that stays recall)"""
    K = np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    for img in _images(*shape, c=3):
        s = ndi.correlate1d(ndi.correlate1d(img.astype(np.int64), K, axis=1, mode="nearest"), K, axis=0, mode="nearest")
        assert np.array_equal(orc.gaussian7(img), ((s + 32768) >> 16).astype(np.uint8))
        # the kernel is cv::getGaussianKernel(7, sigma = 0.3 * ((7 - 1) * 0.5 - 1) + 0.8 = 1.4) in 8 fractional bits? no: OpenCV uses the
        # fixed table [0.03125, 0.109375, 0.21875, 0.28125, ...] for ksize 7, sigma <= 0 -- exactly K / 256
        assert np.allclose(K / 256.0, [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125])


@pytest.mark.parametrize("shape", SHAPES)
def test_erode3_is_scipys_grey_erosion(orc, shape):
    """erode 3 x 3: replicating the border and padding it with the maximum (OpenCV's default border value for erode) are the same
    minimum -- both scipy modes agree with the oracle, for 1 and 2 iterations."""
    for img in _images(*shape):
        for iters in (1, 2):
            a = b = img
            for _ in range(iters):
                a = ndi.grey_erosion(a, size=(3, 3), mode="nearest")
                b = ndi.grey_erosion(b, size=(3, 3), mode="constant", cval=255)
            got = orc.erode3(img, iters)
            assert np.array_equal(got, a) and np.array_equal(got, b)
        m = (img > 100).astype(np.uint8) * 255
        assert np.array_equal(orc.erode3(m, 1) > 0, ndi.binary_erosion(m > 0, structure=np.ones((3, 3)), border_value=1))


@pytest.mark.parametrize("shape", SHAPES)
def test_dist_c_is_scipys_chessboard_distance_transform(orc, shape):
    rng = np.random.default_rng(shape[1])
    for p in (0.02, 0.3, 0.9):
        src = (rng.random(shape) > p).astype(np.uint8) * 255
        src[rng.integers(0, shape[0]), rng.integers(0, shape[1])] = 0          # at least one zero pixel
        want = ndi.distance_transform_cdt(src > 0, metric="chessboard")
        assert np.array_equal(orc.dist_c(src), want.astype(np.float32))


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("shape", [(48, 64), (23, 91), (12, 12)])
def test_spread_is_a_bitwise_maximum_filter_over_the_window_to_the_right_and_below(orc, shape, T):
    """spread: dst(y, x) = OR of src over [y, y + T) x [x, x + T) inside the image -- per bit a maximum filter whose window starts AT the
    pixel (scipy: origin -(T // 2) moves the centred window there), zeros outside."""
    rng = np.random.default_rng(T)
    src = ((1 << rng.integers(0, 8, shape)) * (rng.random(shape) < 0.3)).astype(np.uint8)
    want = np.zeros(shape, np.uint8)
    for b in range(8):
        bit = (src >> b) & 1
        want |= (ndi.maximum_filter(bit, size=T, mode="constant", cval=0, origin=-(T // 2)) << b).astype(np.uint8)
    assert np.array_equal(orc.spread(src, T), want)
    # and the definition, by brute force (guards the origin convention used above)
    if shape == (12, 12):
        brute = np.zeros(shape, np.uint8)
        for y in range(shape[0]):
            for x in range(shape[1]):
                brute[y, x] = np.bitwise_or.reduce(src[y:y + T, x:x + T].ravel())
        assert np.array_equal(want, brute)


def test_frame0_through_the_independent_primitives(orc, frame0):
    """The reference's own benchmark frame: pyrDown, Sobel of the blurred image and the depth labels' median, oracle vs scipy."""
    bgr, depth = frame0
    K5, K7 = np.array([1, 4, 6, 4, 1], np.int64), np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    s = ndi.correlate1d(ndi.correlate1d(bgr.astype(np.int64), K5, axis=1, mode="mirror"), K5, axis=0, mode="mirror")
    assert np.array_equal(orc.pyrdown(bgr), ((s[::2, ::2] + 128) >> 8).astype(np.uint8))
    g = ndi.correlate1d(ndi.correlate1d(bgr.astype(np.int64), K7, axis=1, mode="nearest"), K7, axis=0, mode="nearest")
    blurred = ((g + 32768) >> 16).astype(np.uint8)
    assert np.array_equal(orc.gaussian7(bgr), blurred)
    dx, dy = orc.sobel3(blurred)
    for ch in range(3):
        assert np.array_equal(dx[:, :, ch], ndi.sobel(blurred[:, :, ch].astype(np.int64), axis=1, mode="nearest"))
        assert np.array_equal(dy[:, :, ch], ndi.sobel(blurred[:, :, ch].astype(np.int64), axis=0, mode="nearest"))
    q = orc.depth_quantize(depth)                                   # = median5 of the raw labels; idempotent check of the tail:
    assert np.array_equal(orc.median5(q), ndi.median_filter(q, size=5, mode="nearest"))
