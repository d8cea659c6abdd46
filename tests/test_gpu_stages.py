"""GPU parity, stage by stage: every HIP kernel against the CPU oracle on the same inputs, bit-exact.
All calls go through the C ABI (include/linemod_hip.h) of liblinemod_hip.so."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det(lm):
    d = lm.Detector(color_only=False)
    yield d
    d.close()


def _rand_bgr(rng, h, w, smooth=True):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if smooth:  # blocky structure so that the >=5/9 vote keeps something
        img = np.kron(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8),
                      np.ones((8, 8, 1), np.uint8))[:h, :w]
        img = (img.astype(np.int32) + rng.integers(-3, 4, img.shape)).clip(0, 255).astype(np.uint8)
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (960, 1280), (37, 53), (16, 64), (130, 70), (8, 8)])
def test_color_quantize_parity(det, orc, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for smooth in (True, False):
        bgr = _rand_bgr(rng, shape[0], shape[1], smooth)
        q, mag = det.stage_color_quantize(bgr, want_magnitude=True)
        eq, emag = orc.color_quantize(bgr, want_magnitude=True)
        assert np.array_equal(q, eq)
        assert np.array_equal(mag, emag)


@pytest.mark.parametrize("variant", [1, 3, 4])
def test_color_quantize_both_blur_kernels(lm, det, orc, frame0, variant):
    """The Gaussian blur has three kernels -- one shot (few frames, 1), the row walker whose column sums travel between neighbouring
    lanes (batches, 3) and the matrix-core form (4); r02's plain sliding window (2) was deleted in r05 -- chosen by batch size:
    force each (LM_TUNE_CBLUR_VARIANT) on shapes that hit strip ends, row ends and both pyramid levels' widths."""
    det.set_tuning(lm.TUNE_CBLUR_VARIANT, variant)
    try:
        rng = np.random.default_rng(variant)
        # (variant 4, the matrix-core blur, takes rows of a multiple of 32 bytes -- 3 w % 32 == 0; other shapes fall back to variant 3)
        for shape in [(480, 640), (240, 320), (960, 1280), (16, 64), (17, 64), (50, 16), (130, 160), (33, 48), (97, 128), (200, 192), (1, 64), (31, 64), (96, 64), (193, 256)]:
            for smooth in (True, False):
                bgr = _rand_bgr(rng, shape[0], shape[1], smooth)
                assert np.array_equal(det.stage_color_quantize(bgr), orc.color_quantize(bgr)), (variant, shape, smooth)
        for extreme in (0, 255):
            bgr = np.full((64, 128, 3), extreme, np.uint8); bgr[20:40, 30:90] = 255 - extreme
            assert np.array_equal(det.stage_color_quantize(bgr), orc.color_quantize(bgr)), (variant, extreme)
        assert np.array_equal(det.stage_color_quantize(frame0[0]), orc.color_quantize(frame0[0]))
    finally:
        det.set_tuning(lm.TUNE_CBLUR_VARIANT, 0)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_color_quantize_both_gradient_kernels(lm, det, orc, frame0, variant):
    """Orientation + vote run as two kernels (few frames) or as the fused strip kernel with the integer orientation rule
    (batches): force each (LM_TUNE_CGRAD_VARIANT) on shapes that hit strip ends, wave ends (62 segments) and thresholds."""
    det.set_tuning(lm.TUNE_CGRAD_VARIANT, variant)
    try:
        rng = np.random.default_rng(10 + variant)
        for shape in [(480, 640), (240, 320), (960, 1280), (16, 64), (17, 64), (50, 16), (130, 160), (33, 48), (35, 1008)]:
            for smooth in (True, False):
                bgr = _rand_bgr(rng, shape[0], shape[1], smooth)
                assert np.array_equal(det.stage_color_quantize(bgr), orc.color_quantize(bgr)), (variant, shape, smooth)
        for thr in (0.0, 3.5, 10.0, 30.0, 200.0, 1e6):
            assert np.array_equal(det.stage_color_quantize(frame0[0], weak_threshold=thr), orc.color_quantize(frame0[0], thr)), thr
    finally:
        det.set_tuning(lm.TUNE_CGRAD_VARIANT, 0)


def test_color_quantize_frame0_and_thresholds(det, orc, frame0):
    bgr, _ = frame0
    assert np.array_equal(det.stage_color_quantize(bgr), orc.color_quantize(bgr))
    for thr in (0.0, 3.5, 30.0, 200.0):
        assert np.array_equal(det.stage_color_quantize(bgr, weak_threshold=thr), orc.color_quantize(bgr, thr))


def test_color_quantize_all_angles(det, orc):
    """Synthetic gradients in every direction (incl. exact bin boundaries of the fastAtan2 polynomial)."""
    h, w = 64, 512
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3), np.uint8)
    for k in range(8):
        a = 2 * np.pi * (k * 64 + xx[:, k * 64:(k + 1) * 64] - k * 64) / 512.0
        blk = 128 + 100 * np.sin(0.35 * ((xx[:, k * 64:(k + 1) * 64]) * np.cos(a) + yy[:, k * 64:(k + 1) * 64] * np.sin(a)))
        img[:, k * 64:(k + 1) * 64, k % 3] = blk.clip(0, 255)
    q = det.stage_color_quantize(img)
    e = orc.color_quantize(img)
    assert np.array_equal(q, e)
    assert len(np.unique(e)) >= 8


@pytest.mark.parametrize("shape", [(480, 640), (960, 1280), (36, 52), (17, 80)])
def test_pyrdown_parity(det, orc, shape):
    rng = np.random.default_rng(shape[1])
    bgr = rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
    assert np.array_equal(det.stage_pyrdown(bgr), orc.pyrdown(bgr))


@pytest.mark.parametrize("variant", [1, 2])
def test_pyrdown_both_kernels(lm, det, orc, frame0, variant):
    """cv::pyrDown has a one-shot kernel (a lane per 8 output pixels) and, for batches, a row-walking one whose column sums
    travel between neighbouring lanes (r03): force each (LM_TUNE_PYRDOWN_VARIANT) on shapes that hit strip ends, row ends,
    the reflected borders and both pyramid levels' sizes, on random and on extreme contents."""
    det.set_tuning(lm.TUNE_PYRDOWN_VARIANT, variant)
    try:
        rng = np.random.default_rng(variant)
        for shape in [(480, 640), (240, 320), (960, 1280), (4, 16), (6, 32), (34, 48), (66, 160), (130, 1024), (32, 16)]:
            for kind in range(3):
                if kind == 0:
                    bgr = rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
                elif kind == 1:
                    bgr = (rng.integers(0, 2, (shape[0], shape[1], 3)) * 255).astype(np.uint8)
                else:
                    bgr = np.full((shape[0], shape[1], 3), 255, np.uint8)
                assert np.array_equal(det.stage_pyrdown(bgr), orc.pyrdown(bgr)), (variant, shape, kind)
        assert np.array_equal(det.stage_pyrdown(frame0[0]), orc.pyrdown(frame0[0]))
    finally:
        det.set_tuning(lm.TUNE_PYRDOWN_VARIANT, 0)


@pytest.mark.parametrize("median_rows", [0, 2])     # 0: by batch size (one frame: 4 output rows per lane), 2: the batch form (16)
@pytest.mark.parametrize("shape", [(480, 640), (960, 1280), (40, 48), (23, 91), (12, 12), (17, 16), (33, 8)])
def test_depth_quantize_parity(lm, det, orc, synth, shape, median_rows, request):
    rng = np.random.default_rng(shape[0])
    h, w = shape
    det.set_tuning(lm.TUNE_DMEDIAN_VARIANT, median_rows)
    request.addfinalizer(lambda: det.set_tuning(lm.TUNE_DMEDIAN_VARIANT, 0))
    _, depth = synth.make_frame(max(w, 64), max(h, 64), seed=shape[0] + 5)
    depth = np.ascontiguousarray(depth[:h, :w])
    assert np.array_equal(det.stage_depth_quantize(depth), orc.depth_quantize(depth))
    noisy = rng.integers(0, 2600, (h, w)).astype(np.uint16)     # hits d >= 2000, |delta| >= 50, zeros
    assert np.array_equal(det.stage_depth_quantize(noisy), orc.depth_quantize(noisy))
    steps = (600 + 49 * rng.integers(0, 3, (h, w))).astype(np.uint16)   # right at the difference threshold
    assert np.array_equal(det.stage_depth_quantize(steps), orc.depth_quantize(steps))


@pytest.mark.parametrize("diff_thr", [1, 50, 249, 250, 5461, 5462, 40000])   # 249 / 250: integer vs double products of the normal; 5461 / 5462: packed vs per-pixel taps
def test_depth_quantize_thresholds_and_full_range(lm, orc, diff_thr):
    """k_dnormal runs its eight taps on packed pixel pairs (saturating u16 subtracts, i16 sums) up to
    difference_threshold 5461 and per pixel above: both sides of the switch, depths over the whole u16 range (deltas that
    wrap as i16), and steps right at the gate."""
    h, w = 96, 160
    d = lm.Detector(color_only=False, width=w, height=h, T=[4, 8], difference_threshold=diff_thr, distance_threshold=70000)
    rng = np.random.default_rng(diff_thr)
    full = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    near = (30000 + rng.integers(-diff_thr, diff_thr + 1, (h, w)).clip(-30000, 35535)).astype(np.uint16)
    edge = (1000 + (diff_thr - 1) * rng.integers(0, 3, (h, w)) + rng.integers(0, 2, (h, w))).clip(0, 65535).astype(np.uint16)
    for depth in (full, near, edge):
        assert np.array_equal(d.stage_depth_quantize(depth), orc.depth_quantize(depth, 70000, diff_thr)), diff_thr
    d.close()


@pytest.mark.parametrize("diff_thr", [50, 250, 6000])     # the three bodies of k_dnormal: packed taps + integer products, packed + double products, per-pixel taps
def test_depth_quantize_zero_length_normal(lm, orc, diff_thr):
    """a pixel whose normal has all three components 0 -- depth 0 among zeros (nz = -det * 0), or every one of the eight
    taps outside the bilateral gate (det = ddx = ddy = 0) -- makes the float tail's length 0; dn_sqrt(0) is a NaN and a float -> int
    conversion of a NaN is undefined.  The kernel now selects len = 1 for it, which lands on the table's "outside" entry = label 0,
    the oracle's `len > 0` result; whole images of such pixels, and lattices that mix them with ordinary ones."""
    h, w = 64, 96
    d = lm.Detector(color_only=False, width=w, height=h, T=[4, 8], difference_threshold=diff_thr)
    zeros = np.zeros((h, w), np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    lattice = np.where(((yy // 5) + (xx // 5)) % 2 == 0, 300, 300 + 4 * diff_thr).clip(0, 65535).astype(np.uint16)   # every tap 5 away crosses the gate
    mixed = lattice.copy(); mixed[20:40, 30:70] = 0; mixed[5:15, 5:25] = 700
    for depth in (zeros, lattice, mixed):
        got, exp = d.stage_depth_quantize(depth), orc.depth_quantize(depth, 2000, diff_thr)
        assert np.array_equal(got, exp), diff_thr
    assert not d.stage_depth_quantize(zeros).any()
    d.close()


def test_depth_normal_float_tail_sequences_are_exact(lm):
    """k_dnormal takes 1 / len and sqrt by short sequences (r04: dn_rcp = v_rcp + ONE Newton step, dn_sqrt = v_rsq + one coupled
    step g + (x - g g) y / 2; r03's longer forms -- v_rcp + six fused steps, v_sqrt_f32 + the +-1 ulp fix-up -- are swept beside
    them).  Every float of the
    tail's domain -- len in [1, 2^42], squared lengths in [1, 2^84] and 0 -- goes through both forms on the device, against
    the CORRECTLY ROUNDED 1.0f / x and sqrtf (r04: the reference used to be __fsqrt_rn, which this build lowers to the same
    bare v_sqrt_f32 the kernel used -- a test that could not fail); none may differ."""
    d = lm.Detector(color_only=False)
    assert d.selftest_float_tail() == (0, 0)
    print("floats on which the bare v_sqrt_f32 is not the correctly rounded root: %d" % d.last_bare_sqrt_mismatches)
    print("the longer sequences of r03 / r04a, floats that differ:", d.last_candidate_mismatches)
    assert all(v == 0 for k, v in d.last_candidate_mismatches.items() if not k.startswith("1 / root"))
    d.close()


def test_depth_quantize_frame0_and_custom_lut(lm, orc, frame0):
    _, depth = frame0
    d = lm.Detector(color_only=False)
    assert np.array_equal(d.stage_depth_quantize(depth), orc.depth_quantize(depth))
    rng = np.random.default_rng(9)
    lut = rng.integers(0, 256, 8000).astype(np.uint8)            # arbitrary bytes: median must be a true median
    d.set_normal_lut(lut)
    assert np.array_equal(d.stage_depth_quantize(depth), orc.depth_quantize(depth, lut=lut))
    # 0 / one-hot entries in another arrangement: stays on the streaming kernels (counting median)
    lut2 = np.where(rng.random(8000) < 0.1, 0, 1 << rng.integers(0, 8, 8000)).astype(np.uint8)
    d.set_normal_lut(lut2)
    assert np.array_equal(d.stage_depth_quantize(depth), orc.depth_quantize(depth, lut=lut2))
    d.close()


@pytest.mark.parametrize("w,h,T", [(640, 480, 5), (320, 240, 8), (640, 480, 2), (1280, 960, 2), (640, 480, 8),
                                   (40, 16, 8), (35, 21, 7), (48, 36, 3), (64, 64, 1), (24, 8, 4)])
def test_linear_memories_parity(det, orc, w, h, T):
    rng = np.random.default_rng(w * 7 + T)
    q = ((1 << rng.integers(0, 8, (h, w))) * (rng.random((h, w)) < 0.35)).astype(np.uint8)
    got = det.stage_linear_memories(q, T)
    spr = orc.spread(q, T)
    resp = orc.response_maps(spr)
    exp = np.stack([orc.linearize(resp[o], T) for o in range(8)])
    assert np.array_equal(got, exp)
    # not only one-hot input: any byte is a valid spread source
    q2 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    got2 = det.stage_linear_memories(q2, T)
    resp2 = orc.response_maps(orc.spread(q2, T))
    assert np.array_equal(got2, np.stack([orc.linearize(resp2[o], T) for o in range(8)]))


def test_linear_memories_custom_lut(lm, orc):
    d = lm.Detector(color_only=False)
    lut = orc.similarity_lut(1)
    d.set_similarity_lut(lut)
    rng = np.random.default_rng(4)
    q = ((1 << rng.integers(0, 8, (48, 80))) * (rng.random((48, 80)) < 0.4)).astype(np.uint8)
    resp = orc.response_maps(orc.spread(q, 8), lut)
    assert np.array_equal(d.stage_linear_memories(q, 8), np.stack([orc.linearize(resp[o], 8) for o in range(8)]))
    d.close()


@pytest.mark.parametrize("flags", [0, 1])   # 0: nibble-packed responses on the scanned level, 1: LM_FLAG_BYTE_RESPONSES
@pytest.mark.parametrize("color_only,size", [(False, (640, 480)), (True, (640, 480)), (True, (1280, 960))])
def test_full_preprocess_parity(lm, orc, synth, frame0, color_only, size, flags):
    """a3-a10 end to end on a resident frame: quantised images and all linear memories, every level."""
    w, h = size
    if size == (640, 480):
        bgr, depth = frame0
    else:
        bgr, depth = synth.make_frame(w, h, seed=77)
    d = lm.Detector(color_only=color_only, width=w, height=h, flags=flags)
    o = orc.Detector(color_only=color_only)
    d.upload_frame(0, bgr, None if color_only else depth)
    d.prepare_slot(0)
    o.prepare(bgr, None if color_only else depth)
    for level in range(2):
        for mod in range(1 if color_only else 2):
            assert np.array_equal(d.debug_read(0, 0, level, mod), o.stage(0, level, mod)), (level, mod)
            assert np.array_equal(d.debug_read(0, 2, level, mod), o.stage(2, level, mod)), (level, mod)
    d.close()
