"""The fused gradient kernel (k_cgrad, lm_dev_color.h) replaces fastAtan2 + rint by an integer rule:

    s = (1282 min > 255 max) + (1384 min > 925 max),  q = |dy| > |dx| ? 4 - s : s,  label = (sign(dx) != sign(dy) ? -q : q) & 7

with min / max of (|dx|, |dy|) -- computed on the device as q = [255 a < 1282 b] + [925 a < 1384 b] + [1384 a < 925 b] + [1282 a < 255 b]
for (a, b) = (|dx|, |dy|), the same rule spelled out for both octants (four_sign_rule below).  A 3x3 Sobel of 8-bit data gives |dx|, |dy| <= 1020; this test sweeps ALL of those
(2041 x 2041 pairs) against the oracle's float code (cv::phase -> convertTo(CV_8U, 16/360) -> & 7).
"""
import numpy as np
import pytest


def integer_rule(dx, dy):
    ax, ay = np.abs(dx), np.abs(dy)
    mn, mx = np.minimum(ax, ay), np.maximum(ax, ay)
    s = (mn * 1282 > mx * 255).astype(np.int32) + (mn * 1384 > mx * 925)
    q = np.where(ay > ax, 4 - s, s)
    q = np.where((dx < 0) ^ (dy < 0), -q, q)
    return (q & 7).astype(np.uint8)


def four_sign_rule(dx, dy):
    """The form k_cgrad computes since r03: q = how many of the four sector bounds of the first quadrant |dy| / |dx| exceeds
    (one dot product per bound, its sign bit counted) -- the min / max rule spelled out for both octants."""
    a, b = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64)
    q = ((255 * a - 1282 * b < 0).astype(np.int32) + (925 * a - 1384 * b < 0) + (1384 * a - 925 * b < 0) + (1282 * a - 255 * b < 0))
    q = np.where((dx < 0) ^ (dy < 0), -q, q)
    return (q & 7).astype(np.uint8)


def test_integer_orientation_rule_matches_float_path_everywhere():
    from oracle import oracle as orc
    r = 1020
    dx, dy = np.meshgrid(np.arange(-r, r + 1, dtype=np.int32), np.arange(-r, r + 1, dtype=np.int32))
    want = orc.orientation_labels(dx, dy)
    got = integer_rule(dx, dy)
    assert want.shape == got.shape == (2 * r + 1, 2 * r + 1)
    assert np.array_equal(got, want)
    assert np.array_equal(four_sign_rule(dx, dy), want)
    assert set(np.unique(want).tolist()) == set(range(8))


def test_rule_thresholds_are_strictly_between_realised_ratios():
    """255/1282 and 925/1384 are mediants of neighbouring fractions with denominators <= 1020, so no realisable
    min/max equals them: the strict compare never sees equality except at (0, 0), where the label is 0."""
    from fractions import Fraction
    for lo, hi, mid in ((Fraction(182, 915), Fraction(73, 367), Fraction(255, 1282)),
                        (Fraction(661, 989), Fraction(264, 395), Fraction(925, 1384))):
        assert lo < mid < hi
        assert hi.numerator * lo.denominator - lo.numerator * hi.denominator == 1   # neighbours in the Farey sequence
        assert mid.denominator > 1020
    assert integer_rule(np.array([0]), np.array([0]))[0] == 0


def test_fused_and_unfused_fastatan2_give_the_same_labels_everywhere():
    """upstream's vector code path (v_atan_f32, what cv::phase runs on rows) evaluates the polynomial's
    three Horner steps as v_fma -- truly fused on an AVX2 build -- while the oracle and the float kernels restate the
    scalar atan_f32.  Swept over every gradient a 3x3 Sobel of 8-bit data can produce: neither the 8-bin label nor the
    16-bin value before `& 7` differs, with the scale of convertTo in float or in double.  So the integer rule's two
    bound pairs hold for either build of OpenCV and the kernels need no second variant."""
    from oracle import oracle as orc
    r = 1020
    dx, dy = np.meshgrid(np.arange(-r, r + 1, dtype=np.int32), np.arange(-r, r + 1, dtype=np.int32))
    base, base16 = orc.orientation_labels_variant(dx, dy, 0, want_raw16=True)
    assert np.array_equal(base, orc.orientation_labels(dx, dy))
    report = []
    for variant, name in ((1, "fused polynomial"), (2, "double convertTo"), (3, "fused polynomial + double convertTo")):
        lab, raw = orc.orientation_labels_variant(dx, dy, variant, want_raw16=True)
        report.append((name, int((lab != base).sum()), int((raw != base16).sum())))
    print("pairs swept: %d" % dx.size)
    for name, nl, nr in report:
        print("%-40s labels differing: %d   16-bin values differing: %d" % (name, nl, nr))
    assert all(nl == 0 and nr == 0 for _, nl, nr in report), report
    assert np.array_equal(integer_rule(dx, dy), base)


def test_color_quantize_is_unchanged_by_the_fused_variant():
    """The whole a3 stage on the reference's benchmark frame and on noise, with the oracle's fastAtan2 switched to the
    fused form: identical quantised images (the polynomial's values differ in the last bits, the labels never)."""
    import os
    from oracle import oracle as orc
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame0.npz"))
    rng = np.random.default_rng(7)
    imgs = [f["bgr"], rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)]
    for img in imgs:
        a = orc.color_quantize(img)
        old = orc.set_atan_variant(1)
        try:
            b = orc.color_quantize(img)
        finally:
            orc.set_atan_variant(old)
        assert np.array_equal(a, b)
