"""CPU tests of the oracle itself: golden vectors on the reference's own frame, known-answer tests,
algebraic properties of each stage and brute-force cross-checks written independently in numpy.
(The reference has no tests for this path -- SURVEY.md section 4 -- so this pyramid is ours.)"""
import hashlib

import numpy as np
import pytest

from conftest import assert_matches_equal, crop_masks


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# golden vectors (pin the oracle against silent drift)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,color_only", [("rgbd", False), ("color", True)])
def test_golden_frame0(orc, frame0, golden0, name, color_only):
    bgr, depth = frame0
    det = orc.Detector(color_only=color_only)
    det.add_class("obj", golden0[name + "_descs"], golden0[name + "_features"])
    m = det.match(bgr, None if color_only else depth, 80.0)
    assert_matches_equal(m, golden0[name + "_matches"])
    hashes = dict(s.split(":") for s in golden0[name + "_hashes"])
    for l in range(det.pyramid_levels):
        for mod in range(det.num_modalities):
            assert sha(det.stage(0, l, mod)) == hashes["q%d%d" % (l, mod)]
            assert sha(det.stage(2, l, mod)) == hashes["lm%d%d" % (l, mod)]


@pytest.mark.parametrize("color_only", [False, True])
def test_golden_extraction_is_reproduced(orc, frame0, golden0, color_only):
    """addTemplate on the same crops reproduces the committed template bank bit for bit."""
    bgr, depth = frame0
    name = "color" if color_only else "rgbd"
    det = orc.Detector(color_only=color_only)
    for m in crop_masks(640, 480, 7, 6):
        tid, _ = det.add_template("obj", bgr, None if color_only else depth, m)
        assert tid >= 0
    descs, feats = det.export_class(0)
    assert np.array_equal(descs, golden0[name + "_descs"])
    assert np.array_equal(feats, golden0[name + "_features"])


# ---------------------------------------------------------------------------------------------
# known-answer: a template cut out of the frame is found where it was cut, with similarity 100
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("color_only", [False, True])
def test_known_answer_self_extracted(orc, frame0, color_only):
    bgr, depth = frame0
    det = orc.Detector(color_only=color_only)
    T0 = det.cfg.T[0]
    boxes = []
    for m in crop_masks(640, 480, 11, 8):
        tid, bb = det.add_template("obj", bgr, None if color_only else depth, m)
        if tid >= 0:          # a featureless crop legitimately fails (upstream returns -1)
            assert tid == len(boxes)
            boxes.append(bb)
    assert len(boxes) >= 3
    res = det.match(bgr, None if color_only else depth, 90.0)
    for tid, bb in enumerate(boxes):
        mine = res[res["template_id"] == tid]
        assert len(mine) >= 1
        best = mine[0]
        assert best["similarity"] == 100.0
        # match (x, y) is the template bbox origin up to the T0 lattice the response maps live on
        assert abs(int(best["x"]) - bb[0]) < T0 and abs(int(best["y"]) - bb[1]) < T0
    # sorted by the total order, unique
    sims = res["similarity"]
    assert np.all(sims[:-1] >= sims[1:])


def test_threads_give_identical_matches(orc, frame0, golden0):
    bgr, depth = frame0
    det = orc.Detector(color_only=False)
    det.add_class("obj", golden0["rgbd_descs"], golden0["rgbd_features"])
    det.prepare(bgr, depth)
    a = det.match_prepared(60.0, threads=1)
    b = det.match_prepared(60.0, threads=4)
    assert len(a) > 0
    assert_matches_equal(a, b)


def test_shard_merge_equals_single(orc, frame0, golden0):
    """union of per-shard lists, merged, equals the unsharded list."""
    bgr, depth = frame0
    det = orc.Detector(color_only=False)
    det.add_class("obj", golden0["rgbd_descs"], golden0["rgbd_features"])
    det.prepare(bgr, depth)
    full = det.match_prepared(60.0)
    n = det.class_num_templates(0)
    for R in (2, 3, 4):
        parts = [det.match_prepared(60.0, tid_lo=n * r // R, tid_hi=n * (r + 1) // R) for r in range(R)]
        assert_matches_equal(orc.merge(parts), full)


# ---------------------------------------------------------------------------------------------
# stage properties / independent numpy restatements
# ---------------------------------------------------------------------------------------------
def test_gaussian_constant_and_impulse(orc):
    img = np.full((20, 24, 3), 137, np.uint8)
    assert np.array_equal(orc.gaussian7(img), img)          # kernel sums to 1 -> constants survive
    imp = np.zeros((21, 21, 3), np.uint8)
    imp[10, 10] = 255
    out = orc.gaussian7(imp).astype(np.int64)
    k = np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    exp = (255 * np.outer(k, k) + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14, 0], exp)
    assert out[:7].sum() == 0


def test_sobel_on_ramp(orc):
    x = np.arange(32, dtype=np.uint8)[None, :, None].repeat(16, 0).repeat(3, 2) * 3
    dx, dy = orc.sobel3(x)
    assert np.all(dx[:, 1:-1] == 24) and np.all(dy == 0)   # (1+2+1) * 2 * slope 3
    assert np.all(dx[:, 0] == 12) and np.all(dx[:, -1] == 12)  # replicate border halves it


def test_color_quantize_vertical_edge(orc):
    img = np.zeros((32, 32, 3), np.uint8)
    img[:, 16:] = 200
    q = orc.color_quantize(img)
    # gradient along +x -> angle 0 -> bin 0 -> one-hot 1; only near the edge, never on the border
    assert set(np.unique(q)) == {0, 1}
    assert q[:, :10].sum() == 0 and q[:, 22:].sum() == 0
    assert q[0].sum() == 0 and q[-1].sum() == 0
    assert np.all(q[5:27, 15:17] == 1)
    img_t = np.ascontiguousarray(img.transpose(1, 0, 2))
    qt = orc.color_quantize(img_t)  # gradient along +y -> 90 deg -> 16-bin 4 -> label 4
    assert set(np.unique(qt[:, 2:-2])) == {0, 16}
    # where the zeroed border column and the gradient-free outer row meet, bin 0 collects 5 of 9 votes
    assert qt[12, 1] == 1 and qt[19, 30] == 1


def test_opposite_gradients_share_a_bin(orc):
    img = np.zeros((32, 32, 3), np.uint8)
    img[:, 16:] = 200
    a = orc.color_quantize(img)
    b = orc.color_quantize(255 - img)   # gradient flips by 180 deg -> same label after &7
    assert np.array_equal(a, b)


def test_pyrdown_constant_and_reference(orc):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (18, 22, 3), dtype=np.uint8)
    out = orc.pyrdown(img)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    pad = np.pad(img.astype(np.int64), ((2, 2), (2, 2), (0, 0)), mode="reflect")   # reflect == REFLECT_101
    exp = np.zeros((9, 11, 3), np.int64)
    for y in range(9):
        for x in range(11):
            win = pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]
            exp[y, x] = (win * np.outer(k, k)[:, :, None]).sum((0, 1))
    assert np.array_equal(out, ((exp + 128) >> 8).astype(np.uint8))


def test_depth_quantize_plane_and_holes(orc):
    H, W = 40, 48
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (800 + 2 * xx).astype(np.uint16)       # plane tilted along +x
    q = orc.depth_quantize(depth)
    inner = q[8:-9, 8:-9]
    assert len(np.unique(inner)) == 1 and inner[0, 0] in (1, 2, 4, 8, 16, 32, 64, 128)
    assert q[:3].sum() == 0                          # 5-px border of normals is 0, median keeps most of it 0
    far = np.full((H, W), 2500, np.uint16)           # beyond distance_threshold -> no normal
    assert orc.depth_quantize(far).sum() == 0
    zeros = np.zeros((H, W), np.uint16)              # depth 0: nz == 0 -> LUT index out of table -> 0
    assert orc.depth_quantize(zeros).sum() == 0


def test_normal_lut_is_one_hot_and_settable(orc):
    lut = orc.normal_lut()
    assert lut.shape == (8000,) and set(np.unique(lut)) <= {1, 2, 4, 8, 16, 32, 64, 128}
    depth = (800 + 2 * np.mgrid[0:40, 0:48][1]).astype(np.uint16)
    a = orc.depth_quantize(depth)
    b = orc.depth_quantize(depth, lut=np.roll(lut, 1))
    assert a.shape == b.shape


def test_spread_bruteforce(orc):
    rng = np.random.default_rng(1)
    q = (1 << rng.integers(0, 8, (24, 40))).astype(np.uint8) * (rng.random((24, 40)) < 0.3)
    for T in (2, 5, 8):
        out = orc.spread(q.astype(np.uint8), T)
        exp = np.zeros_like(out)
        for y in range(24):
            for x in range(40):
                exp[y, x] = np.bitwise_or.reduce(q[y:y + T, x:x + T].astype(np.uint8).ravel())
        assert np.array_equal(out, exp)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_similarity_lut_consistency(orc, variant):
    lut = orc.similarity_lut(variant)
    assert lut.max() == 4 and lut.min() == 0
    for ori in range(8):
        for half in range(2):
            row = lut[32 * ori + 16 * half:32 * ori + 16 * half + 16]
            assert row[0] == 0
            for v in range(1, 16):   # every entry = max over its set bits of the single-bit entries
                assert row[v] == max(row[1 << b] for b in range(4) if v & (1 << b))
        assert lut[32 * ori + 16 * (ori // 4) + (1 << (ori % 4))] == 4   # own orientation scores 4


# SIMILARITY_LUT as cv::linemod ships it, typed out independently of the oracle's rule
UPSTREAM_SIMILARITY_LUT = """
0 4 3 4 2 4 3 4 1 4 3 4 2 4 3 4 | 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0
0 3 4 4 3 3 4 4 2 3 4 4 3 3 4 4 | 0 1 0 1 0 1 0 1 0 1 0 1 0 1 0 1
0 2 3 3 4 4 4 4 3 3 3 3 4 4 4 4 | 0 2 1 2 0 2 1 2 0 2 1 2 0 2 1 2
0 1 2 2 3 3 3 3 4 4 4 4 4 4 4 4 | 0 3 2 3 1 3 2 3 0 3 2 3 1 3 2 3
0 0 1 1 2 2 2 2 3 3 3 3 3 3 3 3 | 0 4 3 4 2 4 3 4 1 4 3 4 2 4 3 4
0 1 0 1 1 1 1 1 2 2 2 2 2 2 2 2 | 0 3 4 4 3 3 4 4 2 3 4 4 3 3 4 4
0 2 1 2 0 2 1 2 1 2 1 2 1 2 1 2 | 0 2 3 3 4 4 4 4 3 3 3 3 4 4 4 4
0 3 2 3 1 3 2 3 0 3 2 3 1 3 2 3 | 0 1 2 2 3 3 3 3 4 4 4 4 4 4 4 4
"""


def test_default_lut_is_upstreams(orc):
    exp = np.array([int(t) for t in UPSTREAM_SIMILARITY_LUT.replace("|", " ").split()], np.uint8)
    assert exp.size == 256
    assert np.array_equal(orc.similarity_lut(), exp)
    det = orc.Detector(color_only=True)          # what orc_create installs
    spr = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(orc.response_maps(spr), orc.response_maps(spr, exp))
    det.close()


def test_linear_lut_variant(orc):
    lut = orc.similarity_lut(0)
    for ori in range(8):
        for bit in range(8):
            assert lut[32 * ori + 16 * (bit // 4) + (1 << (bit % 4))] == max(0, 4 - abs(ori - bit))


def test_response_and_linearize(orc):
    rng = np.random.default_rng(2)
    spr = rng.integers(0, 256, (16, 40), dtype=np.uint8)
    lut = orc.similarity_lut()
    resp = orc.response_maps(spr, lut)
    exp = np.maximum(lut[(32 * np.arange(8))[:, None, None] + (spr & 15)[None]],
                     lut[(32 * np.arange(8) + 16)[:, None, None] + (spr >> 4)[None]])
    assert np.array_equal(resp, exp)
    for T in (2, 4, 8):
        lin = orc.linearize(resp[3], T)
        W = 40 // T
        for y in range(16):
            for x in range(40):
                assert lin[(y % T) * T + x % T, (y // T) * W + x // T] == resp[3, y, x]


def test_similarity_wraps_and_threshold(orc, synth):
    """similarity() scans template_positions CONTIGUOUS bytes: positions run across the right border
    into the next row (SURVEY.md 'wrap-around semantics'); the oracle reproduces that, so a template
    of width ~ the whole image still reports candidates in the wrapped columns."""
    bgr, depth = synth.make_frame(640, 480, seed=3)
    det = orc.Detector(color_only=True)
    descs, feats, _ = synth.make_bank(20, 1, 2, seed=5, fixed_l0_size=(96, 96))
    det.add_class("c", descs, feats)
    res = det.match(bgr, None, 0.0)       # threshold 0: raw > 2n, plenty of candidates
    assert len(res) > 0
    xs = res["x"]
    assert xs.min() >= 0 and xs.max() < 640


def test_dimension_asserts(orc):
    det = orc.Detector(color_only=False)
    bad = np.zeros((482, 640, 3), np.uint8)
    with pytest.raises(RuntimeError):
        det.prepare(bad, np.zeros((482, 640), np.uint16))
    with pytest.raises(RuntimeError):   # depth missing: sources.size() != modalities.size()
        det.prepare(np.zeros((480, 640, 3), np.uint8), None)


def test_no_templates_no_matches(orc, frame0):
    bgr, depth = frame0
    det = orc.Detector(color_only=False)
    assert len(det.match(bgr, depth, 80.0)) == 0


def test_scan_modes_agree(orc, synth):
    """The oracle's two shapes of the similarity sums -- one bounds check per byte (scalar) and the hoisted check with
    vectorisable byte adds (what bench.py's cpu_baseline times beside it) -- give the same candidates and matches,
    including templates whose scan runs past an orientation's block (reads of 0) and refinement patches at the border."""
    bgr, depth = synth.make_frame(640, 480, seed=9)
    o = orc.Detector(color_only=False)
    o.prepare(bgr, depth)
    q = {(l, m): o.stage(0, l, m).reshape(480 >> l, 640 >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(60, 2, 2, seed=3, quantized=q, crop_fraction=0.3)
    o.add_class("c", descs, feats)
    d2, f2, _ = synth.make_bank(6, 2, 2, seed=5, fixed_l0_size=(620, 470))     # span <= 0 / clamped templates
    o.add_class("big", d2, f2)
    d3, f3, _ = synth.make_bank(6, 2, 2, seed=6, fixed_l0_size=(700, 500))     # larger than the frame: negative spans
    o.add_class("toobig", d3, f3)
    res = {}
    try:
        for mode in (0, 1):
            orc.set_scan_mode(mode)
            res[mode] = (o.scan_candidates(40.0), o.match(bgr, depth, 40.0), o.match(bgr, depth, 0.0, class_idx=1),
                         o.match(bgr, depth, 0.0, class_idx=2))
    finally:
        orc.set_scan_mode(1)
    assert len(res[0][1]) > 0
    for a, b in zip(res[0], res[1]):
        assert a.tobytes() == b.tobytes()
