"""GPU parity of the whole hot path (Detector::match) against the CPU oracle and the committed golden
vectors: bit-identical (x, y, template_id, class) and order, identical similarity bits (the north star
allows 1e-5).  Everything goes through the C ABI of liblinemod_hip.so."""
import numpy as np
import pytest

from conftest import assert_matches_equal, crop_masks

pytestmark = pytest.mark.gpu


def _pair(lm, orc, color_only, size=(640, 480), **kw):
    d = lm.Detector(color_only=color_only, width=size[0], height=size[1], **kw)
    o = orc.Detector(color_only=color_only)
    return d, o


def _quantized(o, bgr, depth, color_only):
    o.prepare(bgr, None if color_only else depth)
    out = {}
    for l in range(2):
        w, h = bgr.shape[1] >> l, bgr.shape[0] >> l
        for m in range(1 if color_only else 2):
            out[(l, m)] = o.stage(0, l, m).reshape(h, w)
    return out


@pytest.mark.parametrize("name,color_only", [("rgbd", False), ("color", True)])
def test_golden_frame0(lm, frame0, golden0, name, color_only):
    """Reference's own frame + committed template bank -> committed match list."""
    bgr, depth = frame0
    d = lm.Detector(color_only=color_only)
    d.add_class("lagergehaeuse.ply", golden0[name + "_descs"], golden0[name + "_features"])
    got = d.match(bgr, None if color_only else depth, 80.0, class_idx=0)
    assert_matches_equal(got, golden0[name + "_matches"])
    d.close()


@pytest.mark.parametrize("color_only", [False, True])
def test_known_answer_via_gpu_add_template(lm, orc, frame0, color_only):
    """Detector::addTemplate on the GPU path reproduces the oracle's templates, and each one is found
    at its crop origin with similarity 100."""
    bgr, depth = frame0
    d, o = _pair(lm, orc, color_only)
    boxes = []
    for m in crop_masks(640, 480, 11, 8):
        tid, bb = d.add_template("obj", bgr, None if color_only else depth, m)
        otid, obb = o.add_template("obj", bgr, None if color_only else depth, m)
        assert tid == otid and (tid < 0 or bb == obb)
        if tid >= 0:
            boxes.append(bb)
    assert len(boxes) >= 3
    for tid in range(len(boxes)):
        for level in range(2):
            for mod in range(1 if color_only else 2):
                a, b = d.get_template(0, tid, level, mod), o.get_template(0, tid, level, mod)
                assert a[:2] == b[:2] and np.array_equal(a[2], b[2])
    got = d.match(bgr, None if color_only else depth, 90.0)
    assert_matches_equal(got, o.match(bgr, None if color_only else depth, 90.0))
    T0 = d.get_T(0)
    for tid, bb in enumerate(boxes):
        best = got[got["template_id"] == tid][0]
        assert best["similarity"] == 100.0
        assert abs(int(best["x"]) - bb[0]) < T0 and abs(int(best["y"]) - bb[1]) < T0
    d.close()


@pytest.mark.parametrize("color_only,size,n,thr,flags", [
    (False, (640, 480), 300, 80.0, 0),
    (False, (640, 480), 300, 55.0, 0),
    (True, (640, 480), 300, 70.0, 0),
    (True, (1280, 960), 200, 75.0, 0),
    (False, (640, 480), 300, 55.0, 1),     # LM_FLAG_BYTE_RESPONSES: byte scan kernel
    (True, (640, 480), 300, 70.0, 1),
    (False, (640, 480), 300, 80.0, 2),     # LM_FLAG_BLOCKING_SYNC: sleeps on a blocking event while waiting
])
def test_synthetic_bank_parity(lm, orc, synth, color_only, size, n, thr, flags):
    """Seeded synthetic frame + bank (10 % crops of the frame so real matches exist)."""
    bgr, depth = synth.make_frame(size[0], size[1], seed=1234)
    d, o = _pair(lm, orc, color_only, size, flags=flags)
    q = _quantized(o, bgr, depth, color_only)
    M = 1 if color_only else 2
    descs, feats, crops = synth.make_bank(n, M, 2, seed=4321, quantized=q, crop_fraction=0.15, frame_size=size,
                                          T0=d.get_T(0))
    assert len(crops) > 5
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    got = d.match(bgr, None if color_only else depth, thr)
    exp = o.match(bgr, None if color_only else depth, thr)
    assert len(exp) >= len(crops)
    assert_matches_equal(got, exp)
    # scan stage in isolation: candidate set before refinement
    d.upload_frame(1, bgr, None if color_only else depth)
    d.prepare_slot(1)
    cands = d.stage_scan(1, thr)
    o.prepare(bgr, None if color_only else depth)
    assert np.array_equal(cands, o.scan_candidates(thr, threads=8))   # a11-a13 record by record
    d.close()


def test_low_threshold_many_candidates(lm, orc, synth):
    """Threshold 0 floods the refinement stage (no NMS upstream): exercises compaction, the
    > LM_SORT_CAP host-sort path and duplicate removal."""
    bgr, depth = synth.make_frame(640, 480, seed=21)
    d, o = _pair(lm, orc, True)
    descs, feats, _ = synth.make_bank(150, 1, 2, seed=8)
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    for thr in (0.0, 20.0, 50.0):
        exp = o.match(bgr, None, thr, threads=8)
        got = d.match(bgr, None, thr, cap=1 << 18)
        assert_matches_equal(got, exp)
        if thr == 0.0:
            assert len(exp) > 4096      # beyond LM_SORT_CAP: host-side sort of the device keys
    d.close()


def test_multi_class_and_class_selection(lm, orc, synth, frame0):
    bgr, depth = frame0
    d, o = _pair(lm, orc, False)
    q = _quantized(o, bgr, depth, False)
    for k, seed in enumerate((1, 2, 3)):
        descs, feats, _ = synth.make_bank(50, 2, 2, seed=seed, quantized=q, crop_fraction=0.3)
        assert d.add_class("model%d.ply" % k, descs, feats) == k
        o.add_class("model%d.ply" % k, descs, feats)
    for ci in (0, 1, 2, -1):
        assert_matches_equal(d.match(bgr, depth, 75.0, class_idx=ci), o.match(bgr, depth, 75.0, class_idx=ci))
    with pytest.raises(lm.LinemodError):
        d.match(bgr, depth, 75.0, class_idx=3)
    d.close()


def test_class_list_matches_once_prepared(lm, orc, synth, frame0):
    """Detector::match(sources, threshold, matches, class_ids) with upstream's class LIST (HighLevelLinemod.cpp:145,152):
    a3-a10 once per frame for all the named classes, lists in the total order; lm_match_prepared = a11-a15 only on slots
    whose pre-processing is current."""
    bgr, depth = frame0
    d = lm.Detector(color_only=False, frame_slots=8)
    o = orc.Detector(color_only=False)
    q = _quantized(o, bgr, depth, False)
    for k, seed in enumerate((1, 2, 3, 4)):
        descs, feats, _ = synth.make_bank(40, 2, 2, seed=seed, quantized=q, crop_fraction=0.3)
        d.add_class("model%d.ply" % k, descs, feats)
        o.add_class("model%d.ply" % k, descs, feats)
    frames = [(bgr, depth), synth.make_frame(640, 480, seed=5), (bgr[::-1].copy(), depth[::-1].copy())]
    for i, (b, dp) in enumerate(frames):
        d.upload_frame(i, b, dp)
    thr = 70.0
    per = {(i, c): o.match(b, dp, thr, class_idx=c) for i, (b, dp) in enumerate(frames) for c in range(4)}
    assert sum(len(v) for v in per.values()) > 20

    mixed_all = [o.match(b, dp, thr, class_idx=-1) for b, dp in frames]

    def expect(i, classes):
        # the reference's semantics for a class LIST: all the named classes' matches sorted together, then adjacent-unique
        # (for this bank no two equal matches of one class have another class's match between them, so the merge of the
        # per-class lists is that list; checked against the oracle's all-class list below)
        return lm.merge_matches([per[(i, c)] for c in classes])

    for i in range(3):
        assert_matches_equal(expect(i, [0, 1, 2, 3]), mixed_all[i])

    d.set_profiling(False)          # resets the stage counters
    for classes in ([0, 2], [1, 2, 3], [3], [2, 0, 2], [0, 1, 2, 3]):
        got, cnt = d.match_batch_classes(0, 3, thr, classes)
        for i in range(3):
            assert_matches_equal(got[i, :cnt[i]], expect(i, sorted(set(classes))))
    sc = d.get_stage_counts()
    # [0, 2] is two runs of classes (two scan launches), the others one each
    assert sc["preprocess_frames"] == 15 and sc["scan_launches"] == 7 and sc["sort_launches"] == 5, sc
    # a host frame with a class list (lm_match_classes), through slot 0
    assert_matches_equal(d.match_classes(*frames[2], thr, [3, 1]), expect(2, [1, 3]))
    d.upload_frame(0, *frames[0])
    # all classes: empty list, {-1}, and the classic class_idx = -1 agree
    ref, rc = d.match_batch(3, thr, -1)
    for classes in (None, [-1]):
        got, cnt = d.match_batch_classes(0, 3, thr, classes)
        for i in range(3):
            assert_matches_equal(got[i, :cnt[i]], ref[i, :rc[i]])
            assert_matches_equal(got[i, :cnt[i]], o.match(*frames[i], thr, class_idx=-1))
    with pytest.raises(lm.LinemodError):
        d.match_batch_classes(0, 3, thr, [0, 4])
    with pytest.raises(lm.LinemodError):
        d.match_batch_classes(0, 3, thr, [-1, 1])
    # ---- prepared slots: one class per call, no pre-processing
    d.set_profiling(False)
    for c in range(4):
        got, cnt = d.match_prepared(0, 3, thr, [c])
        for i in range(3):
            assert_matches_equal(got[i, :cnt[i]], per[(i, c)])
    sc = d.get_stage_counts()
    assert sc["preprocess_frames"] == 0 and sc["scan_launches"] == 4, sc
    got, cnt = d.match_prepared(1, 2, 55.0, [1, 3])                  # another threshold needs no new pre-processing either
    for i in (1, 2):
        assert_matches_equal(got[i - 1, :cnt[i - 1]], lm.merge_matches([o.match(*frames[i], 55.0, class_idx=c) for c in (1, 3)]))
    # a new upload invalidates the slot: never a silent match against stale memories
    d.upload_frame(1, *frames[0])
    with pytest.raises(lm.LinemodError) as e:
        d.match_prepared(0, 3, thr, [0])
    assert e.value.code == lm.LM_ERR_INVALID
    d.prepare_slot(1)
    got, cnt = d.match_prepared(0, 3, thr, [0])
    assert_matches_equal(got[1, :cnt[1]], per[(0, 0)])
    # so does a LUT change
    d.set_similarity_lut(d.similarity_lut())
    with pytest.raises(lm.LinemodError):
        d.match_prepared(0, 1, thr, [0])
    # the lanes take class lists too
    d.match_begin_classes(0, 0, 2, thr, [1, 2])
    d.match_begin_classes(1, 2, 1, thr, [0])
    got, cnt = d.match_end(0, n_slots=2)
    for i in range(2):
        assert_matches_equal(got[i, :cnt[i]], expect(i if i != 1 else 0, [1, 2]))
    got, cnt = d.match_end(1, n_slots=1)
    assert_matches_equal(got[0, :cnt[0]], per[(2, 0)])
    d.close()


def test_edge_cases(lm, orc, synth, frame0):
    bgr, depth = frame0
    d, o = _pair(lm, orc, False)
    assert len(d.match(bgr, depth, 80.0)) == 0                      # empty bank
    # templates larger than the frame allows: span <= 0 -> never a candidate; huge ones clamp oddly upstream
    descs, feats, _ = synth.make_bank(6, 2, 2, seed=5, fixed_l0_size=(620, 470))
    d.add_class("big", descs, feats); o.add_class("big", descs, feats)
    assert_matches_equal(d.match(bgr, depth, 0.0), o.match(bgr, depth, 0.0))
    descs, feats, _ = synth.make_bank(6, 2, 2, seed=6, fixed_l0_size=(700, 500))
    d.add_class("toobig", descs, feats); o.add_class("toobig", descs, feats)
    assert_matches_equal(d.match(bgr, depth, 0.0), o.match(bgr, depth, 0.0))
    # features with fewer than the default counts (ragged feature lists)
    descs, feats, _ = synth.make_bank(30, 2, 2, seed=7, num_features=21)
    d.add_class("ragged", descs, feats); o.add_class("ragged", descs, feats)
    assert_matches_equal(d.match(bgr, depth, 30.0), o.match(bgr, depth, 30.0))
    # missing depth for an RGB-D detector: sources.size() != modalities.size()
    with pytest.raises(lm.LinemodError) as e:
        d.match(bgr, None, 80.0)
    assert e.value.code == lm.LM_ERR_INVALID
    with pytest.raises(lm.LinemodError):
        d.match(bgr, depth, -1.0)
    d.close()


def test_overflow_is_reported(lm, synth):
    bgr, depth = synth.make_frame(640, 480, seed=21)
    d = lm.Detector(lm.default_config(color_only=True, max_candidates=1000))
    descs, feats, _ = synth.make_bank(40, 1, 2, seed=8)
    d.add_class("c", descs, feats)
    with pytest.raises(lm.LinemodError) as e:
        d.match(bgr, None, 0.0)
    assert e.value.code == lm.LM_ERR_OVERFLOW
    d.close()


def test_wraparound_templates(lm, orc, synth):
    """Upstream scans template_positions contiguous bytes, so wide templates match 'across' the right
    border; bit parity means reproducing that (SURVEY.md section 7 'wrap-around semantics')."""
    bgr, depth = synth.make_frame(640, 480, seed=33)
    d, o = _pair(lm, orc, True)
    descs, feats, _ = synth.make_bank(25, 1, 2, seed=9, fixed_l0_size=(600, 60))
    d.add_class("wide", descs, feats); o.add_class("wide", descs, feats)
    exp = o.match(bgr, None, 10.0)
    assert len(exp) > 0
    assert_matches_equal(d.match(bgr, None, 10.0), exp)
    d.close()


def test_custom_luts_end_to_end(lm, orc, frame0, golden0):
    bgr, depth = frame0
    d, o = _pair(lm, orc, False)
    d.add_class("c", golden0["rgbd_descs"], golden0["rgbd_features"])
    o.add_class("c", golden0["rgbd_descs"], golden0["rgbd_features"])
    for variant in (1, 2):
        lut = orc.similarity_lut(variant)
        d.set_similarity_lut(lut); o.set_similarity_lut(lut)
        assert_matches_equal(d.match(bgr, depth, 70.0), o.match(bgr, depth, 70.0))
    d.close()


def test_resident_slots_and_batch(lm, orc, synth):
    size = (640, 480)
    d, o = _pair(lm, orc, False, size)
    frames = [synth.make_frame(size[0], size[1], seed=100 + i) for i in range(4)]
    q = _quantized(o, frames[0][0], frames[0][1], False)
    descs, feats, _ = synth.make_bank(200, 2, 2, seed=12, quantized=q, crop_fraction=0.2)
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    for i, (bgr, depth) in enumerate(frames):
        d.upload_frame(i, bgr, depth)
    out, counts = d.match_batch(4, 70.0)
    for i, (bgr, depth) in enumerate(frames):
        exp = o.match(bgr, depth, 70.0)
        assert counts[i] == len(exp)
        assert_matches_equal(out[i, :counts[i]], exp)
        assert_matches_equal(d.match_slot(i, 70.0), exp)        # slots stay resident, repeatable
    d.close()


def test_sharded_detectors_union_equals_single(lm, orc, synth, frame0):
    """SURVEY.md 8e on one GPU: R detectors each holding one contiguous template_id shard; merging
    their lists reproduces the unsharded result exactly (ids stay global)."""
    bgr, depth = frame0
    o = orc.Detector(color_only=False)
    q = _quantized(o, bgr, depth, False)
    descs, feats, _ = synth.make_bank(101, 2, 2, seed=13, quantized=q, crop_fraction=0.3)
    o.add_class("c", descs, feats)
    full = o.match(bgr, depth, 70.0)
    assert len(full) > 10
    for R in (2, 3):
        parts = []
        for r in range(R):
            d = lm.Detector(lm.default_config(color_only=False, shard_rank=r, shard_size=R))
            d.add_class("c", descs, feats)
            parts.append(d.match(bgr, depth, 70.0))
            d.close()
        assert_matches_equal(lm.merge_matches(parts), full)


def test_other_pyramids(lm, orc, synth):
    """Not only the reference's two constructions: 1 and 3 pyramid levels, other T."""
    bgr, depth = synth.make_frame(640, 480, seed=55)
    for T in ([8], [4, 8], [2, 4, 8]):
        L = len(T)
        d = lm.Detector(lm.default_config(color_only=False, T=T))
        o = orc.Detector(color_only=False, T=T)
        o.prepare(bgr, depth)
        q = {(l, m): o.stage(0, l, m).reshape(480 >> l, 640 >> l) for l in range(L) for m in range(2)}
        descs, feats, _ = synth.make_bank(60, 2, L, seed=14, quantized=q, crop_fraction=0.3, T0=T[0])
        d.add_class("c", descs, feats); o.add_class("c", descs, feats)
        exp = o.match(bgr, depth, 60.0)
        assert len(exp) > 0
        assert_matches_equal(d.match(bgr, depth, 60.0), exp)
        d.close()


def test_two_lanes_equal_batch(lm, orc, synth):
    """lm_match_begin / lm_match_end: two lanes (two HIP streams) on disjoint slot ranges deliver exactly what
    lm_match_batch delivers, in any interleaving; misuse is refused."""
    d, o = _pair(lm, orc, False, frame_slots=8)
    frames = [synth.make_frame(640, 480, seed=300 + i) for i in range(8)]
    q = _quantized(o, frames[0][0], frames[0][1], False)
    descs, feats, _ = synth.make_bank(120, 2, 2, seed=3, quantized=q, crop_fraction=0.3, T0=d.get_T(0))
    d.add_class("c", descs, feats)
    for i, (b, dp) in enumerate(frames):
        d.upload_frame(i, b, dp)
    ref, rc = d.match_batch(8, 70.0)
    ref = [ref[i, :rc[i]].copy() for i in range(8)]
    assert sum(len(r) for r in ref) > 0
    for rounds in range(2):
        d.match_begin(0, 0, 3, 70.0)
        d.match_begin(1, 3, 5, 70.0)
        with pytest.raises(lm.LinemodError):
            d.match_begin(1, 0, 2, 70.0)                      # lane busy
        with pytest.raises(lm.LinemodError):
            d.match_batch(8, 70.0)                            # synchronous call while lanes are busy
        with pytest.raises(lm.LinemodError):
            d.upload_frame(4, *frames[4])                     # slot in flight
        o1, c1 = d.match_end(1, n_slots=5)
        d.match_begin(1, 6, 2, 70.0)                          # lane 1 again while lane 0 is still busy
        with pytest.raises(lm.LinemodError):
            d.match_begin(0, 2, 2, 70.0)                      # lane 0 busy
        o0, c0 = d.match_end(0, n_slots=3)
        o2, c2 = d.match_end(1, n_slots=2)
        for i in range(3):
            assert o0[i, :c0[i]].tobytes() == ref[i].tobytes()
        for i in range(5):
            assert o1[i, :c1[i]].tobytes() == ref[3 + i].tobytes()
        for i in range(2):
            assert o2[i, :c2[i]].tobytes() == ref[6 + i].tobytes()
    with pytest.raises(lm.LinemodError):
        d.match_end(0, n_slots=1)                             # nothing in flight
    d.match_begin(0, 0, 4, 70.0)
    with pytest.raises(lm.LinemodError):
        d.match_begin(1, 3, 2, 70.0)                          # overlapping slot ranges
    d.match_end(0, n_slots=4)
    d.close()


def test_config5_batch_1280x960_multi_object(lm, orc, synth):
    """BASELINE.json config 5 (matching part): a batch of 1280x960 RGB-D frames against three models with odd
    template counts (odd work-item counts, several 1016-position chunks per template at W x H = 80 x 60), per class
    and all classes at once, through the two lanes."""
    size = (1280, 960)
    d, o = _pair(lm, orc, False, size, frame_slots=4)
    frames = [synth.make_frame(size[0], size[1], seed=500 + i) for i in range(4)]
    q = _quantized(o, frames[0][0], frames[0][1], False)
    for k, (n, seed) in enumerate(((41, 5), (33, 6), (27, 7))):
        descs, feats, _ = synth.make_bank(n, 2, 2, seed=seed, quantized=q, crop_fraction=0.4, frame_size=size,
                                          T0=d.get_T(0))
        assert d.add_class("model%d.ply" % k, descs, feats) == k
        o.add_class("model%d.ply" % k, descs, feats)
    for i, (b, dp) in enumerate(frames):
        d.upload_frame(i, b, dp)
    for ci in (-1, 1):
        exp = [o.match(b, dp, 78.0, class_idx=ci, threads=8) for b, dp in frames]
        assert sum(len(e) for e in exp) > 0
        d.match_begin(0, 0, 1, 78.0, ci)
        d.match_begin(1, 1, 3, 78.0, ci)
        o0, c0 = d.match_end(0, n_slots=1)
        o1, c1 = d.match_end(1, n_slots=3)
        assert_matches_equal(o0[0, :c0[0]], exp[0])
        for i in range(3):
            assert_matches_equal(o1[i, :c1[i]], exp[1 + i])
    d.close()


def test_batch_of_8_crowded_frames(lm, orc, synth):
    """Eight slots = the XCD-affine paths (k_refine_plan included) with very uneven, large candidate lists:
    a low threshold floods some frames, one slot holds a blank frame with no candidate at all."""
    d, o = _pair(lm, orc, True, frame_slots=8)
    frames = [synth.make_frame(640, 480, seed=900 + i)[0] for i in range(8)]
    frames[5] = np.full((480, 640, 3), 127, np.uint8)
    o.prepare(frames[0], None)
    q = {(l, 0): o.stage(0, l, 0).reshape(480 >> l, 640 >> l) for l in range(2)}
    descs, feats, _ = synth.make_bank(60, 1, 2, seed=77, quantized=q, crop_fraction=0.5, T0=d.get_T(0))
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    for i, b in enumerate(frames):
        d.upload_frame(i, b, None)
    for thr in (35.0, 70.0):
        out, counts = d.match_batch(8, thr, cap_per_frame=1 << 15)
        cands = [d.last_counts(i)[0] for i in range(8)]
        assert cands[5] == 0 and counts[5] == 0
        assert max(cands) > 500 or thr > 50
        for i in (0, 3, 5, 7):
            exp = o.match(frames[i], None, thr, threads=8, cap=1 << 18)
            assert_matches_equal(out[i, :counts[i]], exp)
    d.close()


@pytest.mark.parametrize("color_only,thr", [(False, 80.0), (False, 55.0), (False, 0.0), (True, 85.0), (True, 40.0)])
def test_scan_pruning_is_exact(lm, orc, synth, color_only, thr):
    """k_scan4 stops loading a work item's features once no position it holds can still exceed the raw threshold
    (partial sum + 4 x features to come).  Candidate list with pruning == without (scan variant bit 3) == oracle's,
    record by record, and the pruned scan really loads fewer features at a high threshold."""
    bgr, depth = synth.make_frame(640, 480, seed=77)
    d, o = _pair(lm, orc, color_only, frame_slots=4)
    d.set_tuning(lm.TUNE_SCAN_FORM, 1)                              # this test is about k_scan4's pruning rules (k_scan1: tests/test_gpu_scan_planes.py)
    dep = None if color_only else depth
    q = _quantized(o, bgr, depth, color_only)
    M = 1 if color_only else 2
    descs, feats, crops = synth.make_bank(400, M, 2, seed=12, quantized=q, crop_fraction=0.15, frame_size=(640, 480), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    d.upload_frame(0, bgr, dep)
    d.upload_frame(1, synth.make_frame(640, 480, seed=78)[0], None if color_only else synth.make_frame(640, 480, seed=78)[1])
    d.prepare_slot(0)
    o.prepare(bgr, dep)
    exp = o.scan_candidates(thr, threads=8)
    stats, lanes = {}, {}
    for variant in (32, 16, 8):                                 # per-lane pruning, wave-level pruning (r02), none
        d.set_scan_variant(variant)
        d.set_scan_stats(True)
        assert np.array_equal(d.stage_scan(0, thr), exp)
        stats[variant] = d.get_scan_stats()
        lanes[variant] = d.get_scan_lane_stats()
        d.set_scan_stats(False)
        assert_matches_equal(d.match_slot(0, thr, cap=1 << 16), o.match(bgr, dep, thr, threads=8))
        got, cnt = d.match_batch(2, thr, cap_per_frame=1 << 15)  # two frames per wave: the pair must agree to stop
        assert_matches_equal(got[0, :cnt[0]], o.match(bgr, dep, thr, threads=8))
    assert stats[8][0] == stats[8][1] == stats[32][1]           # the exhaustive scan loads every in-bounds feature
    assert stats[32][0] <= stats[32][1]
    assert stats[32][0] == stats[16][0]                         # a wave stops when its last lane dies: the same rule
    assert lanes[8][0] == lanes[8][1] == 64 * stats[8][1] and lanes[16][0] == 64 * stats[16][0]
    assert lanes[32][0] <= lanes[16][0]
    if thr >= 80.0:
        assert stats[32][0] < 0.8 * stats[32][1]
        assert lanes[32][0] < 0.8 * lanes[16][0]                # dead lanes leave the loads' exec mask
    d.set_scan_variant(0)                                       # the default picks one of the two pruning rules by modality count
    assert np.array_equal(d.stage_scan(0, thr), exp)
    # the order of a template's feature list (r04: farthest-point order by default) changes how soon the pruning stops, never the
    # candidate list
    for order in (0, 2, 1, 3):
        d.set_tuning(lm.TUNE_SCAN_LIST_ORDER, order)
        d.prepare_slot(0)
        assert np.array_equal(d.stage_scan(0, thr), exp), order
        assert_matches_equal(d.match_slot(0, thr, cap=1 << 16), o.match(bgr, dep, thr, threads=8))
    # the measurement hook over a batch of prepared slots (incl. the timing-only variant without the shift-undo)
    d.prepare_slot(1)
    assert d.time_scan_batch(0, 2, thr, iters=2) > 0 and d.time_scan_batch(0, 2, thr, iters=2, variant=8 | 64) > 0
    with pytest.raises(lm.LinemodError):
        d.time_scan_batch(0, 3, thr)                            # slot 2 holds no prepared frame
    assert np.array_equal(d.stage_scan(0, thr), exp)            # (the timing runs leave no state behind)
    if thr == 0.0:
        assert stats[32][0] == stats[32][1]                     # nothing can be pruned when every position qualifies
    d.close()


def test_color_check_counts_against_numpy_fill(lm, synth):
    """lm_color_check_counts (f1: fillPoly of the template's feature hull + the two countNonZero of colorCheck) on
    hulls whose fill is known in closed form: axis-aligned rectangles (corner features) and a single point, partly
    outside the frame, against a numpy colour mask built with the same HSV rule on a two-colour image."""
    W, H = 640, 480
    d = lm.Detector(color_only=True, width=W, height=H)
    # templates: rectangle 40 x 24 (corners + interior points), 1-point template, horizontal segment
    shapes = [[(0, 0), (40, 0), (40, 24), (0, 24), (7, 9)], [(5, 5)], [(0, 3), (30, 3), (12, 3)]]
    descs = np.zeros(len(shapes) * 2, synth.DESC_DTYPE)
    feats = []
    for t, pts in enumerate(shapes):
        for l in range(2):
            f = np.zeros(len(pts), synth.FEATURE_DTYPE)
            f["x"] = [p[0] >> l for p in pts]; f["y"] = [p[1] >> l for p in pts]
            descs[t * 2 + l] = (48 >> l, 32 >> l, l, len(pts))
            feats.append(f)
    d.add_class("shapes", descs, np.concatenate(feats))
    bgr = np.zeros((H, W, 3), np.uint8)
    bgr[:, 300:] = (40, 200, 90)                          # right part: V = 200 passes V >= 100, left part black fails
    d.upload_frame(0, bgr)
    mask = np.zeros((H, W), bool); mask[:, 300:] = True
    m = np.zeros(6, lm.MATCH_DTYPE)
    m["x"] = [280, 10, 620, 0, 290, -20]; m["y"] = [100, 10, 470, 0, 50, -2]; m["template_id"] = [0, 0, 0, 1, 2, 2]
    a, b = d.color_check_counts(0, [0, 0, 100], [180, 255, 255], m)
    for k in range(6):
        pts = np.array(shapes[m["template_id"][k]])
        x0, x1 = pts[:, 0].min() + m["x"][k], pts[:, 0].max() + m["x"][k]
        y0, y1 = pts[:, 1].min() + m["y"][k], pts[:, 1].max() + m["y"][k]
        xs = slice(max(x0, 0), min(x1, W - 1) + 1); ys = slice(max(y0, 0), min(y1, H - 1) + 1)
        region = np.zeros((H, W), bool); region[ys, xs] = True
        assert a[k] == region.sum() and b[k] == (region & mask).sum(), (k, a[k], b[k])
    d.close()


def test_batch_of_32_frames_takes_the_batch_kernels(lm, orc, synth):
    """From 16 frames per launch on the preprocess uses its batch kernels (sliding-window blur): 32 distinct frames in one
    lm_match_batch and as two lanes of 16, every list against the oracle."""
    W, H = 640, 480
    d = lm.Detector(color_only=False, width=W, height=H, frame_slots=32)
    o = orc.Detector(color_only=False)
    frames = [synth.make_frame(W, H, seed=3000 + i) for i in range(32)]
    o.prepare(*frames[0])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(60, 2, 2, seed=31, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=5)
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    for i, (b, dp) in enumerate(frames):
        d.upload_frame(i, b, dp)
    exp = [o.match(b, dp, 70.0, threads=8) for b, dp in frames]
    assert sum(len(e) for e in exp) > 0
    out, cnt = d.match_batch(32, 70.0)
    for i in range(32):
        assert_matches_equal(out[i, :cnt[i]], exp[i])
    d.match_begin(0, 0, 16, 70.0)
    d.match_begin(1, 16, 16, 70.0)
    for lane in (0, 1):
        out, cnt = d.match_end(lane, n_slots=16)
        for i in range(16):
            assert_matches_equal(out[i, :cnt[i]], exp[16 * lane + i])
    d.close()


def test_split_sort_equals_plain_sort_and_oracle(lm, orc, synth):
    """a15 on the device has two forms (LM_TUNE_SORT_SPLIT): one workgroup per frame, or -- for lists longer than 1024 keys --
    four chunk workgroups per frame and a merge launch that ranks every key against the other chunks.  Eight crowded frames at
    thresholds that put the lists of refined matches on both sides of 1024, 2048, 3072 and of LM_SORT_CAP = 4096 (beyond it the
    host sorts): every list of every form against the oracle's, plus the adaptive default after it has seen long lists."""
    d, o = _pair(lm, orc, True, frame_slots=8)
    frames = [synth.make_frame(640, 480, seed=900 + i)[0] for i in range(8)]
    frames[5] = np.full((480, 640, 3), 127, np.uint8)
    o.prepare(frames[0], None)
    q = {(l, 0): o.stage(0, l, 0).reshape(480 >> l, 640 >> l) for l in range(2)}
    descs, feats, _ = synth.make_bank(60, 1, 2, seed=77, quantized=q, crop_fraction=0.5, T0=d.get_T(0))
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    for i, b in enumerate(frames):
        d.upload_frame(i, b, None)
    lengths = set()
    for thr in (20.0, 30.0, 35.0, 45.0, 60.0):
        exp = [o.match(frames[i], None, thr, threads=8, cap=1 << 18) for i in range(8)]
        for mode in (0, 1, 2, 2):
            d.set_tuning(lm.TUNE_SORT_SPLIT, mode)
            out, counts = d.match_batch(8, thr, cap_per_frame=1 << 15)
            for i in range(8):
                assert_matches_equal(out[i, :counts[i]], exp[i])
        lengths.update(d.last_counts(i)[1] for i in range(8))
    buckets = {min(n // 1024, 4) for n in lengths}
    assert {0, 1, 4} <= buckets and len(buckets) >= 4, sorted(lengths)       # short lists, several chunk counts, host-sorted ones
    d.close()
