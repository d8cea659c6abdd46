"""The bit-plane scan k_scan1 (r05, LM_TUNE_SCAN_FORM) against the oracle and against the nibble scan k_scan4: the same candidate
lists and match lists, bit for bit -- whatever the frame size (one or several chunks per template, few or many lanes per frame),
the number of frames in the call (full and ragged groups of frames per wave), the threshold, the modality count and the
similarity table (the miss bound follows the table's largest response below 4)."""
import numpy as np
import pytest

from conftest import assert_matches_equal, class_sublist

pytestmark = pytest.mark.gpu


def _quantized(o, bgr, depth, color_only):
    o.prepare(bgr, None if color_only else depth)
    out = {}
    for l in range(2):
        w, h = bgr.shape[1] >> l, bgr.shape[0] >> l
        for m in range(1 if color_only else 2):
            out[(l, m)] = o.stage(0, l, m).reshape(h, w)
    return out


def _setup(lm, orc, synth, color_only, size, T, n_templates, slots, seed, nf=63, size_range=(48, 160)):
    w, h = size
    size_range = (min(size_range[0], h // 6), min(size_range[1], h // 3))
    d = lm.Detector(lm.default_config(color_only=color_only, width=w, height=h, T=T, frame_slots=slots))
    o = orc.Detector(color_only=color_only, T=T)
    frames = [synth.make_frame(w, h, seed=seed + k) for k in range(min(slots, 3))]
    q = _quantized(o, frames[0][0], frames[0][1], color_only)
    M = 1 if color_only else 2
    descs, feats, _ = synth.make_bank(n_templates, M, 2, seed=seed + 100, quantized=q, crop_fraction=0.3, frame_size=(w, h), T0=T[0],
                                      num_features=nf, size_range=size_range)
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    return d, o, frames


@pytest.mark.parametrize("color_only,size,T,thr", [
    (False, (640, 480), [5, 8], 80.0),       # config 2's shape: 1200 positions per memory, one chunk, 9-10 lanes per frame
    (True, (640, 480), [2, 8], 85.0),
    (False, (384, 256), [4, 8], 75.0),       # 384 positions per memory: 3-4 lanes per frame, 16+ frames per wave
    (True, (1280, 960), [2, 8], 88.0),       # 4800 positions per memory
    (False, (640, 480), [4, 4], 80.0),       # T = 4 at the scanned level: 4800 positions
    (False, (640, 480), [5, 8], 55.0),       # a low threshold: many survivors of the miss bound
    (False, (640, 480), [5, 8], 97.0),
])
def test_scan1_candidates_and_matches(lm, orc, synth, color_only, size, T, thr):
    nb = 9
    d, o, frames = _setup(lm, orc, synth, color_only, size, T, 60, nb, seed=900)
    dep = lambda k: None if color_only else frames[k % len(frames)][1]
    bgr = lambda k: frames[k % len(frames)][0]
    exp_m = [o.match(bgr(k), dep(k), thr, threads=8, cap=1 << 18) for k in range(len(frames))]
    o.prepare(bgr(0), dep(0))
    exp_c = o.scan_candidates(thr, threads=8)
    # one frame, the candidate list record by record
    d.upload_frame(0, bgr(0), dep(0)); d.prepare_slot(0)
    d.set_tuning(lm.TUNE_SCAN_FORM, 1)
    assert np.array_equal(d.stage_scan(0, thr), exp_c)
    d.set_tuning(lm.TUNE_SCAN_FORM, 2)
    d.prepare_slot(0)                                    # (the miss planes are written only for the form that reads them)
    d.set_scan_stats(True)
    assert np.array_equal(d.stage_scan(0, thr), exp_c)
    loaded, unpruned = d.get_scan_stats()
    assert 0 < loaded <= unpruned
    d.set_scan_stats(False)
    # whole matches, every group shape: 1 frame, a ragged group, several groups
    for n in (1, 2, 5, nb):
        for k in range(n):
            d.upload_frame(k, bgr(k), dep(k))
        before = d.get_scan_form_stats()
        got, cnt = d.match_batch(n, thr, cap_per_frame=1 << 15)
        after = d.get_scan_form_stats()
        assert after[0] - before[0] == after[1] - before[1] >= 1 and after[3] > 0       # every scan launch was k_scan1
        for k in range(n):
            assert_matches_equal(got[k, :cnt[k]], exp_m[k % len(frames)])
    # the survivors' exact sums taken by the waves themselves (scan variant bit 8: what a wave does when its reservation in the survivor queue
    # does not fit) give the same lists
    d.set_scan_variant(256)
    got, cnt = d.match_batch(nb, thr, cap_per_frame=1 << 15)
    for k in range(nb):
        assert_matches_equal(got[k, :cnt[k]], exp_m[k % len(frames)])
    d.prepare_slot(0)
    assert np.array_equal(d.stage_scan(0, thr), exp_c)
    d.set_scan_variant(0)
    # the default picks by cost: the lists are the same either way
    d.set_tuning(lm.TUNE_SCAN_FORM, 0)
    got, cnt = d.match_batch(nb, thr, cap_per_frame=1 << 15)
    for k in range(nb):
        assert_matches_equal(got[k, :cnt[k]], exp_m[k % len(frames)])
    d.close()


def test_scan1_similarity_tables(lm, orc, synth):
    """The miss bound is 4 - the table's largest response below 4: tables whose misses cost 1, 2, 3 or 4, and a table without a 4."""
    d, o, frames = _setup(lm, orc, synth, False, (640, 480), [5, 8], 50, 4, seed=950)
    base = orc.similarity_lut()
    tables = [base]
    for lo, hi in ((3, 4), (2, 4), (0, 4), (1, 3)):      # (response of a neighbouring orientation, of the orientation itself)
        t = base.copy()
        t[base == 1] = lo; t[base == 4] = hi
        tables.append(t)
    bgr, dep = frames[0]
    for t in tables:
        d.set_similarity_lut(t); o.set_similarity_lut(t)
        for thr in (70.0, 90.0):
            exp = o.match(bgr, dep, thr, threads=8, cap=1 << 18)
            o.prepare(bgr, dep)
            exp_c = o.scan_candidates(thr, threads=8)
            for form in (1, 2):
                d.set_tuning(lm.TUNE_SCAN_FORM, form)
                for k in range(4):
                    d.upload_frame(k, bgr, dep)
                d.prepare_slot(0)
                assert np.array_equal(d.stage_scan(0, thr), exp_c), (form, thr)
                assert (d.get_scan_form_stats()[3] > 0) == (form == 2)
                got, cnt = d.match_batch(4, thr, cap_per_frame=1 << 15)
                for k in range(4):
                    assert_matches_equal(got[k, :cnt[k]], exp)
    d.close()


def test_scan1_few_features_and_classes(lm, orc, synth):
    """Templates of few features (tail rounds of 4, 2 and 1), a class list (one launch per run of classes) and the per-class calls."""
    for nf in (63, 21, 13, 9, 2):
        d, o, frames = _setup(lm, orc, synth, False, (640, 480), [5, 8], 30, 8, seed=970 + nf, nf=nf)
        bgr, dep = frames[0]
        q = _quantized(o, bgr, dep, False)
        d2, f2, _ = synth.make_bank(17, 2, 2, seed=1234 + nf, quantized=q, crop_fraction=0.5, frame_size=(640, 480), T0=5, num_features=nf)
        d.add_class("c2", d2, f2); o.add_class("c2", d2, f2)
        d.set_tuning(lm.TUNE_SCAN_FORM, 2)
        for k in range(8):
            d.upload_frame(k, bgr, dep)
        for thr in (60.0, 85.0):
            for classes in ([-1], [1], [0], [1, 0]):
                exp = o.match(bgr, dep, thr, class_idx=classes[0] if len(classes) == 1 else -1, threads=8, cap=1 << 18)
                got, cnt = d.match_batch_classes(0, 8, thr, classes, cap_per_frame=max(len(exp), 1))
                for k in range(8):
                    assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (nf, thr, classes, k)
        d.close()


@pytest.mark.parametrize("color_only,size,T", [
    (False, (576, 432), [4, 4]),      # 72 memory columns at T = 4: segments of 64 + 8 columns (the fuzzer's catch: a 16-column unit must not run over the row end)
    (False, (576, 472), [4, 4]),
    (True, (1280, 960), [2, 8]),      # 80 memory columns at T = 8: k_lm_fast<8, 80, ..> with 16-column units
    (False, (1280, 960), [5, 8]),
    (False, (640, 480), [5, 8]),
])
def test_linear_memories_after_a_batch_call(lm, orc, synth, color_only, size, T):
    """The batch kernels' linear memories (what lm_match_batch leaves in the slots) against the oracle, every level and modality, with
    the miss planes being written beside them (scan form 2) and without (1); then the scan on them."""
    w, h = size
    nb = 16
    d = lm.Detector(lm.default_config(color_only=color_only, width=w, height=h, T=T, frame_slots=nb))
    o = orc.Detector(color_only=color_only, T=T)
    frames = [synth.make_frame(w, h, seed=400 + k) for k in range(2)]
    q = _quantized(o, frames[0][0], frames[0][1], color_only)
    M = 1 if color_only else 2
    descs, feats, _ = synth.make_bank(20, M, 2, seed=77, quantized=q, crop_fraction=0.5, frame_size=(w, h), T0=T[0])
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    exp = [o.match(f[0], None if color_only else f[1], 85.0, threads=8, cap=1 << 18) for f in frames]
    for form in (2, 1):
        d.set_tuning(lm.TUNE_SCAN_FORM, form)
        for k in range(nb):
            d.upload_frame(k, frames[k % 2][0], None if color_only else frames[k % 2][1])
        got, cnt = d.match_batch(nb, 85.0, cap_per_frame=1 << 15)
        for k in (0, 1, nb - 1):
            assert_matches_equal(got[k, :cnt[k]], exp[k % 2])
            o.prepare(frames[k % 2][0], None if color_only else frames[k % 2][1])
            for level in range(2):
                for mod in range(M):
                    assert np.array_equal(d.debug_read(k, 2, level, mod), o.stage(2, level, mod)), (form, k, level, mod)
    d.close()


def test_scan_form_by_cost_and_key_validation(lm, orc, synth):
    """LM_TUNE_SCAN_FORM 0 picks the bit-plane scan for batches (8+ frames, whole groups) of a one-modality detector and the nibble scan for a
    single frame and for two modalities; the keys reject values outside their range and a change while a lane has a match in flight."""
    d, o, frames = _setup(lm, orc, synth, True, (640, 480), [2, 8], 40, 16, seed=990)
    for k in range(16):
        d.upload_frame(k, frames[k % 3][0], None)
    d.match_batch(16, 85.0, cap_per_frame=1 << 15)
    assert d.get_scan_form_stats()[3] > 0                      # 16 colour-only frames: k_scan1
    # the call's scan was k_scan1 by cost: its slots hold the spread byte + the planes, no response memories.  They are scanned by k_scan1 from
    # then on -- also by a prepared match below the threshold rule -- and lm_debug_read still shows the same linear memories
    exp60 = o.match(frames[1][0], None, 60.0, threads=8, cap=1 << 18)
    d.set_tuning(lm.TUNE_SCAN1_MIN_THRESHOLD, 70)
    got, cnt = d.match_prepared(0, 16, 60.0, [-1], cap_per_frame=1 << 15)
    assert d.get_scan_form_stats()[3] > 0
    assert cnt[1] == len(exp60) and got[1, :cnt[1]].tobytes() == exp60.tobytes()
    d.set_tuning(lm.TUNE_SCAN1_MIN_THRESHOLD, 50)
    o.prepare(frames[2][0], None)
    assert np.array_equal(d.debug_read(2, 2, 1, 0), o.stage(2, 1, 0))
    d.upload_frame(0, frames[0][0], None); d.prepare_slot(0)   # one frame: prepared without planes -> cannot be scanned together with the others
    with pytest.raises(lm.LinemodError):
        d.match_prepared(0, 16, 85.0, [-1], cap_per_frame=1 << 12)
    got, cnt = d.match_prepared(1, 15, 85.0, [-1], cap_per_frame=1 << 14)          # ... apart they can
    assert d.get_scan_form_stats()[3] > 0
    d.match_batch(1, 85.0, cap_per_frame=1 << 15)
    assert d.get_scan_form_stats()[3] == 0                     # one frame: k_scan4 (three launches instead of one would double its scan time)
    d.match_batch(16, 48.0, cap_per_frame=1 << 15)
    assert d.get_scan_form_stats()[3] == 0                     # below LM_TUNE_SCAN1_MIN_THRESHOLD (50): too many survivors of the miss bound
    d.set_tuning(lm.TUNE_SCAN1_MIN_THRESHOLD, 40)
    d.match_batch(16, 48.0, cap_per_frame=1 << 15)
    assert d.get_scan_form_stats()[3] > 0
    for key, bad in ((lm.TUNE_SCAN_FORM, -1), (lm.TUNE_SCAN_FORM, 4), (lm.TUNE_SCAN1_MIN_THRESHOLD, 101), (lm.TUNE_SCAN1_MIN_THRESHOLD, -1)):
        with pytest.raises(lm.LinemodError):
            d.set_tuning(key, bad)
    d.match_begin(0, 0, 8, 85.0)
    with pytest.raises(lm.LinemodError):
        d.set_tuning(lm.TUNE_SCAN_FORM, 2)                     # the slots in flight were prepared for the other form
    out = np.zeros((8, 1 << 12), lm.MATCH_DTYPE); cnt = np.zeros(8, np.int32)
    d.match_end(0, 1 << 12, out=out, counts=cnt)
    d.set_tuning(lm.TUNE_SCAN_FORM, 2)
    d.close()
    d2, o2, frames2 = _setup(lm, orc, synth, False, (640, 480), [5, 8], 40, 16, seed=991)
    for k in range(16):
        d2.upload_frame(k, frames2[k % 3][0], frames2[k % 3][1])
    d2.match_batch(16, 85.0, cap_per_frame=1 << 15)
    assert d2.get_scan_form_stats()[3] == 0                    # two modalities: k_scan4's exact pruning stops sooner than the miss bound
    d2.close()


def test_slots_with_planes_and_slots_with_spread_bytes_do_not_mix(lm, orc, synth):
    """r06: a launch reads ONE layout of the scanned level.  Under the default form, slots 0..7 matched at threshold 85 keep only the
    spread byte (+ planes: the call's scan is k_scan1 by cost); slots 8..15 matched at threshold 40 -- below LM_TUNE_SCAN1_MIN_THRESHOLD, so the
    call's scan is k_scan4 -- keep the response memories (+ planes: 8 colour-only frames).  One prepared call over slots of both kinds is
    refused; apart, each half gives the oracle's lists at either threshold."""
    d, o, frames = _setup(lm, orc, synth, True, (640, 480), [2, 8], 40, 16, seed=993)
    for k in range(16):
        d.upload_frame(k, frames[k % 3][0], None)
    exp85 = [o.match(f[0], None, 85.0, threads=8, cap=1 << 18) for f in frames]
    exp40 = [o.match(f[0], None, 40.0, threads=8, cap=1 << 18) for f in frames]
    d.match_begin(0, 0, 8, 85.0)
    got, cnt = d.match_end(0, 1 << 15, n_slots=8)
    assert d.get_scan_form_stats()[3] > 0                                     # k_scan1 by cost: no response memories in slots 0..7
    for k in range(8):
        assert_matches_equal(got[k, :cnt[k]], exp85[k % 3])
    d.match_begin(0, 8, 8, 40.0)
    got, cnt = d.match_end(0, 1 << 16, n_slots=8)
    assert d.get_scan_form_stats()[3] == 0                                    # k_scan4: planes AND response memories in slots 8..15
    for k in range(8):
        assert_matches_equal(got[k, :cnt[k]], exp40[(8 + k) % 3])
    for first, n in ((0, 16), (4, 8), (7, 2)):
        with pytest.raises(lm.LinemodError) as e:
            d.match_prepared(first, n, 85.0, [-1], cap_per_frame=1 << 15)
        assert "different scan forms" in str(e.value)
    got, cnt = d.match_prepared(0, 8, 40.0, [-1], cap_per_frame=1 << 16)      # (spread-byte slots: k_scan1 whatever the threshold rule says)
    assert d.get_scan_form_stats()[3] > 0
    for k in range(8):
        assert_matches_equal(got[k, :cnt[k]], exp40[k % 3])
    got, cnt = d.match_prepared(8, 8, 85.0, [-1], cap_per_frame=1 << 15)      # (planes + response memories: either kernel may scan them)
    for k in range(8):
        assert_matches_equal(got[k, :cnt[k]], exp85[(8 + k) % 3])
    d.close()


def test_scan_variants_that_change_the_lists_are_refused(lm, orc, synth):
    """r06 (VERDICT r5 #1c): lm_set_scan_variant takes only variants whose lists are variant 0's; the two timing experiments that skip work
    (bit 6: no shift-undo, bit 7: no exact sums of the survivors) are arguments of lm_time_scan / lm_time_scan_batch alone."""
    d, o, frames = _setup(lm, orc, synth, True, (640, 480), [2, 8], 30, 8, seed=994)
    for bad in (64, 8 | 64, 128, 128 | 256, 512, -1, 1 << 20):
        with pytest.raises(lm.LinemodError):
            d.set_scan_variant(bad)
    for k in range(8):
        d.upload_frame(k, frames[k % 3][0], None)
    exp = [o.match(f[0], None, 85.0, threads=8, cap=1 << 18) for f in frames]
    for ok in (0, 1, 2, 8, 16, 32, 34, 256, 0):
        d.set_scan_variant(ok)
        got, cnt = d.match_batch(8, 85.0, cap_per_frame=1 << 15)
        for k in range(8):
            assert_matches_equal(got[k, :cnt[k]], exp[k % 3])
    # the timing hooks still take them (candidates counted, none stored) and leave the detector's lists alone
    d.set_tuning(lm.TUNE_SCAN_FORM, 2)
    got, cnt = d.match_batch(8, 85.0, cap_per_frame=1 << 15)
    assert d.time_scan_batch(0, 8, 85.0, iters=2, variant=128) > 0
    got, cnt = d.match_prepared(0, 8, 85.0, [-1], cap_per_frame=1 << 15)
    for k in range(8):
        assert_matches_equal(got[k, :cnt[k]], exp[k % 3])
    d.close()


# ---- r06: the bit-plane scan with a frame's planes in LDS (k_scanl, LM_TUNE_SCAN_FORM 3) ------------------------------------------------------
@pytest.mark.parametrize("color_only,size,T,thr,fits", [
    (False, (640, 480), [5, 8], 80.0, True),      # config 2's shape: 153 600 bytes of planes, the whole LDS; 8 units of 128 positions per template
    (True, (640, 480), [2, 8], 85.0, True),       # one modality: half the image
    (False, (384, 256), [4, 8], 75.0, True),      # 384 positions per memory: 3 units per template, 21 templates per wave
    (False, (640, 480), [4, 4], 80.0, True),      # T = 4 at the scanned level: 4800 positions per memory, up to 38 units per template
    (False, (640, 480), [5, 8], 97.0, True),
    (False, (640, 480), [5, 8], 30.0, True),      # a flood of survivors: the LDS queue overflows, the waves take the sums themselves
    (True, (1280, 960), [2, 8], 88.0, False),     # 307 200 bytes of planes: does not fit, form 3 runs k_scan1
])
def test_scanl_candidates_and_matches(lm, orc, synth, color_only, size, T, thr, fits):
    nb = 9
    d, o, frames = _setup(lm, orc, synth, color_only, size, T, 60, nb, seed=1900)
    dep = lambda k: None if color_only else frames[k % len(frames)][1]
    bgr = lambda k: frames[k % len(frames)][0]
    exp_m = [o.match(bgr(k), dep(k), thr, threads=8, cap=1 << 18) for k in range(len(frames))]
    o.prepare(bgr(0), dep(0))
    exp_c = o.scan_candidates(thr, threads=8, cap=1 << 20)
    d.set_tuning(lm.TUNE_SCAN_FORM, 3)
    d.upload_frame(0, bgr(0), dep(0)); d.prepare_slot(0)
    d.set_scan_stats(True)
    assert np.array_equal(d.stage_scan(0, thr, cap=1 << 20), exp_c)
    loaded, unpruned = d.get_scan_stats()
    st = d.get_scan_form_stats()
    assert 0 < loaded <= unpruned and (st[3] >= 1000) == fits and st[3] > 0, st
    assert st[2] >= len(exp_c)                              # the survivors of the miss bound are a superset of the candidates
    d.set_scan_stats(False)
    # the linear memories lm_debug_read shows are the oracle's although the slot keeps spread bytes + planes only
    for level in range(2):
        for mod in range(1 if color_only else 2):
            assert np.array_equal(d.debug_read(0, 2, level, mod), o.stage(2, level, mod))
    for n in (1, 2, 5, nb):
        for k in range(n):
            d.upload_frame(k, bgr(k), dep(k))
        before = d.get_scan_form_stats()
        got, cnt = d.match_batch(n, thr, cap_per_frame=1 << 16)
        after = d.get_scan_form_stats()
        assert after[0] - before[0] == after[1] - before[1] >= 1 and (after[3] >= 1000) == fits
        for k in range(n):
            assert_matches_equal(got[k, :cnt[k]], exp_m[k % len(frames)])
    # two lanes at once, and the prepared slots once more by the default rule (the slots keep no response memories: a bit-plane form either way)
    d.match_begin(0, 0, 4, thr); d.match_begin(1, 4, 5, thr)
    g0, c0 = d.match_end(0, 1 << 16, n_slots=4)
    g1, c1 = d.match_end(1, 1 << 16, n_slots=5)
    for k in range(4):
        assert_matches_equal(g0[k, :c0[k]], exp_m[k % len(frames)])
    for k in range(5):
        assert_matches_equal(g1[k, :c1[k]], exp_m[(4 + k) % len(frames)])
    d.close()


def test_scanl_similarity_tables_and_class_ranges(lm, orc, synth):
    """k_scanl with similarity tables whose misses cost 1, 2, 3 or 4 (the miss bound follows the table) and with a class list: the lane items
    of a class range start at the class's first template."""
    d, o, frames = _setup(lm, orc, synth, False, (640, 480), [5, 8], 50, 4, seed=1950)
    bgr, dep = frames[0]
    q = _quantized(o, bgr, dep, False)
    for c in (1, 2):
        descs, feats, _ = synth.make_bank(17 + 5 * c, 2, 2, seed=2000 + c, quantized=q, crop_fraction=0.4, frame_size=(640, 480), T0=5)
        d.add_class("c%d" % c, descs, feats); o.add_class("c%d" % c, descs, feats)
    d.set_tuning(lm.TUNE_SCAN_FORM, 3)
    base = orc.similarity_lut()
    tables = [base]
    for lo, hi in ((3, 4), (2, 4), (0, 4), (1, 3)):
        t = base.copy()
        t[base == 1] = lo; t[base == 4] = hi
        tables.append(t)
    for t in tables:
        d.set_similarity_lut(t); o.set_similarity_lut(t)
        for thr in (70.0, 90.0):
            for k in range(4):
                d.upload_frame(k, bgr, dep)
            got, cnt = d.match_batch_classes(0, 4, thr, [-1], cap_per_frame=1 << 16)
            assert d.get_scan_form_stats()[3] >= 1000
            exp = o.match(bgr, dep, thr, -1, threads=8, cap=1 << 18)
            for k in range(4):
                assert_matches_equal(got[k, :cnt[k]], exp)
            per_class = [o.match(bgr, dep, thr, c, threads=8, cap=1 << 18) for c in range(3)]
            for cls in ([1], [0, 2], [2]):
                got, cnt = d.match_prepared(0, 4, thr, cls, cap_per_frame=1 << 16)
                assert d.get_scan_form_stats()[3] >= 1000
                for k in (0, 3):
                    for c in cls:
                        assert_matches_equal(class_sublist(got[k, :cnt[k]], c), per_class[c])
            o.prepare(bgr, dep)
            d.prepare_slot(0)
            for ci in (-1, 0, 1, 2):
                assert np.array_equal(d.stage_scan(0, thr, ci, cap=1 << 20), o.scan_candidates(thr, ci, threads=8, cap=1 << 20))
    d.close()


def test_scanl_full_queues_many_times(lm, orc, synth):
    """r06: a workgroup of k_scanl whose survivors FILL its LDS queue (1916 entries) -- concurrent reservations at the capacity, partial fits, the waves' own
    sums for the rest -- must lose nothing: 24 copies of one frame against a bank cut from it, at a threshold low enough for thousands of survivors per
    workgroup, 12 times over; every list equals the oracle's.  (The first form of the queue gave failed reservations back by an atomic subtract and lost one
    candidate in four million on config 2's heavy frame: profiles/r06_ab_experiments.log section 5.)"""
    d, o, frames = _setup(lm, orc, synth, False, (640, 480), [5, 8], 400, 24, seed=2100)
    bgr, dep = frames[0]
    thr = 62.0
    exp = o.match(bgr, dep, thr, threads=8, cap=1 << 19)
    assert len(exp) > 200
    for k in range(24):
        d.upload_frame(k, bgr, dep)
    d.set_tuning(lm.TUNE_SCAN_FORM, 3)
    d.set_scan_stats(True)
    got, cnt = d.match_batch(24, thr, cap_per_frame=1 << 16)
    assert d.get_scan_form_stats()[3] >= 1000 and d.get_scan_form_stats()[2] > 24 * 3000      # thousands of survivors per frame: the queues overflow
    d.set_scan_stats(False)
    for rep in range(12):
        got, cnt = d.match_batch(24, thr, cap_per_frame=1 << 16)
        for k in range(24):
            assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (rep, k, cnt[k], len(exp))
    d.close()


@pytest.mark.parametrize("color_only", [True, False])
def test_scan1_full_queues_many_times(lm, orc, synth, color_only):
    """r06: k_scan1's survivor queues at their capacity.  The default queues (131 072 entries per XCD) never fill in the suite, so the code the round
    changed -- a reservation that does not fit writes as many survivors as fit, the wave sums the rest itself, the counter only grows -- ran nowhere;
    LM_TUNE_SURVIVOR_QUEUE shrinks them to 64 entries per XCD queue, every launch overflows all eight, and every list must still equal the oracle's,
    12 times over (both second stages: nibble memories for the RGB-D detector's mixed call, spread bytes for the colour-only one)."""
    d, o, frames = _setup(lm, orc, synth, color_only, (640, 480), [5, 8], 400, 16, seed=2200)
    bgr, dep = frames[0][0], (None if color_only else frames[0][1])
    thr = 62.0
    exp = o.match(bgr, dep, thr, threads=8, cap=1 << 19)
    assert len(exp) > 100
    for k in range(16):
        d.upload_frame(k, bgr, dep)
    with pytest.raises(lm.LinemodError):
        d.set_tuning(lm.TUNE_SURVIVOR_QUEUE, 8)
    d.set_tuning(lm.TUNE_SCAN_FORM, 2)
    d.set_tuning(lm.TUNE_SURVIVOR_QUEUE, 512)
    d.set_scan_stats(True)
    got, cnt = d.match_batch(16, thr, cap_per_frame=1 << 16)
    st = d.get_scan_form_stats()
    assert st[0] >= 1 and 0 < st[3] < 1000 and st[2] > 16 * 1000        # k_scan1 ran, thousands of survivors per frame against 512 queue entries
    d.set_scan_stats(False)
    for rep in range(12):
        got, cnt = d.match_batch(16, thr, cap_per_frame=1 << 16)
        for k in range(16):
            assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (rep, k, cnt[k], len(exp))
    d.set_tuning(lm.TUNE_SURVIVOR_QUEUE, 1 << 20)                      # back to the default: the queues are re-allocated on the next scan
    got, cnt = d.match_batch(16, thr, cap_per_frame=1 << 16)
    for k in range(16):
        assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes()
    d.close()


def test_scanl_feature_counts(lm, orc, synth):
    """r06: k_scanl with templates of few features (42, 21, 13, 9, 2 at level 0, i.e. 21 .. 1 per modality at the scanned level: lists padded with entries of
    the zero block, tail rounds), with a second class and a class list, at two thresholds; and with MORE than 64 features at the scanned level -- a
    one-level pyramid on a 320 x 240 frame scans level 0 itself, 63 features per modality = 126 per template: 16 rounds of eight, and the second stage
    takes the second half of a survivor's list too (a lane requests 64 entries at once).  Lists and candidate lists equal the oracle's."""
    for nf in (42, 21, 13, 9, 2):
        d, o, frames = _setup(lm, orc, synth, False, (640, 480), [5, 8], 40, 8, seed=3070 + nf, nf=nf)
        bgr, dep = frames[0]
        q = _quantized(o, bgr, dep, False)
        d2, f2, _ = synth.make_bank(17, 2, 2, seed=3234 + nf, quantized=q, crop_fraction=0.5, frame_size=(640, 480), T0=5, num_features=nf)
        d.add_class("c2", d2, f2); o.add_class("c2", d2, f2)
        d.set_tuning(lm.TUNE_SCAN_FORM, 3)
        for k in range(8):
            d.upload_frame(k, bgr, dep)
        for thr in (60.0, 85.0):
            for classes in ([-1], [1], [1, 0]):
                exp = o.match(bgr, dep, thr, class_idx=classes[0] if len(classes) == 1 else -1, threads=8, cap=1 << 18)
                got, cnt = d.match_batch_classes(0, 8, thr, classes, cap_per_frame=max(len(exp), 1))
                assert d.get_scan_form_stats()[3] >= 1000, (nf, thr, classes)                      # k_scanl ran
                for k in range(8):
                    assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (nf, thr, classes, k, cnt[k], len(exp))
            o.prepare(bgr, dep)
            d.prepare_slot(0)
            assert np.array_equal(d.stage_scan(0, thr, -1, cap=1 << 20), o.scan_candidates(thr, -1, threads=8, cap=1 << 20)), (nf, thr)
        d.close()
    # one pyramid level: the scanned level is level 0 with all 63 features per modality
    w, h, T = 320, 240, [8]
    d = lm.Detector(lm.default_config(color_only=False, width=w, height=h, T=T, frame_slots=8))
    o = orc.Detector(color_only=False, T=T)
    bgr, dep = synth.make_frame(w, h, seed=3333)
    o.prepare(bgr, dep)
    q = {(0, m): o.stage(0, 0, m).reshape(h, w) for m in range(2)}
    descs, feats, _ = synth.make_bank(60, 2, 1, seed=3334, quantized=q, crop_fraction=0.4, frame_size=(w, h), T0=T[0], size_range=(40, 80))
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    d.set_tuning(lm.TUNE_SCAN_FORM, 3)
    for k in range(8):
        d.upload_frame(k, bgr, dep)
    for thr in (60.0, 80.0):
        exp = o.match(bgr, dep, thr, threads=8, cap=1 << 18)
        assert len(exp) > 0
        got, cnt = d.match_batch(8, thr, cap_per_frame=max(len(exp), 1))
        assert d.get_scan_form_stats()[3] >= 1000, thr                                              # k_scanl ran on the one-level pyramid
        for k in range(8):
            assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (thr, k, cnt[k], len(exp))
        o.prepare(bgr, dep)
        d.prepare_slot(0)
        assert np.array_equal(d.stage_scan(0, thr, -1, cap=1 << 20), o.scan_candidates(thr, -1, threads=8, cap=1 << 20)), thr
    d.close()


def test_scan1_first_use_of_a_lanes_queue(lm, orc, synth):
    """r06: a lane's survivor queue is allocated -- and its counters zeroed -- on the lane's first bit-plane scan.  The zeroing was a hipMemset on the null
    stream, which is not ordered against the lanes' non-blocking streams: when the lane's stream was idle at that moment (its pre-processing over, or none to do)
    k_scan1 started within microseconds of the fill and could count survivors before the counters were zeroed -- k_scan1_exact then summed fewer than were
    queued (a lost match about once in ten runs of the suite, always a lane's first scan).  LM_TUNE_SURVIVOR_QUEUE frees the queues, so every repetition here IS
    a first use, and lm_match_prepared launches the scan on an idle stream right behind the allocation."""
    nb = 16
    d, o, frames = _setup(lm, orc, synth, True, (640, 480), [5, 8], 300, nb, seed=2300)
    bgr = frames[0][0]
    thr = 70.0
    exp = o.match(bgr, None, thr, threads=8, cap=1 << 19)
    assert len(exp) > 50
    for k in range(nb):
        d.upload_frame(k, bgr, None)
    d.set_tuning(lm.TUNE_SCAN_FORM, 2)
    got, cnt = d.match_batch(nb, thr, cap_per_frame=1 << 15)       # prepares the slots
    for rep in range(60):
        d.set_tuning(lm.TUNE_SURVIVOR_QUEUE, (1 << 20) + 8 * (rep & 1))     # frees every lane's queue (and un-prepares nothing)
        got, cnt = d.match_prepared(0, nb, thr, [-1], cap_per_frame=1 << 15)
        for k in range(nb):
            assert cnt[k] == len(exp) and got[k, :cnt[k]].tobytes() == exp.tobytes(), (rep, k, cnt[k], len(exp))
    assert 0 < d.get_scan_form_stats()[3] < 1000                    # k_scan1 ran
    d.close()
