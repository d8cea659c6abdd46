"""Streaming input path (the reference's real call pattern: a fresh camera frame per detect(), detector.cpp:17-42):
asynchronous uploads on the detector's copy stream, ordered against the lanes by per-slot events.  Every list is
compared with the CPU oracle's on the same frame."""
import numpy as np
import pytest

from conftest import assert_matches_equal

pytestmark = pytest.mark.gpu

W, H, THR = 640, 480, 75.0


def _setup(lm, orc, synth, n_templates=120, slots=8, n_frames=6):
    d = lm.Detector(color_only=False, width=W, height=H, frame_slots=slots)
    o = orc.Detector(color_only=False)
    frames = [synth.make_frame(W, H, seed=500 + i) for i in range(n_frames)]
    o.prepare(*frames[0])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(n_templates, 2, 2, seed=99, quantized=q, crop_fraction=0.3, frame_size=(W, H),
                                      T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    exp = [o.match(b, dp, THR, threads=8) for b, dp in frames]
    assert sum(len(e) for e in exp) > 0
    return d, frames, exp


def test_upload_immediately_followed_by_begin_on_lane1(lm, orc, synth):
    """an upload into lane 1's slots directly followed by lm_match_begin(1) used to race (the copy
    ran on lane 0's stream, the kernels on lane 1's).  Changing frames every round: a stale or half-copied frame
    shows up as a wrong list."""
    d, frames, exp = _setup(lm, orc, synth)
    nf = len(frames)
    for rnd in range(12):
        # lane 0 keeps its stream busy with slots 0..3 while lane 1's slots 4..7 are refilled and started at once
        ids0 = [(rnd + k) % nf for k in range(4)]
        ids1 = [(3 * rnd + 2 * k + 1) % nf for k in range(4)]
        for k, f in enumerate(ids0):
            d.upload_frame(k, *frames[f])
        d.match_begin(0, 0, 4, THR, 0)
        for k, f in enumerate(ids1):
            d.upload_frame(4 + k, *frames[f])
        d.match_begin(1, 4, 4, THR, 0)
        out0, c0 = d.match_end(0, n_slots=4)
        out1, c1 = d.match_end(1, n_slots=4)
        for k, f in enumerate(ids0):
            assert_matches_equal(out0[k, :c0[k]], exp[f])
        for k, f in enumerate(ids1):
            assert_matches_equal(out1[k, :c1[k]], exp[f])
    d.close()


def test_upload_refused_for_busy_lane_and_allowed_elsewhere(lm, orc, synth):
    d, frames, exp = _setup(lm, orc, synth, n_frames=3)
    for k in range(8):
        d.upload_frame(k, *frames[k % 3])
    d.match_begin(0, 0, 4, THR, 0)
    with pytest.raises(lm.LinemodError):
        d.upload_frame(2, *frames[1])            # slot of the lane in flight
    d.upload_frame(6, *frames[2])                # another slot: allowed while lane 0 computes
    out0, c0 = d.match_end(0, n_slots=4)
    for k in range(4):
        assert_matches_equal(out0[k, :c0[k]], exp[k % 3])
    got = d.match_slot(6, THR, 0)
    assert_matches_equal(got, exp[2])
    d.close()


def test_double_buffered_streaming_pinned(lm, orc, synth):
    """The serving loop of include/linemod_hip.h: while a lane computes slot range A the next frames are uploaded
    into range B from PINNED host memory (no staging copy), then the roles swap.  Frames change every step."""
    d, frames, exp = _setup(lm, orc, synth)
    nf, B = len(frames), 4
    pb = lm.PinnedBuffer(2 * B * (W * H * 3 + W * H * 2))
    hb = [[pb.view(np.uint8, (H, W, 3), offset=(r * B + k) * W * H * 5) for k in range(B)] for r in range(2)]
    hd = [[pb.view(np.uint16, (H, W), offset=(r * B + k) * W * H * 5 + W * H * 3) for k in range(B)] for r in range(2)]

    def fill(r, step):
        ids = [(step * B + k) % nf for k in range(B)]
        for k, f in enumerate(ids):
            hb[r][k][...] = frames[f][0]
            hd[r][k][...] = frames[f][1]
            d.upload_frame_pinned(r * B + k, hb[r][k], hd[r][k])
        return ids

    ids = {0: fill(0, 0)}
    d.match_begin(0, 0, B, THR, 0)
    for step in range(1, 9):
        r = step & 1
        ids[r] = fill(r, step)                   # overlaps the compute of the other range
        out, cnt = d.match_end(0, n_slots=B)
        for k, f in enumerate(ids[r ^ 1]):
            assert_matches_equal(out[k, :cnt[k]], exp[f])
        d.match_begin(0, r * B, B, THR, 0)
    out, cnt = d.match_end(0, n_slots=B)
    for k, f in enumerate(ids[8 & 1]):
        assert_matches_equal(out[k, :cnt[k]], exp[f])
    d.close()
    pb.close()


@pytest.mark.parametrize("chunks", [1, 2, 5])
def test_single_frame_calls_with_staging_chunks(lm, orc, synth, chunks):
    """lm_match (host frame in, the drop-in call): the staging memcpy is cut into pieces that overlap the DMA."""
    d, frames, exp = _setup(lm, orc, synth, n_frames=3)
    d.set_stage_chunks(chunks)
    for rnd in range(3):
        for f in range(3):
            assert_matches_equal(d.match(frames[f][0], frames[f][1], THR, 0), exp[f])
    # strided sources (a cv::Mat ROI): row pitch larger than the row
    big = np.zeros((H, W + 16, 3), np.uint8)
    bigd = np.zeros((H, W + 8), np.uint16)
    big[:, :W] = frames[1][0]
    bigd[:, :W] = frames[1][1]
    out = np.zeros(4096, lm.MATCH_DTYPE)
    import ctypes as C
    n = C.c_size_t()
    rc = d.lib.lm_match(d.h, big.ctypes.data_as(C.c_void_p), big.strides[0], bigd.ctypes.data_as(C.c_void_p),
                        bigd.strides[0], THR, 0, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    assert rc == 0
    assert_matches_equal(out[:n.value], exp[1])
    d.close()


@pytest.mark.parametrize("color_only", [False, True])
def test_single_frame_host_and_resident_calls(lm, orc, synth, color_only):
    """Single-frame calls -- lm_match on a host frame (copies inline on the compute stream) and an upload immediately followed by
    lm_match_slot -- with frames changing every call: same lists.  (r05: the launch-shape knobs this test used to sweep, three
    concurrent pre-processing chains and lm_match's copies on the copy streams, lost their A/B runs in r02 and were deleted;
    lm_set_tuning rejects their keys.)"""
    d = lm.Detector(color_only=color_only, width=W, height=H, frame_slots=4)
    o = orc.Detector(color_only=color_only)
    M = 1 if color_only else 2
    frames = [synth.make_frame(W, H, seed=900 + i) for i in range(4)]
    o.prepare(frames[0][0], None if color_only else frames[0][1])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(150, M, 2, seed=5, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    for removed_key in (1, 2, 10):
        with pytest.raises(lm.LinemodError):
            d.set_tuning(removed_key, 0)
    with pytest.raises(lm.LinemodError):
        d.set_tuning(lm.TUNE_CBLUR_VARIANT, 2)             # r02's sliding-window blur: deleted
    exp = [o.match(b, None if color_only else dp, 70.0, threads=8) for b, dp in frames]
    for rnd in range(5):
        for f, (b, dp) in enumerate(frames):
            assert_matches_equal(d.match(b, None if color_only else dp, 70.0), exp[f])
        for f, (b, dp) in enumerate(frames):
            d.upload_frame(f, b, None if color_only else dp)
            assert_matches_equal(d.match_slot(f, 70.0), exp[f])       # upload immediately followed by the match
    d.close()


@pytest.mark.parametrize("color_only", [False, True])
def test_batch_upload_one_strided_transfer(lm, orc, synth, color_only):
    """lm_upload_frames_pinned: a run of [colour | depth] host frames into consecutive slots with one hipMemcpy2DAsync,
    immediately followed by the match (per-slot events order it); host frame stride larger than a frame."""
    M = 1 if color_only else 2
    d = lm.Detector(color_only=color_only, width=W, height=H, frame_slots=8)
    o = orc.Detector(color_only=color_only)
    frames = [synth.make_frame(W, H, seed=700 + i) for i in range(5)]
    o.prepare(frames[0][0], None if color_only else frames[0][1])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(100, M, 2, seed=6, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    exp = [o.match(b, None if color_only else dp, THR, threads=8) for b, dp in frames]
    fb = W * H * (3 if color_only else 5)
    stride = fb + 4096
    pb = lm.PinnedBuffer(5 * stride)
    for rnd in range(3):
        order = [(rnd + 2 * k) % 5 for k in range(5)]
        for k, f in enumerate(order):
            pb.view(np.uint8, (H, W, 3), offset=k * stride)[...] = frames[f][0]
            if not color_only:
                pb.view(np.uint16, (H, W), offset=k * stride + W * H * 3)[...] = frames[f][1]
        d.upload_frames_pinned(2, 5, pb.ptr.value, stride)
        d.match_begin(1, 2, 5, THR, -1)
        out, cnt = d.match_end(1, n_slots=5)
        for k, f in enumerate(order):
            assert_matches_equal(out[k, :cnt[k]], exp[f])
    d.close()
    pb.close()


@pytest.mark.parametrize("color_only", [False, True])
def test_few_frames_take_one_launch_per_dependency_level(lm, orc, synth, color_only):
    """Calls of fewer than 16 frames run a3-a10 as five launches, each holding the independent kernels of one dependency
    level (LM_TUNE_PHASE_MAX_SLOTS, default 15), instead of the plain sequence: both must give the oracle's lists, for
    the RGB-D pyramid T = {5, 8} and the colour-only one T = {2, 8}, at every batch size on either side of 8."""
    M = 1 if color_only else 2
    d = lm.Detector(color_only=color_only, width=W, height=H, frame_slots=12)
    o = orc.Detector(color_only=color_only)
    frames = [synth.make_frame(W, H, seed=700 + i) for i in range(5)]
    o.prepare(frames[0][0], None if color_only else frames[0][1])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(100, M, 2, seed=77, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    exp = [o.match(b, None if color_only else dp, THR, threads=8) for b, dp in frames]
    assert sum(len(e) for e in exp) > 0
    for k in range(12):
        b, dp = frames[(2 * k + 1) % 5]
        d.upload_frame(k, b, None if color_only else dp)
    for phases in (15, 0):
        d.set_tuning(lm.TUNE_PHASE_MAX_SLOTS, phases)
        for n in (1, 2, 3, 8, 12):
            out, cnt = d.match_batch(n, THR, 0)
            for k in range(n):
                try:
                    assert_matches_equal(out[k, :cnt[k]], exp[(2 * k + 1) % 5])
                except AssertionError as e:
                    raise AssertionError("phases %d, batch of %d, slot %d: %s" % (phases, n, k, str(e)[:80]))
        b, dp = frames[4]
        assert_matches_equal(d.match(b, None if color_only else dp, THR, 0), exp[4])     # (lm_match goes through slot 0)
        d.upload_frame(0, frames[1][0], None if color_only else frames[1][1])
    d.close()


def test_lm_match_takes_pinned_frames_without_staging(lm, orc, synth):
    """lm_match recognises frames inside blocks from lm_host_alloc: they go to the device in one DMA transfer without
    the staging copy, pageable memory goes through the staging buffer -- same lists."""
    d, frames, exp = _setup(lm, orc, synth, n_frames=3)
    pb = lm.PinnedBuffer(3 * W * H * 5)
    for k, (b, dp) in enumerate(frames):
        pc = pb.view(np.uint8, (H, W, 3), offset=k * W * H * 5)
        pd = pb.view(np.uint16, (H, W), offset=k * W * H * 5 + W * H * 3)
        pc[...] = b; pd[...] = dp
        assert_matches_equal(d.match(pc, pd, THR, 0), exp[k])              # contiguous [colour | depth]: one transfer
        assert_matches_equal(d.match(b, dp, THR, 0), exp[k])               # pageable
        assert_matches_equal(d.match(pc, dp, THR, 0), exp[k])              # mixed: staged
    d.close()
    pb.close()


@pytest.mark.parametrize("color_only,size", [(False, (640, 480)), (True, (640, 480)), (True, (1280, 960)), (False, (320, 240))])
def test_batches_take_one_launch_per_dependency_level(lm, orc, synth, color_only, size):
    """Calls of 16 or more frames run a3-a10 as four launches in which the batch kernels of one dependency level share a
    grid (LM_TUNE_BATCH_PHASES, default on; r03: the level-1 kernels fill the tail of the level-0 ones).  Same buffers and
    lists as one launch per kernel and as the oracle: quantised images and linear memories of every level and modality
    are diffed for the first, a middle and the last slot of the batch, the lists for all."""
    w, h = size
    M = 1 if color_only else 2
    n = 24 if w <= 640 else 16
    d = lm.Detector(color_only=color_only, width=w, height=h, frame_slots=n)
    o = orc.Detector(color_only=color_only)
    frames = [synth.make_frame(w, h, seed=900 + i) for i in range(4)]
    o.prepare(frames[0][0], None if color_only else frames[0][1])
    q = {(l, m): o.stage(0, l, m).reshape(h >> l, w >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(60, M, 2, seed=78, size_range=(48, min(160, h // 2)), quantized=q, crop_fraction=0.3,
                                      frame_size=(w, h), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    exp = [o.match(b, None if color_only else dp, THR, threads=8) for b, dp in frames]
    assert sum(len(e) for e in exp) > 0
    for k in range(n):
        b, dp = frames[(3 * k + 1) % 4]
        d.upload_frame(k, b, None if color_only else dp)
    for phases, blur_pyr, pairs in ((0, 1, 1), (1, 1, 0), (0, 1, 0), (2, 1, 0), (0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 2, 0), (1, 2, 0), (2, 2, 1)):      # (pairs: only varies the blur strip since r05)
        d.set_tuning(lm.TUNE_BATCH_PHASES, phases)
        d.set_tuning(lm.TUNE_BLUR_STRIP, (0, 16, 32, 64)[(phases + 2 * pairs + blur_pyr) % 4])      # rows per blur strip inside k_blur_pyr
        d.set_tuning(lm.TUNE_BLUR_PYR, blur_pyr)     # level-0 blur + pyrDown apart (0), in one launch back to back per slot (1), or dealt out evenly (2)
        d.set_tuning(lm.TUNE_CGRAD_LEVELS, (phases + pairs) % 2)      # r06: the two levels' gradients in one launch (1) or one launch per level (0); used when phases == 0
        for nb in (n, 16):
            out, cnt = d.match_batch(nb, THR, 0)
            for k in range(nb):
                try:
                    assert_matches_equal(out[k, :cnt[k]], exp[(3 * k + 1) % 4])
                except AssertionError as e:
                    raise AssertionError("batch phases %d, blur_pyr %d, pairs %d, batch of %d, slot %d: %s" % (phases, blur_pyr, pairs, nb, k, str(e)[:80]))
        for k in (0, n // 2 + 1, n - 1):
            b, dp = frames[(3 * k + 1) % 4]
            o.prepare(b, None if color_only else dp)
            for level in range(2):
                for mod in range(M):
                    assert np.array_equal(d.debug_read(k, 0, level, mod), o.stage(0, level, mod)), (phases, blur_pyr, pairs, k, level, mod)
                    assert np.array_equal(d.debug_read(k, 2, level, mod), o.stage(2, level, mod)), (phases, blur_pyr, pairs, k, level, mod)
    d.set_tuning(lm.TUNE_BLUR_PYR, 3)
    d.set_tuning(lm.TUNE_BLUR_STRIP, 0)
    d.set_tuning(lm.TUNE_CGRAD_LEVELS, 1)
    d.close()


def test_shifted_upload_equals_upload_of_the_shifted_frame(lm, orc, synth):
    """lm_upload_frame_shifted = the reference's translateImg (zeros shifted in) folded into the staging copy: every shift, also beyond the
    frame, must give the match list of the host-shifted frame."""
    W, H = 640, 480
    bgr, depth = synth.make_frame(W, H, seed=4242)
    d = lm.Detector(color_only=False, width=W, height=H, frame_slots=2)
    o = orc.Detector(color_only=False)
    o.prepare(bgr, depth)
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(40, 2, 2, seed=9, quantized=q, crop_fraction=0.5, frame_size=(W, H), T0=5)
    d.add_class("c", descs, feats)

    def host_shift(img, ox, oy):
        out = np.zeros_like(img)
        x0, x1, y0, y1 = max(ox, 0), min(W + ox, W), max(oy, 0), min(H + oy, H)
        if x1 > x0 and y1 > y0:
            out[y0:y1, x0:x1] = img[y0 - oy:y1 - oy, x0 - ox:x1 - ox]
        return out

    for ox, oy in ((0, 0), (-12, 10), (37, -5), (-640, 3), (5, 480), (1, 1)):
        d.upload_frame_shifted(0, bgr, depth, ox, oy)
        d.upload_frame(1, host_shift(bgr, ox, oy), host_shift(depth, ox, oy))
        out, cnt = d.match_batch(2, 70.0)
        assert cnt[0] == cnt[1] and out[0, :cnt[0]].tobytes() == out[1, :cnt[1]].tobytes(), (ox, oy)
        assert np.array_equal(d.debug_read(0, 0, 0, 1), d.debug_read(1, 0, 0, 1)), (ox, oy)      # the depth modality's quantised image
    d.close()


def _translate(img, ox, oy):
    """numpy restatement of the reference's translateImg (warpAffine with a pure integer translation, zeros shifted in)."""
    out = np.zeros_like(img)
    h, w = img.shape[:2]
    x0, x1 = max(ox, 0), min(w + ox, w)
    y0, y1 = max(oy, 0), min(h + oy, h)
    if x1 > x0 and y1 > y0:
        out[y0:y1, x0:x1] = img[y0 - oy:y1 - oy, x0 - ox:x1 - ox]
    return out


def test_staged_and_pinned_shifted_uploads_equal_the_translated_frame(lm, orc, synth):
    """r05: the three-step staged upload (lm_stage_reserve, lm_stage_rows over row ranges filled by several threads,
    lm_upload_staged) and the pinned row-offset copy (lm_upload_frame_pinned_shifted) put the same frame into a slot as uploading
    the host-translated frame: the match lists equal the oracle's on the numpy-translated frame, for shifts of both signs,
    a shift larger than the frame (clamped: an all-zero frame) and no shift."""
    import threading
    d, frames, _ = _setup(lm, orc, synth, n_frames=2)
    o = orc.Detector(color_only=False)
    o.add_class("c", *_bank_of(d, lm))
    pb = lm.PinnedBuffer(W * H * 5)
    pc, pd_ = pb.view(np.uint8, (H, W, 3)), pb.view(np.uint16, (H, W), offset=W * H * 3)
    for (sx, sy) in ((-12, 10), (7, -5), (0, 0), (33, 0), (0, -41), (5 * W, 3)):
        bgr, depth = frames[(sx + sy) % 2]
        tb, td = _translate(bgr, max(-W, min(W, sx)), max(-H, min(H, sy))), _translate(depth, max(-W, min(W, sx)), max(-H, min(H, sy)))
        exp = o.match(tb, td, THR, threads=8)
        # staged: rows dealt to four threads in interleaved pieces
        d.stage_reserve(2, 1)
        pieces = [(r, min(r + 37, H)) for r in range(0, H, 37)]
        th = [threading.Thread(target=lambda k=k: [d.stage_rows(2, bgr, depth, sx, sy, a, b) for (a, b) in pieces[k::4]]) for k in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        d.upload_staged(2)
        out, c = d.match_batch_classes(2, 1, THR, [0])
        assert_matches_equal(out[0, :c[0]], exp)
        # pinned: the DMA engine's row-offset copy
        pc[...] = bgr
        pd_[...] = depth
        d.upload_frame_pinned_shifted(3, pc, pd_, sx, sy)
        out, c = d.match_batch_classes(3, 1, THR, [0])
        assert_matches_equal(out[0, :c[0]], exp)
        # and the r04 one-call form
        d.upload_frame_shifted(4, bgr, depth, sx, sy)
        out, c = d.match_batch_classes(4, 1, THR, [0])
        assert_matches_equal(out[0, :c[0]], exp)
    with pytest.raises(lm.LinemodError):
        d.stage_rows(5, frames[0][0], frames[0][1], 0, 0, 0, H)      # no lm_stage_reserve for slot 5
    with pytest.raises(lm.LinemodError):
        d.upload_staged(5)
    d.stage_reserve(5, 1)
    with pytest.raises(lm.LinemodError):
        d.stage_rows(5, frames[0][0], frames[0][1], 0, 0, 0, H + 1)  # row range outside the frame
    pb.close([d])
    d.close()


def _bank_of(d, lm):
    """(descs, features) of class 0 of detector d, read back through lm_get_template (whole pyramids, [template][level * M + modality])."""
    M, L = d.num_modalities, d.pyramid_levels
    descs, feats = [], []
    for t in range(d.class_num_templates(0)):
        for l in range(L):
            for m in range(M):
                w, h, f = d.get_template(0, t, l, m)
                descs.append((w, h, l, len(f)))
                feats.append(f)
    return np.array(descs, lm.DESC_DTYPE), np.concatenate(feats)


def test_match_collect_redelivers_after_overflow(lm, orc, synth):
    """r05: lm_match_end with too small a buffer reports LM_ERR_OVERFLOW and the lengths; lm_match_collect then hands over the same lists
    without a second pass over the GPU, until the slot is uploaded to again."""
    d, frames, exp = _setup(lm, orc, synth, n_frames=3)
    for k in range(3):
        d.upload_frame(k, *frames[k])
    big = max(len(e) for e in exp)
    assert big > 4
    d.match_begin(1, 0, 3, THR, 0)
    with pytest.raises(lm.LinemodError) as ei:
        d.match_end(1, cap_per_frame=2, n_slots=3)
    assert ei.value.code == lm.LM_ERR_OVERFLOW
    stage0 = d.get_stage_counts()
    out, c = d.match_collect(0, 3, cap_per_frame=big)
    assert d.get_stage_counts() == stage0                  # nothing ran on the GPU
    for k in range(3):
        assert_matches_equal(out[k, :c[k]], exp[k])
    d.upload_frame(1, *frames[2])
    with pytest.raises(lm.LinemodError):
        d.match_collect(0, 3, cap_per_frame=big)           # slot 1 holds a frame no match has run on
    out, c = d.match_collect(0, 1, cap_per_frame=big)
    assert_matches_equal(out[0, :c[0]], exp[0])
    d.close()


def test_colour_check_of_a_batch_in_one_call_beside_a_busy_lane(lm, orc, synth):
    """r05: lm_color_check_counts_slots (a list spanning several resident frames: one mask launch, one hull launch) returns, per match,
    what lm_color_check_counts returns for that match's slot alone -- also while ANOTHER lane has a match in flight (own stream and
    buffers); slots of the busy lane are refused."""
    d, frames, exp = _setup(lm, orc, synth, n_frames=4)
    for k in range(8):
        d.upload_frame(k, *frames[k % 4])
    lo, hi = (0, 0, 60), (255, 200, 255)
    out, c = d.match_batch_classes(0, 4, THR, [0])
    lists = [out[k, :min(c[k], 300)].copy() for k in range(4)]
    assert sum(len(l) for l in lists) > 20
    single = [d.color_check_counts(k, lo, hi, lists[k]) for k in range(4)]
    allm = np.concatenate(lists)
    slot_of = np.concatenate([np.full(len(l), k, np.int32) for k, l in enumerate(lists)])
    perm = np.random.default_rng(5).permutation(len(allm))             # any order of the list
    d.match_begin(1, 4, 4, THR, 0)                                       # lane 1 busy on slots 4..7
    a, b = d.color_check_counts_slots(slot_of[perm], lo, hi, allm[perm])
    inv = np.argsort(perm)
    assert np.array_equal(a[inv], np.concatenate([s[0] for s in single])) and np.array_equal(b[inv], np.concatenate([s[1] for s in single]))
    assert a.sum() > 0 and 0 < b.sum() < a.sum()
    a1, b1 = d.color_check_counts(2, lo, hi, lists[2])                  # the one-slot form beside the busy lane
    assert np.array_equal(a1, single[2][0]) and np.array_equal(b1, single[2][1])
    with pytest.raises(lm.LinemodError):
        d.color_check_counts_slots(np.full(len(lists[0]), 5, np.int32), lo, hi, lists[0])     # slot 5 belongs to the match in flight
    out1, c1 = d.match_end(1, n_slots=4)
    for k in range(4):
        assert_matches_equal(out1[k, :c1[k]], exp[k])
    # lm_color_mask_prepare: the masks computed on the lane AHEAD of the match; the check afterwards (same range) skips its mask launch
    # and counts the same; another range recomputes; an upload invalidates the slot's mask (a stale mask would give the old frame's counts)
    d.color_mask_prepare(1, 0, 4, lo, hi)
    d.match_begin(1, 0, 4, THR, 0)
    with pytest.raises(lm.LinemodError):
        d.color_mask_prepare(1, 0, 4, lo, hi)                            # the lane is busy
    out0, c0 = d.match_end(1, n_slots=4)
    a2, b2 = d.color_check_counts_slots(slot_of, lo, hi, allm)
    assert np.array_equal(a2, np.concatenate([s[0] for s in single])) and np.array_equal(b2, np.concatenate([s[1] for s in single]))
    lo2, hi2 = (0, 0, 120), (255, 255, 255)
    a3, b3 = d.color_check_counts_slots(slot_of, lo2, hi2, allm)
    ref3 = [d.color_check_counts(k, lo2, hi2, lists[k]) for k in range(4)]
    assert np.array_equal(b3, np.concatenate([r[1] for r in ref3])) and not np.array_equal(b3, b2)
    d.color_mask_prepare(0, 0, 4, lo, hi)
    d.upload_frame(0, *frames[3])                                        # slot 0 now holds another frame: its prepared mask is stale
    d.match_begin(0, 0, 4, THR, 0)
    d.match_end(0, n_slots=4)
    a4, b4 = d.color_check_counts_slots(np.zeros(len(lists[0]), np.int32), lo, hi, lists[0])
    d.upload_frame(5, *frames[3])
    a5, b5 = d.color_check_counts(5, lo, hi, lists[0])                   # the same frame in a slot that never had a prepared mask
    assert np.array_equal(a4, a5) and np.array_equal(b4, b5)
    # r06: (i) masks prepared on a lane whose match is never begun: the check that reuses them waits for the mask launch (an event behind it) and
    # counts the same; (ii) while a colour check is in flight no upload goes to a slot it reads, any other slot is free; after its end the slot is free again
    for k in range(4):
        d.upload_frame(k, *frames[k % 4])
    d.color_mask_prepare(2, 0, 4, lo, hi)
    a6, b6 = d.color_check_counts_slots(slot_of, lo, hi, allm)
    assert np.array_equal(a6, np.concatenate([s[0] for s in single])) and np.array_equal(b6, np.concatenate([s[1] for s in single]))
    import ctypes as C
    m = np.ascontiguousarray(allm); sl = np.ascontiguousarray(slot_of)
    clo, chi = (C.c_double * 3)(*lo), (C.c_double * 3)(*hi)
    d._check(d.lib.lm_color_check_begin_slots(d.h, sl.ctypes.data_as(C.c_void_p), clo, chi, m.ctypes.data_as(C.c_void_p), len(m)))
    with pytest.raises(lm.LinemodError):
        d.upload_frame(1, *frames[2])
    d.upload_frame(6, *frames[2])
    a7, b7 = np.zeros(len(m), np.int64), np.zeros(len(m), np.int64)
    d._check(d.lib.lm_color_check_end(d.h, a7.ctypes.data_as(C.c_void_p), b7.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(a7, a6) and np.array_equal(b7, b6)
    d.upload_frame(1, *frames[2])
    d.close()


def test_depth_counts_of_a_batch_in_one_call(lm, synth):
    """r06: lm_depth_counts_begin / _end -- per query the crop's values below the window and inside it, depths <= 1 counted as 65535 -- against numpy on
    frames with holes, in several slots, also through a shifted upload (the resident frame is the translated one) and beside a busy lane; refusals."""
    W, H = 640, 480
    d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=6))
    rng = np.random.default_rng(5)
    frames = []
    for k in range(4):
        bgr, depth = synth.make_frame(W, H, seed=700 + k)
        depth = depth.copy()
        depth[rng.integers(0, H, 4000), rng.integers(0, W, 4000)] = rng.integers(0, 2, 4000)       # holes: 0 and 1
        frames.append((bgr, depth))
    resident = []
    for k in range(3):
        d.upload_frame(k, *frames[k]); resident.append(frames[k][1])
    sx, sy = 12, -9
    d.upload_frame_shifted(3, frames[3][0], frames[3][1], sx, sy)
    tr = np.zeros_like(frames[3][1])
    ys, xs = max(sy, 0), max(sx, 0)
    tr[ys:H + min(sy, 0), xs:W + min(sx, 0)] = frames[3][1][max(-sy, 0):H - max(sy, 0), max(-sx, 0):W - max(sx, 0)]
    resident.append(tr)
    d.upload_wait(-1)
    n = 3000
    q = np.zeros(n, lm.DEPTH_QUERY_DTYPE)
    q["slot"] = rng.integers(0, 4, n)
    q["x0"] = rng.integers(0, W - 1, n); q["y0"] = rng.integers(0, H - 1, n)
    q["x1"] = np.minimum(q["x0"] + rng.integers(0, 300, n), W); q["y1"] = np.minimum(q["y0"] + rng.integers(0, 260, n), H)
    q["lo"] = rng.integers(0, 1400, n); q["hi"] = np.minimum(q["lo"] + rng.integers(0, 400, n), 65535)
    q["lo"][::17] = 65535; q["hi"][::17] = 65535                     # the holes' own value
    q["lo"][::19] = 0
    q[5]["x1"] = q[5]["x0"]                                           # an empty crop
    q[6] = (0, 0, W, H, 600, 900, 2, 0)                               # the whole frame
    # beside a busy lane: a match of the slots 4..5 is in flight while the counts of slots 0..3 are taken
    d.upload_frame(4, *frames[0]); d.upload_frame(5, *frames[1])
    descs, feats, _ = synth.make_bank(50, 2, 2, seed=3, frame_size=(W, H), T0=5)
    d.add_class("c", descs, feats)
    d.match_begin(1, 4, 2, 85.0)
    below, inside = d.depth_counts(q)
    d.match_end(1, 4096, n_slots=2)
    for i in range(n):
        crop = resident[q[i]["slot"]][q[i]["y0"]:q[i]["y1"], q[i]["x0"]:q[i]["x1"]].astype(np.int64)
        t = np.where(crop <= 1, 65535, crop)
        assert below[i] == int((t < q[i]["lo"]).sum()) and inside[i] == int(((t >= q[i]["lo"]) & (t <= q[i]["hi"])).sum()), (i, q[i])
    assert below.sum() > 0 and inside.sum() > 0
    b0, i0 = d.depth_counts(q[:0])
    assert len(b0) == 0
    bad = q[:2].copy(); bad[1]["x1"] = W + 1
    with pytest.raises(lm.LinemodError):
        d.depth_counts(bad)
    bad = q[:2].copy(); bad[0]["slot"] = 6
    with pytest.raises(lm.LinemodError):
        d.depth_counts(bad)
    below2, inside2 = d.depth_counts(q)                               # (a refused call leaves nothing in flight)
    assert np.array_equal(below2, below) and np.array_equal(inside2, inside)
    d.close()
    c = lm.Detector(lm.default_config(color_only=True, width=W, height=H, frame_slots=2))
    c.upload_frame(0, frames[0][0], None)
    with pytest.raises(lm.LinemodError):
        c.depth_counts(q[:4])                                         # no depth frame on the device
    c.close()
