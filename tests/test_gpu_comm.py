"""The RCCL exchange of the C ABI (lm_comm_init, lm_match_begin_gathered / lm_match_end_gathered) on what
a 1-GPU box can run: a single-rank communicator.  The gathered lists must equal lm_match_batch's (a one-shard "merge"
is the identity), overflow of the fixed gather capacity must be an error, never a truncation."""
import socket

import numpy as np
import pytest

from conftest import assert_matches_equal

pytestmark = pytest.mark.gpu
W, H = 640, 480


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _detector(lm, orc, synth, slots=8, n_templates=200):
    d = lm.Detector(color_only=False, width=W, height=H, frame_slots=slots)
    o = orc.Detector(color_only=False)
    frames = [synth.make_frame(W, H, seed=40 + i) for i in range(slots)]
    o.prepare(*frames[0])
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(n_templates, 2, 2, seed=3, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=5)
    d.add_class("c", descs, feats)
    o.add_class("c", descs, feats)
    for i, (b, dp) in enumerate(frames):
        d.upload_frame(i, b, dp)
    return d, o, frames


def test_gathered_equals_batch_on_single_rank_communicator(lm, orc, synth):
    d, o, frames = _detector(lm, orc, synth)
    thr = 65.0
    exp = [o.match(b, dp, thr, 0, threads=8) for b, dp in frames]
    assert sum(len(e) for e in exp) > 8
    d.comm_init(0, 1, "127.0.0.1", _free_port())
    out = np.zeros(1 << 16, lm.MATCH_DTYPE)
    cnt = np.zeros(8, np.int32)
    for rnd in range(3):
        # both lanes in flight, each with its own communicator
        d.match_begin_gathered(0, 0, 4, thr, 0)
        d.match_begin_gathered(1, 4, 4, thr, 0)
        for lane in (0, 1):
            f0, nf, tot = d.match_end_gathered(lane, out, cnt)
            assert (f0, nf) == (0, 4) and tot == cnt[:4].sum()
            pos = 0
            for i in range(4):
                assert_matches_equal(out[pos:pos + cnt[i]], exp[4 * lane + i])
                pos += cnt[i]
    # an ungathered call on the same detector still works, and the two kinds of end cannot be mixed up
    d.match_begin_gathered(0, 0, 4, thr, 0)
    with pytest.raises(lm.LinemodError):
        d.match_end(0, n_slots=4)
    d.match_end_gathered(0, out, cnt)
    got, c = d.match_batch(8, thr, 0)
    for i in range(8):
        assert_matches_equal(got[i, :c[i]], exp[i])
    assert d.comm_max([3.5, -1.0]) == [3.5, -1.0]
    d.comm_barrier()
    d.comm_destroy()
    d.close()


def _gathered_lists(d, lm, lane, first, n, thr, cap=1 << 20):
    out = np.zeros(cap, lm.MATCH_DTYPE)
    cnt = np.zeros(n, np.int32)
    d.match_begin_gathered(lane, first, n, thr, 0)
    f0, nf, tot = d.match_end_gathered(lane, out, cnt)
    assert (f0, nf) == (0, n) and tot == cnt.sum()
    ends = np.cumsum(cnt)
    return [out[e - c:e].copy() for e, c in zip(ends, cnt)]


def test_gather_capacity_overflow_takes_the_sized_exchange(lm, orc, synth):
    """More records than the fixed gather capacity: the gathered path runs its second, exactly sized exchange and returns
    what lm_match_batch returns (the reference consumes ALL matches, HighLevelLinemod.cpp:206-253) -- never an error,
    never a truncation."""
    d, o, frames = _detector(lm, orc, synth)
    thr = 30.0                                           # hundreds of matches per frame
    exp = [o.match(b, dp, thr, 0, threads=8) for b, dp in frames[:2]]
    assert len(exp[0]) + len(exp[1]) > 16
    d.comm_init(0, 1, "127.0.0.1", _free_port(), recs_per_frame_cap=8)
    for rnd in range(2):
        got = _gathered_lists(d, lm, rnd, 0, 2, thr)
        for g, e in zip(got, exp):
            assert_matches_equal(g, e)
    ref, c = d.match_batch(2, thr, 0, cap_per_frame=1 << 15)
    for i in range(2):
        assert_matches_equal(got[i], ref[i, :c[i]])
    # the fast path still works afterwards on the same communicator
    got = _gathered_lists(d, lm, 0, 0, 2, 95.0)
    for i in range(2):
        assert_matches_equal(got[i], o.match(*frames[i], 95.0, 0, threads=8))
    d.comm_destroy()
    d.close()


def test_gathered_host_sorted_frames(lm, orc, synth):
    """Threshold 0: far more than the 4096 matches the device sorts per frame; lm_match_batch hands such frames to the
    host sort, and the gathered path must deliver the same lists (VERDICT r2 missing #5)."""
    d, o, frames = _detector(lm, orc, synth, slots=8, n_templates=120)
    ref, c = d.match_batch(3, 0.0, 0, cap_per_frame=1 << 17)
    assert c.max() > 4096
    exp = o.match(*frames[0], 0.0, 0, threads=8)
    assert_matches_equal(ref[0, :c[0]], exp)
    d.comm_init(0, 1, "127.0.0.1", _free_port())
    got = _gathered_lists(d, lm, 1, 0, 3, 0.0, cap=1 << 19)
    for i in range(3):
        assert_matches_equal(got[i], ref[i, :c[i]])
    d.comm_destroy()
    d.close()


def test_comm_init_failure_leaves_nothing_behind(lm, orc, synth):
    """a failed lm_comm_init must be repeatable and must not leave a communicator without buffers."""
    d, o, frames = _detector(lm, orc, synth, slots=8, n_templates=20)
    with pytest.raises(lm.LinemodError):
        d.comm_init(3, 2, "127.0.0.1", _free_port())      # rank outside the world
    with pytest.raises(lm.LinemodError):
        d.match_begin_gathered(0, 0, 2, 80.0, 0)          # no communicator: refused, nothing dereferenced
    d.comm_init(0, 1, "127.0.0.1", _free_port())
    assert d.comm_info() == (0, 1)
    _gathered_lists(d, lm, 0, 0, 2, 80.0)
    d.comm_destroy()
    d.close()


@pytest.mark.parametrize("color_only,form,slots", [(True, 0, 16), (True, 2, 8), (False, 3, 8), (False, 0, 32)])
def test_gathered_call_through_the_bit_plane_scans(lm, orc, synth, color_only, form, slots):
    """r06: the gathered path (k_pack_lists + 2 x ncclAllGather on a single-rank communicator) behind the bit-plane scans -- a colour-only
    detector whose 16-frame call takes k_scan1 by cost (8 frames: forced), an RGB-D detector under the LDS-resident form k_scanl (forced on 8 frames, by cost on 32):
    the gathered lists equal lm_match_batch's and the oracle's."""
    d = lm.Detector(color_only=color_only, width=W, height=H, frame_slots=slots)
    o = orc.Detector(color_only=color_only)
    frames = [synth.make_frame(W, H, seed=140 + i) for i in range(4)]
    dep = lambda k: None if color_only else frames[k % 4][1]
    o.prepare(frames[0][0], dep(0))
    M = 1 if color_only else 2
    q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(150, M, 2, seed=5, quantized=q, crop_fraction=0.3, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    for i in range(slots):
        d.upload_frame(i, frames[i % 4][0], dep(i))
    thr = 82.0
    exp = [o.match(frames[k][0], dep(k), thr, 0, threads=8) for k in range(4)]
    assert sum(len(e) for e in exp) > 4
    d.set_tuning(lm.TUNE_SCAN_FORM, form)
    d.comm_init(0, 1, "127.0.0.1", _free_port())
    before = d.get_scan_form_stats()
    got = _gathered_lists(d, lm, 0, 0, slots, thr)
    after = d.get_scan_form_stats()
    assert after[0] - before[0] >= 1                                    # the launch was a bit-plane scan ...
    assert (after[3] >= 1000) == (not color_only)                       # ... k_scanl for the RGB-D detector, k_scan1 for the colour-only one
    for i in range(slots):
        assert_matches_equal(got[i], exp[i % 4])
    ref, c = d.match_batch(slots, thr, 0, cap_per_frame=1 << 14)
    for i in range(slots):
        assert_matches_equal(ref[i, :c[i]], got[i])
    d.comm_destroy()
    d.close()
