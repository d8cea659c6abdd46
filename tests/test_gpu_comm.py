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


def test_gather_capacity_overflow_is_an_error(lm, orc, synth):
    d, o, frames = _detector(lm, orc, synth)
    thr = 30.0                                           # hundreds of matches per frame
    exp = [o.match(b, dp, thr, 0, threads=8) for b, dp in frames[:2]]
    assert len(exp[0]) + len(exp[1]) > 16
    d.comm_init(0, 1, "127.0.0.1", _free_port(), recs_per_frame_cap=8)
    out = np.zeros(1 << 16, lm.MATCH_DTYPE)
    cnt = np.zeros(8, np.int32)
    d.match_begin_gathered(0, 0, 2, thr, 0)
    with pytest.raises(lm.LinemodError) as e:
        d.match_end_gathered(0, out, cnt)
    assert e.value.code == lm.LM_ERR_OVERFLOW
    d.comm_destroy()
    d.close()
