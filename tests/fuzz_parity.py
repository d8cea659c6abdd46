"""Fuzzer (kept under tests/ because it drives the oracle; not collected by pytest): stage kernels and whole matches against the oracle on random shapes and
contents.  usage: python tests/fuzz_parity.py [seconds] [seed]"""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
det = lm.Detector(color_only=False)
t0 = time.time()
n_stage = n_match = n_scan = n_batch = 0


def rand_bgr(h, w):
    k = rng.integers(0, 4)
    if k == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if k == 1:   # blocks with sharp edges
        b = rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 3), dtype=np.uint8)
        return np.ascontiguousarray(np.kron(b, np.ones((8, 8, 1), np.uint8))[:h, :w])
    if k == 2:   # extremes
        return (rng.integers(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    f, _ = synth.make_frame(max(w, 64), max(h, 64), seed=int(rng.integers(1 << 30)))
    return np.ascontiguousarray(f[:h, :w])


def rand_depth(h, w):
    k = rng.integers(0, 3)
    if k == 0:
        return rng.integers(0, 2600, (h, w)).astype(np.uint16)
    if k == 1:
        return (600 + 49 * rng.integers(0, 3, (h, w))).astype(np.uint16)
    _, d = synth.make_frame(max(w, 64), max(h, 64), seed=int(rng.integers(1 << 30)))
    return np.ascontiguousarray(d[:h, :w])


while time.time() - t0 < budget * 0.5:
    w = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 160, 192, 320, 640, 24, 40, 56, 37, 53, 70]))
    h = int(rng.integers(8, 130))
    bgr = rand_bgr(h, w)
    thr = float(rng.choice([10.0, 0.0, 30.0, 200.0]))
    kb, kg = int(rng.choice([0, 1, 3, 4])), int(rng.integers(0, 4))      # blur / gradient kernels: by batch size, few-frame, batch; blur 4 = matrix cores
    det.set_tuning(lm.TUNE_CBLUR_VARIANT, kb); det.set_tuning(lm.TUNE_CGRAD_VARIANT, kg)
    assert np.array_equal(det.stage_color_quantize(bgr, thr), orc.color_quantize(bgr, thr)), ("colour", h, w, thr, kb, kg)
    if h % 2 == 0 and w % 2 == 0 and h >= 4:
        kp = int(rng.integers(0, 3))
        det.set_tuning(lm.TUNE_PYRDOWN_VARIANT, kp)
        assert np.array_equal(det.stage_pyrdown(bgr), orc.pyrdown(bgr)), ("pyrdown", h, w, kp)
    dep = rand_depth(h, w)
    det.set_tuning(lm.TUNE_DMEDIAN_VARIANT, int(rng.integers(0, 3)))   # median: 4 or 16 output rows per lane
    assert np.array_equal(det.stage_depth_quantize(dep), orc.depth_quantize(dep)), ("depth", h, w)
    n_stage += 1
det.set_tuning(lm.TUNE_CBLUR_VARIANT, 0); det.set_tuning(lm.TUNE_CGRAD_VARIANT, 0); det.set_tuning(lm.TUNE_PYRDOWN_VARIANT, 0)
det.set_tuning(lm.TUNE_DMEDIAN_VARIANT, 0)
det.close()

while time.time() - t0 < budget:
    color_only = bool(rng.integers(0, 2))
    T = [[5, 8], [2, 8], [4, 4], [8, 8]][int(rng.integers(0, 4))] if not color_only else [[2, 8], [4, 8], [2, 4]][int(rng.integers(0, 3))]
    unit = int(np.lcm.reduce([T[0], 2 * T[1], 8]))
    w, h = unit * int(rng.integers(1, 1 + 640 // unit)), unit * int(rng.integers(1, 1 + 480 // unit))
    w, h = max(w, 4 * unit), max(h, 4 * unit)
    try:
        nb = int(rng.choice([1, 1, 3, 8, 16, 19]))               # frames per call: phase launches below 16, batch kernels from 16
        d = lm.Detector(lm.default_config(color_only=color_only, width=w, height=h, T=T, flags=int(rng.integers(0, 2)),
                                          frame_slots=max(nb, 2)))
    except lm.LinemodError:
        continue
    o = orc.Detector(color_only=color_only, T=T)
    bgr, depth = synth.make_frame(w, h, seed=int(rng.integers(1 << 30)))
    o.prepare(bgr, None if color_only else depth)
    M = 1 if color_only else 2
    q = {(l, m): o.stage(0, l, m).reshape(h >> l, w >> l) for l in range(2) for m in range(M)}
    n = int(rng.integers(5, 120))
    try:
        nfeat = int(rng.choice([63, 63, 63, 42, 21, 9]))                # r06: also tail rounds of the bit-plane scans (63 is upstream's maximum per template)
        descs, feats, _ = synth.make_bank(n, M, 2, seed=int(rng.integers(1 << 30)), quantized=q, crop_fraction=0.3,
                                          frame_size=(w, h), T0=T[0], num_features=nfeat)
    except Exception:
        d.close(); continue
    d.add_class("c", descs, feats); o.add_class("c", descs, feats)
    thr = float(rng.choice([60.0, 75.0, 85.0, 40.0]))
    variant = int(rng.choice([0, 0, 32, 8, 1, 34, 9, 16, 17, 18]))   # load-block sizes; per-lane pruning (default), none (bit 3), wave-level (bit 4)
    d.set_scan_variant(variant)
    d.set_tuning(lm.TUNE_SCAN_FORM, int(rng.choice([0, 1, 2, 3, 3])))        # r05 / r06: nibble scan / bit-plane scan / bit-plane scan with the planes in LDS (by cost, forced)
    got = d.match(bgr, None if color_only else depth, thr, cap=1 << 18)
    exp = o.match(bgr, None if color_only else depth, thr, threads=8, cap=1 << 18)
    assert got.tobytes() == exp.tobytes(), ("match", color_only, T, w, h, n, thr, variant, len(got), len(exp))
    if nb > 1:
        d.set_tuning(lm.TUNE_PHASE_MAX_SLOTS, int(rng.choice([15, 0])))
        d.set_tuning(lm.TUNE_BATCH_PHASES, int(rng.choice([0, 1, 2])))     # 16+ frames: level-fused batch launches or one per kernel
        d.set_tuning(lm.TUNE_BLUR_PYR, int(rng.choice([0, 1, 2, 3])))         # level-0 blur + pyrDown apart, in one launch back to back, or dealt out evenly
        d.set_tuning(lm.TUNE_BLUR_STRIP, int(rng.choice([0, 16, 32, 64])))      # rows per blur strip inside k_blur_pyr
        d.set_tuning(lm.TUNE_CGRAD_LEVELS, int(rng.choice([0, 1])))             # r06: the two levels' gradients of a batch in one launch or two
        d.set_tuning(lm.TUNE_SCAN_LIST_ORDER, int(rng.choice([0, 1, 2, 3])))    # order of the scan's feature lists (same sums)
        d.set_tuning(lm.TUNE_SCAN_FORM, int(rng.choice([0, 1, 2, 3, 3])))
        d.set_tuning(lm.TUNE_SURVIVOR_QUEUE, int(rng.choice([1 << 20, 1 << 20, 64, 200, 4096])))   # r06: k_scan1's survivor queues, also far too small (partial fits, the waves' own sums)
        # r05: every way a frame reaches a slot -- the one-call upload, the staged upload (rows in random pieces; a zero shift here, the
        # shifted forms are swept in tests/test_gpu_stream.py) -- and the lists once more through lm_match_collect
        staged = bool(rng.integers(0, 2))
        if staged:
            d.stage_reserve(0, nb)
        for k in range(nb):
            if staged:
                cut = sorted(set([0, h] + [int(v) for v in rng.integers(0, h + 1, 3)]))
                for a, b in zip(cut[:-1], cut[1:]):
                    d.stage_rows(k, bgr, None if color_only else depth, 0, 0, a, b)
                d.upload_staged(k)
            else:
                d.upload_frame(k, bgr, None if color_only else depth)
        outb, cntb = d.match_batch(nb, thr, 0, cap_per_frame=max(len(exp), 1))
        outc2, cntc2 = d.match_collect(0, nb, cap_per_frame=max(len(exp), 1))
        assert np.array_equal(cntb, cntc2) and outb.tobytes() == outc2.tobytes(), ("collect", color_only, T, w, h, n, thr, nb)
        for k in range(nb):
            assert cntb[k] == len(exp) and outb[k, :cntb[k]].tobytes() == exp.tobytes(), ("batch", color_only, T, w, h, n, thr, nb, k)
        n_batch += 1
    if n_match % 4 == 0:
        # a second class: Detector::match with a class list = the merged per-class lists, one pre-processing (lm_match_batch_classes),
        # and the prepared slot again per class (lm_match_prepared)
        try:
            d2, f2, _ = synth.make_bank(max(n // 2, 3), M, 2, seed=int(rng.integers(1 << 30)), quantized=q, crop_fraction=0.3, frame_size=(w, h), T0=T[0])
        except Exception:
            d.close(); continue
        d.add_class("c2", d2, f2); o.add_class("c2", d2, f2)
        d.upload_frame(0, bgr, None if color_only else depth)
        exp2 = o.match(bgr, None if color_only else depth, thr, class_idx=-1, threads=8, cap=1 << 18)
        outc, cntc = d.match_batch_classes(0, 1, thr, [1, 0], cap_per_frame=max(len(exp2), 1))
        assert outc[0, :cntc[0]].tobytes() == exp2.tobytes(), ("class list", color_only, T, w, h, n, thr)
        for c in (0, 1):
            outp, cntp = d.match_prepared(0, 1, thr, [c], cap_per_frame=max(len(exp2), 1))
            # (std::unique removes ADJACENT duplicates: in the mixed list a template of the other class can sit between two
            # equal matches of this class, so the class's own list is the filtered mixed list made unique once more)
            sub = exp2[exp2["class_idx"] == c]
            keep = np.ones(len(sub), bool)
            if len(sub) > 1:
                same = (sub["x"][1:] == sub["x"][:-1]) & (sub["y"][1:] == sub["y"][:-1]) & (sub["similarity"][1:] == sub["similarity"][:-1])
                keep[1:] = ~same
            expc = o.match(bgr, None if color_only else depth, thr, class_idx=c, threads=8, cap=1 << 18)
            assert sub[keep].tobytes() == expc.tobytes(), ("filter + unique", color_only, T, w, h, n, thr, c)
            assert outp[0, :cntp[0]].tobytes() == expc.tobytes(), ("prepared", color_only, T, w, h, n, thr, c)
    # a11-a13 alone: the scan kernel's candidate list, record by record
    d.upload_frame(1, bgr, None if color_only else depth)
    d.prepare_slot(1)
    assert np.array_equal(d.stage_scan(1, thr, 0), o.scan_candidates(thr, 0, threads=8)), ("scan", color_only, T, w, h, n, thr, variant)
    d.close()
    n_match += 1
    n_scan += 1
print("fuzz ok: %d stage rounds, %d whole matches, %d scan candidate lists, %d multi-frame calls in %.0f s (seed %s)" % (
    n_stage, n_match, n_scan, n_batch, time.time() - t0, sys.argv[2] if len(sys.argv) > 2 else "1"))
