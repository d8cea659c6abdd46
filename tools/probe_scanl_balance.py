"""r06 probe: is k_scanl's second stage slow because of ONE heavy frame (the frame the bank's crop templates were cut from)?  All frames distinct;
the scan launch over slots [0, 88) (with frame 0) against [8, 96) (without)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, M, NB = 640, 480, 2, 96
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=NB))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(NB)]
d.upload_frame(0, frames[0][0], frames[0][1]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
d.add_class("c", descs, feats)
for i in range(NB):
    d.upload_frame(i, *frames[i])
thr = 80.0
for form in (3,):
    d.set_tuning(lm.TUNE_SCAN_FORM, form)
    out, cnt = d.match_batch(NB, thr, cap_per_frame=4096)
    cands = [d.last_counts(i)[0] for i in range(NB)]
    print("form", form, "candidates per frame: frame 0 =", cands[0], " others mean %.0f max %d" % (np.mean(cands[1:]), max(cands[1:])), flush=True)
    for first, n in ((0, 88), (8, 88), (0, 96)):
        res = {v: min(d.time_scan_batch(first, n, thr, iters=20, variant=v) for _ in range(3)) for v in ((0,) if form == 1 else (0, 128))}
        print("   slots [%d, %d): %s us" % (first, first + n, {k: round(v, 1) for k, v in res.items()}), flush=True)
    if form == 3:
        d.set_scan_stats(True)
        d.time_scan_batch(0, 96, thr, iters=1, variant=7 << 9)      # (2 warm-ups + 1 timed launch: 3 launches x 768 workgroups x 16 waves)
        t0, t1 = d.get_scan_stats(); t2 = d.get_scan_lane_stats()[0]; t3 = d.get_scan_form_stats()[2]
        nw = 3 * 96 * int(os.environ.get('LM_SCANL_R', '8')) * 16
        print("   per wave, us at 100 MHz: planes copy %.2f  counting %.2f  barrier wait %.2f  spread copy %.2f  exact sums %.2f" % (
            t0 / nw / 100, t1 / nw / 100, (t2 & 0xFFFFFFFF) / nw / 100, (t2 >> 32) / nw / 100, t3 / nw / 100))
        d.set_scan_stats(False)
        d.set_scan_stats(True); d.match_prepared(0, 1, thr, [-1], cap_per_frame=4096); s0 = d.get_scan_form_stats()[2]; d.set_scan_stats(False)
        d.set_scan_stats(True); d.match_prepared(1, 8, thr, [-1], cap_per_frame=4096); s1 = d.get_scan_form_stats()[2] / 8; d.set_scan_stats(False)
        print("   survivors: frame 0 %d, frames 1..8 mean %.0f" % (s0, s1))
d.close()
