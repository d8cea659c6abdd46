"""r06 probe: the scan launch alone on config 2's workload (640x480 RGB-D, 3000 fixed-geometry templates, threshold 80) for several batch
sizes and both scan forms: k_scan4 (form 1), k_scan1 with and without its second stage (form 2; lm_time_scan_batch variant 0 / 128).
usage: python tools/probe_scan1.py [config 2|3|5] [n_slots ...]"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sizes = [int(a) for a in sys.argv[2:]] or [96]
if cfgn == 2:
    W, H, M, seedf, seedb, l0 = 640, 480, 2, 1234, 4321, (96, 96)
else:
    W, H, M, seedf, seedb, l0 = 1280, 960, 1, 2234, 77, (192, 192)
NB = max(sizes)
d = lm.Detector(lm.default_config(color_only=(M == 1), width=W, height=H, frame_slots=NB))
frames = [synth.make_frame(W, H, seed=seedf + i) for i in range(min(NB, 32))]
d.upload_frame(0, frames[0][0], frames[0][1] if M == 2 else None); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, _ = synth.make_bank(3000, M, 2, seed=seedb, fixed_l0_size=l0, quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
d.add_class("c", descs, feats)
for i in range(NB):
    b, dp = frames[i % len(frames)]
    d.upload_frame(i, b, dp if M == 2 else None)
thr = 80.0
for form in (1, 2, 3):
    d.set_tuning(lm.TUNE_SCAN_FORM, form)
    for n in sizes:
        out, cnt = d.match_batch(n, thr, cap_per_frame=4096)
        st = d.get_scan_form_stats()
        d.set_scan_stats(True)
        d.match_prepared(0, n, thr, [-1], cap_per_frame=4096)
        kept = d.get_scan_stats(); lanes = d.get_scan_lane_stats(); fs = d.get_scan_form_stats()
        d.set_scan_stats(False)
        res = {}
        for v in ((0,) if form == 1 else (0, 128) if form == 2 else (0, 128, 512, 1024, 128 | 1536, 128 | 2048)):
            res[v] = min(d.time_scan_batch(0, n, thr, iters=20, variant=v) for _ in range(3))
        print("config %d form %d n %3d L1 %2d | scan %s us per launch = %s us per frame | kept %.3f lanes %.3f | survivors/frame %.0f | matches0 %d" % (
            cfgn, form, n, st[3], {k: round(v, 1) for k, v in res.items()}, {k: round(v / n, 3) for k, v in res.items()},
            kept[0] / max(kept[1], 1), lanes[0] / max(lanes[1], 1), fs[2] / n if form >= 2 else 0, cnt[0]), flush=True)
d.close()
