"""r06: repeat a 96-frame config-2 batch (variable geometry) under every scan form and diff every frame's list with the oracle's: hunts an intermittent
difference (1 match of 562 missing once in ~6 runs of tests/test_gpu_fullsize.py::test_config2_batch_of_96_frames_at_stated_size)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from oracle import oracle as orc
W, H, M, NB = 640, 480, 2, 96
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
forms = [int(a) for a in sys.argv[2:]] or [0, 1, 2]
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(4)]
d = lm.Detector(color_only=False, width=W, height=H, frame_slots=NB)
o = orc.Detector(color_only=False)
o.prepare(frames[0][0], frames[0][1])
q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats); o.add_class("c", descs, feats)
exp = [o.match(frames[k][0], frames[k][1], 80.0, 0, threads=16, cap=1 << 18) for k in range(4)]
expc = []
for k in range(4):
    o.prepare(frames[k][0], frames[k][1]); expc.append(o.scan_candidates(80.0, 0, threads=16))
for k in range(NB):
    d.upload_frame(k, frames[k % 4][0], frames[k % 4][1])
d.upload_wait(-1)
for form in forms:
    d.set_tuning(lm.TUNE_SCAN_FORM, form)
    bad = 0
    for rep in range(reps):
        got, cnt = d.match_batch(NB, 80.0, 0, cap_per_frame=1 << 15)
        for k in range(NB):
            g = got[k, :cnt[k]]; e = exp[k % 4]
            if len(g) != len(e) or g.tobytes() != e.tobytes():
                bad += 1
                gs = set(map(tuple, g[["x", "y", "template_id"]].tolist())); es = set(map(tuple, e[["x", "y", "template_id"]].tolist()))
                print("form %d rep %d frame %d (slot %d): %d vs %d; missing %s extra %s" % (form, rep, k % 4, k, len(g), len(e), sorted(es - gs)[:4], sorted(gs - es)[:4]), flush=True)
                # is the scan's candidate list complete for that slot?
                c = d.stage_scan(k, 80.0, 0)
                print("      stage_scan of the slot now: %d candidates, oracle %d, equal %s" % (len(c), len(expc[k % 4]), np.array_equal(c, expc[k % 4])), flush=True)
    print("form %d: %d repetitions x %d frames, %d lists differ (scan form stats %s)" % (form, reps, NB, bad, d.get_scan_form_stats()), flush=True)
d.close()
