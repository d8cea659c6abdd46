"""Scratch: sort / refine stage time with and without the crowded frame 0 in the batch (one lane, 128 frames)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, M, B = 640, 480, 2, 128
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(B + 1)]
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B))
d.upload_frame(0, *frames[0]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
for start in (0, 1):
    for i in range(B):
        d.upload_frame(i, *frames[start + i])
    for _ in range(5): d.match_batch(B, 80.0, 0)
    d.set_profiling(True)
    for _ in range(30): out, cnt = d.match_batch(B, 80.0, 0)
    p = d.get_profile(); d.set_profiling(False)
    print("frames %d..%d: stages/frame %s  max matches %d" % (start, start + B - 1, [round(v / p["frames"], 2) for v in p["stage_us"]], cnt.max()))
