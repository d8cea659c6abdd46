"""Scratch: per-XCD candidate totals of the refine kernel with the fixed slot -> XCD mapping vs a balanced one."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, M, B = 640, 480, 2, 256
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(B)]
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B))
d.upload_frame(0, *frames[0]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
for i in range(B):
    d.upload_frame(i, *frames[i])
d.match_batch(B, 80.0, 0)
c = np.array([d.last_counts(i)[0] for i in range(B)])
print("candidates per slot: mean %.0f median %.0f max %d  p90 %.0f" % (c.mean(), np.median(c), c.max(), np.percentile(c, 90)))
for lane in range(2):
    cl = c[lane * 128:(lane + 1) * 128]
    per_xcd = np.array([cl[x::8].sum() for x in range(8)])
    print("lane %d: per-XCD totals %s  max/mean %.2f" % (lane, per_xcd.tolist(), per_xcd.max() / per_xcd.mean()))
