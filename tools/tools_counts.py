"""Scratch: candidates of the scan vs matches that survive the refinement, per frame of the bench workload."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, B = 640, 480, 32
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(B)]
d.upload_frame(0, *frames[0]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
descs, feats, _ = synth.make_bank(3000, 2, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
for i, (b, dp) in enumerate(frames):
    d.upload_frame(i, b, dp)
out, cnt = d.match_batch(B, 80.0, 0)
c = np.array([d.last_counts(i) for i in range(B)])
print("candidates per frame:", c[:, 0].tolist())
print("matches before unique:", c[:, 1].tolist())
print("unique matches:", cnt.tolist())
print("mean candidates %.1f, surviving %.1f, unique %.1f" % (c[:, 0].mean(), c[:, 1].mean(), cnt.mean()))
