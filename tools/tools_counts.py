"""Scratch: candidate / match counts and level-0 feature counts of the bench workload."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
W, H, M, B = 640, 480, 2, 8
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(B)]
q = quantized_from_gpu(d, frames[0][0], frames[0][1], M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
print("features per (level, modality) of template 0:", [int(descs[k]["num_features"]) for k in range(4)], "template 1:", [int(descs[4 + k]["num_features"]) for k in range(4)])
for i, (b, dp) in enumerate(frames):
    d.upload_frame(i, b, dp)
out, counts = d.match_batch(B, 80.0)
for i in range(B):
    print("slot", i, "candidates, matches before unique:", d.last_counts(i), "final", counts[i])
