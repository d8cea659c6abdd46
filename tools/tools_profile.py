"""Scratch: run batches of config 2 for rocprofv3 kernel traces.  usage: tools_profile.py [batch] [iters]"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
size=(640,480); M=2
d = lm.Detector(lm.default_config(color_only=False, width=size[0], height=size[1], frame_slots=max(B, 8)))
bgr, depth = synth.make_frame(size[0], size[1], seed=1234)
q = quantized_from_gpu(d, bgr, depth, M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96,96), quantized=q, crop_fraction=0.1, frame_size=size, T0=5)
d.add_class("c", descs, feats)
for i in range(B):
    b, dp = synth.make_frame(size[0], size[1], seed=1234 + i)
    d.upload_frame(i, b, dp)
for _ in range(iters): out, counts = d.match_batch(B, 80.0)
print(counts)
