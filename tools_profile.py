"""Scratch: run N frames of config 2 for rocprofv3 kernel traces."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
size=(640,480); M=2
d = lm.Detector(color_only=False, width=size[0], height=size[1])
bgr, depth = synth.make_frame(size[0], size[1], seed=1234)
q = quantized_from_gpu(d, bgr, depth, M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96,96), quantized=q, crop_fraction=0.1, frame_size=size, T0=5)
d.add_class("c", descs, feats)
d.upload_frame(0, bgr, depth)
for _ in range(30): m = d.match_slot(0, 80.0)
print(len(m))
