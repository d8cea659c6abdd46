"""Scratch: scan-kernel variant sweep in batch mode via the live profile."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
size=(640,480); M=2
NB = 32
d = lm.Detector(lm.default_config(color_only=False, width=size[0], height=size[1], frame_slots=NB))
bgr, depth = synth.make_frame(size[0], size[1], seed=1234)
q = quantized_from_gpu(d, bgr, depth, M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96,96), quantized=q, crop_fraction=0.1, frame_size=size, T0=5)
d.add_class("c", descs, feats)
for i in range(NB):
    b, dp = synth.make_frame(size[0], size[1], seed=1234 + i)
    d.upload_frame(i, b, dp)
ref = None
for variant in (0, 1, 2):
    d.set_scan_variant(variant)
    for B in (8, 32):
        for _ in range(5): out, counts = d.match_batch(B, 80.0)
        d.set_profiling(True)
        t=time.time()
        for _ in range(30): out, counts = d.match_batch(B, 80.0)
        dt=time.time()-t
        p = d.get_profile(); d.set_profiling(False)
        print("variant %d B %2d: scan %.1f us/frame (%.2f TB/s)  stages/frame %s  wall %.1f us/frame  matches0 %d" % (
            variant, B, p["stage_us"][1]/p["frames"], p["scan_bytes"]/p["stage_us"][1]/1e6,
            [round(v/p["frames"],1) for v in p["stage_us"]], dt/30/B*1e6, counts[0]))
