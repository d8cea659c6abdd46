"""Scratch: upper bound of what removing the scan's shift-undo (4 x v_alignbit + DPP per feature) could buy.
The exhaustive scan of a 96-frame launch is timed with and without those instructions (the second gives WRONG sums: same
loads, same adds, no realignment) -- what pre-shifted copies of the linear memories would save at best, before their
8 x footprint costs anything.  Config 2 and config 3 workloads of bench.py."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
for name, W, H, color_only, l0, n in (("config 2", 640, 480, False, (96, 96), 96), ("config 3", 1280, 960, True, (192, 192), 128)):
    M = 1 if color_only else 2
    d = lm.Detector(lm.default_config(color_only=color_only, width=W, height=H, frame_slots=n))
    fr = [synth.make_frame(W, H, seed=1234 + i) for i in range(8)]
    d.upload_frame(0, fr[0][0], None if color_only else fr[0][1]); d.prepare_slot(0)
    q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=l0, quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    for i in range(n):
        b, dp = fr[i % 8]
        d.upload_frame(i, b, None if color_only else dp)
    d.match_batch(n, 80.0, 0)                      # prepares every slot
    res = {}
    for label, variant in (("pruned (default)", 0), ("exhaustive", 8), ("exhaustive, no shift-undo (wrong sums)", 8 | 64)):
        res[label] = min(d.time_scan_batch(0, n, 80.0, 0, iters=10, variant=variant) for _ in range(3))
        print("%s, %d frames per launch: %-40s %8.1f us per launch" % (name, n, label, res[label]))
    d.close()
