"""Scratch: throughput of the reference's shipped configuration (colour-only, T = {2, 8}) on two lanes."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, BT, NL = 640, 480, 256, 2
B = BT // NL
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(BT)]
d = lm.Detector(lm.default_config(color_only=True, width=W, height=H, frame_slots=BT))
d.upload_frame(0, frames[0][0], None); d.prepare_slot(0)
q = {(l, 0): d.debug_read(0, 0, l, 0).reshape(H >> l, W >> l) for l in range(2)}
descs, feats, crops = synth.make_bank(3000, 1, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=2)
d.add_class("c", descs, feats)
for i in range(BT):
    d.upload_frame(i, frames[i][0], None)
outs = [(np.zeros((B, 4096), lm.MATCH_DTYPE), np.zeros(B, np.int32)) for _ in range(NL)]
def run(n):
    for l in range(NL):
        d.match_begin(l, l * B, B, 80.0, 0)
    for k in range(n):
        for l in range(NL):
            d.match_end(l, 4096, out=outs[l][0], counts=outs[l][1])
            if k + 1 < n:
                d.match_begin(l, l * B, B, 80.0, 0)
run(5)
d.set_profiling(True)
t0 = time.perf_counter(); run(50); dt = time.perf_counter() - t0
p = d.get_profile()
print("colour-only: %.1f detections/s (%.2f us/frame) stages/frame %s matches0 %d cands %s" % (
    BT * 50 / dt, dt / 50 / BT * 1e6, [round(v / p["frames"], 2) for v in p["stage_us"]], outs[0][1][0], d.last_counts(0)))
