"""Scratch: does running two detectors (two streams, two host threads) on one GPU overlap the stages?"""
import importlib, sys, os, time, threading
import numpy as np
if os.environ.get("TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
if os.environ.get("SWI"):
    sys.setswitchinterval(float(os.environ["SWI"]))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
W, H, M = 640, 480, 2
NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = BT // NCTX
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(BT)]
dets = [lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B)) for _ in range(NCTX)]
q = quantized_from_gpu(dets[0], frames[0][0], frames[0][1], M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
for k, d in enumerate(dets):
    d.add_class("c", descs, feats)
    for i in range(B):
        d.upload_frame(i, *frames[k * B + i])
outs = [(np.zeros((B, 4096), lm.MATCH_DTYPE), np.zeros(B, np.int32)) for _ in range(NCTX)]
NBUF = 3
bufs = [(np.zeros((BT, 4096), lm.MATCH_DTYPE), np.zeros(BT, np.int32)) for _ in range(NBUF)]
views = [[(o[c * B:(c + 1) * B], cn[c * B:(c + 1) * B]) for c in range(NCTX)] for o, cn in bufs]
MIMIC = os.environ.get("MIMIC")
def run(k, steps):
    for j in range(steps):
        if MIMIC:
            o, cn = views[j % NBUF][k]
            dets[k].match_batch(B, 80.0, 0, cap_per_frame=4096, out=o, counts=cn)
        else:
            dets[k].match_batch(B, 80.0, 0, cap_per_frame=4096, out=outs[k][0], counts=outs[k][1])
def timed(steps):
    th = [threading.Thread(target=run, args=(k, steps)) for k in range(NCTX)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
timed(10)
if os.environ.get('TSYNC'):
    import torch
    torch.cuda.synchronize()
if os.environ.get('TTENSOR'):
    import torch
    _t = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
if os.environ.get('PROF'):
    for d in dets: d.set_profiling(True)
dt = timed(100)
print("contexts %d, %d frames per step: %.1f detections/s  (%.2f us/frame)  matches0 %d" % (NCTX, BT, BT * 100 / dt, dt / 100 / BT * 1e6, outs[0][1][0]))
