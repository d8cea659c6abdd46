"""Scratch: one detector, two lanes (lm_match_begin / lm_match_end) from one host thread."""
import importlib, sys, os, time
import numpy as np
import ctypes
if os.environ.get("PRELOAD"):   # bind the process to the system ROCm runtime before torch can load its bundled copy
    for lib in ("libhsa-runtime64.so.1", "libamdhip64.so.7"):
        ctypes.CDLL("/opt/rocm/lib/" + lib, mode=ctypes.RTLD_GLOBAL)
if os.environ.get("TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from tools_probe import quantized_from_gpu
W, H, M = 640, 480, 2
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = BT // NL
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(BT)]
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=BT))
q = quantized_from_gpu(d, frames[0][0], frames[0][1], M)
descs, feats, crops = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
for i in range(BT):
    d.upload_frame(i, *frames[i])
outs = [(np.zeros((B, 4096), lm.MATCH_DTYPE), np.zeros(B, np.int32)) for _ in range(NL)]
tb = te = 0.0
def run(n):
    global tb, te
    for l in range(NL):
        d.match_begin(l, l * B, B, 80.0, 0)
    for k in range(n):
        for l in range(NL):
            t = time.perf_counter()
            d.match_end(l, 4096, out=outs[l][0], counts=outs[l][1])
            te += time.perf_counter() - t
            if k + 1 < n:
                t = time.perf_counter()
                d.match_begin(l, l * B, B, 80.0, 0)
                tb += time.perf_counter() - t
run(10)
t0 = time.perf_counter(); run(100); dt = time.perf_counter() - t0
print("host time per lane-step: begin %.1f us, end (wait + collect) %.1f us" % (tb / (110 * NL) * 1e6, te / (110 * NL) * 1e6))
print("lanes %d, %d frames per step: %.1f detections/s  (%.2f us/frame)  matches0 %d" % (NL, BT, BT * 100 / dt, dt / 100 / BT * 1e6, outs[0][1][0]))

v = ctypes.c_int()
ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(v))
print("HIP runtime version", v.value, "| mapped:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l}))
