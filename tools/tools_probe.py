"""Scratch perf probe (not part of the product): stage timings and scan-variant sweep on the GPU."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")

def quantized_from_gpu(d, bgr, depth, M, L=2):
    d.upload_frame(0, bgr, depth if M == 2 else None)
    d.prepare_slot(0)
    q = {}
    for l in range(L):
        for m in range(M):
            q[(l, m)] = d.debug_read(0, 0, l, m).reshape(d.height >> l, d.width >> l)
    return q

def run(color_only, size, n, fixed):
    M = 1 if color_only else 2
    d = lm.Detector(color_only=color_only, width=size[0], height=size[1])
    bgr, depth = synth.make_frame(size[0], size[1], seed=1234)
    q = quantized_from_gpu(d, bgr, depth, M)
    t = time.time()
    descs, feats, crops = synth.make_bank(n, M, 2, seed=4321, fixed_l0_size=fixed, quantized=q, crop_fraction=0.1,
                                          frame_size=size, T0=d.get_T(0))
    print("bank gen %.1fs crops %d" % (time.time() - t, len(crops)))
    d.add_class("c", descs, feats)
    for i in range(8):
        b, dp = synth.make_frame(size[0], size[1], seed=1234 + i)
        d.upload_frame(i, b, dp if M == 2 else None)
    m = d.match_slot(0, 80.0)
    print("matches", len(m), m[:3], "counts", d.last_counts(0))
    for v in (0, 1, 2):
        us, by = d.time_scan(0, 80.0, iters=50, variant=v)
        print("scan variant %d: %.2f us, %.1f MB algorithmic -> %.2f TB/s" % (v, us, by / 1e6, by / us / 1e6))
    st = d.time_stages(0, 80.0, iters=20)
    print("stages us: preprocess %.1f scan %.1f refine %.1f sort+copy %.1f" % tuple(st))
    for nb in (1, 2, 4, 8):
        d.match_batch(nb, 80.0)
        t = time.time(); K = 20
        for _ in range(K): d.match_batch(nb, 80.0)
        dt = time.time() - t
        print("batch %d: %.1f us/frame -> %.0f detections/s" % (nb, dt / K / nb * 1e6, K * nb / dt))
    d.close()

if __name__ == "__main__":
    run(False, (640, 480), 3000, (96, 96))
    run(True, (1280, 960), 3000, (192, 192))
