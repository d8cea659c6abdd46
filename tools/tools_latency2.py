"""Scratch: single-frame and small-batch latency with the pre-processing as one launch per dependency level
(LM_TUNE_PHASE_MAX_SLOTS) against the plain 14-launch sequence."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H = 640, 480
OUT = np.zeros(4096, lm.MATCH_DTYPE)     # caller-owned result buffer: the loop below measures the library, not numpy allocations
for color_only in (False, True):
    M = 1 if color_only else 2
    d = lm.Detector(lm.default_config(color_only=color_only, width=W, height=H, frame_slots=16))
    frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(8)]
    d.upload_frame(0, frames[0][0], None if color_only else frames[0][1]); d.prepare_slot(0)
    q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1,
                                      frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    for i in range(16):
        b, dp = frames[i % 8]
        d.upload_frame(i, b, None if color_only else dp)
    for phase, upmode in ((0, 0), (16, 0), (16, 2)):
        d.set_tuning(lm.TUNE_PHASE_MAX_SLOTS, phase)
        d.set_tuning(lm.TUNE_MATCH_UPLOAD_MODE, upmode)
        for _ in range(20):
            d.match_slot(1, 80.0, 0, out=OUT)
        t = time.perf_counter()
        for k in range(300):
            d.match_slot(1 + k % 7, 80.0, 0, out=OUT)
        t_slot = (time.perf_counter() - t) / 300
        for _ in range(10):
            d.match(frames[1][0], None if color_only else frames[1][1], 80.0, 0, out=OUT)
        t = time.perf_counter()
        for k in range(150):
            b, dp = frames[1 + k % 7]
            d.match(b, None if color_only else dp, 80.0, 0, out=OUT)
        t_host = (time.perf_counter() - t) / 150
        pb = lm.PinnedBuffer(7 * W * H * 5)
        pf = []
        for k in range(7):
            pc = pb.view(np.uint8, (H, W, 3), offset=k * W * H * 5); pd = pb.view(np.uint16, (H, W), offset=k * W * H * 5 + W * H * 3)
            pc[...] = frames[1 + k][0]; pd[...] = frames[1 + k][1]
            pf.append((pc, pd))
        for _ in range(10):
            d.match(pf[0][0], None if color_only else pf[0][1], 80.0, 0, out=OUT)
        t = time.perf_counter()
        for k in range(150):
            b, dp = pf[k % 7]
            d.match(b, None if color_only else dp, 80.0, 0, out=OUT)
        t_pin = (time.perf_counter() - t) / 150
        line = ("%s phases<=%d upload-mode " + str(upmode) + ": resident frame %.0f us, host frame in (lm_match) %.0f us, from pinned memory %.0f us") % (
            "colour-only" if color_only else "RGB-D", phase, t_slot * 1e6, t_host * 1e6, t_pin * 1e6)
        for nb in (2, 4, 8, 12):
            for _ in range(5):
                d.match_batch(nb, 80.0, 0)
            t = time.perf_counter()
            for _ in range(50):
                d.match_batch(nb, 80.0, 0)
            line += ", batch %d: %.0f us" % (nb, (time.perf_counter() - t) / 50 * 1e6)
        print(line)
    d.close()
