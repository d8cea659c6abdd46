"""Scratch: host cost and link rate of lm_upload_frame_pinned / lm_upload_frame (256 frames of 640x480 RGB-D)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
W, H, B = 640, 480, 256
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=B))
fb = W * H * 5
pb = lm.PinnedBuffer(B * fb)
hb = [pb.view(np.uint8, (H, W, 3), offset=i * fb) for i in range(B)]
hd = [pb.view(np.uint16, (H, W), offset=i * fb + W * H * 3) for i in range(B)]
SEQ = (1,) * 5 + (2,) * 5 + (3,) * 5 + (4,) * 8 + (2,) * 4 + (4,) * 4
for rep in range(len(SEQ)):
    d.set_tuning(lm.TUNE_COPY_STREAMS, SEQ[rep])
    print("copy streams", SEQ[rep], end=": ")
    t0 = time.perf_counter()
    for i in range(B):
        d.upload_frame_pinned(i, hb[i], hd[i])
    t1 = time.perf_counter()
    d.upload_wait(-1)
    t2 = time.perf_counter()
    print("pinned: enqueue %.2f ms (%.1f us per frame), done after %.2f ms: %.1f GB/s" % ((t1 - t0) * 1e3, (t1 - t0) / B * 1e6, (t2 - t0) * 1e3, B * fb / (t2 - t0) / 1e9))
for rep in range(4):
    t0 = time.perf_counter()
    d.upload_frames_pinned(0, B, pb.ptr.value, fb)
    t1 = time.perf_counter()
    d.upload_wait(-1)
    t2 = time.perf_counter()
    print("batch 2D: enqueue %.2f ms, done after %.2f ms: %.1f GB/s" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, B * fb / (t2 - t0) / 1e9))
for nb in (2, 4, 8, 16):
    t0 = time.perf_counter()
    for k in range(nb):
        d.upload_frames_pinned(k * B // nb, B // nb, pb.ptr.value + k * (B // nb) * fb, fb)
    t1 = time.perf_counter()
    d.upload_wait(-1)
    t2 = time.perf_counter()
    print("batch 2D in %d pieces: enqueue %.2f ms, done after %.2f ms: %.1f GB/s" % (nb, (t1 - t0) * 1e3, (t2 - t0) * 1e3, B * fb / (t2 - t0) / 1e9))
pg = [(np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.uint16)) for _ in range(B)]
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(B):
        d.upload_frame(i, *pg[i])
    t1 = time.perf_counter()
    d.upload_wait(-1)
    t2 = time.perf_counter()
    print("pageable (staged): enqueue %.2f ms (%.1f us per frame), done after %.2f ms: %.1f GB/s" % ((t1 - t0) * 1e3, (t1 - t0) / B * 1e6, (t2 - t0) * 1e3, B * fb / (t2 - t0) / 1e9))
d.close()
