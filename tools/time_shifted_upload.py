"""How fast is the row-offset DMA copy of a pinned frame (lm_upload_frame_pinned_shifted) against the plain one?  8 frames of 1280x960 RGB-D."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
W, H, N = 1280, 960, 8
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=N))
fb = W * H * 3 + W * H * 2
pb = lm.PinnedBuffer(N * fb)
hb = [pb.view(np.uint8, (H, W, 3), offset=i * fb) for i in range(N)]
hd = [pb.view(np.uint16, (H, W), offset=i * fb + W * H * 3) for i in range(N)]
rng = np.random.default_rng(1)
for i in range(N):
    hb[i][...] = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); hd[i][...] = rng.integers(0, 3000, (H, W), dtype=np.uint16)
def run(fn, reps=20):
    for _ in range(3):
        for i in range(N): fn(i)
        d.upload_wait(-1)
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(N): fn(i)
        d.upload_wait(-1)
    return (time.perf_counter() - t) / reps / N * 1e6
print("plain   %.1f us per frame" % run(lambda i: d.upload_frame_pinned(i, hb[i], hd[i])))
for sx, sy in ((0, 0), (7, 0), (0, 5), (7, 5), (-13, -9), (40, 30)):
    print("shift (%d, %d): %.1f us per frame" % (sx, sy, run(lambda i: d.upload_frame_pinned_shifted(i, hb[i], hd[i], sx, sy))))
pb.close([d]); d.close()
