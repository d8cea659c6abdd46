"""Scratch: host enqueue time (lm_match_begin returns) vs completion (lm_match_end) for one resident frame."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H = 640, 480
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=8))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(2)]
d.upload_frame(0, *frames[0]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
descs, feats, _ = synth.make_bank(3000, 2, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1,
                                  frame_size=(W, H), T0=d.get_T(0))
d.add_class("c", descs, feats)
d.upload_frame(1, *frames[1])
out = np.zeros((1, 4096), lm.MATCH_DTYPE); cn = np.zeros(1, np.int32)
for fork in (0, 2):
    d.set_tuning(lm.TUNE_FORK_MAX_SLOTS, fork)
    for _ in range(20):
        d.match_begin(0, 1, 1, 80.0, 0); d.match_end(0, 4096, out=out, counts=cn)
    tb = te = 0.0
    N = 200
    for _ in range(N):
        t0 = time.perf_counter()
        d.match_begin(0, 1, 1, 80.0, 0)
        t1 = time.perf_counter()
        d.match_end(0, 4096, out=out, counts=cn)
        t2 = time.perf_counter()
        tb += t1 - t0; te += t2 - t1
    print("fork<=%d: enqueue %.1f us, wait+collect %.1f us" % (fork, tb / N * 1e6, te / N * 1e6))
d.close()
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=8))
d.add_class("c", descs, feats)
d.upload_frame(1, *frames[1])
for fork in (0, 2, 0, 2):
    d.set_tuning(lm.TUNE_FORK_MAX_SLOTS, fork)
    d.time_stages(1, 80.0, 0, iters=5)
    print("fork<=%d stage us (preprocess, scan, refine, sort):" % fork, ["%.1f" % v for v in d.time_stages(1, 80.0, 0, iters=50)])
