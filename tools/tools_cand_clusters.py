"""Scratch: how clustered are the scan's candidates?  (Would refining runs / blocks of neighbouring lattice positions of one
template on their union patch pay?)  Bench workload of config 2, a few frames; prints run-length statistics."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H, M = 640, 480, 2
d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=8))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(6)]
d.upload_frame(0, *frames[0]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=5)
d.add_class("c", descs, feats)
T1 = 16   # a level-1 lattice step of T = 8 is 16 level-0 pixels: candidates are reported at level-1 coordinates * ... (x, y as the scan emits them)
for i, (b, dp) in enumerate(frames):
    d.upload_frame(1, b, dp); d.prepare_slot(1)
    c = d.stage_scan(1, 80.0, 0)                     # (tid, cls, x, y) sorted by (cls, tid, y, x); x, y at level 1, step 8
    m = d.match_slot(1, 80.0, 0)
    n = len(c)
    key = set(map(tuple, c[:, [0, 2, 3]]))
    step = 8
    right = sum((t, x + step, y) in key for t, x, y in key)
    down = sum((t, x, y + step) in key for t, x, y in key)
    # horizontal runs
    runs = []
    for t, x, y in key:
        if (t, x - step, y) not in key:
            k = 1
            while (t, x + k * step, y) in key: k += 1
            runs.append(k)
    runs = np.array(runs)
    # 2-D clusters (4-connected)
    seen, sizes = set(), []
    for p in key:
        if p in seen: continue
        stack, sz = [p], 0
        seen.add(p)
        while stack:
            t, x, y = stack.pop(); sz += 1
            for dx, dy in ((step, 0), (-step, 0), (0, step), (0, -step)):
                nb = (t, x + dx, y + dy)
                if nb in key and nb not in seen: seen.add(nb); stack.append(nb)
        sizes.append(sz)
    sizes = np.array(sizes)
    print("frame %d: %d candidates, %d matches after refine+unique; with right neighbour %d, with lower neighbour %d; horizontal runs: %d (mean %.2f, max %d, hist %s); 4-connected clusters: %d (mean %.2f, max %d)" % (
        i, n, len(m), right, down, len(runs), runs.mean() if len(runs) else 0, runs.max() if len(runs) else 0, np.bincount(runs)[:8].tolist(), len(sizes), sizes.mean() if len(sizes) else 0, sizes.max() if len(sizes) else 0))
