"""Counting experiment for the scan's stopping rules (kept under tests/ because it drives the oracle; not collected by pytest; no GPU):
for templates x frames of the config-2 workload, after how many features does a work item stop under
  (a) k_scan4's exact rule (partial sum + 4 x features to come <= threshold at every position), tested every 6 features, 2 frames per wave;
  (b) k_scan1's miss bound (misses > (4 F - thr - 1) / 3), every 8 features, G frames per wave;
  (c) the exact deficit (3 x misses + zeros) every 4 features, G frames per wave (the two-plane form DESIGN section 8 costs).
usage: python tests/sim_scan_rules.py [templates] [frames] [G]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from oracle import oracle as orc

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 150
NFR = int(sys.argv[2]) if len(sys.argv) > 2 else 14
G = int(sys.argv[3]) if len(sys.argv) > 3 else 7
W, H, T0, T1, THR = 640, 480, 5, 8, 80.0
o = orc.Detector(color_only=False)
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(NFR)]
o.prepare(frames[0][0], frames[0][1])
q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
descs, feats, _ = synth.make_bank(NT, 2, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=T0)
w1, h1 = W // 2, H // 2
Wm, Hm = w1 // T1, h1 // T1
wh = Wm * Hm
# response memories of level 1 per frame: [m][o][g][wh]
resp = []
for bgr, dep in frames:
    o.prepare(bgr, dep)
    resp.append([np.asarray(o.stage(2, 1, m)).reshape(8, T1 * T1, wh) for m in range(2)])
# walk the bank: descs are [t][level*M + m]
fo = 0
surv = {}
# lower bounds of a feature's deficit d = 4 - response from ONE plane (w<k>: k x [d >= k], tested every 8 features) or TWO planes (every 4)
BOUNDS = {"w1": lambda d: (d >= 1) * 1, "w2": lambda d: (d >= 2) * 2, "w3": lambda d: (d >= 3) * 3, "w4": lambda d: (d >= 4) * 4,
          "p13": lambda d: (d >= 1) * 1 + (d >= 3) * 2, "p24": lambda d: (d >= 2) * 2 + (d >= 4) * 2, "p12": lambda d: (d >= 1) * 1 + (d >= 2) * 1,
          "p23": lambda d: (d >= 2) * 2 + (d >= 3) * 1, "p14": lambda d: (d >= 1) * 1 + (d >= 4) * 3}
stops = {"exact6": [], "miss8": [], "exact4": []}
Fs = []
for t in range(NT):
    lists = []
    for k in range(4):
        ds = descs[t * 4 + k]
        f = feats[fo:fo + ds["num_features"]] if isinstance(feats, np.ndarray) else None
        lists.append((ds, fo))
        fo += int(ds["num_features"])
    feat_all = []
    for m in range(2):
        ds, start = lists[2 + m]                                         # level 1, modality m
        ff = feats[start:start + int(ds["num_features"])]
        wf, hf = (int(ds["width"]) - 1) // T1 + 1, (int(ds["height"]) - 1) // T1 + 1
        P = (Hm - hf) * Wm + (Wm - wf) + 1
        for x, y, lab in zip(ff["x"], ff["y"], ff["label"]):
            if 0 <= x < w1 and 0 <= y < h1:
                feat_all.append((m, int(lab), (int(y) % T1) * T1 + int(x) % T1, (int(y) // T1) * Wm + int(x) // T1))
    n = sum(int(lists[2 + m][0]["num_features"]) for m in range(2))
    F = len(feat_all)
    thr = int(2 * n + np.float32(THR / 100.0) * np.float32(2 * n) + np.float32(0.5))
    K0 = 4 * F - thr - 1
    mmax = K0 // 3
    Fs.append(F)
    P = max(min(P, wh), 0)
    per_frame = {"exact6": [], "miss8": [], "exact4": []}
    def stop(alive, step):
        for done in range(step, F, step):
            if not alive[done - 1]:
                return done
        return F
    for fr in range(NFR):
        vals = np.zeros((F, P), np.int16)
        for i, (m, lab, g, base) in enumerate(feat_all):
            seg = resp[fr][m][lab, g, base:base + P]
            vals[i, :len(seg)] = seg
        d1 = 4 - vals
        deficit = np.cumsum(d1, axis=0)                               # [F][P]
        alive_exact = (deficit <= K0).any(axis=1)                           # after feature i+1: some position still in reach
        alive_miss = (np.cumsum(d1 >= 1, axis=0) <= K0).any(axis=1)         # the shipped table's largest response below 4 is 3: a miss costs at least 1
        for name, lb in BOUNDS.items():
            a = (np.cumsum(lb(d1), axis=0) <= K0)
            per_frame.setdefault(name, []).append(stop(a.any(axis=1), 8 if name.startswith("w") else 4))
            surv.setdefault(name, 0); surv[name] += int(a[-1].sum())
        surv.setdefault("exact", 0); surv["exact"] += int((deficit[-1] <= K0).sum())
        per_frame["exact6"].append(stop(alive_exact, 6))
        per_frame["miss8"].append(stop(alive_miss, 8))
        per_frame["exact4"].append(stop(alive_exact, 4))
    for k in per_frame:
        stops.setdefault(k, []).append(per_frame[k])
Fs = np.array(Fs)
def kept(key, g):
    s = np.array(stops[key])                                                # [NT][NFR]
    tot = 0
    for a in range(0, NFR, g):
        tot += s[:, a:a + g].max(axis=1).sum() * 1.0
    return tot / (Fs.sum() * ((NFR + g - 1) // g))
print("templates %d, frames %d, features per template %.1f" % (NT, NFR, Fs.mean()))
print("k_scan4 rule (exact, every 6 features), 2 frames per wave : feature loads kept %.3f   (1 frame: %.3f)" % (kept("exact6", 2), kept("exact6", 1)))
print("k_scan1 rule (miss bound, every 8),     %d frames per wave : %.3f   (2 frames: %.3f, 1 frame: %.3f)" % (G, kept("miss8", G), kept("miss8", 2), kept("miss8", 1)))
for name in BOUNDS:
    print("bound %-4s %d frames per wave: kept %.3f (1 frame %.3f) | survivors per candidate %.1f" % (name, G, kept(name, G), kept(name, 1), surv[name] / max(surv["exact"], 1)))
print("candidates per frame and template: %.4f" % (surv["exact"] / (NT * NFR)))
print("exact deficit, every 4 features,        %d frames per wave : %.3f   (2 frames: %.3f, 1 frame: %.3f)" % (G, kept("exact4", G), kept("exact4", 2), kept("exact4", 1)))
