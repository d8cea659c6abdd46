"""r06 counting experiments for the bit-plane scan on config 2 (CPU, drives the oracle; test infrastructure, not product):
per-modality miss rates, the effect of the feature ORDER on when the miss bound kills a wave, round sizes, counter widths, and how the survivors
cluster (what a second stage working on runs of positions would load).
usage: python tools/sim_scan_r06.py [templates] [frames] [G]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("line-mod-pipeline_amd.synth")
from oracle import oracle as orc

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NFR = int(sys.argv[2]) if len(sys.argv) > 2 else 8
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H, T0, T1, THR = 640, 480, 5, 8, 80.0
o = orc.Detector(color_only=False)
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(NFR)]
o.prepare(frames[0][0], frames[0][1])
q = {(l, m): o.stage(0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(2)}
descs, feats, _ = synth.make_bank(NT, 2, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=T0)
w1, h1 = W // 2, H // 2
Wm, Hm = w1 // T1, h1 // T1
wh = Wm * Hm
resp = []
for bgr, dep in frames:
    o.prepare(bgr, dep)
    resp.append([np.asarray(o.stage(2, 1, m)).reshape(8, T1 * T1, wh) for m in range(2)])
fo = 0
miss_rate = np.zeros(2); miss_n = np.zeros(2)
d_hist = np.zeros((2, 5))
orders = ["as_is", "depth_first", "interleave", "color_only_first16_then_depth"]
stops = {k: [] for k in orders}
stops2 = {k: [] for k in orders}          # p12 bound
surv_total = 0; cand_total = 0; runs8 = 0; runs32 = 0; lanes128 = 0; surv_p12 = 0
Fs = []
for t in range(NT):
    lists = []
    for k in range(4):
        ds = descs[t * 4 + k]
        lists.append((ds, fo))
        fo += int(ds["num_features"])
    feat_all = []
    for m in range(2):
        ds, start = lists[2 + m]
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
    Fs.append(F)
    P = max(min(P, wh), 0)
    mods = np.array([f[0] for f in feat_all])
    idx_c = np.where(mods == 0)[0]; idx_d = np.where(mods == 1)[0]
    perm = {"as_is": np.arange(F), "depth_first": np.concatenate([idx_d, idx_c])}
    il = []
    for a, b in zip(idx_c, idx_d): il += [b, a]
    il += list(idx_c[len(idx_d):]) + list(idx_d[len(idx_c):])
    perm["interleave"] = np.array(il)
    perm["color_only_first16_then_depth"] = np.concatenate([idx_c[:16], idx_d, idx_c[16:]])
    pf = {k: [] for k in orders}; pf2 = {k: [] for k in orders}
    for fr in range(NFR):
        vals = np.zeros((F, P), np.int16)
        for i, (m, lab, g, base) in enumerate(feat_all):
            seg = resp[fr][m][lab, g, base:base + P]
            vals[i, :len(seg)] = seg
        d1 = 4 - vals
        for m in range(2):
            sel = d1[mods == m]
            miss_rate[m] += (sel >= 1).sum(); miss_n[m] += sel.size
            for dd in range(5): d_hist[m, dd] += (sel == dd).sum()
        for k in orders:
            dm = (d1[perm[k]] >= 1)
            alive = (np.cumsum(dm, axis=0) <= K0).any(axis=1)
            st = F
            for done in range(8, F, 8):
                if not alive[done - 1]: st = done; break
            pf[k].append(st)
            d2 = np.minimum(d1[perm[k]], 2)
            alive2 = (np.cumsum(d2, axis=0) <= K0).any(axis=1)
            st = F
            for done in range(8, F, 8):
                if not alive2[done - 1]: st = done; break
            pf2[k].append(st)
        tot = d1.sum(axis=0)
        misses = (d1 >= 1).sum(axis=0)
        sv = misses <= K0
        surv_total += int(sv.sum()); cand_total += int((tot <= K0).sum())
        surv_p12 += int((np.minimum(d1, 2).sum(axis=0) <= K0).sum())
        pos = np.where(sv)[0]
        runs8 += len(np.unique(pos // 8)); runs32 += len(np.unique(pos // 32)); lanes128 += len(np.unique(pos // 128))
    for k in orders:
        stops[k].append(pf[k]); stops2[k].append(pf2[k])
Fs = np.array(Fs)
def kept(st, g):
    s = np.array(st); tot = 0
    for a in range(0, NFR, g): tot += s[:, a:a + g].max(axis=1).sum() * 1.0
    return tot / (Fs.sum() * ((NFR + g - 1) // g))
print("templates %d frames %d G %d  F mean %.1f" % (NT, NFR, G, Fs.mean()))
print("miss rate colour %.3f depth %.3f" % tuple(miss_rate / miss_n))
print("deficit histogram colour", np.round(d_hist[0] / d_hist[0].sum(), 3), "depth", np.round(d_hist[1] / d_hist[1].sum(), 3))
for k in orders:
    print("order %-32s miss bound kept: G=%d %.3f  G=1 %.3f | p12 bound kept: G=%d %.3f G=1 %.3f" % (k, G, kept(stops[k], G), kept(stops[k], 1), G, kept(stops2[k], G), kept(stops2[k], 1)))
print("per frame: candidates %.1f  survivors(miss bound) %.1f  survivors(p12) %.1f" % (cand_total / NFR * 3000 / NT, surv_total / NFR * 3000 / NT, surv_p12 / NFR * 3000 / NT))
print("survivors per 8-block %.2f per 32-block %.2f per 128-lane %.2f" % (surv_total / max(runs8, 1), surv_total / max(runs32, 1), surv_total / max(lanes128, 1)))
print("128-position lane-units with survivors per frame: %.1f (of %d)" % (lanes128 / NFR * 3000 / NT, 3000 * 8))
