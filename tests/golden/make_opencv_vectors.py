"""PINNING HOOK -- run this where OpenCV with the contrib `rgbd` module is installed (`import cv2; cv2.linemod`).

The reference's arithmetic lives in cv::linemod (opencv_contrib, version unpinned: /root/reference/CMakeLists.txt:53);
neither OpenCV nor the reference can be built in the image this repo is developed in, so the CPU oracle
(oracle/linemod_oracle.cpp) is a restatement whose parity with OpenCV is UNPINNED.  This script turns "unpinned" into
"pinned" the day an OpenCV box exists: it feeds the reference's own frame (tests/golden/frame0.npz = benchmark/img0.png +
depth0.png) to the real cv::linemod in the two configurations the reference builds (HighLevelLinemod.cpp:26-43) and
writes tests/golden/opencv_vectors.npz (+ tests/golden/opencv_linemod_templates.yml.gz, a template file written by real
OpenCV; + opencv_normal_lut.npy, the 8000 bytes of NORMAL_LUT recovered by probing DepthNormal with synthetic planes; +
the f1 vectors: cvtColor / inRange masks and convexHull + fillPoly counts).  tests/test_opencv_vectors.py consumes these
files when present and is skipped otherwise.

    python tests/golden/make_opencv_vectors.py        # needs: pip install opencv-contrib-python

Nothing here runs in this repo's CI and nothing of OpenCV is copied: the outputs are data (inputs -> outputs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def crop_masks(width, height, seed, n):          # same windows as make_golden.py / tests/conftest.py
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        w = int(rng.integers(60, 160)); h = int(rng.integers(60, 160))
        x0 = int(rng.integers(45, width - w - 45)); y0 = int(rng.integers(45, height - h - 45))
        m = np.zeros((height, width), np.uint8)
        m[y0:y0 + h, x0:x0 + w] = 255
        out.append(m)
    return out


def main():
    try:
        import cv2
        lm = cv2.linemod
    except Exception as e:                        # noqa: BLE001
        sys.exit("cv2.linemod is not available here (%s): run this on a box with opencv-contrib-python" % e)
    f = np.load(os.path.join(HERE, "frame0.npz"))
    bgr, depth = np.ascontiguousarray(f["bgr"]), np.ascontiguousarray(f["depth"])
    H, W = depth.shape
    out = {"opencv_version": np.array(cv2.__version__)}
    for name, color_only, T in (("rgbd", False, [5, 8]), ("color", True, [2, 8])):
        mods = [lm.ColorGradient_create(10.0, 63, 55.0)]          # the defaults the reference relies on
        if not color_only:
            mods.append(lm.DepthNormal_create(2000, 50, 63, 2))   # A.3
        det = lm.Detector(mods, T) if hasattr(lm, "Detector") else cv2.linemod_Detector(mods, T)
        sources = [bgr] if color_only else [bgr, depth]
        # ---- quantised images of every modality and level, straight from Modality::process
        for m, (mod, src) in enumerate(zip(mods, sources)):
            qp = mod.process(src, np.array([], np.uint8))
            for level in range(len(T)):
                if level > 0:
                    qp.pyrDown()
                out["%s_q%d%d" % (name, level, m)] = qp.quantize()
        # ---- templates of the six seeded crop windows + the match list at the reference's threshold
        descs, feats, boxes = [], [], []
        for mask in crop_masks(W, H, 7, 6):
            tid, bb = det.addTemplate(sources, "obj", mask)
            boxes.append((tid,) + tuple(bb))
            if tid < 0:
                continue
            for t in det.getTemplates("obj", tid):               # [level * M + modality]
                descs.append((t.width, t.height, t.pyramid_level, len(t.features)))
                feats.extend((ft.x, ft.y, ft.label) for ft in t.features)
        out[name + "_boxes"] = np.array(boxes, np.int32)
        out[name + "_descs"] = np.array(descs, np.int32).reshape(-1, 4)
        out[name + "_features"] = np.array(feats, np.int32).reshape(-1, 3)
        for thr in (80.0, 60.0):
            res = det.match(sources, thr)
            matches = res[0] if isinstance(res, tuple) else res
            out["%s_matches_%d" % (name, int(thr))] = np.array(
                [(m.x, m.y, m.similarity, m.template_id) for m in matches], np.float64).reshape(-1, 4)
        if name == "rgbd":
            try:                                                   # a file as HighLevelLinemod.cpp:256-270 writes it
                fs = cv2.FileStorage(os.path.join(HERE, "opencv_linemod_templates.yml.gz"), cv2.FILE_STORAGE_WRITE)
                det.write(fs)
                fs.startWriteStruct("classes", cv2.FileNode_SEQ)
                for cid in det.classIds():
                    fs.startWriteStruct("", cv2.FileNode_MAP)
                    det.writeClass(cid, fs)
                    fs.endWriteStruct()
                fs.endWriteStruct()
                fs.release()
            except Exception as e:                                 # noqa: BLE001 -- not every binding wraps write()
                print("could not write the template file through these bindings:", e)
        print(name, "templates", det.numTemplates(), "matches@80", len(out[name + "_matches_80"]))
    phase_vectors(cv2, out)
    f1_vectors(cv2, bgr, out)
    np.savez_compressed(os.path.join(HERE, "opencv_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "opencv_vectors.npz"))
    recover_normal_lut(cv2)


# ---------------------------------------------------------------------------------------------------------------------
# a3 steps 4-5 on their own: cv::phase + convertTo(CV_8U, 16/360) over EVERY gradient a 3x3 Sobel of 8-bit data can produce
# (|dx|, |dy| <= 1020).  The labels pin the oracle's rule and k_cgrad's integer rule; a sample of the raw angles tells which
# form of the fastAtan2 polynomial this OpenCV build runs (fused v_fma on AVX2 builds, plain multiply-add otherwise:
# tests/test_opencv_vectors.py compares the bits with both forms of the oracle and reports the one that matches).
def phase_vectors(cv2, out):
    r = 1020
    dx, dy = np.meshgrid(np.arange(-r, r + 1, dtype=np.float32), np.arange(-r, r + 1, dtype=np.float32))
    ang = cv2.phase(dx, dy, angleInDegrees=True)
    q = cv2.convertScaleAbs(ang, alpha=16.0 / 360.0)     # |x| of a non-negative value: saturate_cast<uchar>(cvRound(..)) as convertTo does
    out["phase_labels_2041"] = (q & 7).astype(np.uint8)
    out["phase_raw16_2041"] = q.astype(np.uint8)
    out["phase_sample_angles"] = ang.ravel()[::37].astype(np.float32)
    out["opencv_build_cpu"] = np.array(" ".join(l.strip() for l in cv2.getBuildInformation().splitlines()
                                                if "CPU/HW" in l or "Baseline" in l or "Dispatched" in l or "requested" in l))


# ---------------------------------------------------------------------------------------------------------------------
# f1: the OpenCV calls of the reference's colour check on the reference's own frame
#   cvtColor(BGR2HSV) + inRange                       HighLevelLinemod.cpp:159-161
#   convexHull + fillPoly + the two countNonZero      HighLevelLinemod.cpp:113-135, 424-434
# Polygons: feature-like point sets (<= 126 points inside a box) moved to offsets inside and partly outside the frame.
# Consumers: tests/test_opencv_vectors.py -> host/PostProcess.cpp (bgr2hsv_inrange, convex_hull, hull_counts) on the CPU
# and lm_color_check_counts on the GPU.
def f1_vectors(cv2, bgr, out):
    H, W = bgr.shape[:2]
    ranges = [((0, 0, 0), (255, 150, 255)),            # the shipped models/<name>.yml: S <= 150
              ((0, 0, 50), (255, 150, 255)),
              ((20, 30, 40), (110, 255, 200)),
              ((100, 0, 0), (180, 255, 255))]
    hsv = cv2.cvtColor(bgr, cv2.COLOR_BGR2HSV)
    out["f1_hsv"] = hsv
    out["f1_ranges"] = np.array(ranges, np.float64)
    masks = [cv2.inRange(hsv, np.array(lo, np.float64), np.array(hi, np.float64)) for lo, hi in ranges]
    out["f1_masks"] = np.packbits(np.stack(masks) != 0, axis=-1)
    rng = np.random.default_rng(2024)
    pts_all, off_all, n_all, cnt = [], [], [], []
    for k in range(48):
        n = int(rng.integers(3, 127))
        bw, bh = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        pts = np.stack([rng.integers(0, bw, n), rng.integers(0, bh, n)], 1).astype(np.int32)
        if k % 6 == 0:                                  # degenerate: collinear / repeated points
            pts[:, 1] = pts[0, 1]
        if k % 6 == 1:
            pts[:] = pts[0]
        ox = int(rng.integers(-60, W - 40)); oy = int(rng.integers(-60, H - 40))
        moved = (pts + np.array([ox, oy], np.int32)).astype(np.int32)
        hull = cv2.convexHull(moved.reshape(-1, 1, 2))
        m = np.zeros((H, W), np.uint8)
        cv2.fillPoly(m, [hull.reshape(-1, 2)], 255)
        both = cv2.bitwise_and(masks[k % len(masks)], m)
        pad = np.zeros((126, 2), np.int32); pad[:n] = pts
        pts_all.append(pad); off_all.append((ox, oy)); n_all.append(n)
        cnt.append((int(cv2.countNonZero(m)), int(cv2.countNonZero(both)), k % len(masks)))
    out["f1_points"] = np.stack(pts_all)
    out["f1_offsets"] = np.array(off_all, np.int32)
    out["f1_npoints"] = np.array(n_all, np.int32)
    out["f1_counts"] = np.array(cnt, np.int64)          # (countNonZero(mask), countNonZero(colour & mask), range index)


# ---------------------------------------------------------------------------------------------------------------------
# NORMAL_LUT (normal_lut.i, 20 x 20 x 20 bytes) recovered from the BEHAVIOUR of cv::linemod::DepthNormal -- nothing is
# read from OpenCV's sources.  On a plane z = d0 + a x + b y with every neighbour inside the difference threshold the
# accumulators of upstream's loop are A0 = A3 = 150, A1 = 0, b0 = 150 a, b1 = 150 b, so the normal is proportional to
# (1150 a, 1150 b, -d): integer slopes and a depth offset steer it into any cell (v1, v2, v3) =
# (int(10 nx + 10), int(10 ny + 10), int(20 nz + 20)) that the unit hemisphere nz <= 0 reaches; cells it does not
# reach are never read by DepthNormal.  A patch counts for a cell only when all 25 pixels of the 5 x 5 median window
# at its centre fall well inside that cell (margin 0.15 of a cell), so the label quantize() returns there is the
# table's byte for the cell.  Output: opencv_normal_lut.npy (flat index v3 * 400 + v2 * 20 + v1, 0 = never observed)
# and the coverage, for lm_set_normal_lut / orc_set_normal_lut.
def recover_normal_lut(cv2, out_dir=HERE, every=1):
    """every > 1: probe only every n-th reachable cell (the self-test of the method in tests/test_opencv_vectors.py)."""
    lm = cv2.linemod
    mod = lm.DepthNormal_create(70000, 70000, 63, 2)       # thresholds out of the way: every pixel valid, every tap used
    rng = np.random.default_rng(99)
    lut = np.zeros(8000, np.uint8)
    votes = np.zeros((8000, 256), np.int32)
    S = 48                                                 # patch size; centre (24, 24), taps reach +-5, the median +-2
    cells_target = []
    for v3 in range(20):
        for v2 in range(20):
            for v1 in range(20):
                # cell centre -> direction; keep the cells whose box meets the unit sphere (a few neighbours too many are harmless)
                c = np.array([(v1 + 0.5 - 10) / 10, (v2 + 0.5 - 10) / 10, (v3 + 0.5 - 20) / 20])
                lo = np.array([(v1 - 10) / 10, (v2 - 10) / 10, (v3 - 20) / 20]); hi = lo + np.array([0.1, 0.1, 0.05])
                far = np.maximum(np.abs(lo), np.abs(hi)); near = np.where((lo <= 0) & (hi >= 0), 0, np.minimum(np.abs(lo), np.abs(hi)))
                if (near ** 2).sum() <= 1.0 <= (far ** 2).sum():
                    cells_target.append((v1, v2, v3, c))
    cells_target = cells_target[::every]
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    tried = 0

    def cell_of(u):
        return (np.floor(u[..., 0] * 10 + 10).astype(int), np.floor(u[..., 1] * 10 + 10).astype(int), np.floor(u[..., 2] * 20 + 20).astype(int))

    for v1, v2, v3, c in cells_target:
        flat = v3 * 400 + v2 * 20 + v1
        for margin in (0.15, 0.08, 0.03):
            if votes[flat].sum() >= 3:
                break
            # unit vectors whose cell coordinates lie inside the cell with `margin` to spare: sample the box, normalise, re-test
            u = np.stack([(v1 + rng.uniform(0, 1, 4000) - 10) / 10, (v2 + rng.uniform(0, 1, 4000) - 10) / 10,
                          (v3 + rng.uniform(0, 1, 4000) - 20) / 20], 1)
            nrm = np.linalg.norm(u, axis=1)
            u = u[nrm > 0] / nrm[nrm > 0, None]
            fr = np.stack([u[:, 0] * 10 + 10 - v1, u[:, 1] * 10 + 10 - v2, u[:, 2] * 20 + 20 - v3], 1)
            u = u[np.all((fr > margin) & (fr < 1 - margin), axis=1) & (u[:, 2] < -1e-3)]
            for cand in u[:40]:
                if votes[flat].sum() >= 3:
                    break
                for L in (64000.0, 40000.0, 25000.0):
                    a, b = int(round(cand[0] * L / 1150)), int(round(cand[1] * L / 1150))
                    d0 = -cand[2] * L
                    z = np.rint(d0 + a * (xx - S // 2) + b * (yy - S // 2))
                    if z.min() < 1 or z.max() > 65000:
                        continue
                    # float restatement of the cell of every pixel of the centre window (only used to REJECT patches whose
                    # median window is not entirely, and safely, inside the cell)
                    win = z[S // 2 - 2:S // 2 + 3, S // 2 - 2:S // 2 + 3]
                    nx = 1150.0 * 22500 * a; ny = 1150.0 * 22500 * b; nz = -22500.0 * win
                    ln = np.sqrt(nx * nx + ny * ny + nz * nz)
                    ok = True
                    for fv, vv in ((nx / ln * 10 + 10, v1), (ny / ln * 10 + 10, v2), (nz / ln * 20 + 20, v3)):
                        ok &= bool(np.all((fv - vv > 0.02) & (fv - vv < 0.98)))
                    if not ok:
                        continue
                    q = mod.process(z.astype(np.uint16), np.array([], np.uint8)).quantize()
                    votes[flat, int(q[S // 2, S // 2])] += 1
                    tried += 1
                    break
    seen = votes.sum(1) > 0
    lut[seen] = votes[seen].argmax(1).astype(np.uint8)
    conflict = int(((votes > 0).sum(1) > 1).sum())
    targeted = np.array([(c[0], c[1], c[2]) for c in cells_target], np.int32)
    if out_dir:
        np.save(os.path.join(out_dir, "opencv_normal_lut.npy"), lut)
        np.savez_compressed(os.path.join(out_dir, "opencv_normal_lut_coverage.npz"), votes=votes, targeted=targeted)
    print("NORMAL_LUT: %d cells targeted, %d observed, %d with conflicting labels (must be 0), %d patches"
          % (len(cells_target), int(seen.sum()), conflict, tried))
    return lut, votes, targeted


if __name__ == "__main__":
    main()
