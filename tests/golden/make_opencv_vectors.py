"""PINNING HOOK -- run this where OpenCV with the contrib `rgbd` module is installed (`import cv2; cv2.linemod`).

The reference's arithmetic lives in cv::linemod (opencv_contrib, version unpinned: /root/reference/CMakeLists.txt:53);
neither OpenCV nor the reference can be built in the image this repo is developed in, so the CPU oracle
(oracle/linemod_oracle.cpp) is a restatement whose parity with OpenCV is UNPINNED.  This script turns "unpinned" into
"pinned" the day an OpenCV box exists: it feeds the reference's own frame (tests/golden/frame0.npz = benchmark/img0.png +
depth0.png) to the real cv::linemod in the two configurations the reference builds (HighLevelLinemod.cpp:26-43) and
writes tests/golden/opencv_vectors.npz (+ tests/golden/opencv_linemod_templates.yml.gz, a template file written by real
OpenCV).  tests/test_opencv_vectors.py consumes these files when present and is skipped otherwise.

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
    np.savez_compressed(os.path.join(HERE, "opencv_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "opencv_vectors.npz"))


if __name__ == "__main__":
    main()
