"""Generates tests/golden/frame0_golden.npz: outputs of the CPU oracle on the reference's own data
files (benchmark/img0.png + depth0.png, committed as frame0.npz by make_frame_fixture.py).

The reference holds no golden vectors for this path (parity unpinned), so these vectors
pin OUR oracle: any later change to oracle/linemod_oracle.cpp that alters a result fails
tests/test_oracle.py.  The known-answer part (self-extracted templates must be found at their crop
origin with similarity 100) does not depend on this file.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def crop_masks(width, height, seed, n):
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
    f = np.load(os.path.join(HERE, "frame0.npz"))
    bgr, depth = f["bgr"], f["depth"]
    H, W = depth.shape
    out = {}
    for name, color_only in (("rgbd", False), ("color", True)):
        det = O.Detector(color_only=color_only)
        for m in crop_masks(W, H, 7, 6):
            tid, bb = det.add_template("obj", bgr, None if color_only else depth, m)
            assert tid >= 0
        descs, feats = det.export_class(0)
        matches = det.match(bgr, None if color_only else depth, 80.0)
        out[name + "_descs"] = descs
        out[name + "_features"] = feats
        out[name + "_matches"] = matches
        hashes = []
        for l in range(det.pyramid_levels):
            for m in range(det.num_modalities):
                hashes.append("q%d%d:%s" % (l, m, sha(det.stage(0, l, m))))
                hashes.append("lm%d%d:%s" % (l, m, sha(det.stage(2, l, m))))
        out[name + "_hashes"] = np.array(hashes)
        print(name, "templates", det.num_templates(), "matches", len(matches))
    np.savez_compressed(os.path.join(HERE, "frame0_golden.npz"), **out)


if __name__ == "__main__":
    main()
