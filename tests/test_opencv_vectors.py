"""Consumes tests/golden/opencv_vectors.npz -- outputs of the REAL cv::linemod on the reference's own frame, produced by
tests/golden/make_opencv_vectors.py on a box that has opencv-contrib -- and pins the CPU oracle against them.  The file
cannot be produced in this repo's build image (no OpenCV, no network), so until someone runs the hook these tests are
skipped and the oracle's parity with OpenCV stays UNPINNED."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, crop_masks

VEC = os.path.join(GOLDEN, "opencv_vectors.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(VEC), reason="no OpenCV vectors: run tests/golden/make_opencv_vectors.py "
                                                               "where cv2.linemod imports")


@pytest.fixture(scope="module")
def vec():
    return np.load(VEC)


@pytest.mark.parametrize("name,color_only", [("rgbd", False), ("color", True)])
def test_quantised_images_equal_opencv(orc, frame0, vec, name, color_only):
    bgr, depth = frame0
    o = orc.Detector(color_only=color_only)
    o.prepare(bgr, None if color_only else depth)
    for level in range(2):
        got = o.stage(0, level, 0).reshape(bgr.shape[0] >> level, bgr.shape[1] >> level)
        assert np.array_equal(got, vec["%s_q%d0" % (name, level)]), "ColorGradient level %d differs from OpenCV" % level
    if not color_only:
        # DepthNormal needs OpenCV's NORMAL_LUT (normal_lut.i), which the oracle substitutes: report, do not fail
        got = o.stage(0, 0, 1).reshape(depth.shape)
        same = float((got == vec["rgbd_q01"]).mean())
        if same < 1.0:
            pytest.xfail("DepthNormal labels agree with OpenCV on %.1f %% of the pixels: install normal_lut.i "
                         "(lm_set_normal_lut / orc_set_normal_lut) to pin this modality" % (100 * same))


@pytest.mark.parametrize("name,color_only", [("rgbd", False), ("color", True)])
def test_templates_and_matches_equal_opencv(orc, frame0, vec, name, color_only):
    bgr, depth = frame0
    o = orc.Detector(color_only=color_only)
    boxes = []
    for m in crop_masks(640, 480, 7, 6):
        tid, bb = o.add_template("obj", bgr, None if color_only else depth, m)
        boxes.append((tid,) + tuple(bb))
    assert np.array_equal(np.array(boxes, np.int32), vec[name + "_boxes"])
    descs, feats = o.export_class(0)
    if color_only:      # the depth modality's features depend on NORMAL_LUT
        assert np.array_equal(np.stack([descs[k] for k in ("width", "height", "pyramid_level", "num_features")], 1), vec[name + "_descs"])
        assert np.array_equal(np.stack([feats[k] for k in ("x", "y", "label")], 1), vec[name + "_features"])
    for thr in (80, 60):
        exp = vec["%s_matches_%d" % (name, thr)]
        got = o.match(bgr, None if color_only else depth, float(thr))
        # upstream's order is not total: compare as sets of (x, y, similarity, template_id)
        a = sorted((int(m["x"]), int(m["y"]), round(float(m["similarity"]), 4), int(m["template_id"])) for m in got)
        b = sorted((int(r[0]), int(r[1]), round(float(r[2]), 4), int(r[3])) for r in exp)
        if color_only:
            assert a == b
        elif a != b:
            pytest.xfail("RGB-D match list differs from OpenCV (expected while NORMAL_LUT is substituted)")
