"""Consumes tests/golden/opencv_vectors.npz -- outputs of the REAL cv::linemod on the reference's own frame, produced by
tests/golden/make_opencv_vectors.py on a box that has opencv-contrib -- and pins the CPU oracle against them.  The file
cannot be produced in this repo's build image (no OpenCV, no network), so until someone runs the hook these tests are
skipped and the oracle's parity with OpenCV stays UNPINNED."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, crop_masks

VEC = os.path.join(GOLDEN, "opencv_vectors.npz")
LUT = os.path.join(GOLDEN, "opencv_normal_lut.npy")          # NORMAL_LUT recovered by the hook's DepthNormal probe
YML = os.path.join(GOLDEN, "opencv_linemod_templates.yml.gz")  # a template file written by real OpenCV
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
        if os.path.exists(LUT):
            pytest.skip("pinned by test_depth_normal_equals_opencv_with_the_recovered_lut")
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


def test_phase_labels_equal_opencv_and_report_the_polynomial_form(vec):
    """cv::phase + convertTo over all 2041^2 Sobel gradients: labels (and the 16-bin values) must equal the oracle's -- in
    either form of the fastAtan2 polynomial, they agree on every label (test_orientation_rule.py) -- and the bits of a
    sample of raw angles say which form this OpenCV build evaluates (reported, not asserted)."""
    if "phase_labels_2041" not in vec.files:
        pytest.skip("vectors written by an older hook: no phase section")
    from oracle import oracle as O
    r = 1020
    dx, dy = np.meshgrid(np.arange(-r, r + 1, dtype=np.int32), np.arange(-r, r + 1, dtype=np.int32))
    lab, raw = O.orientation_labels_variant(dx, dy, 0, want_raw16=True)
    assert np.array_equal(lab, vec["phase_labels_2041"])
    assert np.array_equal(raw, vec["phase_raw16_2041"])
    fy, fx = dy.astype(np.float32).ravel()[::37], dx.astype(np.float32).ravel()[::37]
    want = vec["phase_sample_angles"]
    same = {v: int((O.fast_atan2(fy, fx, v).view(np.uint32) == want.view(np.uint32)).sum()) for v in (0, 1)}
    print("raw angles bit-equal to the unfused form: %d, to the fused form: %d of %d  (%s)"
          % (same[0], same[1], want.size, str(vec["opencv_build_cpu"]) if "opencv_build_cpu" in vec.files else "?"))
    assert max(same.values()) == want.size, "cv::phase matches neither form of the restated polynomial bit for bit: %r" % same


# ---- NORMAL_LUT recovered from OpenCV's behaviour (the hook's recover_normal_lut) --------------------------------------
@pytest.mark.skipif(not os.path.exists(LUT), reason="no recovered NORMAL_LUT: run tests/golden/make_opencv_vectors.py")
def test_depth_normal_equals_opencv_with_the_recovered_lut(orc, frame0, vec):
    """With the recovered table installed the DepthNormal modality is PINNED: quantised images of both levels, templates
    and the RGB-D match lists must equal OpenCV's.  Pixels whose LUT cell the probe could not observe (well under 1 % of
    the hemisphere, listed in opencv_normal_lut_coverage.npz) are the only admissible differences and are counted."""
    bgr, depth = frame0
    lut = np.load(LUT)
    assert lut.shape == (8000,) and set(np.unique(lut)) <= {0, 1, 2, 4, 8, 16, 32, 64, 128}
    o = orc.Detector(color_only=False)
    o.set_normal_lut(lut)
    o.prepare(bgr, depth)
    for level in range(2):
        got = o.stage(0, level, 1).reshape(depth.shape[0] >> level, depth.shape[1] >> level)
        diff = float((got != vec["rgbd_q%d1" % level]).mean())
        assert diff < 0.002, "DepthNormal level %d differs from OpenCV on %.3f %% of the pixels" % (level, 100 * diff)
    boxes = []
    for m in crop_masks(640, 480, 7, 6):
        tid, bb = o.add_template("obj", bgr, depth, m)
        boxes.append((tid,) + tuple(bb))
    assert np.array_equal(np.array(boxes, np.int32), vec["rgbd_boxes"])
    for thr in (80, 60):
        got = o.match(bgr, depth, float(thr))
        a = sorted((int(m["x"]), int(m["y"]), round(float(m["similarity"]), 4), int(m["template_id"])) for m in got)
        b = sorted((int(r[0]), int(r[1]), round(float(r[2]), 4), int(r[3])) for r in vec["rgbd_matches_%d" % thr])
        assert a == b


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(LUT), reason="no recovered NORMAL_LUT")
def test_gpu_depth_normal_with_the_recovered_lut(lm, frame0, vec):
    bgr, depth = frame0
    d = lm.Detector(color_only=False)
    d.set_normal_lut(np.load(LUT))
    assert not d.normal_lut_is_substitute()
    d.upload_frame(0, bgr, depth)
    d.prepare_slot(0)
    for level in range(2):
        got = d.debug_read(0, 0, level, 1).reshape(depth.shape[0] >> level, depth.shape[1] >> level)
        assert float((got != vec["rgbd_q%d1" % level]).mean()) < 0.002
    d.close()


# ---- f1: cvtColor / inRange and convexHull / fillPoly / countNonZero --------------------------------------------------
def _f1_bin(vec, frame0, path):
    bgr = frame0[0]
    H, W = bgr.shape[:2]
    masks = np.unpackbits(vec["f1_masks"], axis=-1)[..., :W].astype(np.uint8) * 255
    with open(path, "wb") as f:
        f.write(np.array([W, H, len(vec["f1_ranges"]), len(vec["f1_counts"])], np.int32).tobytes())
        f.write(np.ascontiguousarray(bgr).tobytes())
        f.write(np.ascontiguousarray(vec["f1_ranges"], np.float64).tobytes())
        f.write(np.ascontiguousarray(masks).tobytes())
        for k in range(len(vec["f1_counts"])):
            n = int(vec["f1_npoints"][k])
            f.write(np.array([n, vec["f1_offsets"][k][0], vec["f1_offsets"][k][1], vec["f1_counts"][k][2]], np.int32).tobytes())
            f.write(np.array(vec["f1_counts"][k][:2], np.int64).tobytes())
            f.write(np.ascontiguousarray(vec["f1_points"][k][:n], np.int32).tobytes())


def test_host_colour_check_equals_opencv(lm, frame0, vec, tmp_path):
    """host/PostProcess.cpp (the product's host colour check and the checker of the GPU one) against the OpenCV calls of
    the reference's colorCheck: HSV + inRange masks pixel by pixel, hull fill counts polygon by polygon."""
    if "f1_counts" not in vec.files:
        pytest.skip("vectors written by an older hook: no f1 section")
    _f1_bin(vec, frame0, tmp_path / "f1.bin")
    exe = str(tmp_path / "f1_vectors_check")
    libdir = os.path.dirname(lm.LIB_PATH)
    host = os.path.join(ROOT, "line-mod-pipeline_amd", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "f1_vectors_check.cpp"),
                           os.path.join(host, "PostProcess.cpp"), os.path.join(host, "HighLevelLinemod.cpp"),
                           os.path.join(host, "TemplateGenerator.cpp"), "-L" + libdir, "-llinemod_hip", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, str(tmp_path / "f1.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_colour_check_equals_opencv(lm, frame0, vec):
    """lm_color_check_counts (k_hsv_mask + k_hull_counts) against the same vectors: every polygon becomes a template whose
    level-0 features are the polygon's points."""
    if "f1_counts" not in vec.files:
        pytest.skip("vectors written by an older hook: no f1 section")
    bgr = frame0[0]
    d = lm.Detector(color_only=True)
    descs, feats = [], []
    usable = []
    for k in range(len(vec["f1_counts"])):
        n = int(vec["f1_npoints"][k])
        if n > 63:
            continue                                     # a colour-only template holds at most 63 features per level
        pts = vec["f1_points"][k][:n]
        usable.append(k)
        descs.append((int(pts[:, 0].max()) + 1, int(pts[:, 1].max()) + 1, 0, n))
        feats += [(int(x), int(y), 0) for x, y in pts]
        descs.append((1, 1, 1, 1))
        feats.append((0, 0, 0))
    d.add_class("polys", np.array(descs, lm.DESC_DTYPE), np.array(feats, lm.FEATURE_DTYPE))
    d.upload_frame(0, bgr, None)
    for r in range(len(vec["f1_ranges"])):
        ks = [k for k in usable if int(vec["f1_counts"][k][2]) == r]
        m = np.zeros(len(ks), lm.MATCH_DTYPE)
        for i, k in enumerate(ks):
            m[i] = (int(vec["f1_offsets"][k][0]), int(vec["f1_offsets"][k][1]), 100.0, usable.index(k), 0)
        a, b = d.color_check_counts(0, vec["f1_ranges"][r][0], vec["f1_ranges"][r][1], m)
        assert np.array_equal(a, vec["f1_counts"][ks, 0]) and np.array_equal(b, vec["f1_counts"][ks, 1])
    d.close()


# ---- f2: a linemod_templates.yml.gz written by real OpenCV -------------------------------------------------------------
@pytest.mark.skipif(not os.path.exists(YML), reason="no OpenCV-written template file")
def test_load_yaml_written_by_opencv(lm, vec, tmp_path):
    """lm_load_yaml on a file cv::linemod::Detector::write / writeClass produced (HighLevelLinemod.cpp:256-270): the
    templates must come out as OpenCV's getTemplates reported them, and lm_save_yaml -> lm_load_yaml round-trips them."""
    d = lm.Detector(color_only=False)
    d.load_yaml(YML)
    assert d.num_classes() == 1 and d.class_ids() == ["obj"]
    descs, feats = [], []
    for tid in range(d.class_num_templates(0)):
        for level in range(2):
            for mod in range(2):
                w, h, f = d.get_template(0, tid, level, mod)
                descs.append((w, h, level, len(f)))
                feats += [(int(a), int(b), int(c)) for a, b, c in zip(f["x"], f["y"], f["label"])]
    assert np.array_equal(np.array(descs, np.int32), vec["rgbd_descs"])
    assert np.array_equal(np.array(feats, np.int32).reshape(-1, 3), vec["rgbd_features"])
    d.save_yaml(str(tmp_path / "again.yml.gz"))
    d2 = lm.Detector(color_only=False)
    d2.load_yaml(str(tmp_path / "again.yml.gz"))
    assert d2.num_templates() == d.num_templates()
    d.close(); d2.close()
