"""the reference's template file linemod_templates.yml.gz (cv::FileStorage YAML of
Detector::write + writeClass) and the other FileStorage files it reads.  OpenCV is not available here, so the
layout is pinned by a hand-written file in FileStorage's style and by the reference's own YAML files'
values (committed as numbers below); host-only, no GPU."""
import gzip
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _bank(det):
    out = []
    for ci, cid in enumerate(det.class_ids()):
        tps = []
        for tid in range(det.class_num_templates(ci)):
            tp = []
            for level in range(det.cfg.pyramid_levels):
                for m in range(det.cfg.num_modalities):
                    w, h, feats = det.get_template(ci, tid, level, m)
                    tp.append(((w, h, level), [tuple(int(v) for v in f) for f in feats]))
            tps.append(tp)
        out.append((cid, tps))
    return out


def test_yaml_round_trip_and_layout(lm, golden0, tmp_path):
    d = lm.Detector(color_only=False)
    d.add_class("lagergehaeuse.ply", golden0["rgbd_descs"], golden0["rgbd_features"])
    d.add_class("second", golden0["rgbd_descs"][:8], golden0["rgbd_features"][:int(golden0["rgbd_descs"][:8]["num_features"].sum())])
    path = tmp_path / "linemod_templates.yml.gz"
    d.save_yaml(path)
    text = gzip.open(path, "rt").read()            # really gzip
    assert text.startswith("%YAML:1.0\n---\npyramid_levels: 2\nT: [ 5, 8 ]\nmodalities:\n   -\n      type: ColorGradient\n"
                           "      weak_threshold: 10.\n      num_features: 63\n      strong_threshold: 55.\n   -\n"
                           "      type: DepthNormal\n      distance_threshold: 2000\n      difference_threshold: 50\n"
                           "      num_features: 63\n      extract_threshold: 2\nclasses:\n   -\n"
                           "      class_id: \"lagergehaeuse.ply\"\n      modalities: [ ColorGradient, DepthNormal ]\n"
                           "      pyramid_levels: 2\n      template_pyramids:\n         -\n            template_id: 0\n"
                           "            templates:\n               -\n                  width: ")
    assert "                     - [ " in text
    e = lm.Detector(color_only=False)
    e.load_yaml(path)
    assert _bank(e) == _bank(d)
    # plain (uncompressed) when the path does not end in .gz; loading twice leaves existing classes alone
    plain = tmp_path / "t.yml"
    d.save_yaml(plain)
    assert open(plain).read() == text
    e.load_yaml(plain)
    assert e.num_classes() == 2 and _bank(e) == _bank(d)
    d.close(); e.close()


def test_yaml_opencv_style_file(lm):
    d = lm.Detector(color_only=False)
    d.load_yaml(os.path.join(GOLD, "opencv_style_templates.yml"))
    assert d.class_ids() == ["bolt", "lager gehaeuse.ply"]
    b = _bank(d)
    assert len(b[0][1]) == 2 and len(b[1][1]) == 0
    tp0 = b[0][1][0]
    assert tp0[0] == ((40, 30, 0), [(0, 0, 1), (39, 29, 7), (12, 5, 0)])
    assert tp0[1] == ((40, 30, 0), [(3, 4, 2)])
    assert tp0[2] == ((20, 15, 1), [(1, 1, 6), (19, 14, 3)])
    assert tp0[3] == ((20, 15, 1), [])
    assert b[0][1][1][3] == ((4, 4, 1), [(0, 3, 5)])
    assert d.cfg.strong_threshold == 55.0 and d.cfg.weak_threshold == 10.0
    d.close()


def test_yaml_rejects_other_detectors(lm, tmp_path):
    d = lm.Detector(color_only=True)           # {ColorGradient}, T = {2, 8}
    with pytest.raises(lm.LinemodError):
        d.load_yaml(os.path.join(GOLD, "opencv_style_templates.yml"))   # two modalities, T = {5, 8}
    p = tmp_path / "c.yml.gz"
    d.save_yaml(p)
    e = lm.Detector(lm.default_config(color_only=True, T=[4, 8]))
    with pytest.raises(lm.LinemodError):
        e.load_yaml(p)
    with pytest.raises(lm.LinemodError):
        e.load_yaml(tmp_path / "missing.yml.gz")
    bad = tmp_path / "bad.yml"
    bad.write_text("%YAML:1.0\n---\npyramid_levels: 2\nT: [ 4, 8 ]\nmodalities:\n   -\n      type: ColorGradient\n")
    with pytest.raises(lm.LinemodError):
        e.load_yaml(bad)                         # incomplete modality parameters
    d.close(); e.close()


def test_yaml_filestorage_lookups(lm, tmp_path):
    """The values below are what the reference's own files hold (linemod_settings.yml, models/lagergehaeuse.yml,
    benchmark/pose0.yml); the files are re-created here in the same layout, comments and wrapped matrices included."""
    s = tmp_path / "linemod_settings.yml"
    s.write_text("%YAML:1.0\n---\n# ###### CAMERA PARAMETERS ######\nvideo width: 640\nvideo height: 480\n"
                 "camera fx: 1044.87\ncamera fy: 1045.69141\n\ndistortion parameters: !!opencv-matrix\n   rows: 1\n"
                 "   cols: 5\n   dt: d\n   data: [ -2.7167827743927644e-03, 2.0942424424199252e-01,\n"
                 "       1.1120545920170163e-03, -6.6420567497010334e-03, 0. ]\n# ###### TEMPLATE GENERATION SETTINGS ######\n"
                 "model folder: models/\nmodel file ending: \".ply\"\nonly use color modality: 1\ndetector threshold: 80\n")
    assert lm.yaml_numbers(s, "video width")[0] == 640 and lm.yaml_numbers(s, "camera fy")[0] == 1045.69141
    assert np.array_equal(lm.yaml_numbers(s, "distortion parameters"),
                          [-2.7167827743927644e-03, 2.0942424424199252e-01, 1.1120545920170163e-03,
                           -6.6420567497010334e-03, 0.0])
    assert lm.yaml_string(s, "model folder") == "models/" and lm.yaml_string(s, "model file ending") == ".ply"
    m = tmp_path / "lagergehaeuse.yml"
    m.write_text("%YAML:1.0\n---\n# HSV color range of Objekt \nlower color range: [ 0., 0., 0., 0. ]\n"
                 "upper color range: [ 255., 150., 255., 0. ]\n\n# rotational symmetry 1=yes 0=no\n"
                 "has rotational symmetry: 1\nplanes of symmetry: [ 1, 1, 1 ]\n")
    assert np.array_equal(lm.yaml_numbers(m, "upper color range"), [255, 150, 255, 0])
    assert lm.yaml_numbers(m, "has rotational symmetry")[0] == 1
    with pytest.raises(lm.LinemodError):
        lm.yaml_numbers(m, "no such key")
    try:
        lm.yaml_numbers(m, "no such key")
    except lm.LinemodError as e:
        assert e.code == lm.LM_ERR_INVALID and "no key" in str(e)      # a missing key is not an I/O failure
    try:
        lm.yaml_numbers(str(tmp_path / "absent.yml"), "x")
    except lm.LinemodError as e:
        assert e.code == lm.LM_ERR_IO
    with pytest.raises(lm.LinemodError):
        lm.yaml_numbers(s, "model folder")


def test_reference_data_files_byte_for_byte(lm):
    """r05: the reference's own cv::FileStorage data files, byte for byte (tests/golden/reference_data/, copied by
    make_reference_yaml_fixtures.py) -- not re-typed copies: linemod_settings.yml and models/lagergehaeuse.yml as the reference reads
    them (utility.cpp readSettings, HighLevelLinemod.cpp:523-543), and benchmark/pose0.yml, which cv::FileStorage itself WROTE (an
    !!opencv-matrix of doubles wrapped over five lines and a wrapped flow sequence): every value the pipeline reads comes back exactly,
    the ground-truth rotation is orthonormal, and it equals what tests/golden/lagergehaeuse.npz holds (extracted by a regular expression,
    an independent reading of the same file)."""
    d = os.path.join(GOLD, "reference_data")
    s = os.path.join(d, "linemod_settings.yml")
    want = {"video width": 640, "video height": 480, "camera fx": 1044.87, "camera fy": 1045.69141, "camera cx": 320, "camera cy": 240,
            "only use color modality": 1, "in plane rotation starting angle": -45, "in plane rotation stopping angle": 45,
            "in plane rotation angle step": 10, "distance start": 500, "distance stop": 1200, "distance step": 50,
            "icosahedron subdivisions": 3, "detector threshold": 80, "percent to pass check": 50, "number of poses to compare": 1,
            "distance to match to be considered same object": 45, "ratio to determine if group is too small": 35,
            "use depth improvement": 1, "depth offset": 30, "use icp": 0, "icp subsampling factor": 2}
    for key, v in want.items():
        got = lm.yaml_numbers(s, key)
        assert len(got) == 1 and got[0] == v, (key, got)
    assert np.array_equal(lm.yaml_numbers(s, "distortion parameters"),
                          [-2.7167827743927644e-03, 2.0942424424199252e-01, 1.1120545920170163e-03, -6.6420567497010334e-03, 0.0])
    assert lm.yaml_string(s, "model folder") == "models/" and lm.yaml_string(s, "model file ending") == ".ply"
    m = os.path.join(d, "lagergehaeuse.yml")
    assert np.array_equal(lm.yaml_numbers(m, "lower color range"), [0, 0, 0, 0])
    assert np.array_equal(lm.yaml_numbers(m, "upper color range"), [255, 150, 255, 0])
    assert lm.yaml_numbers(m, "has rotational symmetry")[0] == 1 and np.array_equal(lm.yaml_numbers(m, "planes of symmetry"), [1, 1, 1])
    p = os.path.join(d, "pose0.yml")
    rot = lm.yaml_numbers(p, "rotMat").reshape(3, 3)
    pos = lm.yaml_numbers(p, "position")
    g = np.load(os.path.join(GOLD, "lagergehaeuse.npz"))
    assert np.array_equal(rot, g["gt_rotation"]) and np.array_equal(pos, g["gt_position"])
    assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(rot) - 1.0) < 1e-6
    assert rot[0, 0] == 6.5663456916809082e-02 and pos[2] == 6.1265930523587429e+02


def test_yaml_malformed_input_is_an_error_not_a_crash(lm, tmp_path):
    """Truncated / garbled template files must come back as LM_ERR_IO (or load what is well-formed), never crash."""
    text = open(os.path.join(GOLD, "opencv_style_templates.yml")).read()
    rng = np.random.default_rng(3)
    d = lm.Detector(color_only=False)
    cuts = sorted(set(int(v) for v in rng.integers(40, len(text), 60)))
    for k, cut in enumerate(cuts):
        p = tmp_path / ("t%d.yml" % k)
        p.write_text(text[:cut])
        e = lm.Detector(color_only=False)
        try:
            e.load_yaml(p)
        except lm.LinemodError:
            pass
        e.close()
    for k in range(40):
        b = bytearray(text.encode())
        for pos in rng.integers(0, len(b), 8):
            b[pos] = int(rng.integers(32, 127))
        p = tmp_path / ("g%d.yml" % k)
        p.write_bytes(bytes(b))
        e = lm.Detector(color_only=False)
        try:
            e.load_yaml(p)
        except lm.LinemodError:
            pass
        e.close()
    (tmp_path / "empty.yml").write_text("")
    with pytest.raises(lm.LinemodError):
        d.load_yaml(tmp_path / "empty.yml")
    (tmp_path / "notgz.yml.gz").write_bytes(b"\x1f\x8b\x08\x00garbage")
    with pytest.raises(lm.LinemodError):
        d.load_yaml(tmp_path / "notgz.yml.gz")
    d.close()
