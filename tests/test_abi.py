"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, its host-side logic (bank, persistence, merge, validation, defaults) behaves like the
reference's Detector calls, and compute entry points fail loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "linemod_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_[a-zA-Z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lm):
    lib = lm.load_library()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "liblinemod_hip.so does not export %s" % n
    assert sorted(lm.EXPORTS) == names
    assert b"gfx950" in lib.lm_version()
    # the binding declares the C signature of every export (size_t / pointer arguments never ride on ctypes' int default)
    for n in names:
        assert getattr(lib, n).argtypes is not None, "no argtypes declared for %s" % n


def test_tuning_keys_of_header_and_binding_agree(lm):
    """Every LM_TUNE_* key of the header has its twin in the Python binding, with the same value, and no key is used twice."""
    src = open(os.path.join(ROOT, "include", "linemod_hip.h")).read()
    keys = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define LM_(TUNE_[A-Z0-9_]+)\s+(\d+)\s*$", src, flags=re.M)}
    assert len(keys) >= 12 and len(set(keys.values())) == len(keys)
    for name, value in keys.items():
        assert getattr(lm, name, None) == value, name
    assert sorted(n for n in dir(lm) if n.startswith("TUNE_")) == sorted(keys)


def test_default_config_matches_reference_constructions(lm):
    # HighLevelLinemod.cpp:26-43: {ColorGradient, DepthNormal} T={5,8}; {ColorGradient} T={2,8}
    c = lm.default_config(color_only=False)
    assert (c.num_modalities, c.pyramid_levels, c.T[0], c.T[1]) == (2, 2, 5, 8)
    c = lm.default_config(color_only=True)
    assert (c.num_modalities, c.pyramid_levels, c.T[0], c.T[1]) == (1, 2, 2, 8)
    assert (c.weak_threshold, c.num_features, c.strong_threshold) == (10.0, 63, 55.0)
    assert (c.distance_threshold, c.difference_threshold, c.extract_threshold) == (2000, 50, 2)


def test_create_rejects_upstream_assert_cases(lm):
    with pytest.raises(lm.LinemodError) as e:   # rows % T != 0
        lm.Detector(color_only=False, width=640, height=482)
    assert e.value.code == lm.LM_ERR_INVALID
    with pytest.raises(lm.LinemodError):
        lm.Detector(color_only=False, width=644, height=480)   # 322 % 8 != 0 at level 1
    with pytest.raises(lm.LinemodError):
        lm.Detector(lm.default_config(num_modalities=3))
    with pytest.raises(lm.LinemodError):
        lm.Detector(lm.default_config(shard_rank=2, shard_size=2))
    lm.Detector(color_only=True, width=1280, height=960).close()


def _has_gpu(lm):
    d = lm.Detector(color_only=True, width=64, height=64, T=[2, 8])
    try:
        d.stage_pyrdown(np.zeros((8, 8, 3), np.uint8))
        return True
    except lm.LinemodError:
        return False
    finally:
        d.close()


def test_compute_fails_loudly_without_gpu(lm, frame0):
    if _has_gpu(lm):
        pytest.skip("a HIP device is present")
    bgr, depth = frame0
    d = lm.Detector(color_only=False)
    for call in (lambda: d.match(bgr, depth, 80.0),
                 lambda: d.upload_frame(0, bgr, depth),
                 lambda: d.stage_color_quantize(bgr),
                 lambda: d.stage_depth_quantize(depth),
                 lambda: d.stage_linear_memories(np.zeros((16, 16), np.uint8), 8),
                 lambda: d.add_template("x", bgr, depth),
                 lambda: d.time_scan(0, 80.0)):
        with pytest.raises(lm.LinemodError) as e:
            call()
        assert e.value.code == lm.LM_ERR_NO_DEVICE
        assert "no CPU fallback" in str(e.value)


def test_bank_roundtrip_and_queries(lm, golden0, tmp_path):
    d = lm.Detector(color_only=False)
    assert d.num_classes() == 0 and d.num_templates() == 0
    descs, feats = golden0["rgbd_descs"], golden0["rgbd_features"]
    ci = d.add_class("lagergehaeuse.ply", descs, feats)
    assert ci == 0 and d.class_ids() == ["lagergehaeuse.ply"]
    assert d.num_templates() == 6 and d.class_num_templates(0) == 6
    assert d.find_class("lagergehaeuse.ply") == 0 and d.find_class("nope") == -1
    ci2 = d.add_class("second", descs[:4], feats[:int(descs[:4]["num_features"].sum())])
    assert ci2 == 1 and d.num_classes() == 2 and d.num_templates() == 7
    # appending to an existing class continues its template ids
    d.add_class("second", descs[:4], feats[:int(descs[:4]["num_features"].sum())])
    assert d.class_num_templates(1) == 2
    # getTemplates(class, id)[level*M + modality]
    k = 0
    off = 0
    for tid in range(6):
        for level in range(2):
            for mod in range(2):
                w, h, f = d.get_template(0, tid, level, mod)
                assert (w, h) == (descs[k]["width"], descs[k]["height"])
                assert np.array_equal(f, feats[off:off + descs[k]["num_features"]])
                off += descs[k]["num_features"]
                k += 1
    with pytest.raises(lm.LinemodError):
        d.get_template(0, 6, 0, 0)
    # persistence (Detector::write/read stand-in)
    p = tmp_path / "bank.lmbk"
    d.save_bank(p)
    e = lm.Detector(color_only=False)
    e.load_bank(p)
    assert e.class_ids() == d.class_ids() and e.num_templates() == d.num_templates()
    for tid in range(6):
        a, b = d.get_template(0, tid, 1, 1), e.get_template(0, tid, 1, 1)
        assert a[:2] == b[:2] and np.array_equal(a[2], b[2])
    c = lm.Detector(color_only=True)
    with pytest.raises(lm.LinemodError) as ex:      # written for another modality set
        c.load_bank(p)
    assert ex.value.code == lm.LM_ERR_IO
    with pytest.raises(lm.LinemodError):
        c.load_bank(tmp_path / "missing.lmbk")


def test_add_class_validation(lm):
    d = lm.Detector(color_only=True)
    descs = np.zeros(2, lm.DESC_DTYPE)
    descs["pyramid_level"] = [0, 1]
    descs["width"], descs["height"] = [40, 20], [40, 20]
    descs["num_features"] = [64, 4]                        # > 63: upstream CV_Assert
    feats = np.zeros(68, lm.FEATURE_DTYPE)
    with pytest.raises(lm.LinemodError) as e:
        d.add_class("c", descs, feats)
    assert e.value.code == lm.LM_ERR_INVALID and d.num_classes() == 0
    descs["num_features"] = [4, 4]
    feats = np.zeros(8, lm.FEATURE_DTYPE)
    feats["label"][3] = 8
    with pytest.raises(lm.LinemodError):
        d.add_class("c", descs, feats)
    feats["label"][3] = 7
    bad = descs.copy()
    bad["pyramid_level"] = [1, 0]
    with pytest.raises(lm.LinemodError):
        d.add_class("c", bad, feats)
    descs["num_features"] = [8, 0]                         # a level without features
    with pytest.raises(lm.LinemodError):
        d.add_class("c", descs, feats)
    descs["num_features"] = [4, 4]
    assert d.add_class("c", descs, feats) == 0


def test_tables_agree_with_oracle_defaults(lm, orc):
    d = lm.Detector(color_only=False)
    assert np.array_equal(d.similarity_lut(), orc.similarity_lut())   # both default to upstream's table
    assert np.array_equal(d.normal_lut(), orc.normal_lut())
    lut = orc.similarity_lut(1)
    d.set_similarity_lut(lut)
    assert np.array_equal(d.similarity_lut(), lut)
    with pytest.raises(lm.LinemodError):
        d.set_similarity_lut(np.full(256, 5, np.uint8))   # 63 * 5 would overflow the byte accumulators


def test_merge_matches_equals_oracle_merge(lm, orc):
    rng = np.random.default_rng(0)
    lists = []
    for r in range(4):
        n = int(rng.integers(0, 50))
        m = np.zeros(n, lm.MATCH_DTYPE)
        m["x"] = rng.integers(0, 8, n) * 5 + 2
        m["y"] = rng.integers(0, 8, n) * 5 + 2
        m["similarity"] = rng.integers(80, 101, n).astype(np.float32)
        m["template_id"] = rng.integers(0, 10, n) + 10 * r
        m["class_idx"] = rng.integers(0, 2, n)
        lists.append(orc.merge([m]))       # each shard list sorted + unique
    got = lm.merge_matches(lists)
    exp = orc.merge(lists)
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes()
    assert len(lm.merge_matches([np.zeros(0, lm.MATCH_DTYPE)] * 3)) == 0


def test_struct_layouts(lm, orc):
    assert lm.MATCH_DTYPE.itemsize == 20 and lm.FEATURE_DTYPE.itemsize == 12 and lm.DESC_DTYPE.itemsize == 16
    assert C.sizeof(lm.Config) == 4 * 22   # 22 int32/float fields of lm_config


def test_scan_variant_refuses_list_changing_bits(lm):
    """r06: lm_set_scan_variant accepts only variants that leave the match lists as they are (no GPU needed: it is a host-side check)."""
    d = lm.Detector(color_only=True)
    for ok in (0, 1, 2, 8, 16, 32, 63, 256, 256 | 34):
        d.set_scan_variant(ok)
    for bad in (64, 72, 128, 384, 512, -1):
        with pytest.raises(lm.LinemodError):
            d.set_scan_variant(bad)
    d.set_scan_variant(0)
    d.close()


def test_build_lists_cover_every_source_file():
    """r06: the kernel and detector sources are split by stage; a source or header missing from build.py's lists would build a library without it (or
    rebuild too little).  Every csrc/*.hip / *.cpp is compiled, every csrc/*.h is a dependency of the objects."""
    import glob
    import importlib
    b = importlib.import_module("line-mod-pipeline_amd.build")
    on_disk = {os.path.basename(p) for p in glob.glob(os.path.join(b.CSRC, "*")) if os.path.isfile(p)}
    compiled = set(b.HIP_SOURCES) | set(b.CXX_SOURCES)
    assert {f for f in on_disk if f.endswith((".hip", ".cpp"))} == compiled
    headers = {os.path.basename(h) for h in b.HEADERS}
    assert {f for f in on_disk if f.endswith(".h")} <= headers
