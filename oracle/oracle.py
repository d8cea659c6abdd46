"""ctypes binding of the CPU oracle (oracle/linemod_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

*** PARITY UNPINNED *** (see linemod_oracle.h).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblinemod_oracle.so")
INT32_MAX = 2**31 - 1


class Feature(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("label", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("similarity", C.c_float),
                ("template_id", C.c_int32), ("class_idx", C.c_int32)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class TemplateDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pyramid_level", C.c_int32),
                ("num_features", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("num_modalities", C.c_int32), ("pyramid_levels", C.c_int32), ("T", C.c_int32 * 4),
                ("weak_threshold", C.c_float), ("num_features", C.c_int32), ("strong_threshold", C.c_float),
                ("distance_threshold", C.c_int32), ("difference_threshold", C.c_int32),
                ("depth_num_features", C.c_int32), ("extract_threshold", C.c_int32)]


MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                        ("class_idx", "<i4")])
FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
DESC_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("pyramid_level", "<i4"), ("num_features", "<i4")])


def build(force=False, arch=None, out=None):
    """Compile the oracle with g++ (oracle/Makefile).  `arch` e.g. '-march=native' builds a
    separate library used by bench.py's cpu_baseline leg on the machine it runs on."""
    if arch is None:
        override = os.environ.get("LINEMOD_ORACLE_LIB")      # e.g. the sanitizer build (make -C oracle asan) in tests/test_sanitize.py
        if override:
            return override
        src_newer = os.path.exists(_LIB_PATH) and any(os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
                                                       for f in ("linemod_oracle.cpp", "linemod_oracle.h"))
        if force or src_newer or not os.path.exists(_LIB_PATH):     # (an edited source must not meet a stale library)
            subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
        return _LIB_PATH
    out = out or os.path.join(_HERE, "_build", "liblinemod_oracle_native.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(_HERE, "linemod_oracle.cpp")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O3", arch, "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off",
                               "-shared", "-o", out, src])
    return out


_lib_cache = {}


def load(path=None):
    path = path or build()
    if path in _lib_cache:
        return _lib_cache[path]
    lib = C.CDLL(path)
    u8p, u16p, f32p, i16p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_float),
                             C.POINTER(C.c_int16))
    vp = C.c_void_p
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_default_config.argtypes = [C.POINTER(Config), C.c_int]
    lib.orc_default_similarity_lut.argtypes = [vp, C.c_int]
    lib.orc_default_normal_lut.argtypes = [vp]
    lib.orc_gaussian7_u8c3.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.orc_sobel3_s16c3.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.orc_color_quantize.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.orc_pyrdown_u8c3.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.orc_orientation_labels.argtypes = [vp, vp, C.c_size_t, vp]
    lib.orc_orientation_labels_variant.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    lib.orc_fast_atan2.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    lib.orc_set_atan_variant.argtypes = [C.c_int]
    lib.orc_set_atan_variant.restype = C.c_int
    lib.orc_depth_quantize.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.orc_resize_nn_half.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.orc_spread.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.orc_response_maps.argtypes = [vp, C.c_int, vp, vp]
    lib.orc_linearize.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.orc_create.restype = vp
    lib.orc_create.argtypes = [C.POINTER(Config)]
    lib.orc_destroy.argtypes = [vp]
    lib.orc_set_similarity_lut.argtypes = [vp, vp]
    lib.orc_set_normal_lut.argtypes = [vp, vp]
    lib.orc_num_classes.argtypes = [vp]
    lib.orc_num_templates.argtypes = [vp]
    lib.orc_class_num_templates.argtypes = [vp, C.c_int]
    lib.orc_add_class.argtypes = [vp, C.c_char_p, C.c_int, vp, vp]
    lib.orc_add_template.argtypes = [vp, C.c_char_p, vp, vp, vp, C.c_int, C.c_int, C.POINTER(Rect)]
    lib.orc_get_template.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), vp, C.POINTER(C.c_int)]
    lib.orc_match_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                    vp, C.c_int]
    lib.orc_prepare_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    lib.orc_match_prepared.argtypes = [vp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    lib.orc_scan_candidates.argtypes = [vp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    lib.orc_get_stage.restype = C.c_int64
    lib.orc_get_stage.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64]
    lib.orc_merge.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]
    lib.orc_set_scan_mode.argtypes = [C.c_int]
    lib.orc_set_scan_mode.restype = None
    _lib_cache[path] = lib
    return lib


def set_scan_mode(mode, lib_path=None):
    """0: scalar similarity sums, one bounds check per byte (upstream-faithful loop shape); 1 (default): the bounds check
    hoisted, vectorisable byte adds.  Identical results; process-wide per loaded library."""
    load(lib_path).orc_set_scan_mode(int(mode))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---------------------------------------------------------------------------------------------
# stage wrappers (numpy in, numpy out)
# ---------------------------------------------------------------------------------------------
def similarity_lut(variant=2, lib=None):
    lib = lib or load()
    out = np.zeros(256, np.uint8)
    lib.orc_default_similarity_lut(_ptr(out), variant)
    return out


def normal_lut(lib=None):
    lib = lib or load()
    out = np.zeros(8000, np.uint8)
    lib.orc_default_normal_lut(_ptr(out))
    return out


def gaussian7(bgr):
    lib = load(); bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
    out = np.empty_like(bgr); lib.orc_gaussian7_u8c3(_ptr(bgr), w, h, _ptr(out)); return out


def sobel3(img):
    lib = load(); img = _c(img, np.uint8); h, w, _ = img.shape
    dx = np.empty((h, w, 3), np.int16); dy = np.empty((h, w, 3), np.int16)
    lib.orc_sobel3_s16c3(_ptr(img), w, h, _ptr(dx), _ptr(dy)); return dx, dy


def color_quantize(bgr, weak_threshold=10.0, want_magnitude=False):
    lib = load(); bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
    q = np.empty((h, w), np.uint8)
    mag = np.empty((h, w), np.float32) if want_magnitude else None
    lib.orc_color_quantize(_ptr(bgr), w, h, weak_threshold, _ptr(q), _ptr(mag))
    return (q, mag) if want_magnitude else q


def orientation_labels(dx, dy):
    """Orientation label (0..7) of the float path (fastAtan2 -> x 16/360 -> rint -> & 7) for int32 gradient arrays."""
    lib = load(); dx = _c(dx, np.int32); dy = _c(dy, np.int32)
    out = np.empty(dx.shape, np.uint8); lib.orc_orientation_labels(_ptr(dx), _ptr(dy), dx.size, _ptr(out)); return out


def orientation_labels_variant(dx, dy, variant, want_raw16=False):
    """As orientation_labels with upstream's OTHER code shapes: variant bit 0 = fused multiply-adds in the fastAtan2 polynomial
    (v_atan_f32 on an AVX2 build), bit 1 = convertTo's scale applied in double."""
    lib = load(); dx = _c(dx, np.int32); dy = _c(dy, np.int32)
    out = np.empty(dx.shape, np.uint8); raw = np.empty(dx.shape, np.uint8) if want_raw16 else None
    lib.orc_orientation_labels_variant(_ptr(dx), _ptr(dy), dx.size, int(variant), _ptr(out), _ptr(raw))
    return (out, raw) if want_raw16 else out


def fast_atan2(y, x, variant=0):
    """cv::phase(x, y, angleInDegrees=True) for float32 arrays; variant 1 = the fused form."""
    lib = load(); y = _c(y, np.float32); x = _c(x, np.float32)
    out = np.empty(x.shape, np.float32); lib.orc_fast_atan2(_ptr(y), _ptr(x), x.size, int(variant), _ptr(out)); return out


def set_atan_variant(variant):
    """Selects the fastAtan2 form orc_color_quantize uses (0 unfused, 1 fused); returns the previous value."""
    return load().orc_set_atan_variant(int(variant))


def pyrdown(bgr):
    lib = load(); bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
    out = np.empty((h // 2, w // 2, 3), np.uint8); lib.orc_pyrdown_u8c3(_ptr(bgr), w, h, _ptr(out)); return out


def depth_quantize(depth, distance_threshold=2000, difference_threshold=50, lut=None):
    lib = load(); depth = _c(depth, np.uint16); h, w = depth.shape
    lut = normal_lut() if lut is None else _c(lut, np.uint8)
    out = np.empty((h, w), np.uint8)
    lib.orc_depth_quantize(_ptr(depth), w, h, distance_threshold, difference_threshold, _ptr(lut), _ptr(out))
    return out


def median5(img):
    """medianBlur(img, 5) on 8-bit, BORDER_REPLICATE (the tail of DepthNormal's quantizedNormals)."""
    lib = load(); img = _c(img, np.uint8); h, w = img.shape
    out = np.empty((h, w), np.uint8); lib.orc_median5_u8(_ptr(img), w, h, _ptr(out)); return out


def erode3(mask, iters=1):
    lib = load(); mask = _c(mask, np.uint8); h, w = mask.shape
    out = np.empty((h, w), np.uint8); lib.orc_erode3_u8(_ptr(mask), w, h, int(iters), _ptr(out)); return out


def dist_c(src):
    """distanceTransform(src, DIST_C, 3): chessboard distance to the nearest zero pixel."""
    lib = load(); src = _c(src, np.uint8); h, w = src.shape
    out = np.empty((h, w), np.float32); lib.orc_dist_c(_ptr(src), w, h, _ptr(out)); return out


def resize_nn_half(img):
    lib = load(); img = _c(img, np.uint8); h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8); lib.orc_resize_nn_half(_ptr(img), w, h, _ptr(out)); return out


def spread(q, T):
    lib = load(); q = _c(q, np.uint8); h, w = q.shape
    out = np.empty_like(q); lib.orc_spread(_ptr(q), w, h, T, _ptr(out)); return out


def response_maps(spr, lut=None):
    lib = load(); spr = _c(spr, np.uint8); lut = similarity_lut() if lut is None else _c(lut, np.uint8)
    out = np.empty((8,) + spr.shape, np.uint8); lib.orc_response_maps(_ptr(spr), spr.size, _ptr(lut), _ptr(out))
    return out


def linearize(resp, T):
    lib = load(); resp = _c(resp, np.uint8); h, w = resp.shape
    out = np.empty((T * T, (h // T) * (w // T)), np.uint8); lib.orc_linearize(_ptr(resp), w, h, T, _ptr(out))
    return out


# ---------------------------------------------------------------------------------------------
# detector
# ---------------------------------------------------------------------------------------------
class Detector:
    """Mirror of cv::linemod::Detector as the reference uses it (HighLevelLinemod.cpp:26-43,93,152)."""

    def __init__(self, color_only=False, T=None, lib_path=None, **overrides):
        self.lib = load(lib_path)
        self.cfg = Config()
        self.lib.orc_default_config(C.byref(self.cfg), 1 if color_only else 0)
        if T is not None:
            self.cfg.pyramid_levels = len(T)
            for i, t in enumerate(T):
                self.cfg.T[i] = t
        for k, v in overrides.items():
            setattr(self.cfg, k, v)
        self.h = self.lib.orc_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError(self.lib.orc_last_error().decode())

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_modalities(self):
        return self.cfg.num_modalities

    @property
    def pyramid_levels(self):
        return self.cfg.pyramid_levels

    def set_similarity_lut(self, lut):
        lut = _c(lut, np.uint8); assert lut.size == 256
        self.lib.orc_set_similarity_lut(self.h, _ptr(lut))

    def set_normal_lut(self, lut):
        lut = _c(lut, np.uint8); assert lut.size == 8000
        self.lib.orc_set_normal_lut(self.h, _ptr(lut))

    def num_classes(self):
        return self.lib.orc_num_classes(self.h)

    def num_templates(self):
        return self.lib.orc_num_templates(self.h)

    def class_num_templates(self, ci):
        return self.lib.orc_class_num_templates(self.h, ci)

    def add_class(self, class_id, descs, features):
        descs = _c(descs, DESC_DTYPE); features = _c(features, FEATURE_DTYPE)
        per = self.cfg.pyramid_levels * self.cfg.num_modalities
        assert descs.size % per == 0 and int(descs["num_features"].sum()) == features.size
        r = self.lib.orc_add_class(self.h, class_id.encode(), descs.size // per, _ptr(descs), _ptr(features))
        if r < 0:
            raise RuntimeError(self.lib.orc_last_error().decode())
        return r

    def add_template(self, class_id, bgr, depth=None, mask=None):
        bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
        depth = None if depth is None else _c(depth, np.uint16)
        mask = None if mask is None else _c(mask, np.uint8)
        bb = Rect()
        tid = self.lib.orc_add_template(self.h, class_id.encode(), _ptr(bgr), _ptr(depth), _ptr(mask), w, h, C.byref(bb))
        return tid, (bb.x, bb.y, bb.width, bb.height)

    def get_template(self, ci, tid, level, modality):
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        if self.lib.orc_get_template(self.h, ci, tid, level, modality, C.byref(w), C.byref(h), None, C.byref(n)) != 0:
            raise IndexError("no such template")
        feats = np.zeros(n.value, FEATURE_DTYPE)
        self.lib.orc_get_template(self.h, ci, tid, level, modality, C.byref(w), C.byref(h), _ptr(feats), C.byref(n))
        return w.value, h.value, feats

    def export_class(self, ci):
        """(descs, features) arrays in the add_class layout."""
        per = self.cfg.pyramid_levels * self.cfg.num_modalities
        M = self.cfg.num_modalities
        descs, feats = [], []
        for tid in range(self.class_num_templates(ci)):
            for k in range(per):
                w, h, f = self.get_template(ci, tid, k // M, k % M)
                descs.append((w, h, k // M, f.size)); feats.append(f)
        return (np.array(descs, DESC_DTYPE), np.concatenate(feats) if feats else np.zeros(0, FEATURE_DTYPE))

    def prepare(self, bgr, depth=None):
        bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
        depth = None if depth is None else _c(depth, np.uint16)
        if self.lib.orc_prepare_frame(self.h, _ptr(bgr), _ptr(depth), w, h) != 0:
            raise RuntimeError(self.lib.orc_last_error().decode())

    def match_prepared(self, threshold, class_idx=-1, tid_lo=0, tid_hi=INT32_MAX, threads=1, cap=1 << 20):
        out = np.zeros(cap, MATCH_DTYPE)
        n = self.lib.orc_match_prepared(self.h, threshold, class_idx, tid_lo, tid_hi, threads, _ptr(out), cap)
        if n < 0:
            raise RuntimeError(self.lib.orc_last_error().decode())
        if n > cap:
            return self.match_prepared(threshold, class_idx, tid_lo, tid_hi, threads, cap=n)
        return out[:n].copy()

    def scan_candidates(self, threshold, class_idx=-1, tid_lo=0, tid_hi=INT32_MAX, threads=1, cap=1 << 18):
        """a11-a13 of the prepared frame: [n, 4] int32 (template_id, class_idx, x, y), sorted."""
        out = np.zeros((cap, 4), np.int32)
        n = self.lib.orc_scan_candidates(self.h, threshold, class_idx, tid_lo, tid_hi, threads, _ptr(out), cap)
        if n < 0:
            raise RuntimeError(self.lib.orc_last_error().decode())
        if n > cap:
            return self.scan_candidates(threshold, class_idx, tid_lo, tid_hi, threads, cap=n)
        return out[:n].copy()

    def match(self, bgr, depth, threshold, class_idx=-1, tid_lo=0, tid_hi=INT32_MAX, threads=1, cap=1 << 16):
        """Detector::match; threads > 1 also threads the image stages (orc_match_frame)."""
        bgr = _c(bgr, np.uint8); h, w, _ = bgr.shape
        depth = None if depth is None else _c(depth, np.uint16)
        out = np.zeros(cap, MATCH_DTYPE)
        n = self.lib.orc_match_frame(self.h, _ptr(bgr), _ptr(depth), w, h, threshold, class_idx, tid_lo, tid_hi,
                                     threads, _ptr(out), cap)
        if n < 0:
            raise RuntimeError(self.lib.orc_last_error().decode())
        if n > cap:
            return self.match(bgr, depth, threshold, class_idx, tid_lo, tid_hi, threads, cap=n)
        return out[:n].copy()

    def stage(self, what, level, modality):
        n = self.lib.orc_get_stage(self.h, what, level, modality, None, 0)
        if n < 0:
            raise RuntimeError("no frame prepared")
        out = np.zeros(n, np.uint8)
        self.lib.orc_get_stage(self.h, what, level, modality, _ptr(out), n)
        return out


def merge(lists):
    """R-way merge + adjacent-unique of per-shard sorted match arrays."""
    lib = load()
    stride = max([len(l) for l in lists] + [1])
    buf = np.zeros((len(lists), stride), MATCH_DTYPE)
    counts = np.zeros(len(lists), np.int32)
    for i, l in enumerate(lists):
        buf[i, :len(l)] = l; counts[i] = len(l)
    out = np.zeros(int(counts.sum()) + 1, MATCH_DTYPE)
    n = lib.orc_merge(_ptr(buf), _ptr(counts), len(lists), stride, _ptr(out), out.size)
    return out[:n].copy()
