"""line-mod-pipeline_amd -- MI355X-native LINE-MOD detector (host-side Python binding).

The product is liblinemod_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/linemod_hip.h).  This package is only the ctypes view of that ABI used by tests, bench.py
and the multi-GPU shard driver; it contains no compute and no fallback: if the library is missing
the import fails, and if no HIP device is present every compute call raises LinemodError.

Import with importlib (the directory name is not a Python identifier):
    lm = importlib.import_module("line-mod-pipeline_amd")
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblinemod_hip.so")

LM_OK, LM_ERR_INVALID, LM_ERR_NO_DEVICE, LM_ERR_HIP, LM_ERR_OVERFLOW, LM_ERR_IO, LM_ERR_EXTRACT = range(7)

MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                        ("class_idx", "<i4")])
FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
DESC_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("pyramid_level", "<i4"), ("num_features", "<i4")])
DEPTH_QUERY_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("slot", "<i4"), ("reserved", "<i4")])   # lm_depth_query


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_modalities", C.c_int32),
                ("pyramid_levels", C.c_int32), ("T", C.c_int32 * 4), ("weak_threshold", C.c_float),
                ("num_features", C.c_int32), ("strong_threshold", C.c_float), ("distance_threshold", C.c_int32),
                ("difference_threshold", C.c_int32), ("depth_num_features", C.c_int32),
                ("extract_threshold", C.c_int32), ("device", C.c_int32), ("shard_rank", C.c_int32),
                ("shard_size", C.c_int32), ("max_candidates", C.c_int32), ("max_matches", C.c_int32),
                ("frame_slots", C.c_int32), ("flags", C.c_int32)]

FLAG_BYTE_RESPONSES = 1
FLAG_BLOCKING_SYNC = 2
TUNE_COPY_STREAMS = 3
TUNE_CBLUR_VARIANT = 4
TUNE_CGRAD_VARIANT = 5
TUNE_PHASE_MAX_SLOTS = 6
TUNE_BATCH_PHASES = 7
TUNE_PYRDOWN_VARIANT = 8
TUNE_BLUR_PYR = 9
TUNE_DMEDIAN_VARIANT = 11
TUNE_BLUR_STRIP = 12
TUNE_WORK_WEIGHT = 13
TUNE_SORT_SPLIT = 14
TUNE_SCAN_LIST_ORDER = 15
TUNE_SCAN_FORM = 16
TUNE_SCAN1_MIN_THRESHOLD = 17
TUNE_CGRAD_LEVELS = 18
TUNE_SURVIVOR_QUEUE = 19


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class LinemodError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liblinemod_hip error %d: %s" % (code, msg))
        self.code = code


# Every symbol include/linemod_hip.h declares (tests check that the library exports all of them).
EXPORTS = [
    "lm_last_error", "lm_version", "lm_default_config", "lm_create", "lm_destroy", "lm_set_similarity_lut",
    "lm_set_normal_lut", "lm_get_similarity_lut", "lm_get_normal_lut", "lm_num_classes", "lm_num_templates",
    "lm_class_num_templates", "lm_class_id", "lm_find_class", "lm_get_T", "lm_num_modalities",
    "lm_pyramid_levels", "lm_add_class", "lm_add_template", "lm_get_template", "lm_match", "lm_upload_frame",
    "lm_match_slot", "lm_match_batch", "lm_merge_matches", "lm_save_bank", "lm_load_bank",
    "lm_stage_color_quantize", "lm_stage_pyrdown", "lm_stage_depth_quantize", "lm_stage_linear_memories",
    "lm_prepare_slot", "lm_debug_read", "lm_stage_scan", "lm_time_scan", "lm_time_stages", "lm_set_scan_variant",
    "lm_last_counts", "lm_set_profiling", "lm_get_profile", "lm_scan_load_bytes",
    "lm_save_yaml", "lm_load_yaml", "lm_yaml_numbers", "lm_yaml_string", "lm_pack_matches", "lm_merge_batch",
    "lm_upload_frame_shifted", "lm_match_begin", "lm_match_end", "lm_synchronize", "lm_merge_frames", "lm_gather_plan", "lm_gather_max_total",
    "lm_upload_frame_pinned", "lm_upload_wait", "lm_host_alloc", "lm_host_free", "lm_set_stage_chunks",
    "lm_set_tuning", "lm_comm_init", "lm_comm_destroy", "lm_comm_info", "lm_match_begin_gathered",
    "lm_match_end_gathered", "lm_comm_barrier", "lm_comm_max", "lm_upload_frames_pinned",
    "lm_rendezvous_broadcast", "lm_normal_lut_is_substitute",
    "lm_set_scan_stats", "lm_get_scan_stats", "lm_color_check_counts",
    "lm_match_batch_classes", "lm_match_prepared", "lm_match_begin_classes", "lm_device_pci_bus_id",
    "lm_get_exchange_profile", "lm_get_stage_counts", "lm_get_scan_lane_stats", "lm_get_scan_form_stats", "lm_match_classes",
    "lm_time_scan_batch",
    "lm_selftest_float_tail",
    "lm_upload_frame_pinned_shifted", "lm_stage_reserve", "lm_stage_rows", "lm_upload_staged", "lm_match_collect",
    "lm_color_check_counts_slots", "lm_color_check_begin_slots", "lm_color_check_end", "lm_color_mask_prepare",
    "lm_depth_counts_begin", "lm_depth_counts_end",
]

_lib = None


def load_library(path=None):
    """Loads liblinemod_hip.so; raises OSError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError("liblinemod_hip.so not built (%s): run `python line-mod-pipeline_amd/build.py` "
                      "or __graft_entry__.build(); there is no CPU fallback" % p)
    lib = C.CDLL(p)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.lm_last_error.restype = C.c_char_p
    lib.lm_version.restype = C.c_char_p
    lib.lm_default_config.argtypes = [C.POINTER(Config), i, i, i]
    lib.lm_default_config.restype = None
    lib.lm_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.lm_destroy.argtypes = [vp]
    lib.lm_destroy.restype = None
    for name in ("lm_set_similarity_lut", "lm_set_normal_lut", "lm_get_similarity_lut", "lm_get_normal_lut"):
        getattr(lib, name).argtypes = [vp, vp]
    lib.lm_normal_lut_is_substitute.argtypes = [vp]
    lib.lm_num_classes.argtypes = [vp]
    lib.lm_num_templates.argtypes = [vp]
    lib.lm_class_num_templates.argtypes = [vp, i]
    lib.lm_class_id.argtypes = [vp, i]
    lib.lm_class_id.restype = C.c_char_p
    lib.lm_find_class.argtypes = [vp, C.c_char_p]
    lib.lm_get_T.argtypes = [vp, i]
    lib.lm_num_modalities.argtypes = [vp]
    lib.lm_pyramid_levels.argtypes = [vp]
    lib.lm_add_class.argtypes = [vp, C.c_char_p, i, vp, vp, C.POINTER(i)]
    lib.lm_add_template.argtypes = [vp, C.c_char_p, vp, sz, vp, sz, vp, sz, C.POINTER(i), C.POINTER(Rect)]
    lib.lm_get_template.argtypes = [vp, i, i, i, i, C.POINTER(i), C.POINTER(i), vp, C.POINTER(i)]
    lib.lm_match.argtypes = [vp, vp, sz, vp, sz, f, i, vp, sz, C.POINTER(sz)]
    lib.lm_upload_frame.argtypes = [vp, i, vp, sz, vp, sz]
    lib.lm_upload_frame_shifted.argtypes = [vp, i, vp, sz, vp, sz, i, i]
    lib.lm_match_slot.argtypes = [vp, i, f, i, vp, sz, C.POINTER(sz)]
    lib.lm_match_batch.argtypes = [vp, i, f, i, vp, sz, vp]
    lib.lm_match_begin.argtypes = [vp, i, i, i, f, i]
    lib.lm_match_end.argtypes = [vp, i, vp, sz, vp]
    lib.lm_pack_matches.argtypes = [vp, sz, vp, i, vp, sz, C.POINTER(sz)]
    lib.lm_merge_batch.argtypes = [vp, sz, vp, i, i, vp, sz, vp, C.POINTER(sz)]
    lib.lm_merge_frames.argtypes = [vp, sz, vp, i, i, i, i, vp, sz, vp, C.POINTER(sz)]
    lib.lm_gather_plan.argtypes = [vp, i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i), vp, vp, vp]
    lib.lm_gather_max_total.argtypes = [vp, i, i, C.POINTER(C.c_uint64)]
    lib.lm_merge_matches.argtypes = [vp, vp, i, sz, vp, sz, C.POINTER(sz)]
    lib.lm_save_bank.argtypes = [vp, C.c_char_p]
    lib.lm_load_bank.argtypes = [vp, C.c_char_p]
    lib.lm_stage_color_quantize.argtypes = [vp, vp, i, i, f, vp, vp]
    lib.lm_stage_pyrdown.argtypes = [vp, vp, i, i, vp]
    lib.lm_stage_depth_quantize.argtypes = [vp, vp, i, i, vp]
    lib.lm_stage_linear_memories.argtypes = [vp, vp, i, i, i, vp]
    lib.lm_prepare_slot.argtypes = [vp, i]
    lib.lm_debug_read.argtypes = [vp, i, i, i, i, vp, sz, C.POINTER(sz)]
    lib.lm_stage_scan.argtypes = [vp, i, f, i, vp, sz, C.POINTER(sz)]
    lib.lm_time_scan.argtypes = [vp, i, f, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.lm_time_stages.argtypes = [vp, i, f, i, i, C.POINTER(C.c_double)]
    lib.lm_time_scan_batch.argtypes = [vp, i, i, f, i, i, i, C.POINTER(C.c_double)]
    lib.lm_selftest_float_tail.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.lm_set_scan_variant.argtypes = [vp, i]
    lib.lm_color_check_counts.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, sz, vp, vp]
    lib.lm_set_scan_stats.argtypes = [vp, i]
    lib.lm_get_scan_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.lm_last_counts.argtypes = [vp, i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.lm_set_profiling.argtypes = [vp, i]
    lib.lm_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]
    lib.lm_last_error.argtypes = []
    lib.lm_version.argtypes = []
    lib.lm_save_yaml.argtypes = [vp, C.c_char_p]
    lib.lm_load_yaml.argtypes = [vp, C.c_char_p]
    lib.lm_yaml_numbers.argtypes = [C.c_char_p, C.c_char_p, vp, sz, C.POINTER(sz)]
    lib.lm_yaml_string.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.lm_scan_load_bytes.argtypes = [vp, i, C.POINTER(C.c_double)]
    lib.lm_synchronize.argtypes = [vp]
    lib.lm_upload_frame_pinned.argtypes = [vp, i, vp, sz, vp, sz]
    lib.lm_upload_wait.argtypes = [vp, i]
    lib.lm_upload_frames_pinned.argtypes = [vp, i, i, vp, sz]
    lib.lm_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.lm_host_free.argtypes = [vp]
    lib.lm_host_free.restype = None
    lib.lm_set_stage_chunks.argtypes = [vp, i]
    lib.lm_set_tuning.argtypes = [vp, i, i]
    lib.lm_comm_init.argtypes = [vp, i, i, C.c_char_p, i, i]
    lib.lm_comm_destroy.argtypes = [vp]
    lib.lm_rendezvous_broadcast.argtypes = [i, i, C.c_char_p, i, vp, sz, i]
    lib.lm_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    lib.lm_match_begin_gathered.argtypes = [vp, i, i, i, f, i]
    lib.lm_match_end_gathered.argtypes = [vp, i, vp, sz, vp, C.POINTER(i), C.POINTER(i), C.POINTER(sz)]
    lib.lm_comm_barrier.argtypes = [vp]
    lib.lm_comm_max.argtypes = [vp, C.POINTER(C.c_double), i]
    lib.lm_match_classes.argtypes = [vp, vp, sz, vp, sz, f, vp, i, vp, sz, C.POINTER(sz)]
    lib.lm_match_batch_classes.argtypes = [vp, i, i, f, vp, i, vp, sz, vp]
    lib.lm_match_prepared.argtypes = [vp, i, i, f, vp, i, vp, sz, vp]
    lib.lm_match_begin_classes.argtypes = [vp, i, i, i, f, vp, i]
    lib.lm_device_pci_bus_id.argtypes = [vp, C.c_char_p, sz]
    lib.lm_get_exchange_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.lm_get_stage_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.lm_get_scan_form_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.lm_get_scan_lane_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.lm_upload_frame_pinned_shifted.argtypes = [vp, i, vp, sz, vp, sz, i, i]
    lib.lm_stage_reserve.argtypes = [vp, i, i]
    lib.lm_stage_rows.argtypes = [vp, i, vp, sz, vp, sz, i, i, i, i]
    lib.lm_upload_staged.argtypes = [vp, i]
    lib.lm_match_collect.argtypes = [vp, i, i, vp, sz, vp]
    lib.lm_color_check_counts_slots.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, sz, vp, vp]
    lib.lm_color_check_begin_slots.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, sz]
    lib.lm_color_check_end.argtypes = [vp, vp, vp]
    lib.lm_depth_counts_begin.argtypes = [vp, vp, C.c_size_t]
    lib.lm_depth_counts_end.argtypes = [vp, vp, vp]
    lib.lm_color_mask_prepare.argtypes = [vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check_out(out, ndim=1):
    """A caller-owned result buffer goes to the library as (pointer, capacity): it must be what the library writes."""
    if not isinstance(out, np.ndarray) or out.dtype != MATCH_DTYPE or not out.flags.c_contiguous or out.ndim != ndim \
            or not out.flags.writeable:
        raise ValueError("out must be a writable C-contiguous %d-d numpy array of MATCH_DTYPE" % ndim)
    return out


def _check_counts(counts, n):
    if not isinstance(counts, np.ndarray) or counts.dtype != np.int32 or not counts.flags.c_contiguous or counts.size < n:
        raise ValueError("counts must be a C-contiguous int32 array with one entry per frame")
    return counts


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def default_config(color_only=False, width=640, height=480, **overrides):
    cfg = Config()
    load_library().lm_default_config(C.byref(cfg), 1 if color_only else 0, width, height)
    for k, v in overrides.items():
        if k == "T":
            cfg.pyramid_levels = len(v)
            for i_, t in enumerate(v):
                cfg.T[i_] = t
        else:
            setattr(cfg, k, v)
    return cfg


def yaml_numbers(path, key):
    """fs[key] of a cv::FileStorage YAML file as a float64 array (scalar, flow sequence or !!opencv-matrix data)."""
    lib = load_library()
    n = C.c_size_t()
    rc = lib.lm_yaml_numbers(str(path).encode(), key.encode(), None, 0, C.byref(n))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    out = np.zeros(n.value, np.float64)
    rc = lib.lm_yaml_numbers(str(path).encode(), key.encode(), _ptr(out), out.size, C.byref(n))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return out


def yaml_string(path, key):
    lib = load_library()
    buf = C.create_string_buffer(4096)
    rc = lib.lm_yaml_string(str(path).encode(), key.encode(), buf, 4096)
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return buf.value.decode()


def rendezvous_broadcast(rank, world, payload, addr="127.0.0.1", port=29511, timeout_s=60):
    """rank 0's `payload` (bytes) to every rank over TCP (what lm_comm_init uses for the ncclUniqueId)."""
    lib = load_library()
    buf = C.create_string_buffer(payload, len(payload))
    rc = lib.lm_rendezvous_broadcast(rank, world, addr.encode(), port, buf, len(payload), timeout_s)
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return buf.raw


def pack_matches(records, counts):
    """[B, cap] per-frame lists + counts -> one contiguous MATCH_DTYPE array (lm_pack_matches)."""
    lib = load_library()
    records = np.ascontiguousarray(records)
    counts = _c(counts, np.int32)
    out = np.zeros(int(counts.sum()), MATCH_DTYPE)
    n = C.c_size_t()
    rc = lib.lm_pack_matches(_ptr(records), records.shape[1], _ptr(counts), len(counts), _ptr(out), out.size, C.byref(n))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return out


def merge_batch(packed, counts, frame_lo=0, frame_hi=None):
    """packed: [R, stride] MATCH_DTYPE (rank r's frames back to back), counts: [R, B] -> (merged packed, counts) of the
    frames [frame_lo, frame_hi) (default: all; lm_merge_frames: per frame R-way merge + adjacent-unique)."""
    lib = load_library()
    packed = np.ascontiguousarray(packed)
    counts = _c(counts, np.int32)
    R, B = counts.shape
    frame_hi = B if frame_hi is None else frame_hi
    out = np.zeros(int(counts[:, frame_lo:frame_hi].sum()), MATCH_DTYPE)
    oc = np.zeros(frame_hi - frame_lo, np.int32)
    n = C.c_size_t()
    rc = lib.lm_merge_frames(_ptr(packed), packed.shape[1], _ptr(counts), R, B, frame_lo, frame_hi, _ptr(out), out.size,
                             _ptr(oc), C.byref(n))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return out[:n.value], oc


def gather_plan(all_cnt, n_ranks, n_frames, rank):
    """lm_gather_plan: the host bookkeeping of lm_match_end_gathered on the all-gathered lengths (n_ranks runs of n_frames + 1
    int32: per-frame counts, then the rank's status word).  Returns a dict: status, bad_rank, f0, f1, counts [R, n],
    piece_start [R], piece_len [R], max_total (records of the largest rank's run)."""
    lib = load_library()
    all_cnt = _c(all_cnt, np.int32)
    if all_cnt.size != n_ranks * (n_frames + 1):
        raise ValueError("all_cnt must hold n_ranks * (n_frames + 1) values")
    st, bad, f0, f1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    counts = np.zeros((n_ranks, n_frames), np.int32)
    ps, pl = np.zeros(n_ranks, np.uint64), np.zeros(n_ranks, np.uint64)
    rc = lib.lm_gather_plan(_ptr(all_cnt), n_ranks, n_frames, rank, C.byref(st), C.byref(bad), C.byref(f0), C.byref(f1), _ptr(counts),
                            _ptr(ps), _ptr(pl))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    mt = C.c_uint64()
    rc = lib.lm_gather_max_total(_ptr(counts), n_ranks, n_frames, C.byref(mt))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return {"status": st.value, "bad_rank": bad.value, "f0": f0.value, "f1": f1.value, "counts": counts,
            "piece_start": ps, "piece_len": pl, "max_total": int(mt.value)}


def merge_matches(lists):
    """R-way merge + adjacent-unique of per-shard sorted match arrays (host side of SURVEY.md 8e)."""
    lib = load_library()
    stride = max([len(l) for l in lists] + [1])
    buf = np.zeros((len(lists), stride), MATCH_DTYPE)
    counts = np.zeros(len(lists), np.int32)
    for k, l in enumerate(lists):
        buf[k, :len(l)] = l
        counts[k] = len(l)
    out = np.zeros(int(counts.sum()) + 1, MATCH_DTYPE)
    n = C.c_size_t()
    rc = lib.lm_merge_matches(_ptr(buf), _ptr(counts), len(lists), stride, _ptr(out), out.size, C.byref(n))
    if rc:
        raise LinemodError(rc, lib.lm_last_error().decode())
    return out[:n.value].copy()


class PinnedBuffer:
    """Pinned host memory from lm_host_alloc, viewed as numpy arrays (sources of Detector.upload_frame_pinned)."""

    def __init__(self, nbytes):
        self.lib = load_library()
        p = C.c_void_p()
        rc = self.lib.lm_host_alloc(nbytes, C.byref(p))
        if rc:
            raise LinemodError(rc, self.lib.lm_last_error().decode())
        self.ptr, self.nbytes = p, nbytes
        self._raw = (C.c_uint8 * nbytes).from_address(p.value)

    def view(self, dtype, shape, offset=0):
        """A numpy view of the block.  Views do NOT own the memory: they dangle after close()."""
        if not self.ptr:
            raise ValueError("PinnedBuffer is closed")
        return np.frombuffer(self._raw, dtype=dtype, count=int(np.prod(shape)), offset=offset).reshape(shape)

    def close(self, detectors=()):
        """Frees the block.  An upload that still reads it must have landed first: pass the detectors that were handed
        views of it (their upload_wait(-1) runs here) or wait yourself."""
        if self.ptr:
            for det in detectors:
                det.upload_wait(-1)
            self._raw = None
            self.lib.lm_host_free(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Detector:
    """cv::linemod::Detector as the reference uses it, on the GPU (see include/linemod_hip.h)."""

    def __init__(self, cfg=None, **kw):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        h = C.c_void_p()
        self._check(self.lib.lm_create(C.byref(self.cfg), C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc:
            raise LinemodError(rc, self.lib.lm_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.lm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries -------------------------------------------------------------------------------
    @property
    def width(self):
        return self.cfg.width

    @property
    def height(self):
        return self.cfg.height

    @property
    def num_modalities(self):
        return self.lib.lm_num_modalities(self.h)

    @property
    def pyramid_levels(self):
        return self.lib.lm_pyramid_levels(self.h)

    def get_T(self, level):
        return self.lib.lm_get_T(self.h, level)

    def num_classes(self):
        return self.lib.lm_num_classes(self.h)

    def num_templates(self):
        return self.lib.lm_num_templates(self.h)

    def class_num_templates(self, ci):
        return self.lib.lm_class_num_templates(self.h, ci)

    def class_ids(self):
        return [self.lib.lm_class_id(self.h, k).decode() for k in range(self.num_classes())]

    def find_class(self, class_id):
        return self.lib.lm_find_class(self.h, class_id.encode())

    # ---- tables --------------------------------------------------------------------------------
    def set_similarity_lut(self, lut):
        lut = _c(lut, np.uint8)
        assert lut.size == 256
        self._check(self.lib.lm_set_similarity_lut(self.h, _ptr(lut)))

    def set_normal_lut(self, lut):
        lut = _c(lut, np.uint8)
        assert lut.size == 8000
        self._check(self.lib.lm_set_normal_lut(self.h, _ptr(lut)))

    def similarity_lut(self):
        out = np.zeros(256, np.uint8)
        self._check(self.lib.lm_get_similarity_lut(self.h, _ptr(out)))
        return out

    def normal_lut_is_substitute(self):
        return bool(self.lib.lm_normal_lut_is_substitute(self.h))

    def normal_lut(self):
        out = np.zeros(8000, np.uint8)
        self._check(self.lib.lm_get_normal_lut(self.h, _ptr(out)))
        return out

    # ---- bank ----------------------------------------------------------------------------------
    def add_class(self, class_id, descs, features):
        descs = _c(descs, DESC_DTYPE)
        features = _c(features, FEATURE_DTYPE)
        per = self.cfg.pyramid_levels * self.cfg.num_modalities
        if descs.size % per or int(descs["num_features"].sum()) != features.size:
            raise ValueError("descs/features do not describe whole template pyramids")
        ci = C.c_int()
        self._check(self.lib.lm_add_class(self.h, class_id.encode(), descs.size // per, _ptr(descs), _ptr(features),
                                          C.byref(ci)))
        return ci.value

    def add_template(self, class_id, bgr, depth=None, mask=None):
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        mask = None if mask is None else _c(mask, np.uint8)
        tid, bb = C.c_int(-1), Rect()
        rc = self.lib.lm_add_template(self.h, class_id.encode(), _ptr(bgr), 0, _ptr(depth), 0, _ptr(mask), 0,
                                      C.byref(tid), C.byref(bb))
        if rc == LM_ERR_EXTRACT:
            return -1, (0, 0, 0, 0)
        self._check(rc)
        return tid.value, (bb.x, bb.y, bb.width, bb.height)

    def get_template(self, ci, tid, level, modality):
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.lm_get_template(self.h, ci, tid, level, modality, C.byref(w), C.byref(h), None, C.byref(n)))
        feats = np.zeros(n.value, FEATURE_DTYPE)
        self._check(self.lib.lm_get_template(self.h, ci, tid, level, modality, C.byref(w), C.byref(h), _ptr(feats),
                                             C.byref(n)))
        return w.value, h.value, feats

    def save_bank(self, path):
        self._check(self.lib.lm_save_bank(self.h, str(path).encode()))

    def load_bank(self, path):
        self._check(self.lib.lm_load_bank(self.h, str(path).encode()))

    def save_yaml(self, path):
        """cv::FileStorage layout of the reference's linemod_templates.yml.gz (gzip when path ends in .gz)."""
        self._check(self.lib.lm_save_yaml(self.h, str(path).encode()))

    def load_yaml(self, path):
        self._check(self.lib.lm_load_yaml(self.h, str(path).encode()))

    # ---- matching ------------------------------------------------------------------------------
    def match(self, bgr, depth, threshold, class_idx=-1, cap=1 << 16, out=None):
        """out: a caller-owned MATCH_DTYPE array to fill (no allocation, the result is a view of it; overflow raises)."""
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        if bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("frame size does not match the detector")
        if out is not None:
            _check_out(out)
            n = C.c_size_t()
            self._check(self.lib.lm_match(self.h, _ptr(bgr), 0, _ptr(depth), 0, threshold, class_idx, _ptr(out), out.size, C.byref(n)))
            return out[:n.value]
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_size_t()
        rc = self.lib.lm_match(self.h, _ptr(bgr), 0, _ptr(depth), 0, threshold, class_idx, _ptr(out), cap, C.byref(n))
        if rc == LM_ERR_OVERFLOW and n.value > cap:
            return self.match(bgr, depth, threshold, class_idx, cap=n.value)
        self._check(rc)
        return out[:n.value].copy()

    def upload_frame(self, slot, bgr, depth=None):
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        if bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("frame size does not match the detector")
        self._check(self.lib.lm_upload_frame(self.h, slot, _ptr(bgr), 0, _ptr(depth), 0))

    def upload_frame_shifted(self, slot, bgr, depth, shift_x, shift_y):
        """upload_frame of the frame translated by (shift_x, shift_y) pixels, zeros shifted in (the reference's principal-point shift)."""
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        if bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("frame size does not match the detector")
        self._check(self.lib.lm_upload_frame_shifted(self.h, slot, _ptr(bgr), 0, _ptr(depth), 0, int(shift_x), int(shift_y)))

    def upload_frame_pinned(self, slot, bgr, depth=None):
        """Source arrays must live in pinned host memory (PinnedBuffer) and stay untouched until upload_wait(slot)
        or until a match that covers the slot has been collected."""
        if bgr.dtype != np.uint8 or not bgr.flags.c_contiguous or bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("pinned colour frame must be a C-contiguous uint8 [h, w, 3] array of the detector's size")
        if depth is not None and (depth.dtype != np.uint16 or not depth.flags.c_contiguous):
            raise ValueError("pinned depth frame must be a C-contiguous uint16 array")
        self._check(self.lib.lm_upload_frame_pinned(self.h, slot, _ptr(bgr), 0, _ptr(depth), 0))

    def upload_frames_pinned(self, first_slot, n_slots, frames_ptr, frame_stride=0):
        """frames_ptr: address (int / c_void_p) of n_slots pinned host frames, each [colour | depth] dense."""
        self._check(self.lib.lm_upload_frames_pinned(self.h, first_slot, n_slots, C.c_void_p(frames_ptr), frame_stride))

    def upload_wait(self, slot=-1):
        self._check(self.lib.lm_upload_wait(self.h, slot))

    def set_stage_chunks(self, chunks):
        self._check(self.lib.lm_set_stage_chunks(self.h, chunks))

    def set_tuning(self, key, value):
        self._check(self.lib.lm_set_tuning(self.h, key, value))

    def match_slot(self, slot, threshold, class_idx=-1, cap=1 << 16, out=None):
        """out: a caller-owned MATCH_DTYPE array to fill (no allocation, the result is a view of it; overflow raises)."""
        if out is not None:
            _check_out(out)
            n = C.c_size_t()
            self._check(self.lib.lm_match_slot(self.h, slot, threshold, class_idx, _ptr(out), out.size, C.byref(n)))
            return out[:n.value]
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_size_t()
        rc = self.lib.lm_match_slot(self.h, slot, threshold, class_idx, _ptr(out), cap, C.byref(n))
        if rc == LM_ERR_OVERFLOW and n.value > cap:
            return self.match_slot(slot, threshold, class_idx, cap=n.value)
        self._check(rc)
        return out[:n.value].copy()

    def match_batch(self, n_slots, threshold, class_idx=-1, cap_per_frame=4096, out=None, counts=None):
        if out is None:
            out = np.zeros((n_slots, cap_per_frame), MATCH_DTYPE)
        if counts is None:
            counts = np.zeros(n_slots, np.int32)
        _check_out(out, 2)
        _check_counts(counts, n_slots)
        if out.shape[0] < n_slots or out.shape[1] != cap_per_frame:
            raise ValueError("out must be [n_slots, cap_per_frame]")
        self._check(self.lib.lm_match_batch(self.h, n_slots, threshold, class_idx, _ptr(out), cap_per_frame, _ptr(counts)))
        return out, counts

    def _class_list(self, classes):
        c = _c([] if classes is None else classes, np.int32).reshape(-1)
        return c, _ptr(c) if c.size else None

    def match_classes(self, bgr, depth, threshold, classes=None, cap=1 << 16):
        """Detector::match(sources, threshold, matches, class_ids) on a host frame with a class list."""
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        if bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("frame size does not match the detector")
        c, cp = self._class_list(classes)
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_size_t()
        rc = self.lib.lm_match_classes(self.h, _ptr(bgr), 0, _ptr(depth), 0, threshold, cp, c.size, _ptr(out), cap, C.byref(n))
        if rc == LM_ERR_OVERFLOW and n.value > cap:
            return self.match_classes(bgr, depth, threshold, classes, cap=n.value)
        self._check(rc)
        return out[:n.value].copy()

    def match_batch_classes(self, first_slot, n_slots, threshold, classes=None, cap_per_frame=4096):
        """Detector::match with upstream's class list: one pre-processing per frame for all the named classes."""
        out = np.zeros((n_slots, cap_per_frame), MATCH_DTYPE)
        counts = np.zeros(n_slots, np.int32)
        c, cp = self._class_list(classes)
        self._check(self.lib.lm_match_batch_classes(self.h, first_slot, n_slots, threshold, cp, c.size, _ptr(out),
                                                    cap_per_frame, _ptr(counts)))
        return out, counts

    def match_prepared(self, first_slot, n_slots, threshold, classes=None, cap_per_frame=4096):
        """a11-a15 only on slots whose a3-a10 results are current (raises LinemodError otherwise)."""
        out = np.zeros((n_slots, cap_per_frame), MATCH_DTYPE)
        counts = np.zeros(n_slots, np.int32)
        c, cp = self._class_list(classes)
        self._check(self.lib.lm_match_prepared(self.h, first_slot, n_slots, threshold, cp, c.size, _ptr(out),
                                               cap_per_frame, _ptr(counts)))
        return out, counts

    def match_begin_classes(self, lane, first_slot, n_slots, threshold, classes=None):
        c, cp = self._class_list(classes)
        self._check(self.lib.lm_match_begin_classes(self.h, lane, first_slot, n_slots, threshold, cp, c.size))

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        self._check(self.lib.lm_device_pci_bus_id(self.h, buf, 64))
        return buf.value.decode()

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self._check(self.lib.lm_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def get_exchange_profile(self):
        """(accumulated HIP-event microseconds of the gathered path's exchange, number of exchanges, lane-steps that needed
        the sized second exchange)."""
        us, n, fb = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self.lib.lm_get_exchange_profile(self.h, C.byref(us), C.byref(n), C.byref(fb)))
        return us.value, n.value, fb.value

    def get_stage_counts(self):
        """dict(preprocess_frames, scan_launches, refine_launches, sort_launches) since set_profiling()."""
        v = (C.c_int64 * 4)()
        self._check(self.lib.lm_get_stage_counts(self.h, v))
        return dict(zip(("preprocess_frames", "scan_launches", "refine_launches", "sort_launches"), list(v)))

    def synchronize(self):
        """hipDeviceSynchronize on the detector's device."""
        self._check(self.lib.lm_synchronize(self.h))

    def match_begin(self, lane, first_slot, n_slots, threshold, class_idx=-1):
        """Enqueue the match of the resident frames in slots [first_slot, first_slot + n_slots) on `lane` (0 or 1)."""
        self._check(self.lib.lm_match_begin(self.h, lane, first_slot, n_slots, threshold, class_idx))

    # ---- multi-GPU exchange (RCCL, include/linemod_hip.h "multi-GPU") ---------------------------
    def comm_init(self, rank, world, addr="127.0.0.1", port=29511, recs_per_frame_cap=0):
        self._check(self.lib.lm_comm_init(self.h, rank, world, addr.encode(), port, recs_per_frame_cap))

    def comm_destroy(self):
        self._check(self.lib.lm_comm_destroy(self.h))

    def comm_barrier(self):
        self._check(self.lib.lm_comm_barrier(self.h))

    def comm_max(self, values):
        v = (C.c_double * len(values))(*values)
        self._check(self.lib.lm_comm_max(self.h, v, len(values)))
        return list(v)

    def match_begin_gathered(self, lane, first_slot, n_slots, threshold, class_idx=-1):
        self._check(self.lib.lm_match_begin_gathered(self.h, lane, first_slot, n_slots, threshold, class_idx))

    def match_end_gathered(self, lane, out, counts):
        """-> (first owned frame, n owned frames, total records): merged lists of the frames this rank owns, back to
        back in `out`, lengths in counts[:n]."""
        _check_out(out)
        _check_counts(counts, 0)
        f0, nf, n = C.c_int(), C.c_int(), C.c_size_t()
        self._check(self.lib.lm_match_end_gathered(self.h, lane, _ptr(out), out.size, _ptr(counts), C.byref(f0),
                                                   C.byref(nf), C.byref(n)))
        return f0.value, nf.value, n.value

    def match_end(self, lane, cap_per_frame=4096, out=None, counts=None, n_slots=None):
        """Wait for `lane` and fetch its lists (same layout as match_batch)."""
        if out is None:
            out = np.zeros((n_slots, cap_per_frame), MATCH_DTYPE)
        if counts is None:
            counts = np.zeros(len(out), np.int32)
        _check_out(out, 2)
        _check_counts(counts, 0)
        if out.shape[1] != cap_per_frame:
            raise ValueError("out must be [n_slots, cap_per_frame]")
        self._check(self.lib.lm_match_end(self.h, lane, _ptr(out), cap_per_frame, _ptr(counts)))
        return out, counts

    # ---- stage hooks ---------------------------------------------------------------------------
    def stage_color_quantize(self, bgr, weak_threshold=10.0, want_magnitude=False):
        bgr = _c(bgr, np.uint8)
        h, w, _ = bgr.shape
        q = np.empty((h, w), np.uint8)
        mag = np.empty((h, w), np.float32) if want_magnitude else None
        self._check(self.lib.lm_stage_color_quantize(self.h, _ptr(bgr), w, h, weak_threshold, _ptr(q), _ptr(mag)))
        return (q, mag) if want_magnitude else q

    def stage_pyrdown(self, bgr):
        bgr = _c(bgr, np.uint8)
        h, w, _ = bgr.shape
        out = np.empty((h // 2, w // 2, 3), np.uint8)
        self._check(self.lib.lm_stage_pyrdown(self.h, _ptr(bgr), w, h, _ptr(out)))
        return out

    def stage_depth_quantize(self, depth):
        depth = _c(depth, np.uint16)
        h, w = depth.shape
        out = np.empty((h, w), np.uint8)
        self._check(self.lib.lm_stage_depth_quantize(self.h, _ptr(depth), w, h, _ptr(out)))
        return out

    def stage_linear_memories(self, quantized, T):
        q = _c(quantized, np.uint8)
        h, w = q.shape
        out = np.empty((8, T * T, (h // T) * (w // T)), np.uint8)
        self._check(self.lib.lm_stage_linear_memories(self.h, _ptr(q), w, h, T, _ptr(out)))
        return out

    def prepare_slot(self, slot):
        self._check(self.lib.lm_prepare_slot(self.h, slot))

    def debug_read(self, slot, what, level, modality):
        n = C.c_size_t()
        self._check(self.lib.lm_debug_read(self.h, slot, what, level, modality, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        self._check(self.lib.lm_debug_read(self.h, slot, what, level, modality, _ptr(out), out.size, C.byref(n)))
        return out

    def stage_scan(self, slot, threshold, class_idx=-1, cap=1 << 20):
        out = np.zeros((cap, 4), np.int32)
        n = C.c_size_t()
        self._check(self.lib.lm_stage_scan(self.h, slot, threshold, class_idx, _ptr(out), cap, C.byref(n)))
        if n.value > cap:
            return self.stage_scan(slot, threshold, class_idx, cap=n.value)
        return out[:n.value].copy()

    # ---- measurement ---------------------------------------------------------------------------
    def time_scan(self, slot, threshold, class_idx=-1, iters=50, variant=0):
        us, by = C.c_double(), C.c_double()
        self._check(self.lib.lm_time_scan(self.h, slot, threshold, class_idx, iters, variant, C.byref(us), C.byref(by)))
        return us.value, by.value

    def time_scan_batch(self, first_slot, n_slots, threshold, class_idx=-1, iters=20, variant=0):
        """Average microseconds of one scan launch over n_slots prepared slots (candidates counted, not stored)."""
        us = C.c_double()
        self._check(self.lib.lm_time_scan_batch(self.h, first_slot, n_slots, threshold, class_idx, iters, variant, C.byref(us)))
        return us.value

    def selftest_float_tail(self):
        """(reciprocals, square roots) of the depth-normal tail's float domain that differ from the correctly rounded forms.
        `self.last_bare_sqrt_mismatches` = floats on which the bare v_sqrt_f32 differs from the correctly rounded root."""
        out = (C.c_uint64 * 8)()
        self._check(self.lib.lm_selftest_float_tail(self.h, out))
        self.last_bare_sqrt_mismatches = int(out[2])
        self.last_candidate_mismatches = {"v_rcp + six steps (r03)": int(out[3]), "v_sqrt + fix-up": int(out[4]), "v_sqrt + v_rsq step": int(out[5]), "1 / root from the root's own v_rsq + one step": int(out[6])}
        return int(out[0]), int(out[1])

    def time_stages(self, slot, threshold, class_idx=-1, iters=20):
        out = (C.c_double * 4)()
        self._check(self.lib.lm_time_stages(self.h, slot, threshold, class_idx, iters, out))
        return list(out)

    def last_counts(self, slot=0):
        """(scan candidates, refined matches before sort+unique) of the last match on `slot`."""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self.lib.lm_last_counts(self.h, slot, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_profiling(self, enable=True):
        self._check(self.lib.lm_set_profiling(self.h, 1 if enable else 0))

    def get_profile(self):
        """dict(stage_us=[preprocess, scan, refine, sort], scan_bytes, launches, frames) accumulated by
        the match calls since set_profiling()."""
        st = (C.c_double * 4)()
        by, la, fr = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self.lib.lm_get_profile(self.h, st, C.byref(by), C.byref(la), C.byref(fr)))
        return dict(stage_us=list(st), scan_bytes=by.value, launches=la.value, frames=fr.value)

    def scan_load_bytes(self, class_idx=-1):
        """Bytes the scan's vector loads request per frame (L2 -> L1 traffic of the hot kernel)."""
        v = C.c_double()
        self._check(self.lib.lm_scan_load_bytes(self.h, class_idx, C.byref(v)))
        return v.value

    def color_check_counts(self, slot, lower_hsv, upper_hsv, matches):
        """(in_hull, in_both) int64 arrays: the two countNonZero of the reference's colorCheck for every match."""
        m = _c(matches, MATCH_DTYPE)
        a, b = np.zeros(len(m), np.int64), np.zeros(len(m), np.int64)
        lo, hi = (C.c_double * 3)(*lower_hsv), (C.c_double * 3)(*upper_hsv)
        self._check(self.lib.lm_color_check_counts(self.h, slot, lo, hi, _ptr(m), len(m), _ptr(a), _ptr(b)))
        return a, b

    def color_check_counts_slots(self, slot_of_match, lower_hsv, upper_hsv, matches):
        """The same for a list whose matches lie in several resident frames: ONE mask launch, one hull launch, one wait."""
        m = _c(matches, MATCH_DTYPE)
        sl = _c(slot_of_match, np.int32)
        if sl.size != len(m):
            raise ValueError("one slot per match")
        a, b = np.zeros(len(m), np.int64), np.zeros(len(m), np.int64)
        lo, hi = (C.c_double * 3)(*lower_hsv), (C.c_double * 3)(*upper_hsv)
        self._check(self.lib.lm_color_check_counts_slots(self.h, _ptr(sl), lo, hi, _ptr(m), len(m), _ptr(a), _ptr(b)))
        return a, b

    def depth_counts(self, queries):
        """r06: per query (DEPTH_QUERY_DTYPE: crop x0, y0, x1, y1 of the frame resident in `slot`, window lo..hi) the number of crop values below lo and
        inside [lo, hi], depths <= 1 counted as 65535 -- the depth check's early verdicts (lm_depth_counts_begin / _end)."""
        q = np.ascontiguousarray(queries, DEPTH_QUERY_DTYPE)
        below = np.zeros(len(q), np.uint32); inside = np.zeros(len(q), np.uint32)
        self._check(self.lib.lm_depth_counts_begin(self.h, _ptr(q) if len(q) else None, len(q)))
        self._check(self.lib.lm_depth_counts_end(self.h, _ptr(below) if len(q) else None, _ptr(inside) if len(q) else None))
        return below, inside

    def color_mask_prepare(self, lane, first_slot, n_slots, lower_hsv, upper_hsv):
        """The slots' colour masks for one HSV range on `lane`'s stream, ahead of the match begun on that lane next."""
        lo, hi = (C.c_double * 3)(*lower_hsv), (C.c_double * 3)(*upper_hsv)
        self._check(self.lib.lm_color_mask_prepare(self.h, lane, first_slot, n_slots, lo, hi))

    def upload_frame_pinned_shifted(self, slot, bgr, depth, shift_x, shift_y):
        """upload_frame_shifted from pinned memory: the DMA engine's row-offset copy, no staging pass."""
        if bgr.dtype != np.uint8 or not bgr.flags.c_contiguous or bgr.shape != (self.cfg.height, self.cfg.width, 3):
            raise ValueError("pinned colour frame must be a C-contiguous uint8 [h, w, 3] array of the detector's size")
        if depth is not None and (depth.dtype != np.uint16 or not depth.flags.c_contiguous):
            raise ValueError("pinned depth frame must be a C-contiguous uint16 array")
        self._check(self.lib.lm_upload_frame_pinned_shifted(self.h, slot, _ptr(bgr), 0, _ptr(depth), 0, int(shift_x), int(shift_y)))

    def stage_reserve(self, first_slot, n_slots):
        self._check(self.lib.lm_stage_reserve(self.h, first_slot, n_slots))

    def stage_rows(self, slot, bgr, depth, shift_x, shift_y, row0, row1):
        """Rows [row0, row1) of both images into the slot's pinned staging buffers (host memory only; any thread)."""
        bgr = _c(bgr, np.uint8)
        depth = None if depth is None else _c(depth, np.uint16)
        self._check(self.lib.lm_stage_rows(self.h, slot, _ptr(bgr), 0, _ptr(depth), 0, int(shift_x), int(shift_y), int(row0), int(row1)))

    def upload_staged(self, slot):
        self._check(self.lib.lm_upload_staged(self.h, slot))

    def match_collect(self, first_slot, n_slots, cap_per_frame=4096):
        """The lists of the last completed match on the slots once more (after LM_ERR_OVERFLOW: with the capacity counts[] named)."""
        out = np.zeros((n_slots, cap_per_frame), MATCH_DTYPE)
        counts = np.zeros(n_slots, np.int32)
        rc = self.lib.lm_match_collect(self.h, first_slot, n_slots, _ptr(out), cap_per_frame, _ptr(counts))
        if rc == LM_ERR_OVERFLOW:
            return None, counts
        self._check(rc)
        return out, counts

    def set_scan_stats(self, enable=True):
        self._check(self.lib.lm_set_scan_stats(self.h, 1 if enable else 0))

    def get_scan_stats(self):
        """(feature loads made, feature loads of an exhaustive scan) since set_scan_stats()."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.lm_get_scan_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get_scan_lane_stats(self):
        """(16-byte lane-loads issued, lane-loads of an exhaustive scan) since set_scan_stats()."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.lm_get_scan_lane_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get_scan_form_stats(self):
        """(scan launches that took the bit-plane kernel, all scan launches, survivors of its miss bound since set_scan_stats(),
        lanes per frame of the last scan launch -- 0: the nibble kernel)."""
        v = (C.c_int64 * 4)()
        self._check(self.lib.lm_get_scan_form_stats(self.h, v))
        return tuple(int(x) for x in v)

    def set_scan_variant(self, variant):
        self._check(self.lib.lm_set_scan_variant(self.h, variant))
