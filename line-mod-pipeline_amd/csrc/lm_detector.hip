// lm_detector.hip -- host side of liblinemod_hip.so: the C ABI of include/linemod_hip.h, the
// template bank (host copy + device upload), resident frame slots and the per-frame launch sequence.
//
// Mirrors cv::linemod::Detector as the reference uses it (/root/reference/src/HighLevelLinemod.cpp:
// 33-34,41-42 ctor; :93 addTemplate; :152 match; :115,181 getTemplates; :55,60,65 class queries).
// There is no CPU fallback: every compute entry point needs a HIP device and fails with
// LM_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/linemod_hip.h"
#include "lm_common.h"
#include "lm_extract.h"
#include "lm_host.h"
#include "lm_kernels.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(LM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    u8* bgr[LM_MAX_LEVELS] = {};
    u16* depth = nullptr;
    u8* quant[LM_MAX_LEVELS][2] = {};
    u8* lm[LM_MAX_LEVELS] = {};
    LmCand* cand = nullptr;
    u64* keys = nullptr;
    lm_match_t* out = nullptr;
    LmHeader* hdr = nullptr;
    int* raw_thr = nullptr;
    float raw_thr_for = -1.0f;  // threshold the device table was built for
    // pinned host staging
    u8* h_bgr = nullptr;
    u16* h_depth = nullptr;
    LmHeader* h_hdr = nullptr;
    lm_match_t* h_out = nullptr;   // inline_n records
    int* h_raw_thr = nullptr;
    bool has_frame = false;
    bool in_flight = false;
};

const u32 INLINE_N = 2048;

}  // namespace

struct lm_detector {
    lm_config cfg;
    LmLevelGeom geom[LM_MAX_LEVELS];
    int lw[LM_MAX_LEVELS], lh[LM_MAX_LEVELS];
    u8 sim_lut[256];
    u8 normal_lut[8000];
    lmh::Bank bank;

    // device state
    bool dev_ready = false;
    std::vector<Slot> slots;
    u64* d_resp_tab = nullptr;
    u8* d_normal_lut = nullptr;
    bool luts_dirty = true;
    // device bank
    bool bank_dirty = true;
    lmh::DeviceBankHost hb;   // host-side arrays of the shard's device bank
    u32* d_item_t = nullptr; u32* d_item_chunk = nullptr;
    u32* d_scan_off = nullptr; int* d_scan_P = nullptr; int* d_scan_n = nullptr;
    int* d_t_global = nullptr; int* d_t_class = nullptr;
    LmRefMeta* d_ref_meta[LM_MAX_LEVELS] = {};
    LmRefFeat* d_ref_feat[LM_MAX_LEVELS] = {};
    // scratch for stage hooks
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    u32 max_cand = 0, max_match = 0;
    int scan_variant = 0;
};

namespace {

void free_device_bank(lm_detector* d) {
    hipFree(d->d_item_t); hipFree(d->d_item_chunk); hipFree(d->d_scan_off); hipFree(d->d_scan_P);
    hipFree(d->d_scan_n); hipFree(d->d_t_global); hipFree(d->d_t_class);
    d->d_item_t = d->d_item_chunk = d->d_scan_off = nullptr;
    d->d_scan_P = d->d_scan_n = d->d_t_global = d->d_t_class = nullptr;
    for (int l = 0; l < LM_MAX_LEVELS; ++l) {
        hipFree(d->d_ref_meta[l]); hipFree(d->d_ref_feat[l]);
        d->d_ref_meta[l] = nullptr; d->d_ref_feat[l] = nullptr;
    }
}

template <typename T>
int upload_vec(T** dptr, const std::vector<T>& v) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return LM_OK;
}

int ensure_device(lm_detector* d) {
    if (d->dev_ready) {
        HIP_TRY(hipSetDevice(d->cfg.device));
        return LM_OK;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(LM_ERR_NO_DEVICE, "no HIP device available: liblinemod_hip has no CPU fallback");
    if (d->cfg.device < 0 || d->cfg.device >= ndev) return fail(LM_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(d->cfg.device));
    const lm_config& c = d->cfg;
    const int M = c.num_modalities, L = c.pyramid_levels;
    d->slots.assign(c.frame_slots, Slot());
    for (Slot& s : d->slots) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        for (auto& ev : s.ev) HIP_TRY(hipEventCreate(&ev));
        for (int l = 0; l < L; ++l) {
            size_t px = (size_t)d->lw[l] * d->lh[l];
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bgr[l]), px * 3));
            for (int m = 0; m < M; ++m) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.quant[l][m]), px));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.lm[l]), d->geom[l].arena_bytes));
            HIP_TRY(hipMemset(s.lm[l], 0, d->geom[l].arena_bytes));  // pads + zero block stay zero forever
        }
        size_t px0 = (size_t)c.width * c.height;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.depth), px0 * 2));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.cand), (size_t)d->max_cand * sizeof(LmCand)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.keys), (size_t)d->max_match * 16));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.out), (size_t)LM_SORT_CAP * sizeof(lm_match_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.hdr), sizeof(LmHeader)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.raw_thr), 128 * sizeof(int)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_bgr), px0 * 3));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_depth), px0 * 2));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_hdr), sizeof(LmHeader)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_out), (size_t)INLINE_N * sizeof(lm_match_t)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_raw_thr), 128 * sizeof(int)));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_resp_tab), 256 * sizeof(u64)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_normal_lut), 8000));
    HIP_TRY(hipDeviceSynchronize());
    d->dev_ready = true;
    d->luts_dirty = true;
    d->bank_dirty = true;
    return LM_OK;
}

int ensure_luts(lm_detector* d) {
    if (!d->luts_dirty) return LM_OK;
    u64 tab[256];
    for (int v = 0; v < 256; ++v) {
        u64 e = 0;
        for (int o = 0; o < 8; ++o) {
            u8 r = std::max(d->sim_lut[32 * o + (v & 15)], d->sim_lut[32 * o + 16 + (v >> 4)]);
            e |= (u64)r << (8 * o);
        }
        tab[v] = e;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(d->d_resp_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->d_normal_lut, d->normal_lut, 8000, hipMemcpyHostToDevice));
    d->luts_dirty = false;
    return LM_OK;
}

int ensure_bank(lm_detector* d) {
    if (!d->bank_dirty) return LM_OK;
    HIP_TRY(hipDeviceSynchronize());
    free_device_bank(d);
    std::string err;
    if (!lmh::build_device_bank(d->bank, d->cfg, d->geom, d->hb, err)) return fail(LM_ERR_INVALID, err);
    int rc;
    if ((rc = upload_vec(&d->d_item_t, d->hb.item_t))) return rc;
    if ((rc = upload_vec(&d->d_item_chunk, d->hb.item_chunk))) return rc;
    if ((rc = upload_vec(&d->d_scan_off, d->hb.scan_off))) return rc;
    if ((rc = upload_vec(&d->d_scan_P, d->hb.scan_P))) return rc;
    if ((rc = upload_vec(&d->d_scan_n, d->hb.scan_n))) return rc;
    if ((rc = upload_vec(&d->d_t_global, d->hb.t_global))) return rc;
    if ((rc = upload_vec(&d->d_t_class, d->hb.t_class))) return rc;
    for (int l = 0; l + 1 < d->cfg.pyramid_levels; ++l) {
        if ((rc = upload_vec(&d->d_ref_meta[l], d->hb.ref_meta[l]))) return rc;
        if ((rc = upload_vec(&d->d_ref_feat[l], d->hb.ref_feat[l]))) return rc;
    }
    d->bank_dirty = false;
    return LM_OK;
}

int check_slot(lm_detector* d, int slot) {
    if (slot < 0 || slot >= (int)d->slots.size()) return fail(LM_ERR_INVALID, "slot out of range");
    return LM_OK;
}

// For levels >= 2 the depth modality's quantised image must exist at level l-1: materialise it.
__global__ void k_nn_half(const u8* __restrict__ src, int sp, u8* __restrict__ dst, int dw, int dh) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < dw && y < dh) dst[(size_t)y * dw + x] = src[(size_t)(2 * y) * sp + 2 * x];
}

void enqueue_depth_pyramid(lm_detector* d, Slot& s) {
    // quant[l][1] for l >= 1 (only read by the stage hooks / levels >= 2 and lm_debug_read)
    for (int l = 1; l < d->cfg.pyramid_levels; ++l) {
        dim3 grid((d->lw[l] + 63) / 64, (d->lh[l] + 3) / 4);
        hipLaunchKernelGGL(k_nn_half, grid, dim3(256), 0, s.stream, s.quant[l - 1][1], d->lw[l - 1], s.quant[l][1],
                           d->lw[l], d->lh[l]);
    }
}

// a3-a10 on the frame resident in the slot.
void enqueue_preprocess(lm_detector* d, Slot& s) {
    const lm_config& c = d->cfg;
    const int M = c.num_modalities, L = c.pyramid_levels;
    for (int l = 0; l < L; ++l) {
        if (l > 0) lmk_pyrdown(s.stream, s.bgr[l - 1], d->lw[l - 1], d->lh[l - 1], s.bgr[l]);
        lmk_color_quantize(s.stream, s.bgr[l], d->lw[l], d->lh[l], c.weak_threshold, s.quant[l][0], nullptr);
        if (M == 2 && l == 0)
            lmk_depth_quantize(s.stream, s.depth, d->lw[0], d->lh[0], c.distance_threshold, c.difference_threshold,
                               d->d_normal_lut, s.quant[0][1]);
    }
    if (M == 2 && L > 2) enqueue_depth_pyramid(d, s);
    for (int l = 0; l < L; ++l) {
        const LmLevelGeom& g = d->geom[l];
        lmk_linear_memories(s.stream, s.quant[l][0], g.w, 0, g.w, g.h, g.T, d->d_resp_tab, s.lm[l], g.ori_stride);
        if (M == 2) {
            // DepthNormalPyramid::pyrDown = NN resize of the quantised image, applied l times: level l
            // reads level 0 at (y << l, x << l); with l in {0,1} that is the SRC_SHIFT template.
            if (l == 0)
                lmk_linear_memories(s.stream, s.quant[0][1], g.w, 0, g.w, g.h, g.T, d->d_resp_tab,
                                    s.lm[l] + g.mod_stride, g.ori_stride);
            else
                lmk_linear_memories(s.stream, s.quant[l - 1][1], d->lw[l - 1], 1, g.w, g.h, g.T, d->d_resp_tab,
                                    s.lm[l] + g.mod_stride, g.ori_stride);
        }
    }
}

void fill_raw_thr(int* tab, float threshold) {
    // A.7: int(2n + (t/100)*2n + 0.5f) in float arithmetic (file is built with -ffp-contract=off)
    for (int n = 0; n < 128; ++n) tab[n] = static_cast<int>(2 * n + (threshold / 100.f) * (2 * n) + 0.5f);
}

struct ItemRange { int lo, n; };
int item_range(lm_detector* d, int class_idx, ItemRange* r) {
    const int nc = (int)d->bank.classes.size();
    if (class_idx >= nc || class_idx < -1) return fail(LM_ERR_INVALID, "class index out of range");
    if (class_idx < 0) { r->lo = 0; r->n = (int)d->hb.item_t.size(); }
    else { r->lo = d->hb.class_item_lo[class_idx]; r->n = d->hb.class_item_hi[class_idx] - r->lo; }
    return LM_OK;
}

LmScanArgs make_scan_args(lm_detector* d, Slot& s, ItemRange r) {
    const int L = d->cfg.pyramid_levels;
    const LmLevelGeom& g = d->geom[L - 1];
    LmScanArgs a;
    a.lm = s.lm[L - 1];
    a.item_t = d->d_item_t; a.item_chunk = d->d_item_chunk;
    a.item_lo = r.lo; a.n_items = r.n;
    a.scan_off = d->d_scan_off; a.scan_P = d->d_scan_P; a.scan_n = d->d_scan_n;
    a.M = d->cfg.num_modalities; a.fpad = d->hb.fpad;
    a.raw_thr_by_n = s.raw_thr;
    a.W = g.W; a.T = g.T;
    a.cand = s.cand; a.cand_count = &s.hdr->cand_count; a.cand_cap = d->max_cand;
    return a;
}

LmRefineArgs make_refine_args(lm_detector* d, Slot& s, int level, float threshold) {
    LmRefineArgs a;
    a.lm = s.lm[level];
    a.g = d->geom[level];
    a.M = d->cfg.num_modalities;
    a.meta = d->d_ref_meta[level]; a.feats = d->d_ref_feat[level];
    a.cand = s.cand; a.cand_count = &s.hdr->cand_count; a.cand_cap = d->max_cand;
    a.threshold = threshold;
    a.t_global = d->d_t_global; a.t_class = d->d_t_class;
    a.keys = s.keys; a.match_count = &s.hdr->match_count; a.match_cap = d->max_match;
    return a;
}

int enqueue_threshold(lm_detector* d, Slot& s, float threshold) {
    if (!(threshold >= 0.0f)) return fail(LM_ERR_INVALID, "threshold must be >= 0");
    if (s.raw_thr_for != threshold) {
        HIP_TRY(hipStreamSynchronize(s.stream));  // pinned table may still be read by an earlier copy
        fill_raw_thr(s.h_raw_thr, threshold);
        HIP_TRY(hipMemcpyAsync(s.raw_thr, s.h_raw_thr, 128 * sizeof(int), hipMemcpyHostToDevice, s.stream));
        s.raw_thr_for = threshold;
    }
    return LM_OK;
}

// a11-a15 on the slot's prepared linear memories; ends with the D2H of header + first records.
int enqueue_match_stages(lm_detector* d, Slot& s, float threshold, ItemRange r, bool timed) {
    const int L = d->cfg.pyramid_levels;
    HIP_TRY(hipMemsetAsync(s.hdr, 0, sizeof(LmHeader), s.stream));
    if (timed) HIP_TRY(hipEventRecord(s.ev[1], s.stream));
    lmk_scan(s.stream, make_scan_args(d, s, r), d->scan_variant);
    if (timed) HIP_TRY(hipEventRecord(s.ev[2], s.stream));
    if (L == 1) {
        lmk_emit_unrefined(s.stream, make_refine_args(d, s, 0, threshold));
    } else {
        for (int l = L - 2; l >= 0; --l) lmk_refine(s.stream, make_refine_args(d, s, l, threshold), l == 0);
    }
    if (timed) HIP_TRY(hipEventRecord(s.ev[3], s.stream));
    lmk_sort_unique(s.stream, s.keys, &s.hdr->match_count, d->max_match, s.out, s.hdr);
    HIP_TRY(hipMemcpyAsync(s.h_hdr, s.hdr, sizeof(LmHeader), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(hipMemcpyAsync(s.h_out, s.out, (size_t)INLINE_N * sizeof(lm_match_t), hipMemcpyDeviceToHost, s.stream));
    if (timed) HIP_TRY(hipEventRecord(s.ev[4], s.stream));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int enqueue_match(lm_detector* d, Slot& s, float threshold, int class_idx, bool timed = false) {
    ItemRange r;
    int rc;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, s, threshold))) return rc;
    if (timed) HIP_TRY(hipEventRecord(s.ev[0], s.stream));
    enqueue_preprocess(d, s);
    if ((rc = enqueue_match_stages(d, s, threshold, r, timed))) return rc;
    s.in_flight = true;
    return LM_OK;
}

inline bool key_less(const u64* a, const u64* b) { return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]); }

// Waits for the slot and delivers the sorted unique matches.
int finish_match(lm_detector* d, Slot& s, lm_match_t* out, size_t cap, size_t* n_out) {
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.in_flight = false;
    const LmHeader h = *s.h_hdr;
    if (h.cand_count > d->max_cand)
        return fail(LM_ERR_OVERFLOW, "scan produced " + std::to_string(h.cand_count) + " candidates, capacity " +
                                         std::to_string(d->max_cand) + " (raise lm_config.max_candidates)");
    if (h.match_count > d->max_match)
        return fail(LM_ERR_OVERFLOW, "refinement produced " + std::to_string(h.match_count) + " matches, capacity " +
                                         std::to_string(d->max_match) + " (raise lm_config.max_matches)");
    size_t n = 0;
    if (h.sorted_on_device) {
        n = h.out_count;
        size_t ncopy = std::min(n, cap);
        size_t inl = std::min<size_t>(ncopy, INLINE_N);
        if (out && inl) std::memcpy(out, s.h_out, inl * sizeof(lm_match_t));
        if (out && ncopy > inl)
            HIP_TRY(hipMemcpy(out + inl, s.out + inl, (ncopy - inl) * sizeof(lm_match_t), hipMemcpyDeviceToHost));
    } else {
        // more than LM_SORT_CAP matches: sort + unique the keys on the host (same total order)
        std::vector<u64> keys((size_t)h.match_count * 2);
        HIP_TRY(hipMemcpy(keys.data(), s.keys, keys.size() * sizeof(u64), hipMemcpyDeviceToHost));
        std::vector<u32> idx(h.match_count);
        for (u32 i = 0; i < h.match_count; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return key_less(&keys[2 * (size_t)a], &keys[2 * (size_t)b]); });
        const u64* prev = nullptr;
        for (u32 i : idx) {
            const u64* k = &keys[2 * (size_t)i];
            if (prev && prev[1] == k[1] && (prev[0] >> 32) == (k[0] >> 32)) { prev = k; continue; }
            prev = k;
            if (out && n < cap) {
                lm_match_t m;
                u32 sb = ~(u32)(k[0] >> 32);
                std::memcpy(&m.similarity, &sb, 4);
                m.template_id = (int)(u32)k[0];
                m.class_idx = (int)(k[1] >> 48);
                m.y = (int)((k[1] >> 24) & 0xFFFFFFu) - 0x800000;
                m.x = (int)(k[1] & 0xFFFFFFu) - 0x800000;
                out[n] = m;
            }
            ++n;
        }
    }
    if (n_out) *n_out = n;
    if (n > cap && out) return fail(LM_ERR_OVERFLOW, "output buffer too small for " + std::to_string(n) + " matches");
    return LM_OK;
}

int upload_frame(lm_detector* d, Slot& s, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                 size_t depth_stride) {
    const lm_config& c = d->cfg;
    if (!bgr) return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): colour image missing");
    if (c.num_modalities == 2 && !depth)
        return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): depth image missing");
    if (bgr_stride == 0) bgr_stride = (size_t)c.width * 3;
    if (depth_stride == 0) depth_stride = (size_t)c.width * 2;
    if (bgr_stride < (size_t)c.width * 3 || depth_stride < (size_t)c.width * 2) return fail(LM_ERR_INVALID, "stride smaller than a row");
    HIP_TRY(hipStreamSynchronize(s.stream));  // staging buffers are reused
    for (int y = 0; y < c.height; ++y) std::memcpy(s.h_bgr + (size_t)y * c.width * 3, bgr + y * bgr_stride, (size_t)c.width * 3);
    HIP_TRY(hipMemcpyAsync(s.bgr[0], s.h_bgr, (size_t)c.width * c.height * 3, hipMemcpyHostToDevice, s.stream));
    if (c.num_modalities == 2) {
        for (int y = 0; y < c.height; ++y)
            std::memcpy(s.h_depth + (size_t)y * c.width, reinterpret_cast<const u8*>(depth) + y * depth_stride, (size_t)c.width * 2);
        HIP_TRY(hipMemcpyAsync(s.depth, s.h_depth, (size_t)c.width * c.height * 2, hipMemcpyHostToDevice, s.stream));
    }
    s.has_frame = true;
    return LM_OK;
}

int ready_for_compute(lm_detector* d) {
    int rc;
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if ((rc = ensure_device(d))) return rc;
    if ((rc = ensure_luts(d))) return rc;
    return LM_OK;
}

int ensure_scratch(lm_detector* d, size_t bytes) {
    if (bytes <= d->scratch_bytes) return LM_OK;
    HIP_TRY(hipDeviceSynchronize());
    hipFree(d->d_scratch);
    d->d_scratch = nullptr; d->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&d->d_scratch, bytes));
    d->scratch_bytes = bytes;
    return LM_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* lm_last_error(void) { return g_err.c_str(); }
const char* lm_version(void) { return "linemod_hip 0.1 (gfx950)"; }

void lm_default_config(lm_config* c, int color_only, int width, int height) {
    std::memset(c, 0, sizeof(*c));
    c->width = width; c->height = height;
    c->num_modalities = color_only ? 1 : 2;
    c->pyramid_levels = 2;
    c->T[0] = color_only ? 2 : 5;
    c->T[1] = 8;
    c->weak_threshold = 10.0f; c->num_features = 63; c->strong_threshold = 55.0f;
    c->distance_threshold = 2000; c->difference_threshold = 50; c->depth_num_features = 63; c->extract_threshold = 2;
    c->device = 0; c->shard_rank = 0; c->shard_size = 1;
    c->max_candidates = 0; c->max_matches = 0; c->frame_slots = 0;
}

int lm_create(const lm_config* cfg, lm_detector** out) {
    if (!cfg || !out) return fail(LM_ERR_INVALID, "null argument");
    *out = nullptr;
    lm_config c = *cfg;
    if (c.num_modalities < 1 || c.num_modalities > 2) return fail(LM_ERR_INVALID, "num_modalities must be 1 or 2");
    if (c.pyramid_levels < 1 || c.pyramid_levels > LM_MAX_LEVELS) return fail(LM_ERR_INVALID, "pyramid_levels out of range");
    if (c.width <= 0 || c.height <= 0) return fail(LM_ERR_INVALID, "bad frame size");
    if (c.shard_size < 1 || c.shard_rank < 0 || c.shard_rank >= c.shard_size) return fail(LM_ERR_INVALID, "bad shard rank/size");
    if (c.max_candidates <= 0) c.max_candidates = 1 << 20;
    if (c.max_matches <= 0) c.max_matches = 1 << 18;
    if (c.frame_slots <= 0) c.frame_slots = 8;
    lm_detector* d = new lm_detector();
    d->cfg = c;
    d->max_cand = (u32)c.max_candidates;
    d->max_match = (u32)c.max_matches;
    int w = c.width, h = c.height;
    for (int l = 0; l < c.pyramid_levels; ++l) {
        if (l > 0) { w /= 2; h /= 2; }
        int T = c.T[l];
        // CV_Assert(rows % T == 0 && cols % T == 0) in linearize, (rows*cols) % 16 == 0 in computeResponseMaps
        if (T <= 0 || w <= 0 || h <= 0 || w % T || h % T || ((long long)w * h) % 16) {
            delete d;
            return fail(LM_ERR_INVALID, "frame size violates cols%T==0, rows%T==0, (rows*cols)%16==0 at level " + std::to_string(l));
        }
        d->lw[l] = w; d->lh[l] = h;
        LmLevelGeom& g = d->geom[l];
        g.w = w; g.h = h; g.T = T; g.W = w / T; g.H = h / T;
        g.wh = (u32)g.W * (u32)g.H;
        size_t pad = align_up((size_t)g.wh + 2 * LM_SCAN_CHUNK + 64, 256);
        size_t ori = align_up((size_t)T * T * g.wh, 256) + pad;
        g.ori_stride = (u32)ori;
        g.mod_stride = (u32)(8 * ori);
        g.zero_off = (u32)((size_t)c.num_modalities * g.mod_stride);
        size_t arena = (size_t)g.zero_off + pad;
        if (arena > 0xFFFFFFFFull) { delete d; return fail(LM_ERR_INVALID, "frame too large for 32-bit arena offsets"); }
        g.arena_bytes = (u32)arena;
    }
    lmh::default_similarity_lut(d->sim_lut);
    lmh::default_normal_lut(d->normal_lut);
    *out = d;
    return LM_OK;
}

void lm_destroy(lm_detector* d) {
    if (!d) return;
    if (d->dev_ready) {
        hipSetDevice(d->cfg.device);
        hipDeviceSynchronize();
        for (Slot& s : d->slots) {
            for (int l = 0; l < LM_MAX_LEVELS; ++l) {
                hipFree(s.bgr[l]); hipFree(s.lm[l]);
                for (int m = 0; m < 2; ++m) hipFree(s.quant[l][m]);
            }
            hipFree(s.depth); hipFree(s.cand); hipFree(s.keys); hipFree(s.out); hipFree(s.hdr); hipFree(s.raw_thr);
            hipHostFree(s.h_bgr); hipHostFree(s.h_depth); hipHostFree(s.h_hdr); hipHostFree(s.h_out); hipHostFree(s.h_raw_thr);
            for (auto& ev : s.ev) if (ev) hipEventDestroy(ev);
            if (s.stream) hipStreamDestroy(s.stream);
        }
        free_device_bank(d);
        hipFree(d->d_resp_tab); hipFree(d->d_normal_lut); hipFree(d->d_scratch);
    }
    delete d;
}

int lm_set_similarity_lut(lm_detector* d, const uint8_t lut[256]) {
    if (!d || !lut) return fail(LM_ERR_INVALID, "null argument");
    for (int i = 0; i < 256; ++i) if (lut[i] > 4) return fail(LM_ERR_INVALID, "similarity LUT entries must be <= 4 (63*4 must fit a byte)");
    std::memcpy(d->sim_lut, lut, 256); d->luts_dirty = true; return LM_OK;
}
int lm_set_normal_lut(lm_detector* d, const uint8_t lut[8000]) {
    if (!d || !lut) return fail(LM_ERR_INVALID, "null argument");
    std::memcpy(d->normal_lut, lut, 8000); d->luts_dirty = true; return LM_OK;
}
int lm_get_similarity_lut(const lm_detector* d, uint8_t lut[256]) { if (!d || !lut) return fail(LM_ERR_INVALID, "null argument"); std::memcpy(lut, d->sim_lut, 256); return LM_OK; }
int lm_get_normal_lut(const lm_detector* d, uint8_t lut[8000]) { if (!d || !lut) return fail(LM_ERR_INVALID, "null argument"); std::memcpy(lut, d->normal_lut, 8000); return LM_OK; }

int lm_num_classes(const lm_detector* d) { return d ? (int)d->bank.classes.size() : -1; }
int lm_num_templates(const lm_detector* d) {
    if (!d) return -1;
    int n = 0;
    for (const auto& c : d->bank.classes) n += (int)c.pyramids.size();
    return n;
}
int lm_class_num_templates(const lm_detector* d, int ci) {
    if (!d || ci < 0 || ci >= (int)d->bank.classes.size()) return -1;
    return (int)d->bank.classes[ci].pyramids.size();
}
const char* lm_class_id(const lm_detector* d, int ci) {
    if (!d || ci < 0 || ci >= (int)d->bank.classes.size()) return nullptr;
    return d->bank.classes[ci].id.c_str();
}
int lm_find_class(const lm_detector* d, const char* id) { return (d && id) ? d->bank.find(id) : -1; }
int lm_get_T(const lm_detector* d, int level) { return (d && level >= 0 && level < d->cfg.pyramid_levels) ? d->cfg.T[level] : -1; }
int lm_num_modalities(const lm_detector* d) { return d ? d->cfg.num_modalities : -1; }
int lm_pyramid_levels(const lm_detector* d) { return d ? d->cfg.pyramid_levels : -1; }

int lm_add_class(lm_detector* d, const char* class_id, int n_templates, const lm_template_desc* descs,
                 const lm_feature* features, int* class_idx_out) {
    if (!d || !class_id || n_templates < 0 || (n_templates && (!descs || !features))) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    int ci = d->bank.add_class(class_id, n_templates, descs, features, d->cfg.pyramid_levels, d->cfg.num_modalities, err);
    if (ci < 0) return fail(LM_ERR_INVALID, err);
    if (class_idx_out) *class_idx_out = ci;
    d->bank_dirty = true;
    return LM_OK;
}

int lm_add_template(lm_detector* d, const char* class_id, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                    size_t depth_stride, const uint8_t* mask, size_t mask_stride, int* template_id_out, lm_rect* bbox_out) {
    if (template_id_out) *template_id_out = -1;
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!class_id) return fail(LM_ERR_INVALID, "null class id");
    const lm_config& c = d->cfg;
    const int M = c.num_modalities, L = c.pyramid_levels;
    Slot& s = d->slots[0];
    if ((rc = upload_frame(d, s, bgr, bgr_stride, depth, depth_stride))) return rc;
    s.has_frame = false;  // slot 0 now holds a template image, not a scene frame
    // quantise every level on the GPU, keeping the gradient magnitude this time
    size_t mag_off[LM_MAX_LEVELS], total = 0;
    for (int l = 0; l < L; ++l) { mag_off[l] = total; total += align_up((size_t)d->lw[l] * d->lh[l] * sizeof(float), 256); }
    if ((rc = ensure_scratch(d, total))) return rc;
    u8* scratch = static_cast<u8*>(d->d_scratch);
    for (int l = 0; l < L; ++l) {
        if (l > 0) lmk_pyrdown(s.stream, s.bgr[l - 1], d->lw[l - 1], d->lh[l - 1], s.bgr[l]);
        lmk_color_quantize(s.stream, s.bgr[l], d->lw[l], d->lh[l], c.weak_threshold, s.quant[l][0],
                           reinterpret_cast<float*>(scratch + mag_off[l]));
    }
    if (M == 2) {
        lmk_depth_quantize(s.stream, s.depth, d->lw[0], d->lh[0], c.distance_threshold, c.difference_threshold,
                           d->d_normal_lut, s.quant[0][1]);
        enqueue_depth_pyramid(d, s);
    }
    std::vector<lmh::ExtractLevel> lv(L);
    for (int l = 0; l < L; ++l) {
        size_t px = (size_t)d->lw[l] * d->lh[l];
        lv[l].w = d->lw[l]; lv[l].h = d->lh[l];
        lv[l].color_q.resize(px); lv[l].color_mag.resize(px);
        HIP_TRY(hipMemcpyAsync(lv[l].color_q.data(), s.quant[l][0], px, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(lv[l].color_mag.data(), scratch + mag_off[l], px * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        if (M == 2) {
            lv[l].depth_q.resize(px);
            HIP_TRY(hipMemcpyAsync(lv[l].depth_q.data(), s.quant[l][1], px, hipMemcpyDeviceToHost, s.stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(s.stream));
    HIP_TRY(hipGetLastError());
    if (mask) {  // mask pyramid: resize(..., INTER_NEAREST) per level
        if (mask_stride == 0) mask_stride = (size_t)c.width;
        lv[0].mask.resize((size_t)c.width * c.height);
        for (int y = 0; y < c.height; ++y) std::memcpy(&lv[0].mask[(size_t)y * c.width], mask + y * mask_stride, (size_t)c.width);
        for (int l = 1; l < L; ++l) {
            lv[l].mask.resize((size_t)lv[l].w * lv[l].h);
            for (int y = 0; y < lv[l].h; ++y)
                for (int x = 0; x < lv[l].w; ++x) lv[l].mask[(size_t)y * lv[l].w + x] = lv[l - 1].mask[(size_t)(2 * y) * lv[l - 1].w + 2 * x];
        }
    }
    lmh::TemplatePyramid tp;
    if (!lmh::extract_pyramid(lv, c, tp)) return fail(LM_ERR_EXTRACT, "not enough features to build a template");
    lm_rect bb = lmh::crop_templates(tp);
    if (bbox_out) *bbox_out = bb;
    int tid = d->bank.add_pyramid(class_id, std::move(tp));
    if (template_id_out) *template_id_out = tid;
    d->bank_dirty = true;
    return LM_OK;
}

int lm_get_template(const lm_detector* d, int ci, int tid, int level, int modality, int* width, int* height,
                    lm_feature* features, int* num_features) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (ci < 0 || ci >= (int)d->bank.classes.size()) return fail(LM_ERR_INVALID, "class index out of range");
    const auto& c = d->bank.classes[ci];
    if (tid < 0 || tid >= (int)c.pyramids.size()) return fail(LM_ERR_INVALID, "template id out of range");
    if (level < 0 || level >= d->cfg.pyramid_levels || modality < 0 || modality >= d->cfg.num_modalities)
        return fail(LM_ERR_INVALID, "level/modality out of range");
    const lmh::Template& t = c.pyramids[tid][level * d->cfg.num_modalities + modality];
    if (width) *width = t.width;
    if (height) *height = t.height;
    if (num_features) *num_features = (int)t.features.size();
    if (features) std::memcpy(features, t.features.data(), t.features.size() * sizeof(lm_feature));
    return LM_OK;
}

int lm_upload_frame(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                    size_t depth_stride) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    return upload_frame(d, d->slots[slot], bgr, bgr_stride, depth, depth_stride);
}

int lm_match_slot(lm_detector* d, int slot, float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    Slot& s = d->slots[slot];
    if (!s.has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    if ((rc = enqueue_match(d, s, threshold, class_idx))) return rc;
    return finish_match(d, s, out, cap, n_out);
}

int lm_match(lm_detector* d, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
             float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    Slot& s = d->slots[0];
    if ((rc = upload_frame(d, s, bgr, bgr_stride, depth, depth_stride))) return rc;
    if ((rc = enqueue_match(d, s, threshold, class_idx))) return rc;
    return finish_match(d, s, out, cap, n_out);
}

int lm_match_batch(lm_detector* d, int n_slots, float threshold, int class_idx, lm_match_t* out, size_t cap_per_frame,
                   int32_t* counts) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (n_slots < 0 || n_slots > (int)d->slots.size()) return fail(LM_ERR_INVALID, "n_slots exceeds frame_slots");
    for (int i = 0; i < n_slots; ++i) {
        if (!d->slots[i].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot " + std::to_string(i));
        if ((rc = enqueue_match(d, d->slots[i], threshold, class_idx))) return rc;
    }
    int first_err = LM_OK;
    std::string first_msg;
    for (int i = 0; i < n_slots; ++i) {
        size_t n = 0;
        rc = finish_match(d, d->slots[i], out ? out + (size_t)i * cap_per_frame : nullptr, cap_per_frame, &n);
        if (counts) counts[i] = (int32_t)n;
        if (rc && !first_err) { first_err = rc; first_msg = g_err; }
    }
    if (first_err) return fail(first_err, first_msg);
    return LM_OK;
}

int lm_merge_matches(const lm_match_t* lists, const int32_t* counts, int n_lists, size_t stride, lm_match_t* out,
                     size_t cap, size_t* n_out) {
    if (!lists || !counts || n_lists < 0) return fail(LM_ERR_INVALID, "null argument");
    std::vector<lm_match_t> all;
    for (int i = 0; i < n_lists; ++i) {
        if (counts[i] < 0 || (size_t)counts[i] > stride) return fail(LM_ERR_INVALID, "count exceeds stride");
        all.insert(all.end(), lists + (size_t)i * stride, lists + (size_t)i * stride + counts[i]);
    }
    lmh::sort_unique(all);
    if (n_out) *n_out = all.size();
    if (out) std::memcpy(out, all.data(), std::min(all.size(), cap) * sizeof(lm_match_t));
    if (all.size() > cap && out) return fail(LM_ERR_OVERFLOW, "output buffer too small");
    return LM_OK;
}

int lm_save_bank(const lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::save_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}
int lm_load_bank(lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::load_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    d->bank_dirty = true;
    return LM_OK;
}

// ---- stage hooks ---------------------------------------------------------------------------------
int lm_stage_color_quantize(lm_detector* d, const uint8_t* bgr, int w, int h, float weak_threshold, uint8_t* quantized,
                            float* magnitude) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!bgr || !quantized || w <= 0 || h <= 0) return fail(LM_ERR_INVALID, "bad argument");
    size_t px = (size_t)w * h;
    size_t o_q = align_up(px * 3, 256), o_m = o_q + align_up(px, 256);
    if ((rc = ensure_scratch(d, o_m + px * 4))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->slots[0].stream;
    HIP_TRY(hipMemcpyAsync(base, bgr, px * 3, hipMemcpyHostToDevice, st));
    lmk_color_quantize(st, base, w, h, weak_threshold, base + o_q, magnitude ? reinterpret_cast<float*>(base + o_m) : nullptr);
    HIP_TRY(hipMemcpyAsync(quantized, base + o_q, px, hipMemcpyDeviceToHost, st));
    if (magnitude) HIP_TRY(hipMemcpyAsync(magnitude, base + o_m, px * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_stage_pyrdown(lm_detector* d, const uint8_t* bgr, int w, int h, uint8_t* out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!bgr || !out || w < 2 || h < 2) return fail(LM_ERR_INVALID, "bad argument");
    size_t px = (size_t)w * h, opx = (size_t)(w / 2) * (h / 2);
    size_t o_o = align_up(px * 3, 256);
    if ((rc = ensure_scratch(d, o_o + opx * 3))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->slots[0].stream;
    HIP_TRY(hipMemcpyAsync(base, bgr, px * 3, hipMemcpyHostToDevice, st));
    lmk_pyrdown(st, base, w, h, base + o_o);
    HIP_TRY(hipMemcpyAsync(out, base + o_o, opx * 3, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_stage_depth_quantize(lm_detector* d, const uint16_t* depth, int w, int h, uint8_t* quantized) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!depth || !quantized || w <= 0 || h <= 0) return fail(LM_ERR_INVALID, "bad argument");
    size_t px = (size_t)w * h;
    size_t o_q = align_up(px * 2, 256);
    if ((rc = ensure_scratch(d, o_q + px))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->slots[0].stream;
    HIP_TRY(hipMemcpyAsync(base, depth, px * 2, hipMemcpyHostToDevice, st));
    lmk_depth_quantize(st, reinterpret_cast<u16*>(base), w, h, d->cfg.distance_threshold, d->cfg.difference_threshold,
                       d->d_normal_lut, base + o_q);
    HIP_TRY(hipMemcpyAsync(quantized, base + o_q, px, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_stage_linear_memories(lm_detector* d, const uint8_t* quantized, int w, int h, int T, uint8_t* out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!quantized || !out || w <= 0 || h <= 0 || T <= 0 || w % T || h % T) return fail(LM_ERR_INVALID, "bad argument");
    size_t px = (size_t)w * h;
    size_t o_l = align_up(px, 256);
    if ((rc = ensure_scratch(d, o_l + 8 * px))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->slots[0].stream;
    HIP_TRY(hipMemcpyAsync(base, quantized, px, hipMemcpyHostToDevice, st));
    lmk_linear_memories(st, base, w, 0, w, h, T, d->d_resp_tab, base + o_l, (u32)px);  // dense: ori_stride = T*T*W*H
    HIP_TRY(hipMemcpyAsync(out, base + o_l, 8 * px, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_prepare_slot(lm_detector* d, int slot) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    Slot& s = d->slots[slot];
    if (!s.has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    enqueue_preprocess(d, s);
    HIP_TRY(hipStreamSynchronize(s.stream));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_debug_read(lm_detector* d, int slot, int what, int level, int modality, uint8_t* out, size_t cap, size_t* size_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    if (level < 0 || level >= d->cfg.pyramid_levels || modality < 0 || modality >= d->cfg.num_modalities)
        return fail(LM_ERR_INVALID, "level/modality out of range");
    Slot& s = d->slots[slot];
    HIP_TRY(hipStreamSynchronize(s.stream));
    const LmLevelGeom& g = d->geom[level];
    if (what == 0) {
        size_t n = (size_t)g.w * g.h;
        if (size_out) *size_out = n;
        if (modality == 1 && level > 0) {  // materialise the NN pyramid of the depth modality on demand
            enqueue_depth_pyramid(d, s);
            HIP_TRY(hipStreamSynchronize(s.stream));
        }
        if (out) HIP_TRY(hipMemcpy(out, s.quant[level][modality], std::min(n, cap), hipMemcpyDeviceToHost));
        return LM_OK;
    }
    if (what == 2) {
        size_t blk = (size_t)g.T * g.T * g.wh;
        size_t n = 8 * blk;
        if (size_out) *size_out = n;
        if (out) {
            if (cap < n) return fail(LM_ERR_INVALID, "buffer too small");
            for (int o = 0; o < 8; ++o)
                HIP_TRY(hipMemcpy(out + o * blk, s.lm[level] + (size_t)modality * g.mod_stride + (size_t)o * g.ori_stride, blk,
                                  hipMemcpyDeviceToHost));
        }
        return LM_OK;
    }
    return fail(LM_ERR_INVALID, "unknown buffer id");
}

int lm_stage_scan(lm_detector* d, int slot, float threshold, int class_idx, int32_t* out, size_t cap_records, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    Slot& s = d->slots[slot];
    ItemRange r;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, s, threshold))) return rc;
    HIP_TRY(hipMemsetAsync(s.hdr, 0, sizeof(LmHeader), s.stream));
    lmk_scan(s.stream, make_scan_args(d, s, r), d->scan_variant);
    LmHeader h;
    HIP_TRY(hipMemcpyAsync(s.h_hdr, s.hdr, sizeof(LmHeader), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));
    HIP_TRY(hipGetLastError());
    h = *s.h_hdr;
    if (h.cand_count > d->max_cand) return fail(LM_ERR_OVERFLOW, "candidate buffer overflow");
    std::vector<LmCand> cand(h.cand_count);
    if (h.cand_count) HIP_TRY(hipMemcpy(cand.data(), s.cand, cand.size() * sizeof(LmCand), hipMemcpyDeviceToHost));
    struct Rec { int32_t tid, cls, x, y; };
    std::vector<Rec> recs(cand.size());
    for (size_t i = 0; i < cand.size(); ++i)
        recs[i] = Rec{d->hb.t_global[cand[i].ti], d->hb.t_class[cand[i].ti], cand[i].x, cand[i].y};
    std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
        if (a.cls != b.cls) return a.cls < b.cls;
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.y != b.y) return a.y < b.y;
        return a.x < b.x;
    });
    if (n_out) *n_out = recs.size();
    if (out) std::memcpy(out, recs.data(), std::min(recs.size(), cap_records) * sizeof(Rec));
    return LM_OK;
}

int lm_time_scan(lm_detector* d, int slot, float threshold, int class_idx, int iters, int variant, double* avg_us_out,
                 double* algorithmic_bytes_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (iters <= 0) return fail(LM_ERR_INVALID, "iters must be positive");
    Slot& s = d->slots[slot];
    ItemRange r;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, s, threshold))) return rc;
    LmScanArgs a = make_scan_args(d, s, r);
    HIP_TRY(hipMemsetAsync(s.hdr, 0, sizeof(LmHeader), s.stream));
    for (int i = 0; i < 3; ++i) lmk_scan(s.stream, a, variant);
    HIP_TRY(hipEventRecord(s.ev[0], s.stream));
    for (int i = 0; i < iters; ++i) lmk_scan(s.stream, a, variant);
    HIP_TRY(hipEventRecord(s.ev[1], s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[0], s.ev[1]));
    if (avg_us_out) *avg_us_out = (double)ms * 1000.0 / iters;
    if (algorithmic_bytes_out) {
        double b = 0;
        if (class_idx < 0) for (double v : d->hb.class_alg_bytes) b += v;
        else b = d->hb.class_alg_bytes[class_idx];
        *algorithmic_bytes_out = b;
    }
    return LM_OK;
}

int lm_time_stages(lm_detector* d, int slot, float threshold, int class_idx, int iters, double out_us[4]) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slot(d, slot))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (iters <= 0 || !out_us) return fail(LM_ERR_INVALID, "bad argument");
    Slot& s = d->slots[slot];
    if (!s.has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if ((rc = enqueue_match(d, s, threshold, class_idx, true))) return rc;
        if ((rc = finish_match(d, s, nullptr, 0, nullptr))) return rc;
        for (int k = 0; k < 4; ++k) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, s.ev[k], s.ev[k + 1]));
            acc[k] += (double)ms * 1000.0;
        }
    }
    for (int k = 0; k < 4; ++k) out_us[k] = acc[k] / iters;
    return LM_OK;
}

// Not part of the public header's stable surface: selects the scan kernel variant used by lm_match_t*.
int lm_set_scan_variant(lm_detector* d, int variant) { if (!d) return LM_ERR_INVALID; d->scan_variant = variant; return LM_OK; }

}  // extern "C"
