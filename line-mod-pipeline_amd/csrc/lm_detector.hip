// lm_detector.hip -- host side of liblinemod_hip.so: the C ABI of include/linemod_hip.h, the
// template bank (host copy + device upload), resident frame slots and the per-batch launch sequence.
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
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

#include "../../include/linemod_hip.h"
#include "lm_common.h"
#include "lm_extract.h"
#include <thread>

#include "lm_host.h"
#include "lm_yaml.h"
#include "lm_kernels.h"
#include "lm_comm.h"

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

#define LM_NCOPY 4
#define LM_NLANES 4      // lanes per detector (lm_match_begin / lm_match_end): HIP streams whose stages overlap

struct Slot {
    u8* h_bgr = nullptr;     // pinned upload staging
    u16* h_depth = nullptr;
    bool has_frame = false;
    bool planes = false;     // ... and that pass wrote the scanned level's miss planes (k_scan1 may read them)
    bool spread_low = false; // ... and ONE spread byte per position INSTEAD of the response memories: only k_scan1 can scan this slot
    bool prepared = false;   // a3-a10 have run on the frame the slot holds with the LUTs / thresholds now in force (lm_match_prepared)
    // Uploads run on the detector's copy stream: ev_up is recorded behind the slot's H2D copies, up_seq is the
    // upload's ticket (0 = never uploaded through the copy stream).  Copies complete in ticket order.
    hipEvent_t ev_up = nullptr;
    hipEvent_t ev_bgr = nullptr;     // recorded behind the colour image alone (RGB-D: the depth copy follows it)
    unsigned long long up_seq = 0;
    int up_stream = 0;               // which copy stream carried the upload (tickets are per stream)
    bool mask_ready = false;         // the slot's colour bit mask holds inRange(HSV(frame), mask_lo, mask_hi) of the frame the slot holds (lm_color_mask_prepare)
    int mask_lane = -1;              // ... written on that lane's stream (its mask_done event orders a later colour check behind the launch)
    int mask_lo[3] = {0, 0, 0}, mask_hi[3] = {0, 0, 0};
    bool staging_open = false;       // lm_stage_reserve has run: lm_stage_rows may fill the staging buffers, lm_upload_staged sends them
    bool matched = false;            // a match on the frame the slot holds has completed: its lists are still in the slot's result block (lm_match_collect)
};

}  // namespace

struct lm_detector {
    lm_config cfg;
    LmLevelGeom geom[LM_MAX_LEVELS];
    int lw[LM_MAX_LEVELS], lh[LM_MAX_LEVELS];
    u8 sim_lut[256];
    u8 normal_lut[8000];
    int lut_onehot = -1;          // cached: every NORMAL_LUT entry is 0 or one-hot (-1 = not evaluated)
    bool normal_lut_substitute = true;   // the built-in table (NOT OpenCV's normal_lut.i) is active: lm_set_normal_lut clears it
    lmh::Bank bank;

    // ---- device state
    bool dev_ready = false;
    // The fields stream / ev / d_raw_thr / h_raw_thr / raw_thr_for below always belong to the ACTIVE lane
    // (activate_lane swaps them); lane 0 is active outside lm_match_begin / lm_match_end.
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int* d_raw_thr = nullptr;
        int* h_raw_thr = nullptr;
        float raw_thr_for = -1.0f;
        bool created = false, busy = false, timed = false;
        hipEvent_t ev_done = nullptr;    // recorded behind the last command of the lane's match in flight (lm_match_end waits for IT, see wait_lane_done)
        int first = 0, n = 0;
        std::vector<int> classes;                       // class list of the match in flight ({-1} = all classes)
        unsigned long long waited_seq[LM_NCOPY] = {};   // newest upload ticket per copy stream this lane's stream waits for
    };
    Lane lanes[LM_NLANES];
    int active = 0;
    hipEvent_t blocking_ev[LM_NLANES] = {};           // LM_FLAG_BLOCKING_SYNC: one per lane
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [0..4] stage boundaries, [5] behind the exchange
    // H2D copies of lm_upload_frame* go through their own stream so that the frames of step k + 1 travel while
    // the lanes compute step k; a lane's stream waits (hipStreamWaitEvent) for the newest upload among its slots.
    // LM_NCOPY copy streams (slot -> stream round-robin): one in-order stream moved 0.6-0.9 MB copies at 25.6 GB/s
    // (per-copy set-up of the DMA engine), several streams keep several engines busy.
    hipStream_t copy_stream[LM_NCOPY] = {};
    unsigned long long up_seq_next[LM_NCOPY], up_seq_done[LM_NCOPY];   // per stream: next ticket / newest ticket known landed
    unsigned long long waited_seq[LM_NCOPY] = {};                      // ACTIVE lane's copy of Lane::waited_seq
    int n_copy_streams = LM_NCOPY;
    int stage_chunks = 1;                                  // pageable source: pieces of the staging memcpy (each piece is its own
                                                           // async copy; measured: every extra hipMemcpyAsync costs more than the overlap wins)
    std::vector<Slot> slots;
    // frame arena: [slot][bgr[l] | depth | quant[l][m] | lm[l]]
    u8* frame_arena = nullptr;
    size_t frame_stride = 0;
    size_t off_bgr[LM_MAX_LEVELS] = {}, off_depth = 0, off_quant[LM_MAX_LEVELS][2] = {}, off_lm[LM_MAX_LEVELS] = {};
    // colour-quantisation scratch S | qn: one region for level 0, one (sized for level 1) shared by the levels above,
    // and the rank-code image of the depth passes -- disjoint, so independent kernels of one dependency level may run in one launch (k_phase)
    size_t off_cscratch = 0, off_cscratch1 = 0, off_dscratch = 0;
    // ---- multi-GPU exchange: RCCL communicator + per-lane gather buffers
    struct Gather {
        int* d_cnt = nullptr; LmOutMatch* d_rec = nullptr;          // this rank's packed lists (k_pack_lists)
        int* d_all_cnt = nullptr; LmOutMatch* d_all_rec = nullptr;  // all ranks', rank-major
        int* h_all_cnt = nullptr; LmOutMatch* h_all_rec = nullptr;  // pinned host copies
        bool active = false;                                        // the lane's match in flight ends with a gather
        u32 cap_lane = 0;
    };
    // ---- f1 colour check on the GPU: hulls of every template, HSV division tables, per-slot colour bit mask
    bool hulls_dirty = true;
    u32* d_hull_class_base = nullptr; u32* d_hull_off = nullptr; int16_t* d_hull_xy = nullptr;
    int* d_hsv_div = nullptr;
    size_t off_cmask = 0; int cmask_wpr = 0;
    // r05: the colour check has its own (high-priority) stream and buffers, so that it runs beside the lanes: the post-processing of
    // batch k overlaps the match of batch k + 1 (HighLevelLineMOD::detectTemplatesBatchBegin / End)
    hipStream_t cc_stream = nullptr;
    u8* cc_dev = nullptr; u8* cc_host = nullptr; size_t cc_cap = 0;       // room for cc_cap matches: records | slot index | two counts
    size_t cc_pending = 0; bool cc_inflight = false;                      // lm_color_check_begin_slots enqueued a check of cc_pending matches
    u8* dc_dev = nullptr; u8* dc_host = nullptr; size_t dc_cap = 0;       // r06, lm_depth_counts_begin: room for dc_cap queries | two counts each
    size_t dc_pending = 0; bool dc_inflight = false;
    int cc_lo = 0, cc_hi = -1, dc_lo = 0, dc_hi = -1;                   // slots a colour check / depth counts in flight read: no upload goes there
    hipEvent_t mask_done[LM_NLANES] = {};                                // behind the mask launch of lm_color_mask_prepare on a lane: a colour check that reuses the masks waits for it
    hipEvent_t cc_done = nullptr, dc_done = nullptr;                      // behind the colour check's / the depth counts' last copy: their `end` waits for the event, not the stream
    LmComm* comm[LM_NLANES] = {};   // one communicator per lane: the lanes' collectives never wait for each other
    int comm_recs_per_frame = 0;
    Gather gather[LM_NLANES];
    double* d_red = nullptr;   // small device buffer of lm_comm_max / lm_comm_barrier
    int batch_phases = 2;            // calls of 16+ frames run a3-a10 as launches of level-fused batch kernels (lmk_preprocess_batch_phases):
                                     // 0 never, 1 always, 2 (default) when no other lane has work in flight -- measured r03: alone on the
                                     // chip the fused launches win (config 2: 4.81 -> 4.66, config 3: 8.54 -> 8.06 us per frame), beside two
                                     // other lanes the separate launches interleave better (config 2: 145 K against 140 K detections/s)
    int scan_list_order = 3;         // LM_TUNE_SCAN_LIST_ORDER (lm_host.h build_device_bank)
    int sort_split_mode = 2;         // device sort: 0 one workgroup per frame (r03), 1 always the split form (chunk workgroups + merge launch),
                                     // 2 (default) the split form while the recently collected lists were longer than LM_SORT_CHUNK keys
    int sort_long_score = 0;         // see note_sort_length
    int work_weight_by_pixels = 1;   // r04: the selection below counts a frame as level-0 pixels / (640 x 480) frames (LM_TUNE_WORK_WEIGHT = 0: by frame count, r03)
    int phase_max_slots = 15;        // calls of up to this many frames run a3-a10 as one launch per dependency level (LmPhaseArgs)
    // aux arena: [slot][LmDevHeader | cand | keys | out]
    u8* aux_arena = nullptr;
    size_t aux_stride = 0;
    size_t off_hdr = 0, off_cand = 0, off_keys = 0, off_out = 0;
    // host-mapped result blocks
    u8* host_blocks = nullptr;
    size_t host_stride = 0;
    int* d_raw_thr = nullptr;
    int* h_raw_thr = nullptr;
    float raw_thr_for = -1.0f;
    u32* d_plan = nullptr;        // k_refine_plan output, one per lane: [8][cap] slots + [8] lengths + [8][cap + 1] running sums + [8][cap] first entries
    int plan_stride_cap = 0;
    u64* d_resp_tab = nullptr;
    int miss_delta = 1;           // 4 - the largest response below 4 of the similarity table (ensure_luts; upstream's table: a neighbouring orientation scores 3)
    u32* d_sim_lut = nullptr;
    u8* d_normal_lut = nullptr;
    bool luts_dirty = true;
    // ---- device bank
    bool bank_dirty = true;
    lmh::DeviceBankHost hb;
    u32* d_item_t = nullptr; u32* d_item_chunk = nullptr;
    u32* d_scan_off = nullptr; int* d_scan_P = nullptr; int* d_scan_n = nullptr;
    int* d_t_global = nullptr; int* d_t_class = nullptr;
    // bit-plane scan (k_scan1): the concatenated offset lists, and the work items of the lane counts used so far
    u32* d_off1 = nullptr; u32* d_offn = nullptr;
    struct Items1 { int L = 0; u32* d_t = nullptr; u32* d_chunk = nullptr; std::vector<int> begin; };
    std::vector<Items1> items1;
    int scan_form = 0;               // LM_TUNE_SCAN_FORM: 0 = by cost (default), 1 = always the nibble scan k_scan4, 2 = the bit-plane scan k_scan1 whenever the level has planes,
                                     //    3 = the bit-plane scan with the planes in LDS (k_scanl) wherever a frame's planes fit (k_scan1 where they do not)
    float scan1_min_threshold = 50.0f;   // below this similarity threshold the miss bound keeps too many positions alive: k_scan4 (LM_TUNE_SCAN1_MIN_THRESHOLD)
    long long cnt_scan1_launches = 0; int last_scan1_lanes = 0;
    bool emit_planes = false;        // the pre-processing being enqueued writes the miss planes (set per call by enqueue_preprocess)
    bool emit_spread_low = false;    // ... and the spread byte instead of the response memories (the call's scan is k_scan1 by the rule below)
    u32* d_offs3 = nullptr;          // [nt][fpad1] orientation << 29 | spread-memory offset of the bit-plane scan's features
    // r06, the bit-plane scan with a frame's planes in LDS (k_scanl; hb.lds_ok): the lists in the LDS image's layout and the lane items
    u32* d_offl = nullptr; u32* d_offsl = nullptr; u32* d_litem = nullptr;
    unsigned long long* d_refine_stat = nullptr;     // LM_REFINE_STAT=1: k_refine's counting experiment (printed by lm_destroy)
    int scanl_min_slots = 24;        // by cost (LM_TUNE_SCAN_FORM 0) from this many frames per call (measured: 16 frames 35.5 us against k_scan4's 35.3, 32 frames 43.4 against 57.5)
    unsigned long long* d_surv[LM_NLANES] = {};      // k_scan1's survivor queues, one per lane, allocated on a lane's first bit-plane scan
    int surv_set[LM_NLANES] = {};                    // which of a queue's two counter sets the lane's next scan launch uses (the other is zeroed behind it)
    u32 surv_cap = 1u << 20;
    LmRefMeta* d_ref_meta[LM_MAX_LEVELS] = {};
    LmRefFeat* d_ref_feat[LM_MAX_LEVELS] = {};
    // scratch for stage hooks
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    u32 max_cand = 0, max_match = 0;
    int scan_variant = 0;
    bool scan_stats = false;                        // lm_set_scan_stats: the scan counts the features it loads
    unsigned long long* d_scan_stat = nullptr;      // [1024][4]: features loaded per wave, features of an exhaustive scan, lane-loads issued
    // live profile of lm_match* (lm_set_profiling): per-stage HIP-event time, scan launches and bytes
    bool profiling = false;
    double prof_us[4] = {0, 0, 0, 0};
    double prof_scan_bytes = 0;
    long long prof_launches = 0, prof_frames = 0;
    long long prof_exch_fallbacks = 0;                           // lane-steps that needed the sized second exchange
    double prof_exch_us = 0; long long prof_exch_launches = 0;   // pack + 2 x all-gather + D2H of the gathered path (ev[4] -> ev[5])
    long long cnt_preprocess_frames = 0, cnt_scan_launches = 0, cnt_refine_launches = 0, cnt_sort_launches = 0;   // lm_get_stage_counts

    u8* bgr(int slot, int l) const { return frame_arena + (size_t)slot * frame_stride + off_bgr[l]; }
    u16* depth(int slot) const { return reinterpret_cast<u16*>(frame_arena + (size_t)slot * frame_stride + off_depth); }
    u8* quant(int slot, int l, int m) const { return frame_arena + (size_t)slot * frame_stride + off_quant[l][m]; }
    u8* lm(int slot, int l) const { return frame_arena + (size_t)slot * frame_stride + off_lm[l]; }
    u8* cscratch(int slot, int l = 0) const { return frame_arena + (size_t)slot * frame_stride + (l == 0 ? off_cscratch : off_cscratch1); }
    u8* dscratch(int slot) const { return frame_arena + (size_t)slot * frame_stride + off_dscratch; }
    u8* aux(int slot, size_t off) const { return aux_arena + (size_t)slot * aux_stride + off; }
    LmHostBlock* host_block(int slot) const { return reinterpret_cast<LmHostBlock*>(host_blocks + (size_t)slot * host_stride); }
};

namespace {

bool any_lane_busy(const lm_detector* d) {
    for (const lm_detector::Lane& ln : d->lanes) if (ln.busy) return true;
    return false;
}

void free_device_bank(lm_detector* d) {
    hipFree(d->d_item_t); hipFree(d->d_item_chunk); hipFree(d->d_scan_off); hipFree(d->d_scan_P);
    hipFree(d->d_scan_n); hipFree(d->d_t_global); hipFree(d->d_t_class);
    hipFree(d->d_off1); hipFree(d->d_offn); hipFree(d->d_offs3); d->d_off1 = d->d_offn = d->d_offs3 = nullptr;
    hipFree(d->d_offl); hipFree(d->d_offsl); hipFree(d->d_litem); d->d_offl = d->d_offsl = d->d_litem = nullptr;
    for (auto& it : d->items1) { hipFree(it.d_t); hipFree(it.d_chunk); }
    d->items1.clear();
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
    const int M = c.num_modalities, L = c.pyramid_levels, S = c.frame_slots;
    // ---- frame arena layout
    size_t off = 0;
    // level-0 colour and depth back to back (no padding between them): a host frame laid out the same way is ONE copy
    d->off_bgr[0] = off; off += (size_t)c.width * c.height * 3;
    d->off_depth = off; off += align_up((size_t)c.width * c.height * 2, 256);
    off = align_up(off, 256);
    for (int l = 1; l < L; ++l) { d->off_bgr[l] = off; off += align_up((size_t)d->lw[l] * d->lh[l] * 3, 256); }
    for (int l = 0; l < L; ++l)
        for (int m = 0; m < M; ++m) { d->off_quant[l][m] = off; off += align_up((size_t)d->lw[l] * d->lh[l], 256); }
    for (int l = 0; l < L; ++l) { d->off_lm[l] = off; off += align_up(d->geom[l].arena_bytes, 256); }
    d->off_cscratch = off; off += align_up(lmk_color_scratch_bytes(c.width, c.height), 256);
    d->off_cscratch1 = off; off += align_up(lmk_color_scratch_bytes(d->lw[L > 1 ? 1 : 0], d->lh[L > 1 ? 1 : 0]), 256);
    d->off_dscratch = off; off += align_up((size_t)c.width * c.height, 256);
    d->cmask_wpr = (c.width + 31) / 32;
    d->off_cmask = off; off += align_up((size_t)d->cmask_wpr * c.height * 4, 256);
    d->frame_stride = align_up(off, 4096);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->frame_arena), d->frame_stride * S));
    HIP_TRY(hipMemset(d->frame_arena, 0, d->frame_stride * S));  // linear-memory pads / zero blocks stay zero forever
    // ---- aux arena layout
    off = 0;
    d->off_hdr = off; off += 256;
    d->off_cand = off; off += align_up((size_t)d->max_cand * sizeof(LmCand), 256);
    d->off_keys = off; off += align_up((size_t)d->max_match * 16, 256);
    d->off_out = off; off += align_up((size_t)LM_SORT_CAP * sizeof(LmOutMatch), 256);
    d->aux_stride = align_up(off, 4096);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->aux_arena), d->aux_stride * S));
    HIP_TRY(hipMemset(d->aux_arena, 0, d->aux_stride * S));      // counters start at zero; k_sort_unique re-arms them
    d->host_stride = align_up(sizeof(LmHostBlock), 256);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d->host_blocks), d->host_stride * S, hipHostMallocMapped));
    std::memset(d->host_blocks, 0, d->host_stride * S);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_raw_thr), 128 * sizeof(int)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d->h_raw_thr), 128 * sizeof(int)));
    d->plan_stride_cap = std::max(S / 8, 1) + 8;    // pieces per XCD list: nslots / 8 + 8 (k_refine_plan)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_plan), LM_NLANES * (24 * (size_t)d->plan_stride_cap + 16) * sizeof(u32)));
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    // lane 1's stream right behind lane 0's: the runtime deals streams to its hardware queues in creation order, and
    // two lanes that land on one queue run strictly one after the other (measured r02: 87 K instead of 102 K det/s)
    for (int l = 1; l < LM_NLANES; ++l) HIP_TRY(hipStreamCreateWithFlags(&d->lanes[l].stream, hipStreamNonBlocking));
    for (int k = 0; k < LM_NCOPY; ++k) {
        HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream[k], hipStreamNonBlocking));
        d->up_seq_next[k] = 1; d->up_seq_done[k] = 0;
    }
    for (auto& ev : d->ev) HIP_TRY(hipEventCreate(&ev));
    d->slots.assign(S, Slot());
    for (Slot& s : d->slots) {
        HIP_TRY(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_bgr, hipEventDisableTiming));
    }
    // pinned staging for pageable sources is allocated on a slot's first staged upload (ensure_staging): a
    // streaming server that hands over pinned frames (lm_upload_frame_pinned) never needs it
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_resp_tab), 256 * sizeof(u64) + 256));   // + the miss masks of the 256 spread bytes (d_lm_fast's planes)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_scan_stat), 4096 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d->d_scan_stat, 0, 4096 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_sim_lut), 256));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_normal_lut), 2 * 8000 + 16));    // the table, then its labels as rank codes (LMK_NORMAL_CODE_OFFSET) + a zero entry for indices outside the table
    HIP_TRY(hipDeviceSynchronize());
    if (const char* ev = getenv("LM_REFINE_STAT")) {
        if (atoi(ev) > 0 && hipMalloc(reinterpret_cast<void**>(&d->d_refine_stat), 8 * sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(d->d_refine_stat, 0, 8 * sizeof(unsigned long long));
    }
    d->dev_ready = true;
    d->luts_dirty = true;
    d->bank_dirty = true; d->hulls_dirty = true;
    return LM_OK;
}

int ensure_luts(lm_detector* d) {
    if (!d->luts_dirty) return LM_OK;
    u64 tab[256 + 32];
    u8* miss = reinterpret_cast<u8*>(tab + 256);       // bit o of miss[v]: orientation o's response to the spread byte v is below 4
    int below4 = 0;                                    // the largest response below 4 the table can give
    for (int v = 0; v < 256; ++v) {
        u64 e = 0;
        miss[v] = 0;
        for (int o = 0; o < 8; ++o) {
            u8 r = std::max(d->sim_lut[32 * o + (v & 15)], d->sim_lut[32 * o + 16 + (v >> 4)]);
            e |= (u64)r << (8 * o);
            if (r < 4) { miss[v] |= (u8)(1u << o); below4 = std::max(below4, (int)r); }
        }
        tab[v] = e;
    }
    d->miss_delta = 4 - below4;                        // what a missed feature costs at least (k_scan1's bound): 1 with upstream's table (responses 0 .. 4)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(d->d_resp_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->d_sim_lut, d->sim_lut, 256, hipMemcpyHostToDevice));
    {
        // k_dnormal writes the label's RANK CODE (lm_kernels.hip, a5 streaming form): looked up directly from a second table
        u8 both[2 * 8000 + 16] = {};                      // [LMK_NORMAL_CODE_OFFSET + 8000 ..] = 0: the code of an index outside the table
        std::memcpy(both, d->normal_lut, 8000);
        for (int i = 0; i < 8000; ++i) {
            const u8 v = d->normal_lut[i];
            const unsigned rank = v ? (unsigned)__builtin_ffs((int)v) : 0u;        // 0 for none, 1 + label otherwise
            both[LMK_NORMAL_CODE_OFFSET + i] = (u8)(rank < 4 ? 8 * rank : (rank < 8 ? 8 * (rank - 4) + 4 : 32u));
        }
        HIP_TRY(hipMemcpy(d->d_normal_lut, both, sizeof(both), hipMemcpyHostToDevice));
    }
    d->luts_dirty = false;
    return LM_OK;
}

int ensure_bank(lm_detector* d) {
    if (!d->bank_dirty) return LM_OK;
    HIP_TRY(hipDeviceSynchronize());
    free_device_bank(d);
    std::string err;
    if (!lmh::build_device_bank(d->bank, d->cfg, d->geom, d->hb, d->scan_list_order, err)) return fail(LM_ERR_INVALID, err);
    int rc;
    if ((rc = upload_vec(&d->d_item_t, d->hb.item_t))) return rc;
    if ((rc = upload_vec(&d->d_item_chunk, d->hb.item_chunk))) return rc;
    if ((rc = upload_vec(&d->d_scan_off, d->hb.scan_off))) return rc;
    if ((rc = upload_vec(&d->d_scan_P, d->hb.scan_P))) return rc;
    if ((rc = upload_vec(&d->d_scan_n, d->hb.scan_n))) return rc;
    if (d->hb.fpad1) {
        if ((rc = upload_vec(&d->d_off1, d->hb.off1))) return rc;
        if ((rc = upload_vec(&d->d_offn, d->hb.offn))) return rc;
        if ((rc = upload_vec(&d->d_offs3, d->hb.offs3))) return rc;
    }
    if (d->hb.lds_ok) {
        if ((rc = upload_vec(&d->d_offl, d->hb.offl))) return rc;
        if ((rc = upload_vec(&d->d_offsl, d->hb.offsl))) return rc;
        if ((rc = upload_vec(&d->d_litem, d->hb.lrec))) return rc;      // (the lane items with their templates' records: 16 bytes each)
    }
    if ((rc = upload_vec(&d->d_t_global, d->hb.t_global))) return rc;
    if ((rc = upload_vec(&d->d_t_class, d->hb.t_class))) return rc;
    for (int l = 0; l + 1 < d->cfg.pyramid_levels; ++l) {
        if ((rc = upload_vec(&d->d_ref_meta[l], d->hb.ref_meta[l]))) return rc;
        if ((rc = upload_vec(&d->d_ref_feat[l], d->hb.ref_feat[l]))) return rc;
    }
    d->bank_dirty = false;
    return LM_OK;
}

int check_slots(lm_detector* d, int first, int n) {
    if (first < 0 || n < 0 || first + n > (int)d->slots.size()) return fail(LM_ERR_INVALID, "slot out of range");
    return LM_OK;
}

// upstream's NORMAL_LUT holds 0 or one-hot bytes; the streaming depth kernels rely on it (counting median)
bool normal_lut_onehot(lm_detector* d) {
    if (d->lut_onehot < 0) {
        d->lut_onehot = 1;
        for (int i = 0; i < 8000; ++i) { const u8 v = d->normal_lut[i]; if (v & (v - 1)) { d->lut_onehot = 0; break; } }
    }
    return d->lut_onehot != 0;
}

// quant[l][1] for l >= 1: DepthNormalPyramid::pyrDown = NN half-size copy of the quantised image
void enqueue_depth_pyramid(lm_detector* d, int first, int n) {
    for (int l = 1; l < d->cfg.pyramid_levels; ++l)
        lmk_nn_half(d->stream, d->quant(first, l - 1, 1), d->lw[l - 1], d->quant(first, l, 1), d->lw[l], d->lh[l],
                    d->frame_stride, n);
}

int wait_uploads(lm_detector* d, hipStream_t stream, int first, int n, unsigned long long* seqs);
// an upload must not overwrite a frame that a colour check / depth-count call in flight is reading on the colour-check stream
int refuse_checked_slots(const lm_detector* d, int first, int n) {
    if (d->cc_inflight && d->cc_pending && first <= d->cc_hi && d->cc_lo < first + n) return fail(LM_ERR_INVALID, "slot is read by a colour check in flight: call lm_color_check_end first");
    if (d->dc_inflight && d->dc_pending && first <= d->dc_hi && d->dc_lo < first + n) return fail(LM_ERR_INVALID, "slot is read by depth counts in flight: call lm_depth_counts_end first");
    return LM_OK;
}

// The miss planes of the scanned level are written (by the pass that writes its nibble memories) only where the bit-plane scan can run: they
// cost k_lm_fast a second set of scattered stores (measured r05: 16.2 -> 24.4 us per 96-frame launch of config 2, 70 -> 115 us per 128 frames
// of config 3).  By cost that is a call of 8+ frames on a one-modality detector (pick_scan1_lanes); LM_TUNE_SCAN_FORM 2 asks for them always,
// 1 never.  A slot remembers whether its pass wrote them (Slot::planes): k_scan1 never reads planes of an older frame.
int scanl_rule(const lm_detector* d, int nslots);
// the scanned level's planes of all modalities fit the LDS image of k_scanl (lm_host.cpp build_device_bank: the same test, there with the bank's size)
bool scanl_geom_ok(const lm_detector* d) {
    const LmLevelGeom& g = d->geom[d->cfg.pyramid_levels - 1];
    const u32 ttwh = (u32)g.T * (u32)g.T * g.wh;
    return g.nibble && g.plane_ori && (ttwh % 128u) == 0 && (size_t)d->cfg.num_modalities * ttwh <= LM_SCANL_IMAGE_MAX && g.wh <= (1u << LM_SCANL_POS_BITS);
}
bool planes_wanted(const lm_detector* d, int n) {
    if (d->scan_form == 1) return false;
    if (d->scan_form >= 2) return true;
    return (d->cfg.num_modalities == 1 && n >= 8) || scanl_rule(d, n) > 0;
}
u32 plane_stride_in_use(const lm_detector* d, int level) {
    const LmLevelGeom& g = d->geom[level];
    if (!g.plane_ori || !d->emit_planes) return 0u;
    return g.plane_ori | (d->emit_spread_low ? 0x80000000u : 0u);      // (bit 31: d_lm_fast writes the spread byte instead of the response memories)
}

// one modality's linear memories of level l (a6-a10)
void enqueue_lm(lm_detector* d, hipStream_t st, int first, int n, int l, int m) {
    const size_t fs = d->frame_stride;
    const LmLevelGeom& g = d->geom[l];
    const int sp = g.spread_only ? 1 : g.nibble ? 2 : 0;
    u8* dst = d->lm(first, l) + (size_t)m * g.mod_stride;
    if (m == 0 || l == 0)
        lmk_linear_memories(st, d->quant(first, l, m), g.w, 0, sp, g.w, g.h, g.T, d->d_resp_tab, dst, g.ori_stride, fs, fs, n, plane_stride_in_use(d, l));
    else   // level l of the depth modality reads the quantised image of level l-1 at (2y, 2x)
        lmk_linear_memories(st, d->quant(first, l - 1, 1), d->lw[l - 1], 1, sp, g.w, g.h, g.T, d->d_resp_tab, dst,
                            g.ori_stride, fs, fs, n, plane_stride_in_use(d, l));
}

// How many 640 x 480 frames one frame of this detector counts as in the few-frame / batch kernel selection (at least 1).
int slot_weight(const lm_detector* d) {
    const long px = (long)d->lw[0] * d->lh[0];
    const int w = d->work_weight_by_pixels ? (int)(px / (640L * 480L)) : 1;
    return w < 1 ? 1 : w;
}

// a3-a10 on the frames resident in slots [first, first + n).
int scan1_rule(const lm_detector* d, int nslots, bool forced);
void enqueue_preprocess(lm_detector* d, int first, int n) {
    d->emit_planes = planes_wanted(d, n) && d->geom[d->cfg.pyramid_levels - 1].plane_ori != 0;
    // by cost, when this call's own scan will be k_scan1: no response memories at all (k_lm_fast is bound by the number of its stores), the
    // second stage of the scan reads the spread byte through the table.  Such a slot can only be scanned by k_scan1 afterwards.
    d->emit_spread_low = d->emit_planes && ((d->scan_form == 0 && !d->bank_dirty && (scanl_rule(d, n) > 0 || scan1_rule(d, n, false) > 0)) ||
                                            (d->scan_form == 3 && scanl_geom_ok(d)));      // (3: whatever the bank; a bank k_scanl cannot take is scanned by k_scan1)
    for (int i = 0; i < n; ++i) { d->slots[first + i].planes = d->emit_planes; d->slots[first + i].spread_low = d->emit_spread_low; }
    const lm_config& c = d->cfg;
    const int M = c.num_modalities, L = c.pyramid_levels;
    const size_t fs = d->frame_stride;
    // ---- few frames: one launch per dependency level (lm_kernels.h LmPhaseArgs): 5 launches instead of 14;
    // ---- batches: the same with the batch kernels, 4 launches instead of 11 (LM_TUNE_BATCH_PHASES)
    // by WORK, not by frame count (r04): a frame counts as level-0 pixels / (640 x 480) frames, so the eight 1280 x 960 frames of a
    // config-5 lane-step (32 VGA frames' worth of pixels) take the batch kernels, not the latency path
    const int weight = slot_weight(d);
    struct WeightGuard { explicit WeightGuard(int w) { lmk_set_slot_weight(w); } ~WeightGuard() { lmk_set_slot_weight(1); } } weight_guard(weight);
    const int n_eff = n * weight;
    const bool few = n_eff <= d->phase_max_slots;
    bool others_busy = false;
    for (int o = 0; o < LM_NLANES; ++o) others_busy |= (o != d->active && d->lanes[o].busy);
    const bool fuse_batch = d->batch_phases == 1 || (d->batch_phases == 2 && !others_busy);
    if ((few || (fuse_batch && n_eff >= 16)) && L == 2) {
        LmPhaseArgs pa{};
        pa.bgr0 = d->bgr(first, 0); pa.bgr1 = d->bgr(first, 1); pa.depth = M == 2 ? d->depth(first) : nullptr;
        pa.cs0 = d->cscratch(first, 0); pa.cs1 = d->cscratch(first, 1); pa.ds = d->dscratch(first);
        pa.qc0 = d->quant(first, 0, 0); pa.qc1 = d->quant(first, 1, 0); pa.qd0 = M == 2 ? d->quant(first, 0, 1) : nullptr;
        pa.lm_c0 = d->lm(first, 0); pa.lm_c1 = d->lm(first, 1);
        pa.lm_d0 = d->lm(first, 0) + d->geom[0].mod_stride; pa.lm_d1 = d->lm(first, 1) + d->geom[1].mod_stride;
        pa.w = d->lw[0]; pa.h = d->lh[0];
        pa.weak_threshold = c.weak_threshold; pa.dist_thr = c.distance_threshold; pa.diff_thr = c.difference_threshold;
        pa.normal_lut = d->d_normal_lut; pa.resp_tab = d->d_resp_tab; pa.ori_stride1 = d->geom[1].ori_stride; pa.plane_ori1 = plane_stride_in_use(d, 1);
        pa.slot_stride = fs; pa.nslots = n;
        auto mode = [&](int l) { return d->geom[l].spread_only ? 1 : d->geom[l].nibble ? 2 : 0; };
        const bool onehot = M == 2 ? normal_lut_onehot(d) : true;
        if (M <= 2 && d->lw[1] * 2 == d->lw[0] && d->lh[1] * 2 == d->lh[0]) {
            if (few && lmk_phases_supported(pa, d->geom[0].T, d->geom[1].T, mode(0), mode(1), onehot)) {
                lmk_preprocess_phases(d->stream, pa, d->geom[0].T);
                return;
            }
            if (!few && lmk_batch_phases_supported(pa, d->geom[0].T, d->geom[1].T, mode(0), mode(1), onehot)) {
                lmk_preprocess_batch_phases(d->stream, pa, d->geom[0].T);
                return;
            }
        }
    }
    // batches: the level-0 blur and pyrDown 0 -> 1 share one slot-interleaved launch (the raw image comes from HBM once)
    const bool blur_pyr = L >= 2 && lmk_blur_pyrdown(d->stream, d->bgr(first, 0), d->lw[0], d->lh[0], d->cscratch(first, 0), d->bgr(first, 1),
                                                      d->quant(first, 0, 0), fs, n);
    // r06: two levels, level-0 blur + pyrDown done: the level-1 blur next, then BOTH levels' gradients in one grid (k_cgrad_levels: level 1's few waves fill the
    // idle SIMDs of level 0's last round instead of a launch of their own)
    bool grads_done = false;
    if (L == 2 && blur_pyr && d->lw[1] * 2 == d->lw[0] && d->lh[1] * 2 == d->lh[0] && lmk_cgrad_levels_wanted(d->lw[0], d->lh[0], n) &&
        lmk_color_blur(d->stream, d->bgr(first, 1), d->lw[1], d->lh[1], d->cscratch(first, 1), fs, n)) {
        grads_done = lmk_cgrad_levels(d->stream, d->cscratch(first, 0), d->lw[0], d->lh[0], d->quant(first, 0, 0), d->cscratch(first, 1), d->lw[1], d->lh[1],
                                      d->quant(first, 1, 0), c.weak_threshold, fs, n);
        if (!grads_done)     // (the blur of level 1 is in its scratch: the gradients one launch per level)
            for (int l = 0; l < L; ++l) lmk_color_quantize(d->stream, d->bgr(first, l), d->lw[l], d->lh[l], c.weak_threshold, d->quant(first, l, 0), nullptr, d->cscratch(first, l), fs, n, true);
        grads_done = true;
    }
    for (int l = 0; l < L; ++l) {
        if (l > 0 && !(l == 1 && blur_pyr)) lmk_pyrdown(d->stream, d->bgr(first, l - 1), d->lw[l - 1], d->lh[l - 1], d->bgr(first, l), fs, n);
        if (!grads_done)
        lmk_color_quantize(d->stream, d->bgr(first, l), d->lw[l], d->lh[l], c.weak_threshold, d->quant(first, l, 0),
                           nullptr, d->cscratch(first, l), fs, n, l == 0 && blur_pyr);
        if (M == 2 && l == 0)
            lmk_depth_quantize(d->stream, d->depth(first), d->lw[0], d->lh[0], c.distance_threshold,
                               c.difference_threshold, d->d_normal_lut, normal_lut_onehot(d), d->quant(first, 0, 1),
                               d->dscratch(first), fs, n);
    }
    if (M == 2 && L > 2) enqueue_depth_pyramid(d, first, n);
    for (int l = 0; l < L; ++l)
        for (int m = 0; m < M; ++m) enqueue_lm(d, d->stream, first, n, l, m);
}

void fill_raw_thr(int* tab, float threshold) {
    // A.7: int(2n + (t/100)*2n + 0.5f) in float arithmetic (file is built with -ffp-contract=off)
    for (int n = 0; n < 128; ++n) tab[n] = static_cast<int>(2 * n + (threshold / 100.f) * (2 * n) + 0.5f);
}

struct ItemRange { int lo, n; int t_lo, t_hi; };   // items of the nibble / byte scan, and the bank-local templates they belong to
int item_range(lm_detector* d, int class_idx, ItemRange* r) {
    const int nc = (int)d->bank.classes.size();
    if (class_idx >= nc || class_idx < -1) return fail(LM_ERR_INVALID, "class index out of range");
    if (class_idx < 0) { r->lo = 0; r->n = (int)d->hb.item_t.size(); r->t_lo = 0; r->t_hi = (int)d->hb.t_global.size(); }
    else {
        r->lo = d->hb.class_item_lo[class_idx]; r->n = d->hb.class_item_hi[class_idx] - r->lo;
        r->t_lo = d->hb.class_t_lo[class_idx]; r->t_hi = d->hb.class_t_hi[class_idx];
    }
    return LM_OK;
}

// Detector::match(..., class_ids): the work-item ranges of a LIST of classes (HighLevelLinemod.cpp:145,152).  {-1} or an
// empty list = every class (upstream: an empty class_ids vector).  Duplicates are dropped (upstream would emit the
// class twice and std::unique would remove the copies again); ranges of neighbouring classes are merged, so the usual
// "all classes of the bank" list is ONE scan launch.  `classes` comes back normalised (sorted, unique).
int item_ranges(lm_detector* d, std::vector<int>& classes, std::vector<ItemRange>& out) {
    const int nc = (int)d->bank.classes.size();
    out.clear();
    for (int c : classes) if (c < -1 || c >= nc) return fail(LM_ERR_INVALID, "class index out of range");
    std::sort(classes.begin(), classes.end());
    classes.erase(std::unique(classes.begin(), classes.end()), classes.end());
    if (classes.empty() || classes[0] < 0) {
        if (classes.size() > 1) return fail(LM_ERR_INVALID, "class index -1 (all classes) cannot be combined with others");
        classes.assign(1, -1);
        out.push_back(ItemRange{0, (int)d->hb.item_t.size(), 0, (int)d->hb.t_global.size()});
        return LM_OK;
    }
    for (int c : classes) {
        const int lo = d->hb.class_item_lo[c], hi = d->hb.class_item_hi[c];
        if (!out.empty() && out.back().lo + out.back().n == lo && out.back().t_hi == d->hb.class_t_lo[c]) { out.back().n += hi - lo; out.back().t_hi = d->hb.class_t_hi[c]; }
        else out.push_back(ItemRange{lo, hi - lo, d->hb.class_t_lo[c], d->hb.class_t_hi[c]});
    }
    return LM_OK;
}

// Which scan a launch over `nslots` frames takes (r05).  The bit-plane scan k_scan1 issues about as many vector instructions per
// wave and feature as the nibble scan k_scan4, and a wave of either is one work item for a group of frames: 64 / L1 frames of
// chunks of 128 L1 - 31 positions there, two frames of chunks of 1016 here.  So the form with fewer waves wins; L1 is the lane count
// with the fewest.  k_scan1 needs a margin (its waves stop when the LAST of their frames is out of reach, and the survivors' exact
// sums come on top), and a threshold high enough for the miss bound to bite.  Returns L1, or 0 for k_scan4 / k_scan.
// forced: the lane count with the fewest waves whatever the rules say (slots without response memories)
int scan1_rule(const lm_detector* d, int nslots, bool forced) {
    const LmLevelGeom& g = d->geom[d->cfg.pyramid_levels - 1];
    if (!g.nibble || !g.plane_ori || !d->hb.fpad1) return 0;
    if (forced) {
        long long best = -1; int bestL = 0;
        for (int L1 = 1; L1 <= 64; ++L1) {
            const int G1 = 64 / L1;
            if ((size_t)(G1 - 1) * d->frame_stride + g.arena_bytes >= 0x7FFFFFFFull) continue;
            const long long waves = d->hb.items1_by_L[L1] * ((nslots + G1 - 1) / G1);
            if (best < 0 || waves < best) { best = waves; bestL = L1; }
        }
        return bestL;
    }
    if (d->scan_form == 1) return 0;
    const bool asked = d->scan_form >= 2;      // (3: the LDS form where a frame's planes fit, this kernel where they do not)
    if (!asked && !(d->raw_thr_for >= d->scan1_min_threshold)) return 0;
    // measured r05 (profiles/r05_ab_experiments.log, three lanes): colour-only config 3 +9 % (the scan launch 389 -> 296 us per 128 frames), but
    // RGB-D config 2 -2 % and config 5 -18 %: with two modalities the exact deficits of k_scan4's pruning stop a work item after 29-46 % of its
    // features, the miss bound after 66-84 %.  By cost = one modality only.
    if (!asked && d->cfg.num_modalities != 1) return 0;
    // ... and batches only: its three launches (queue reset, k_scan1, k_scan1_exact) cost a single 640 x 480 frame 29 instead of 15 us of scan
    // (the reference's one-frame call: 100 against 89 us per call)
    if (!asked && nslots < 8) return 0;
    long long best = -1; int bestL = 0;
    for (int L1 = 1; L1 <= 64; ++L1) {
        const int G1 = 64 / L1;
        // a wave's buffer descriptor starts at its group's first frame: the last frame's arena must end below 2^31 bytes
        if ((size_t)(G1 - 1) * d->frame_stride + g.arena_bytes >= 0x7FFFFFFFull) continue;
        const long long waves = d->hb.items1_by_L[L1] * ((nslots + G1 - 1) / G1);
        if (best < 0 || waves < best) { best = waves; bestL = L1; }
    }
    if (bestL == 0) return 0;
    const long long waves4 = (long long)d->hb.item_t.size() * ((nslots + 1) / 2);
    if (!asked && (best * 5 > waves4 * 4 || nslots < 64 / bestL)) return 0;      // (and whole groups of frames)
    return bestL;
}

// r06: the bit-plane scan with the frame's planes in LDS (k_scanl).  Workgroups (shares of the lane items) per frame: a workgroup takes a CU's whole
// LDS, so the chip runs 256 at a time; each pays a fixed price (the two copies of 150 KB, two barriers, the wait for its last wave: about 14 us)
// plus about 10 us per wave item of its busiest wave.  Measured on config 2's workload (tools/probe_scanl_R.py, profiles/r06_ab_experiments.log):
// the best share count fills ONE round of the chip up to 64 frames and two beyond (32 frames: 8 shares, 64: 4, 96: 5, 128: 4); below 4 shares
// a workgroup's survivors no longer fit its LDS queue.  The model below reproduces those choices.
int scanl_shares(const lm_detector* d, int nslots, int n_litems) {
    const int n_w = (n_litems + 63) / 64;
    const int r_max = std::max(1, std::min(32, n_w / 16));        // (every wave of a workgroup gets an item)
    const int r_min = std::min(4, r_max);
    int best = r_min; double best_t = -1;
    for (int R = r_min; R <= r_max; ++R) {
        const double x = (double)nslots * R / 256.0, rounds = std::max(1.0, 0.7 * std::ceil(x) + 0.3 * x);
        const double t = rounds * (14.2 + 9.8 * ((n_w + 16 * R - 1) / (16 * R)));
        if (best_t < 0 || t < best_t - 1e-9) { best_t = t; best = R; }
    }
    if (const char* ev = getenv("LM_SCANL_R")) best = std::max(1, atoi(ev));       // (experiments)
    return best;
}
// ... by cost (LM_TUNE_SCAN_FORM 0): where a frame's planes fit LDS, for calls of enough frames to fill the chip with such workgroups, at thresholds
// at which the miss bound bites.  Returns the shares per frame, 0 = another form.  (LM_TUNE_SCAN_FORM 3 asks for it wherever it can run.)
int scanl_rule(const lm_detector* d, int nslots) {
    if (d->bank_dirty || !d->hb.lds_ok || !d->d_litem) return 0;
    const LmLevelGeom& g = d->geom[d->cfg.pyramid_levels - 1];
    if (!g.nibble || !g.plane_ori) return 0;
    if (d->scan_form == 1 || d->scan_form == 2) return 0;
    if (d->scan_form == 0 && (!(d->raw_thr_for >= d->scan1_min_threshold) || nslots < d->scanl_min_slots)) return 0;
    return scanl_shares(d, nslots, (int)d->hb.litem.size());
}
// ... for prepared slots: all of them keep the planes and the spread bytes (the form reads both)
int pick_scanl(const lm_detector* d, int first, int nslots, int n_litems) {
    if (d->bank_dirty || !d->hb.lds_ok || !d->d_litem || d->scan_form == 2 || d->scan_form == 1 || n_litems <= 0) return 0;
    for (int i = 0; i < nslots; ++i) if (!d->slots[first + i].planes || !d->slots[first + i].spread_low) return 0;
    // (by cost: calls of few frames stay with k_scan1, which such slots can take as well -- the threshold rule no longer matters: a bit-plane form it is)
    if (d->scan_form == 0 && nslots < d->scanl_min_slots) return 0;
    return scanl_shares(d, nslots, n_litems);
}

// ... for these slots: -1 = they cannot be scanned together.  A launch reads ONE layout of the scanned level (LmScanArgs::exact_spread), so slots
// that keep only the spread byte mix neither with slots without planes nor with slots that have planes AND response memories.
int pick_scan1_lanes(const lm_detector* d, int first, int nslots) {
    bool all_planes = true, any_spread = false, all_spread = true;
    for (int i = 0; i < nslots; ++i) {
        const auto& sl = d->slots[first + i];
        all_planes = all_planes && sl.planes; any_spread = any_spread || sl.spread_low; all_spread = all_spread && sl.spread_low;
    }
    if (any_spread) return (all_planes && all_spread) ? scan1_rule(d, nslots, true) : -1;
    if (!all_planes) return 0;                   // (a frame prepared by a call that did not write them)
    return scan1_rule(d, nslots, false);
}

// the work items of k_scan1 for L1 lanes per frame, built and uploaded on first use
int ensure_items1(lm_detector* d, int L1, const lm_detector::Items1** out) {
    for (const auto& it : d->items1) if (it.L == L1) { *out = &it; return LM_OK; }
    std::vector<u32> t, c;
    lm_detector::Items1 it;
    it.L = L1;
    lmh::build_items1(d->hb, L1, t, c, it.begin);
    int rc;
    if ((rc = upload_vec(&it.d_t, t))) return rc;
    if ((rc = upload_vec(&it.d_chunk, c))) { hipFree(it.d_t); return rc; }
    d->items1.push_back(std::move(it));
    *out = &d->items1.back();
    return LM_OK;
}

LmScanArgs make_scan_args(lm_detector* d, int first, ItemRange r, int nslots = 1) {
    const int L = d->cfg.pyramid_levels;
    const LmLevelGeom& g = d->geom[L - 1];
    LmScanArgs a;
    a.L1 = 0; a.G1 = 1; a.L1_rcp16 = 0; a.delta_rcp16 = 0; a.off1 = a.offn = nullptr; a.fpad1 = 0; a.no_exact = 0; a.surv = nullptr; a.surv_cap = 0; a.surv_set = 0; a.exact_spread = 0; a.offs3 = nullptr; a.resp_tab = nullptr;
    a.lm = d->lm(first, L - 1); a.lm_slot_stride = d->frame_stride;
    a.item_t = d->d_item_t; a.item_chunk = d->d_item_chunk;
    a.item_lo = r.lo; a.n_items = r.n;
    a.scan_off = d->d_scan_off; a.scan_P = d->d_scan_P; a.scan_n = d->d_scan_n;
    a.M = d->cfg.num_modalities; a.fpad = d->hb.fpad; a.nibble = g.nibble;
    a.raw_thr_by_n = d->d_raw_thr;
    a.stat = d->scan_stats ? d->d_scan_stat : nullptr;
    a.W = g.W; a.T = g.T;
    a.hdr = reinterpret_cast<LmDevHeader*>(d->aux(first, d->off_hdr));
    a.cand = reinterpret_cast<LmCand*>(d->aux(first, d->off_cand));
    a.aux_slot_stride = d->aux_stride;
    a.cand_cap = d->max_cand;
    a.dbg = 0; a.lds_form = 0; a.R = 1; a.offl = a.offsl = a.litem = nullptr; a.litem_lo = a.n_litems = 0; a.pb = a.mod_stride = a.planes_off = a.plane_ori = a.tbl_bytes = a.queue_cap = 0;
    {
        const int lo = d->hb.lds_ok ? d->hb.lbegin[(size_t)r.t_lo] : 0, nl = d->hb.lds_ok ? d->hb.lbegin[(size_t)r.t_hi] - lo : 0;
        const int R = pick_scanl(d, first, nslots, nl);
        if (R > 0) {
            a.lds_form = 1; a.R = R;
            a.offl = d->d_offl; a.offsl = d->d_offsl; a.litem = d->d_litem; a.litem_lo = lo; a.n_litems = nl;
            a.pb = (u32)g.T * (u32)g.T * g.wh / 8u; a.mod_stride = g.mod_stride; a.planes_off = 8u * g.ori_stride; a.plane_ori = g.plane_ori;
            const u32 img = (u32)a.M * 8u * a.pb;
            a.tbl_bytes = std::max<u32>(LM_SCANL_TABLE_BYTES, (((g.wh + 127u) / 128u) * 16u + 32u + 15u) & ~15u);
            a.queue_cap = std::min<u32>((LM_SCANL_LDS_BYTES - img - a.tbl_bytes - 16u - 512u) / 4u, 1u << 16);      // (16: queue header, 512: the raw thresholds)
            a.delta_rcp16 = (65536u + (u32)d->miss_delta - 1u) / (u32)d->miss_delta;
            a.fpad1 = d->hb.fpad1; a.exact_spread = 1; a.offs3 = d->d_offs3; a.resp_tab = d->d_resp_tab;
            return a;
        }
    }
    const int L1 = pick_scan1_lanes(d, first, nslots);
    const lm_detector::Items1* it = nullptr;
    if (L1 > 0 && ensure_items1(d, L1, &it) == LM_OK) {
        a.L1 = L1; a.G1 = 64 / L1;
        a.L1_rcp16 = (65536u + (u32)L1 - 1u) / (u32)L1;
        a.delta_rcp16 = (65536u + (u32)d->miss_delta - 1u) / (u32)d->miss_delta;
        a.off1 = d->d_off1; a.offn = d->d_offn; a.fpad1 = d->hb.fpad1;
        a.exact_spread = d->slots[first].spread_low ? 1 : 0; a.offs3 = d->d_offs3; a.resp_tab = d->d_resp_tab;
        a.item_t = it->d_t; a.item_chunk = it->d_chunk;
        a.item_lo = it->begin[(size_t)r.t_lo]; a.n_items = it->begin[(size_t)r.t_hi] - a.item_lo;
        unsigned long long*& q = d->d_surv[d->active];
        if (!q) {
            if (hipMalloc(reinterpret_cast<void**>(&q), (16 + (size_t)d->surv_cap) * sizeof(unsigned long long)) != hipSuccess) { q = nullptr; (void)hipGetLastError(); }
            else if (hipMemset(q, 0, 16 * sizeof(unsigned long long)) != hipSuccess) { hipFree(q); q = nullptr; (void)hipGetLastError(); }
            d->surv_set[d->active] = 0;
        }
        a.surv = q; a.surv_cap = d->surv_cap;       // (no queue: the waves take their survivors' exact sums themselves)
        a.surv_set = d->surv_set[d->active];
        if (g.wh >= (1u << 20) || nslots > 4096) a.surv = nullptr;      // the entry's 20-bit position / 12-bit slot
    }
    return a;
}

// Slots that keep only the spread byte of the scanned level have no response memories: k_scan4 would read garbage there.  When the bit-plane
// form could not be set up for them (no work items: allocation failure, no lane count fits) the match fails instead.
int check_scan_args(const lm_detector* d, int first, const LmScanArgs& a) {
    if (!a.L1 && !a.lds_form && d->slots[first].spread_low)
        return fail(LM_ERR_HIP, "the slots keep only the spread byte of the scanned level and the bit-plane scan could not be set up for them: upload the frames again");
    return LM_OK;
}

// after every launch of lmk_scan with these arguments: the lane's next bit-plane scan takes the other set of queue counters (this launch's
// k_scan1_exact has zeroed it)
void scan_launched(lm_detector* d, LmScanArgs& a) {
    if (!a.L1 || !a.surv) return;
    d->surv_set[d->active] ^= 1;
    a.surv_set = d->surv_set[d->active];
}

LmRefineArgs make_refine_args(lm_detector* d, int first, int level, float threshold) {
    LmRefineArgs a;
    a.lm = d->lm(first, level); a.lm_slot_stride = d->frame_stride;
    a.g = d->geom[level];
    a.M = d->cfg.num_modalities;
    a.meta = d->d_ref_meta[level]; a.feats = d->d_ref_feat[level];
    a.sim_lut = d->d_sim_lut;
    a.hdr = reinterpret_cast<LmDevHeader*>(d->aux(first, d->off_hdr));
    a.cand = reinterpret_cast<LmCand*>(d->aux(first, d->off_cand));
    a.keys = reinterpret_cast<u64*>(d->aux(first, d->off_keys));
    a.aux_slot_stride = d->aux_stride;
    a.cand_cap = d->max_cand; a.match_cap = d->max_match;
    a.threshold = threshold;
    a.t_global = d->d_t_global; a.t_class = d->d_t_class;
    a.plan = nullptr; a.plan_cap = 0;
    a.stat = d->d_refine_stat;
    return a;
}

LmSortArgs make_sort_args(lm_detector* d, int first) {
    LmSortArgs a;
    a.hdr = reinterpret_cast<LmDevHeader*>(d->aux(first, d->off_hdr));
    a.keys = reinterpret_cast<u64*>(d->aux(first, d->off_keys));
    a.out = reinterpret_cast<LmOutMatch*>(d->aux(first, d->off_out));
    // split form (four chunk workgroups per frame + a merge launch) when this detector's recent lists were long enough to need it:
    // the lists are identical either way, so the choice may follow what the last collected frames looked like
    a.split = d->sort_split_mode == 1 || (d->sort_split_mode == 2 && d->sort_long_score > 0);
    a.aux_slot_stride = d->aux_stride;
    a.host = d->host_block(first);
    a.host_slot_stride = d->host_stride;
    a.cand_cap = d->max_cand; a.match_cap = d->max_match;
    return a;
}

int enqueue_threshold(lm_detector* d, float threshold) {
    if (!(threshold >= 0.0f)) return fail(LM_ERR_INVALID, "threshold must be >= 0");
    if (d->raw_thr_for != threshold) {
        HIP_TRY(hipStreamSynchronize(d->stream));  // the pinned table may still feed an earlier copy
        fill_raw_thr(d->h_raw_thr, threshold);
        HIP_TRY(hipMemcpyAsync(d->d_raw_thr, d->h_raw_thr, 128 * sizeof(int), hipMemcpyHostToDevice, d->stream));
        d->raw_thr_for = threshold;
    }
    return LM_OK;
}

// a11-a15 on prepared linear memories; the sort kernel publishes the results to host-mapped memory.
int enqueue_match_stages(lm_detector* d, int first, int n, float threshold, const std::vector<ItemRange>& ranges, bool timed) {
    const int L = d->cfg.pyramid_levels;
    if (pick_scan1_lanes(d, first, n) < 0)
        return fail(LM_ERR_INVALID, "the slots were prepared by calls of different scan forms (some keep only the spread byte of the scanned level): match them apart or upload again");
    if (timed) HIP_TRY(hipEventRecord(d->ev[1], d->stream));
    // one scan launch per run of neighbouring classes; the launches append to the same candidate lists
    for (const ItemRange& r : ranges)
        if (r.n > 0) {
            LmScanArgs sa = make_scan_args(d, first, r, n);
            if (int rc = check_scan_args(d, first, sa)) return rc;
            lmk_scan(d->stream, sa, d->scan_variant, n);
            scan_launched(d, sa);
            d->cnt_scan_launches += 1; d->cnt_scan1_launches += (sa.L1 || sa.lds_form) ? 1 : 0; d->last_scan1_lanes = sa.lds_form ? 1000 + sa.R : sa.L1;
        }
    if (timed) HIP_TRY(hipEventRecord(d->ev[2], d->stream));
    if (L == 1) {
        lmk_emit_unrefined(d->stream, make_refine_args(d, first, 0, threshold), n);
    } else {
        // 8+ slots: balance the slots over the XCDs by their candidate counts (one plan per lane)
        u32* plan = nullptr;
        const int plan_cap = n / 8 + 8;      // pieces per XCD list: its share of the slots + room for the pieces of the heavy ones
        if ((n % 8) == 0 && n <= 1016 && plan_cap <= d->plan_stride_cap && d->d_plan) {
            plan = d->d_plan + (size_t)d->active * (24 * (size_t)d->plan_stride_cap + 16);
            lmk_refine_plan(d->stream, make_refine_args(d, first, L - 2, threshold), n, plan, plan_cap);
        }
        for (int l = L - 2; l >= 0; --l) {
            LmRefineArgs ra = make_refine_args(d, first, l, threshold);
            ra.plan = plan; ra.plan_cap = plan_cap;
            lmk_refine(d->stream, ra, l == 0, n);
            d->cnt_refine_launches += 1;
        }
    }
    if (timed) HIP_TRY(hipEventRecord(d->ev[3], d->stream));
    lmk_sort_unique(d->stream, make_sort_args(d, first), n);
    d->cnt_sort_launches += 1;
    if (timed) HIP_TRY(hipEventRecord(d->ev[4], d->stream));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

// The active lane's stream waits for the copy-stream uploads of the slots it is about to read.
int enqueue_upload_wait(lm_detector* d, int first, int n) {
    int rc;
    unsigned long long seqs[LM_NCOPY];
    if ((rc = wait_uploads(d, d->stream, first, n, seqs))) return rc;
    for (int k = 0; k < LM_NCOPY; ++k) if (seqs[k] > d->waited_seq[k]) d->waited_seq[k] = seqs[k];
    return LM_OK;
}

// `classes` is normalised in place (item_ranges).  prepared: the slots' a3-a10 results are current (checked by the
// caller): a11-a15 only.
int enqueue_match(lm_detector* d, int first, int n, float threshold, std::vector<int>& classes, bool timed = false,
                  bool prepared = false) {
    std::vector<ItemRange> ranges;
    int rc;
    if ((rc = item_ranges(d, classes, ranges))) return rc;
    if ((rc = enqueue_threshold(d, threshold))) return rc;
    if ((rc = enqueue_upload_wait(d, first, n))) return rc;
    if (timed) HIP_TRY(hipEventRecord(d->ev[0], d->stream));
    if (!prepared) {
        enqueue_preprocess(d, first, n);
        d->cnt_preprocess_frames += n;
        for (int i = 0; i < n; ++i) d->slots[first + i].prepared = true;
    }
    return enqueue_match_stages(d, first, n, threshold, ranges, timed);
}
int enqueue_match(lm_detector* d, int first, int n, float threshold, int class_idx, bool timed = false) {
    std::vector<int> classes(1, class_idx);
    return enqueue_match(d, first, n, threshold, classes, timed);
}

inline bool key_less(const u64* a, const u64* b) { return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]); }

// The device sort's split form pays for lists longer than one chunk, and a launch lasts as long as its longest list: the score says
// whether any of the last 4096 collected frames had such a list.
inline void note_sort_length(lm_detector* d, u32 match_count) {
    const int is_long = match_count > LM_SORT_CHUNK && match_count <= LM_SORT_CAP;
    d->sort_long_score = is_long ? 4096 : std::max(d->sort_long_score - 1, 0);
}

// Delivers the sorted unique matches of one slot (the stream has been synchronised).
int collect_slot(lm_detector* d, int slot, lm_match_t* out, size_t cap, size_t* n_out) {
    const LmHostBlock* hb = d->host_block(slot);
    const LmHeader h = hb->hdr;
    if (h.cand_count > d->max_cand)
        return fail(LM_ERR_OVERFLOW, "scan produced " + std::to_string(h.cand_count) + " candidates, capacity " +
                                         std::to_string(d->max_cand) + " (raise lm_config.max_candidates)");
    if (h.match_count > d->max_match)
        return fail(LM_ERR_OVERFLOW, "refinement produced " + std::to_string(h.match_count) + " matches, capacity " +
                                         std::to_string(d->max_match) + " (raise lm_config.max_matches)");
    note_sort_length(d, h.match_count);
    size_t n = 0;
    if (h.sorted_on_device) {
        n = h.out_count;
        size_t ncopy = std::min(n, cap);
        size_t inl = std::min<size_t>(ncopy, LM_INLINE_MATCHES);
        if (out && inl) std::memcpy(out, hb->rec, inl * sizeof(lm_match_t));
        if (out && ncopy > inl)
            HIP_TRY(hipMemcpy(out + inl, reinterpret_cast<lm_match_t*>(d->aux(slot, d->off_out)) + inl,
                              (ncopy - inl) * sizeof(lm_match_t), hipMemcpyDeviceToHost));
    } else {
        // more than LM_SORT_CAP matches: sort + unique the keys on the host (same total order)
        std::vector<u64> keys((size_t)h.match_count * 2);
        if (h.match_count)
            HIP_TRY(hipMemcpy(keys.data(), d->aux(slot, d->off_keys), keys.size() * sizeof(u64), hipMemcpyDeviceToHost));
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

int ensure_staging(lm_detector* d, Slot& s) {
    const lm_config& c = d->cfg;
    if (!s.h_bgr) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_bgr), (size_t)c.width * c.height * 3));
    if (!s.h_depth && c.num_modalities == 2)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_depth), (size_t)c.width * c.height * 2));
    return LM_OK;
}

// Host waits until the slot's last copy-stream upload has landed (its staging buffer / the caller's pinned
// source may then be reused).  A stream's copies complete in ticket order, so this also retires its older tickets.
int wait_slot_upload(lm_detector* d, Slot& s) {
    if (s.up_seq > d->up_seq_done[s.up_stream]) {
        HIP_TRY(hipEventSynchronize(s.ev_up));
        d->up_seq_done[s.up_stream] = s.up_seq;
    }
    return LM_OK;
}

// Makes `stream` wait for the uploads of slots [first, first + n): per copy stream one hipStreamWaitEvent on the
// newest ticket among them (each copy stream is in order).  seqs[k]: that ticket (0: nothing pending on stream k).
int wait_uploads(lm_detector* d, hipStream_t stream, int first, int n, unsigned long long* seqs) {
    unsigned long long best[LM_NCOPY] = {};
    int bi[LM_NCOPY];
    for (int k = 0; k < LM_NCOPY; ++k) bi[k] = -1;
    for (int i = first; i < first + n; ++i) {
        const Slot& s = d->slots[i];
        if (s.up_seq > d->up_seq_done[s.up_stream] && s.up_seq > best[s.up_stream]) { best[s.up_stream] = s.up_seq; bi[s.up_stream] = i; }
    }
    for (int k = 0; k < LM_NCOPY; ++k) {
        if (bi[k] >= 0) HIP_TRY(hipStreamWaitEvent(stream, d->slots[bi[k]].ev_up, 0));
        if (seqs) seqs[k] = best[k];
    }
    return LM_OK;
}

// rows x row_bytes from a strided host image to dense device memory.  Pageable source: rows are packed into the
// slot's pinned staging buffer in `chunks` pieces, each followed by its own async copy, so the host memcpy of
// piece k + 1 overlaps the DMA of piece k.  pinned: the source goes straight to the DMA engine.
int copy_image(hipStream_t st, u8* dst, u8* staging, const u8* src, size_t stride, size_t row_bytes, int rows,
               bool pinned, int chunks) {
    if (pinned) {
        if (stride == row_bytes) HIP_TRY(hipMemcpyAsync(dst, src, row_bytes * rows, hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemcpy2DAsync(dst, row_bytes, src, stride, row_bytes, rows, hipMemcpyHostToDevice, st));
        return LM_OK;
    }
    if (chunks < 1) chunks = 1;
    for (int k = 0; k < chunks; ++k) {
        const int r0 = (int)((long long)rows * k / chunks), r1 = (int)((long long)rows * (k + 1) / chunks);
        if (r1 <= r0) continue;
        if (stride == row_bytes) lmh::copy_stream(staging + (size_t)r0 * row_bytes, src + (size_t)r0 * stride, (size_t)(r1 - r0) * row_bytes);
        else for (int y = r0; y < r1; ++y) lmh::copy_stream(staging + (size_t)y * row_bytes, src + (size_t)y * stride, row_bytes);
        lmh::copy_stream_fence();
        HIP_TRY(hipMemcpyAsync(dst + (size_t)r0 * row_bytes, staging + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes,
                               hipMemcpyHostToDevice, st));
    }
    return LM_OK;
}

// Rows [r0, r1) of the image translated by (ox, oy) pixels, zeros shifted in (cv::warpAffine with a pure translation as the
// reference's translateImg does, PoseDetection.cpp:54-59,192-197), into the dense staging buffer: the shift happens while the staging
// buffer is filled, so a shifted upload costs one pass over the image instead of two.  px = bytes per pixel.  Host memory only:
// any thread may fill disjoint row ranges (lm_stage_rows).  |ox| <= w and |oy| <= h (callers clamp: anything beyond is an all-zero frame).
void stage_rows_shifted(u8* staging, const u8* src, size_t stride, int w, int h, int px, int ox, int oy, int r0, int r1) {
    const size_t row_bytes = (size_t)w * px;
    const int x0 = std::max(ox, 0), x1 = std::min(w + ox, w);        // destination columns [x0, x1) have a source pixel
    const int y0 = std::max(oy, 0), y1 = std::min(h + oy, h);
    for (int y = r0; y < r1; ++y) {
        u8* row = staging + (size_t)y * row_bytes;
        if (y < y0 || y >= y1 || x1 <= x0) { std::memset(row, 0, row_bytes); continue; }
        if (x0 > 0) std::memset(row, 0, (size_t)x0 * px);
        lmh::copy_stream(row + (size_t)x0 * px, src + (size_t)(y - oy) * stride + (size_t)(x0 - ox) * px, (size_t)(x1 - x0) * px);     // non-temporal stores: the DMA engine reads this next
        if (x1 < w) std::memset(row + (size_t)x1 * px, 0, (size_t)(w - x1) * px);
    }
    lmh::copy_stream_fence();
}
inline int clamp_shift(int v, int extent) { return v < -extent ? -extent : (v > extent ? extent : v); }

int copy_image_shifted(hipStream_t st, u8* dst, u8* staging, const u8* src, size_t stride, int w, int h, int px, int ox, int oy) {
    stage_rows_shifted(staging, src, stride, w, h, px, ox, oy, 0, h);
    HIP_TRY(hipMemcpyAsync(dst, staging, (size_t)w * px * (size_t)h, hipMemcpyHostToDevice, st));
    return LM_OK;
}

// The translated image from PINNED host memory: the DMA engine copies the overlapping rectangle row by row (hipMemcpy2DAsync) behind
// a memset of the destination -- no staging copy, no host pass over the pixels at all.
int copy_image_shifted_pinned(hipStream_t st, u8* dst, const u8* src, size_t stride, int w, int h, int px, int ox, int oy) {
    const size_t row_bytes = (size_t)w * px;
    const int x0 = std::max(ox, 0), x1 = std::min(w + ox, w);
    const int y0 = std::max(oy, 0), y1 = std::min(h + oy, h);
    HIP_TRY(hipMemsetAsync(dst, 0, row_bytes * (size_t)h, st));
    if (x1 <= x0 || y1 <= y0) return LM_OK;
    if (stride == row_bytes) {
        // Dense rows: a rectangle copy whose rows start at odd byte offsets runs at a TENTH of the link rate (measured r05,
        // tools/time_shifted_upload.py: 1 148 instead of 115 us per 1280 x 960 RGB-D frame for a shift of 7 pixels).  The rows are contiguous on
        // both sides, so ONE linear copy displaced by ox pixels shifts every row at once; what it wraps from a row's end into the next row's
        // start (or the other way round) is cleared again by a device-side 2-D memset of |ox| columns.
        const size_t rows = (size_t)(y1 - y0), shift_bytes = (size_t)std::abs(ox) * px;
        const u8* s0 = src + (size_t)(y0 - oy) * stride;
        u8* d0 = dst + (size_t)y0 * row_bytes;
        if (ox > 0) HIP_TRY(hipMemcpyAsync(d0 + shift_bytes, s0, rows * row_bytes - shift_bytes, hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemcpyAsync(d0, s0 + shift_bytes, rows * row_bytes - shift_bytes, hipMemcpyHostToDevice, st));
        if (ox != 0) HIP_TRY(hipMemset2DAsync(ox > 0 ? d0 : d0 + row_bytes - shift_bytes, row_bytes, 0, shift_bytes, rows, st));
        return LM_OK;
    }
    HIP_TRY(hipMemcpy2DAsync(dst + (size_t)y0 * row_bytes + (size_t)x0 * px, row_bytes, src + (size_t)(y0 - oy) * stride + (size_t)(x0 - ox) * px, stride,
                             (size_t)(x1 - x0) * px, (size_t)(y1 - y0), hipMemcpyHostToDevice, st));
    return LM_OK;
}

// Frame -> slot.  inline_stream == nullptr: the copies go to the copy stream and the slot gets an upload ticket
// (consumers call wait_uploads); otherwise they are issued on `inline_stream` itself, in order with the kernels
// the caller enqueues behind them (single-frame calls: no cross-stream dependency on the latency path).
int upload_frame(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                 size_t depth_stride, bool pinned = false, hipStream_t inline_stream = nullptr, int shift_x = 0, int shift_y = 0) {
    const lm_config& c = d->cfg;
    Slot& s = d->slots[slot];
    if (!bgr) return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): colour image missing");
    if (c.num_modalities == 2 && !depth)
        return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): depth image missing");
    if (bgr_stride == 0) bgr_stride = (size_t)c.width * 3;
    if (depth_stride == 0) depth_stride = (size_t)c.width * 2;
    if (bgr_stride < (size_t)c.width * 3 || depth_stride < (size_t)c.width * 2) return fail(LM_ERR_INVALID, "stride smaller than a row");
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && slot >= ln.first && slot < ln.first + ln.n) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    if (int crc = refuse_checked_slots(d, slot, 1)) return crc;
    int rc;
    s.prepared = false; s.matched = false; s.mask_ready = false; s.staging_open = false;
    shift_x = clamp_shift(shift_x, c.width); shift_y = clamp_shift(shift_y, c.height);     // (beyond: an all-zero frame either way)
    // the slot's previous upload may still be reading the staging buffer (and must land before this one anyway)
    if ((rc = wait_slot_upload(d, s))) return rc;
    if (!pinned && (rc = ensure_staging(d, s))) return rc;
    const int cs = slot % d->n_copy_streams;
    hipStream_t st = inline_stream ? inline_stream : d->copy_stream[cs];
    const size_t bgr_bytes = (size_t)c.width * c.height * 3;
    const bool shifted = shift_x != 0 || shift_y != 0;
    if (pinned && !shifted && c.num_modalities == 2 && bgr_stride == (size_t)c.width * 3 && depth_stride == (size_t)c.width * 2 &&
        reinterpret_cast<const u8*>(depth) == bgr + bgr_bytes) {
        // [colour | depth] contiguous on the host, as in the frame arena: one DMA transfer
        HIP_TRY(hipMemcpyAsync(d->bgr(slot, 0), bgr, bgr_bytes + (size_t)c.width * c.height * 2, hipMemcpyHostToDevice, st));
        if (!inline_stream) {
            HIP_TRY(hipEventRecord(s.ev_bgr, st));
            HIP_TRY(hipEventRecord(s.ev_up, st));
            s.up_stream = cs;
            s.up_seq = d->up_seq_next[cs]++;
        }
        s.has_frame = true;
        return LM_OK;
    }
    if (shifted && pinned) {
        if ((rc = copy_image_shifted_pinned(st, d->bgr(slot, 0), bgr, bgr_stride, c.width, c.height, 3, shift_x, shift_y))) return rc;
    } else if (shifted) {
        if ((rc = copy_image_shifted(st, d->bgr(slot, 0), s.h_bgr, bgr, bgr_stride, c.width, c.height, 3, shift_x, shift_y))) return rc;
    } else if ((rc = copy_image(st, d->bgr(slot, 0), s.h_bgr, bgr, bgr_stride, (size_t)c.width * 3, c.height, pinned, d->stage_chunks))) return rc;
    if (!inline_stream && c.num_modalities == 2) HIP_TRY(hipEventRecord(s.ev_bgr, st));
    if (c.num_modalities == 2) {
        if (shifted && pinned) rc = copy_image_shifted_pinned(st, reinterpret_cast<u8*>(d->depth(slot)), reinterpret_cast<const u8*>(depth), depth_stride,
                                                               c.width, c.height, 2, shift_x, shift_y);
        else if (shifted) rc = copy_image_shifted(st, reinterpret_cast<u8*>(d->depth(slot)), reinterpret_cast<u8*>(s.h_depth),
                                             reinterpret_cast<const u8*>(depth), depth_stride, c.width, c.height, 2, shift_x, shift_y);
        else rc = copy_image(st, reinterpret_cast<u8*>(d->depth(slot)), reinterpret_cast<u8*>(s.h_depth),
                             reinterpret_cast<const u8*>(depth), depth_stride, (size_t)c.width * 2, c.height, pinned, d->stage_chunks);
        if (rc) return rc;
    }
    if (!inline_stream) {
        HIP_TRY(hipEventRecord(s.ev_up, st));
        s.up_stream = cs;
        s.up_seq = d->up_seq_next[cs]++;
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

// ---- lanes: two HIP streams with their own events and threshold table ---------------------------
void activate_lane(lm_detector* d, int l) {
    if (d->active == l) return;
    lm_detector::Lane& cur = d->lanes[d->active];
    cur.stream = d->stream; cur.d_raw_thr = d->d_raw_thr; cur.h_raw_thr = d->h_raw_thr; cur.raw_thr_for = d->raw_thr_for;
    std::memcpy(cur.waited_seq, d->waited_seq, sizeof(cur.waited_seq));
    for (int k = 0; k < 6; ++k) cur.ev[k] = d->ev[k];
    const lm_detector::Lane& nx = d->lanes[l];
    d->stream = nx.stream; d->d_raw_thr = nx.d_raw_thr; d->h_raw_thr = nx.h_raw_thr; d->raw_thr_for = nx.raw_thr_for;
    std::memcpy(d->waited_seq, nx.waited_seq, sizeof(d->waited_seq));
    for (int k = 0; k < 6; ++k) d->ev[k] = nx.ev[k];
    d->active = l;
}

int ensure_lane(lm_detector* d, int l) {
    lm_detector::Lane& ln = d->lanes[l];
    if (ln.created || l == 0) { ln.created = true; return LM_OK; }   // lane 0 = the detector's own stream (ensure_device)
    if (!ln.stream) HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    for (auto& e : ln.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ln.d_raw_thr), 128 * sizeof(int)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ln.h_raw_thr), 128 * sizeof(int), hipHostMallocDefault));
    ln.raw_thr_for = -1.0f;
    ln.created = true;
    return LM_OK;
}

void account_profile(lm_detector* d, int n, const std::vector<int>& classes, bool gathered = false) {
    for (int k = 0; k < 4; ++k) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, d->ev[k], d->ev[k + 1]) == hipSuccess) d->prof_us[k] += (double)ms * 1000.0;
    }
    if (gathered) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, d->ev[4], d->ev[5]) == hipSuccess) { d->prof_exch_us += (double)ms * 1000.0; d->prof_exch_launches += 1; }
    }
    double b = 0;
    for (int c : classes) {
        if (c < 0) for (double v : d->hb.class_alg_bytes) b += v;
        else if (c < (int)d->hb.class_alg_bytes.size()) b += d->hb.class_alg_bytes[c];
    }
    d->prof_scan_bytes += b * n;
    d->prof_launches += 1;
    d->prof_frames += n;
}

// Waits for everything enqueued on the active lane's stream.  LM_FLAG_BLOCKING_SYNC: sleep on an event created with
// hipEventBlockingSync instead of spinning in hipStreamSynchronize (for hosts with fewer CPUs than busy processes).
int wait_stream(lm_detector* d) {
    if (d->cfg.flags & LM_FLAG_BLOCKING_SYNC) {
        hipEvent_t& ev = d->blocking_ev[d->active];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, d->stream));
        HIP_TRY(hipEventSynchronize(ev));
    } else {
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    // everything this stream was told to wait for has landed
    for (int k = 0; k < LM_NCOPY; ++k) if (d->waited_seq[k] > d->up_seq_done[k]) d->up_seq_done[k] = d->waited_seq[k];
    return LM_OK;
}

// lm_match_end's wait (r05).  hipStreamSynchronize on a stream whose work is long done still enqueues a marker and waits for it, and
// HIP streams share a few hardware queues: the marker lands behind the event-record barriers of a COPY stream on the same queue,
// which wait for the NEXT batch's H2D copies -- measured in the streamed PoseDetection (profiles/r05_e2e_timeline.txt): the lane's
// kernels had finished 4.6 ms earlier, the sync still took 0.43-0.52 ms = the next batch's upload.  An event recorded right behind the
// lane's last command when it was enqueued has no such false dependency.
int wait_lane_done(lm_detector* d, lm_detector::Lane& ln) {
    HIP_TRY(hipEventSynchronize(ln.ev_done));
    for (int k = 0; k < LM_NCOPY; ++k) if (d->waited_seq[k] > d->up_seq_done[k]) d->up_seq_done[k] = d->waited_seq[k];
    return LM_OK;
}

int run_match(lm_detector* d, int first, int n, float threshold, std::vector<int> classes, bool prepared = false) {
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    for (int i = 0; i < n; ++i) {
        if (!d->slots[first + i].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot " + std::to_string(first + i));
        if (prepared && !d->slots[first + i].prepared)
            return fail(LM_ERR_INVALID, "slot " + std::to_string(first + i) + " holds no current a3-a10 results: run lm_prepare_slot or a match on it first");
    }
    int rc;
    if ((rc = enqueue_match(d, first, n, threshold, classes, d->profiling, prepared))) return rc;
    if ((rc = wait_stream(d))) return rc;
    if (d->profiling) account_profile(d, n, classes);   // HIP events on the launch stream bracket every stage
    for (int i = 0; i < n; ++i) d->slots[first + i].matched = true;
    return LM_OK;
}
int run_match(lm_detector* d, int first, int n, float threshold, int class_idx) {
    return run_match(d, first, n, threshold, std::vector<int>(1, class_idx));
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* lm_last_error(void) { return g_err.c_str(); }
const char* lm_version(void) { return "linemod_hip 0.3 (gfx950)"; }

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
    if (c.max_candidates <= 0) c.max_candidates = 1 << 18;
    if (c.max_matches <= 0) c.max_matches = 1 << 18;
    if (c.frame_slots <= 0) c.frame_slots = 8;
    if (c.frame_slots > 1024) return fail(LM_ERR_INVALID, "frame_slots out of range");
    {
        std::string why;   // negative / non-finite thresholds would select kernel paths that were never meant to see them
        if (!lmh::check_modality_params(c, why)) return fail(LM_ERR_INVALID, why);
    }
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
        // pad: one full linear memory (a scan may start W*H-1 bytes into the last memory and read
        // template_positions <= W*H bytes) + the 16-row patch of the refinement + vector-load slack
        size_t pad = align_up((size_t)g.wh + 16 * (size_t)g.W + 2 * LM_SCAN_CHUNK + 64, 256);
        // the lowest level is scanned (8 response memories per modality); the levels above it are only
        // refined at and keep one spread linear memory per modality (1/8 of the bytes)
        g.spread_only = (l + 1 < c.pyramid_levels) ? 1 : 0;
        // responses are <= 4: the scanned level packs two positions per byte when the linearize fast path applies
        g.nibble = (!g.spread_only && !(c.flags & LM_FLAG_BYTE_RESPONSES) && lmk_nibble_supported(w, h, T)) ? 1 : 0;
        size_t ori = align_up(((size_t)T * T * g.wh) >> g.nibble, 256) + pad;
        // r05: next to the nibble memories one miss BIT per position and orientation (k_scan1); a scan may start wh - 1 bits into the last
        // plane's last memory and its 64 lanes read 128 bits each
        size_t plane = g.nibble ? align_up(((size_t)T * T * g.wh + 7) / 8 + ((size_t)g.wh + 64 * 128) / 8 + 64, 256) : 0;
        if (g.nibble && (size_t)c.num_modalities * (8 * ori + 8 * plane) + pad > 0x1FFFFFFFull) plane = 0;     // bit offsets are 32-bit
        size_t mod = g.spread_only ? ori : 8 * ori + 8 * plane;
        size_t arena = (size_t)c.num_modalities * mod + pad;
        if (arena > (g.spread_only ? 0x1FFFFFFFull : g.nibble ? 0x7FFFFFFFull : 0xFFFFFFFFull)) { delete d; return fail(LM_ERR_INVALID, "frame too large for the arena offset encoding"); }
        g.ori_stride = (u32)ori;
        g.plane_ori = (u32)plane;
        g.mod_stride = (u32)mod;
        g.zero_off = (u32)((size_t)c.num_modalities * g.mod_stride);
        g.arena_bytes = (u32)arena;
    }
    lmh::default_similarity_lut(d->sim_lut);
    lmh::default_normal_lut(d->normal_lut);
    *out = d;
    return LM_OK;
}

static void free_gather(lm_detector* d);

void lm_destroy(lm_detector* d) {
    if (!d) return;
    if (d->dev_ready) {
        hipSetDevice(d->cfg.device);
        hipDeviceSynchronize();
        if (d->d_refine_stat) {
            unsigned long long h[8] = {};
            if (hipMemcpy(h, d->d_refine_stat, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "LM_REFINE_STAT: refined alone %llu, in pairs %llu | pair candidates pruned %llu (both of a pair: %llu pairs) | single candidates pruned %llu | dropped by the final test %llu\n",
                        h[0], h[1], h[2], h[3], h[4], h[5]);
            hipFree(d->d_refine_stat);
        }
        for (Slot& s : d->slots) {
            if (s.h_bgr) hipHostFree(s.h_bgr);
            if (s.h_depth) hipHostFree(s.h_depth);
            if (s.ev_up) hipEventDestroy(s.ev_up);
            if (s.ev_bgr) hipEventDestroy(s.ev_bgr);
        }
        for (auto& cs : d->copy_stream) if (cs) hipStreamDestroy(cs);
        hipFree(d->frame_arena); hipFree(d->aux_arena); hipHostFree(d->host_blocks);
        hipFree(d->d_raw_thr); hipHostFree(d->h_raw_thr); hipFree(d->d_plan);
        for (auto& q : d->d_surv) { hipFree(q); q = nullptr; }
        activate_lane(d, 0);
        for (auto& ev : d->blocking_ev) if (ev) hipEventDestroy(ev);
        for (auto& ev : d->ev) if (ev) hipEventDestroy(ev);
        if (d->stream) hipStreamDestroy(d->stream);
        for (int l = 1; l < LM_NLANES; ++l) {
            lm_detector::Lane& ln = d->lanes[l];
            if (ln.created) {
                hipStreamSynchronize(ln.stream);
                for (auto& ev : ln.ev) if (ev) hipEventDestroy(ev);
                hipFree(ln.d_raw_thr); hipHostFree(ln.h_raw_thr);
            }
            if (ln.stream) hipStreamDestroy(ln.stream);
        }
        for (auto& ln : d->lanes) if (ln.ev_done) hipEventDestroy(ln.ev_done);
        free_device_bank(d);
        for (auto& c : d->comm) { delete c; c = nullptr; }
        free_gather(d);
        hipFree(d->d_scan_stat);
        hipFree(d->d_hull_class_base); hipFree(d->d_hull_off); hipFree(d->d_hull_xy); hipFree(d->d_hsv_div);
        if (d->cc_stream) hipStreamDestroy(d->cc_stream);
        hipFree(d->cc_dev); if (d->cc_host) hipHostFree(d->cc_host);
        hipFree(d->dc_dev); if (d->dc_host) hipHostFree(d->dc_host);
        for (hipEvent_t& ev : d->mask_done) if (ev) hipEventDestroy(ev);
        if (d->cc_done) hipEventDestroy(d->cc_done);
        if (d->dc_done) hipEventDestroy(d->dc_done);
        hipFree(d->d_resp_tab); hipFree(d->d_sim_lut); hipFree(d->d_normal_lut); hipFree(d->d_scratch);
    }
    delete d;
}

int lm_set_similarity_lut(lm_detector* d, const uint8_t lut[256]) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !lut) return fail(LM_ERR_INVALID, "null argument");
    for (int i = 0; i < 256; ++i) if (lut[i] > 4) return fail(LM_ERR_INVALID, "similarity LUT entries must be <= 4 (63*4 must fit a byte)");
    // an empty spread value must score 0: reads past a linear memory land in zero padding (upstream: undefined)
    for (int o = 0; o < 8; ++o) if (lut[32 * o] || lut[32 * o + 16]) return fail(LM_ERR_INVALID, "similarity LUT must map an empty nibble to 0");
    std::memcpy(d->sim_lut, lut, 256); d->luts_dirty = true;
    for (Slot& sl : d->slots) sl.prepared = false;
    return LM_OK;
}
int lm_set_normal_lut(lm_detector* d, const uint8_t lut[8000]) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !lut) return fail(LM_ERR_INVALID, "null argument");
    std::memcpy(d->normal_lut, lut, 8000); d->luts_dirty = true; d->lut_onehot = -1; d->normal_lut_substitute = false;
    for (Slot& sl : d->slots) sl.prepared = false;
    return LM_OK;
}
int lm_get_similarity_lut(const lm_detector* d, uint8_t lut[256]) { if (!d || !lut) return fail(LM_ERR_INVALID, "null argument"); std::memcpy(lut, d->sim_lut, 256); return LM_OK; }
int lm_normal_lut_is_substitute(const lm_detector* d) { return (d && d->normal_lut_substitute) ? 1 : 0; }
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
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !class_id || n_templates < 0 || (n_templates && (!descs || !features))) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    int ci = d->bank.add_class(class_id, n_templates, descs, features, d->cfg.pyramid_levels, d->cfg.num_modalities, err);
    if (ci < 0) return fail(LM_ERR_INVALID, err);
    if (class_idx_out) *class_idx_out = ci;
    d->bank_dirty = true; d->hulls_dirty = true;
    return LM_OK;
}

int lm_add_template(lm_detector* d, const char* class_id, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                    size_t depth_stride, const uint8_t* mask, size_t mask_stride, int* template_id_out, lm_rect* bbox_out) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (template_id_out) *template_id_out = -1;
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!class_id) return fail(LM_ERR_INVALID, "null class id");
    const lm_config& c = d->cfg;
    const int M = c.num_modalities, L = c.pyramid_levels;
    if ((rc = upload_frame(d, 0, bgr, bgr_stride, depth, depth_stride, false, d->stream))) return rc;
    d->slots[0].has_frame = false;  // slot 0 now holds a template image, not a scene frame
    d->slots[0].prepared = false;
    // quantise every level on the GPU, keeping the gradient magnitude this time
    size_t mag_off[LM_MAX_LEVELS], total = 0;
    for (int l = 0; l < L; ++l) { mag_off[l] = total; total += align_up((size_t)d->lw[l] * d->lh[l] * sizeof(float), 256); }
    if ((rc = ensure_scratch(d, total))) return rc;
    u8* scratch = static_cast<u8*>(d->d_scratch);
    for (int l = 0; l < L; ++l) {
        if (l > 0) lmk_pyrdown(d->stream, d->bgr(0, l - 1), d->lw[l - 1], d->lh[l - 1], d->bgr(0, l), 0, 1);
        lmk_color_quantize(d->stream, d->bgr(0, l), d->lw[l], d->lh[l], c.weak_threshold, d->quant(0, l, 0),
                           reinterpret_cast<float*>(scratch + mag_off[l]), d->cscratch(0, l), 0, 1);
    }
    if (M == 2) {
        lmk_depth_quantize(d->stream, d->depth(0), d->lw[0], d->lh[0], c.distance_threshold, c.difference_threshold,
                           d->d_normal_lut, normal_lut_onehot(d), d->quant(0, 0, 1), d->dscratch(0), 0, 1);
        enqueue_depth_pyramid(d, 0, 1);
    }
    std::vector<lmh::ExtractLevel> lv(L);
    for (int l = 0; l < L; ++l) {
        size_t px = (size_t)d->lw[l] * d->lh[l];
        lv[l].w = d->lw[l]; lv[l].h = d->lh[l];
        lv[l].color_q.resize(px); lv[l].color_mag.resize(px);
        HIP_TRY(hipMemcpyAsync(lv[l].color_q.data(), d->quant(0, l, 0), px, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipMemcpyAsync(lv[l].color_mag.data(), scratch + mag_off[l], px * sizeof(float), hipMemcpyDeviceToHost, d->stream));
        if (M == 2) {
            lv[l].depth_q.resize(px);
            HIP_TRY(hipMemcpyAsync(lv[l].depth_q.data(), d->quant(0, l, 1), px, hipMemcpyDeviceToHost, d->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
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
    d->bank_dirty = true; d->hulls_dirty = true;
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
    if (features && !t.features.empty()) std::memcpy(features, t.features.data(), t.features.size() * sizeof(lm_feature));
    return LM_OK;
}

int lm_upload_frame(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                    size_t depth_stride) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    return upload_frame(d, slot, bgr, bgr_stride, depth, depth_stride);
}

int lm_upload_frame_shifted(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                            size_t depth_stride, int shift_x, int shift_y) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    return upload_frame(d, slot, bgr, bgr_stride, depth, depth_stride, false, nullptr, shift_x, shift_y);
}

struct PinnedBlock { const u8* p; size_t bytes; };
static std::mutex g_pinned_mu;
static std::vector<PinnedBlock> g_pinned;      // blocks handed out by lm_host_alloc

int lm_upload_frame_pinned(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                           size_t depth_stride) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    return upload_frame(d, slot, bgr, bgr_stride, depth, depth_stride, true);
}

int lm_upload_frames_pinned(lm_detector* d, int first_slot, int n_slots, const uint8_t* frames, size_t frame_stride) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if (!frames || n_slots <= 0) return fail(LM_ERR_INVALID, "bad argument");
    const lm_config& c = d->cfg;
    const size_t fb = (size_t)c.width * c.height * (c.num_modalities == 2 ? 5 : 3);
    if (frame_stride == 0) frame_stride = fb;
    if (frame_stride < fb) return fail(LM_ERR_INVALID, "frame stride smaller than a frame");
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && first_slot < ln.first + ln.n && ln.first < first_slot + n_slots) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    if (int crc = refuse_checked_slots(d, first_slot, n_slots)) return crc;
    for (int i = 0; i < n_slots; ++i) if ((rc = wait_slot_upload(d, d->slots[first_slot + i]))) return rc;
    // one strided transfer: row i = host frame i ([colour | depth] dense), destination pitch = the arena's slot stride
    const int cs = (first_slot / n_slots) % d->n_copy_streams;   // consecutive runs of n_slots slots take turns on the copy streams
    hipStream_t st = d->copy_stream[cs];
    HIP_TRY(hipMemcpy2DAsync(d->bgr(first_slot, 0), d->frame_stride, frames, frame_stride, fb, (size_t)n_slots, hipMemcpyHostToDevice, st));
    const unsigned long long seq = d->up_seq_next[cs]++;
    for (int i = 0; i < n_slots; ++i) {
        Slot& s = d->slots[first_slot + i];
        if (c.num_modalities == 2) HIP_TRY(hipEventRecord(s.ev_bgr, st));
        HIP_TRY(hipEventRecord(s.ev_up, st));
        s.up_stream = cs; s.up_seq = seq; s.has_frame = true; s.prepared = false; s.matched = false; s.mask_ready = false; s.staging_open = false;
    }
    return LM_OK;
}

int lm_upload_frame_pinned_shifted(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                                   size_t depth_stride, int shift_x, int shift_y) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    return upload_frame(d, slot, bgr, bgr_stride, depth, depth_stride, true, nullptr, shift_x, shift_y);
}

// ---- staged uploads (r05): the staging copy of a pageable frame split from the transfer, so that a host thread pool fills the
// pinned staging buffers of a batch (row ranges in parallel) while the thread that owns the detector does something else:
//     lm_stage_reserve(first, n)                       owner thread: the slots' earlier uploads have landed, staging exists
//     lm_stage_rows(slot, ..., row0, row1)  x many     ANY thread, disjoint row ranges: host memory only, no HIP call
//     lm_upload_staged(slot)                           owner thread: the H2D copies + the slot's upload ticket
int lm_stage_reserve(lm_detector* d, int first_slot, int n_slots) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && first_slot < ln.first + ln.n && ln.first < first_slot + n_slots) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    if (int crc = refuse_checked_slots(d, first_slot, n_slots)) return crc;
    for (int i = 0; i < n_slots; ++i) {
        Slot& s = d->slots[first_slot + i];
        if ((rc = wait_slot_upload(d, s))) return rc;
        if ((rc = ensure_staging(d, s))) return rc;
        s.staging_open = true;
    }
    return LM_OK;
}

int lm_stage_rows(lm_detector* d, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                  int shift_x, int shift_y, int row0, int row1) {
    if (!d || !d->dev_ready) return fail(LM_ERR_INVALID, "lm_stage_reserve first");
    if (slot < 0 || slot >= (int)d->slots.size()) return fail(LM_ERR_INVALID, "slot out of range");
    const lm_config& c = d->cfg;
    const Slot& s = d->slots[slot];
    if (!s.staging_open || !s.h_bgr) return fail(LM_ERR_INVALID, "lm_stage_reserve first");
    if (!bgr) return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): colour image missing");
    if (c.num_modalities == 2 && !depth) return fail(LM_ERR_INVALID, "sources.size() != modalities.size(): depth image missing");
    if (bgr_stride == 0) bgr_stride = (size_t)c.width * 3;
    if (depth_stride == 0) depth_stride = (size_t)c.width * 2;
    if (bgr_stride < (size_t)c.width * 3 || depth_stride < (size_t)c.width * 2) return fail(LM_ERR_INVALID, "stride smaller than a row");
    if (row0 < 0 || row1 > c.height || row0 > row1) return fail(LM_ERR_INVALID, "row range outside the frame");
    const int ox = clamp_shift(shift_x, c.width), oy = clamp_shift(shift_y, c.height);
    stage_rows_shifted(s.h_bgr, bgr, bgr_stride, c.width, c.height, 3, ox, oy, row0, row1);
    if (c.num_modalities == 2)
        stage_rows_shifted(reinterpret_cast<u8*>(s.h_depth), reinterpret_cast<const u8*>(depth), depth_stride, c.width, c.height, 2, ox, oy, row0, row1);
    return LM_OK;
}

int lm_upload_staged(lm_detector* d, int slot) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    Slot& s = d->slots[slot];
    if (!s.staging_open) return fail(LM_ERR_INVALID, "lm_stage_reserve first");
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && slot >= ln.first && slot < ln.first + ln.n) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    if (int crc = refuse_checked_slots(d, slot, 1)) return crc;
    const lm_config& c = d->cfg;
    const int cs = slot % d->n_copy_streams;
    hipStream_t st = d->copy_stream[cs];
    s.prepared = false; s.matched = false; s.mask_ready = false; s.staging_open = false;
    HIP_TRY(hipMemcpyAsync(d->bgr(slot, 0), s.h_bgr, (size_t)c.width * c.height * 3, hipMemcpyHostToDevice, st));
    if (c.num_modalities == 2) {
        HIP_TRY(hipEventRecord(s.ev_bgr, st));
        HIP_TRY(hipMemcpyAsync(d->depth(slot), s.h_depth, (size_t)c.width * c.height * 2, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(s.ev_up, st));
    s.up_stream = cs;
    s.up_seq = d->up_seq_next[cs]++;
    s.has_frame = true;
    return LM_OK;
}

int lm_upload_wait(lm_detector* d, int slot) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (slot < 0) {
        for (int k = 0; k < LM_NCOPY; ++k) {
            HIP_TRY(hipStreamSynchronize(d->copy_stream[k]));
            d->up_seq_done[k] = d->up_seq_next[k] - 1;
        }
        return LM_OK;
    }
    if ((rc = check_slots(d, slot, 1))) return rc;
    return wait_slot_upload(d, d->slots[slot]);
}

int lm_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return fail(LM_ERR_INVALID, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LM_ERR_NO_DEVICE, "no HIP device available: pinned host memory needs the HIP runtime");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        g_pinned.push_back({reinterpret_cast<const u8*>(*out), bytes});
    }
    return LM_OK;
}

void lm_host_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].p == p) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}

int lm_set_stage_chunks(lm_detector* d, int chunks) {
    if (!d || chunks < 1 || chunks > 64) return fail(LM_ERR_INVALID, "bad argument");
    d->stage_chunks = chunks;
    return LM_OK;
}

int lm_set_tuning(lm_detector* d, int key, int value) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    switch (key) {
        case LM_TUNE_CBLUR_VARIANT: if (value < 0 || value > 4 || value == 2) break; lmk_set_cblur_variant(value); return LM_OK;
        case LM_TUNE_PHASE_MAX_SLOTS: if (value < 0) break; d->phase_max_slots = value; return LM_OK;
        case LM_TUNE_CGRAD_VARIANT: if (value < 0 || value > 3) break; lmk_set_cgrad_variant(value); return LM_OK;
        case LM_TUNE_COPY_STREAMS: if (value < 1 || value > LM_NCOPY) break; d->n_copy_streams = value; return LM_OK;
        case LM_TUNE_BATCH_PHASES: if (value < 0 || value > 2) break; d->batch_phases = value; return LM_OK;
        case LM_TUNE_PYRDOWN_VARIANT: if (value < 0 || value > 2) break; lmk_set_pyrdown_variant(value); return LM_OK;
        case LM_TUNE_BLUR_PYR: if (value < 0 || value > 3) break; lmk_set_blur_pyr(value != 0); lmk_set_blur_pyr_interleave(value == 2 ? 1 : value == 3 ? 2 : 0); return LM_OK;
        case LM_TUNE_BLUR_STRIP: if (value != 0 && value != 16 && value != 32 && value != 64) break; lmk_set_blur_strip(value); return LM_OK;
        case LM_TUNE_DMEDIAN_VARIANT: if (value < 0 || value > 2) break; lmk_set_dmedian_variant(value); return LM_OK;
        case LM_TUNE_WORK_WEIGHT: if (value < 0 || value > 1) break; d->work_weight_by_pixels = value; return LM_OK;
        case LM_TUNE_SORT_SPLIT: if (value < 0 || value > 2) break; d->sort_split_mode = value; return LM_OK;
        case LM_TUNE_SCAN_LIST_ORDER: if (value < 0 || value > 3) break; d->scan_list_order = value; d->bank_dirty = true; return LM_OK;
        case LM_TUNE_SCAN_FORM:
            if (value < 0 || value > 3) break;
            if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
            d->scan_form = value;
            for (Slot& sl : d->slots) sl.prepared = false;          // (prepared slots may lack the miss planes the new form reads)
            return LM_OK;
        case LM_TUNE_SCAN1_MIN_THRESHOLD: if (value < 0 || value > 100) break; d->scan1_min_threshold = (float)value; return LM_OK;
        case LM_TUNE_CGRAD_LEVELS: if (value < 0 || value > 1) break; lmk_set_cgrad_levels(value); return LM_OK;
        default: return fail(LM_ERR_INVALID, "unknown tuning key");
    }
    return fail(LM_ERR_INVALID, "tuning value out of range");
}

int lm_match_slot(lm_detector* d, int slot, float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if ((rc = run_match(d, slot, 1, threshold, class_idx))) return rc;
    return collect_slot(d, slot, out, cap, n_out);
}

// [p, p + bytes) inside a block from lm_host_alloc?  (A table of our own: asking the runtime about a pageable pointer,
// hipPointerGetAttributes, costs 10-20 us per call -- measured -- on the path this is meant to shorten.)
static bool is_pinned_host(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(g_pinned_mu);
    for (const PinnedBlock& b : g_pinned)
        if (reinterpret_cast<const u8*>(p) >= b.p && reinterpret_cast<const u8*>(p) + bytes <= b.p + b.bytes) return true;
    return false;
}

static int match_host_frame(lm_detector* d, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                            float threshold, std::vector<int> classes, lm_match_t* out, size_t cap, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    // the copies go inline on the compute stream, in order with the kernels (r02 measured the copy-stream forms -- one stream with split
    // events, pieces over all copy streams -- slower for one frame: 133 -> 167 us; they were deleted in r05).  Frames that already live
    // in pinned host memory (lm_host_alloc, hipHostMalloc, hipHostRegister) skip the staging copy
    const size_t hh = (size_t)d->cfg.height;
    const bool pinned = bgr && is_pinned_host(bgr, (bgr_stride ? bgr_stride : (size_t)d->cfg.width * 3) * hh) &&
                        (d->cfg.num_modalities < 2 || (depth && is_pinned_host(depth, (depth_stride ? depth_stride : (size_t)d->cfg.width * 2) * hh)));
    if ((rc = upload_frame(d, 0, bgr, bgr_stride, depth, depth_stride, pinned, d->stream))) return rc;
    if ((rc = run_match(d, 0, 1, threshold, std::move(classes)))) return rc;
    return collect_slot(d, 0, out, cap, n_out);
}

int lm_match(lm_detector* d, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
             float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out) {
    return match_host_frame(d, bgr, bgr_stride, depth, depth_stride, threshold, std::vector<int>(1, class_idx), out, cap, n_out);
}

int lm_match_classes(lm_detector* d, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                     float threshold, const int32_t* class_idxs, int n_classes, lm_match_t* out, size_t cap, size_t* n_out) {
    if (n_classes < 0 || (n_classes && !class_idxs)) return fail(LM_ERR_INVALID, "bad class list");
    return match_host_frame(d, bgr, bgr_stride, depth, depth_stride, threshold, std::vector<int>(class_idxs, class_idxs + n_classes), out, cap, n_out);
}

int lm_match_batch(lm_detector* d, int n_slots, float threshold, int class_idx, lm_match_t* out, size_t cap_per_frame,
                   int32_t* counts) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if ((rc = check_slots(d, 0, n_slots))) return rc;
    if (n_slots == 0) return LM_OK;
    if ((rc = run_match(d, 0, n_slots, threshold, class_idx))) return rc;
    int first_err = LM_OK;
    std::string first_msg;
    for (int i = 0; i < n_slots; ++i) {
        size_t n = 0;
        rc = collect_slot(d, i, out ? out + (size_t)i * cap_per_frame : nullptr, cap_per_frame, &n);
        if (counts) counts[i] = (int32_t)n;
        if (rc && !first_err) { first_err = rc; first_msg = g_err; }
    }
    if (first_err) return fail(first_err, first_msg);
    return LM_OK;
}

// Detector::match(sources, threshold, matches, class_ids) with upstream's class LIST: a3-a10 once per frame, one
// scan launch per run of neighbouring classes, one refinement, one sort; the lists hold the matches of all the named
// classes in the total order (HighLevelLinemod.cpp:145,152).
static int collect_range(lm_detector* d, int first, int n, lm_match_t* out, size_t cap_per_frame, int32_t* counts) {
    int first_err = LM_OK;
    std::string first_msg;
    for (int i = 0; i < n; ++i) {
        size_t k = 0;
        int rc = collect_slot(d, first + i, out ? out + (size_t)i * cap_per_frame : nullptr, cap_per_frame, &k);
        if (counts) counts[i] = (int32_t)k;
        if (rc && !first_err) { first_err = rc; first_msg = g_err; }
    }
    if (first_err) return fail(first_err, first_msg);
    return LM_OK;
}

int lm_match_batch_classes(lm_detector* d, int first_slot, int n_slots, float threshold, const int32_t* class_idxs, int n_classes,
                           lm_match_t* out, size_t cap_per_frame, int32_t* counts) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if (n_classes < 0 || (n_classes && !class_idxs)) return fail(LM_ERR_INVALID, "bad class list");
    if (n_slots == 0) return LM_OK;
    if ((rc = run_match(d, first_slot, n_slots, threshold, std::vector<int>(class_idxs, class_idxs + n_classes)))) return rc;
    return collect_range(d, first_slot, n_slots, out, cap_per_frame, counts);
}

// a11-a15 only, on slots whose a3-a10 results are current (a match or lm_prepare_slot has run on the frame the slot
// holds, no upload and no LUT change since): the reference's per-class detect calls on ONE camera frame
// (PoseDetection.cpp:45-66 per class name) pay the pre-processing once.
int lm_match_prepared(lm_detector* d, int first_slot, int n_slots, float threshold, const int32_t* class_idxs, int n_classes,
                      lm_match_t* out, size_t cap_per_frame, int32_t* counts) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if (n_classes < 0 || (n_classes && !class_idxs)) return fail(LM_ERR_INVALID, "bad class list");
    if (n_slots == 0) return LM_OK;
    if ((rc = run_match(d, first_slot, n_slots, threshold, std::vector<int>(class_idxs, class_idxs + n_classes), true))) return rc;
    return collect_range(d, first_slot, n_slots, out, cap_per_frame, counts);
}

int lm_synchronize(lm_detector* d) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return LM_OK;
}

static int enqueue_gather(lm_detector* d, int lane, int first, int n);

static int begin_lane(lm_detector* d, int lane, int first_slot, int n_slots, float threshold, std::vector<int> classes, bool gathered) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (lane < 0 || lane >= LM_NLANES) return fail(LM_ERR_INVALID, "lane out of range (0 .. 3)");
    if (gathered && !d->comm[0]) return fail(LM_ERR_INVALID, "no communicator: call lm_comm_init first");
    if ((rc = ensure_bank(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if (n_slots <= 0) return fail(LM_ERR_INVALID, "no slots");
    lm_detector::Lane& ln = d->lanes[lane];
    if (ln.busy) return fail(LM_ERR_INVALID, "lane is busy: call lm_match_end first");
    for (int o = 0; o < LM_NLANES; ++o) {
        const lm_detector::Lane& other = d->lanes[o];
        if (o != lane && other.busy && first_slot < other.first + other.n && other.first < first_slot + n_slots)
            return fail(LM_ERR_INVALID, "slot range overlaps the range another lane is working on");
    }
    for (int i = 0; i < n_slots; ++i)
        if (!d->slots[first_slot + i].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot " + std::to_string(first_slot + i));
    if ((rc = ensure_lane(d, lane))) return rc;
    activate_lane(d, lane);
    rc = enqueue_match(d, first_slot, n_slots, threshold, classes, d->profiling);
    if (!rc && gathered) rc = enqueue_gather(d, lane, first_slot, n_slots);
    if (!rc) {
        if (!ln.ev_done && hipEventCreateWithFlags(&ln.ev_done, ((d->cfg.flags & LM_FLAG_BLOCKING_SYNC) ? hipEventBlockingSync : 0) | hipEventDisableTiming) != hipSuccess) {
            ln.ev_done = nullptr;
            rc = fail(LM_ERR_HIP, "hipEventCreate failed");
        }
        if (!rc && hipEventRecord(ln.ev_done, d->stream) != hipSuccess) rc = fail(LM_ERR_HIP, "hipEventRecord failed");
        if (rc) (void)hipStreamSynchronize(d->stream);      // what was enqueued must not outlive the failed call
    }
    if (!rc) {
        ln.busy = true; ln.first = first_slot; ln.n = n_slots; ln.classes = classes; ln.timed = d->profiling;
        d->gather[lane].active = gathered;
    }
    activate_lane(d, 0);
    return rc;
}

int lm_match_begin(lm_detector* d, int lane, int first_slot, int n_slots, float threshold, int class_idx) {
    return begin_lane(d, lane, first_slot, n_slots, threshold, std::vector<int>(1, class_idx), false);
}

int lm_match_begin_classes(lm_detector* d, int lane, int first_slot, int n_slots, float threshold, const int32_t* class_idxs,
                           int n_classes) {
    if (n_classes < 0 || (n_classes && !class_idxs)) return fail(LM_ERR_INVALID, "bad class list");
    return begin_lane(d, lane, first_slot, n_slots, threshold, std::vector<int>(class_idxs, class_idxs + n_classes), false);
}

int lm_match_begin_gathered(lm_detector* d, int lane, int first_slot, int n_slots, float threshold, int class_idx) {
    return begin_lane(d, lane, first_slot, n_slots, threshold, std::vector<int>(1, class_idx), true);
}

int lm_match_end(lm_detector* d, int lane, lm_match_t* out, size_t cap_per_frame, int32_t* counts) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (lane < 0 || lane >= LM_NLANES) return fail(LM_ERR_INVALID, "lane out of range (0 .. 3)");
    lm_detector::Lane& ln = d->lanes[lane];
    if (!ln.busy) return fail(LM_ERR_INVALID, "lane has no match in flight");
    if (d->gather[lane].active) return fail(LM_ERR_INVALID, "the lane's match was begun with lm_match_begin_gathered: collect it with lm_match_end_gathered");
    HIP_TRY(hipSetDevice(d->cfg.device));
    activate_lane(d, lane);
    const int wrc = wait_lane_done(d, ln);
    if (!wrc && ln.timed) account_profile(d, ln.n, ln.classes);
    activate_lane(d, 0);
    ln.busy = false;
    if (wrc) return wrc;
    for (int i = 0; i < ln.n; ++i) d->slots[ln.first + i].matched = true;
    int first_err = LM_OK;
    std::string first_msg;
    for (int i = 0; i < ln.n; ++i) {
        size_t n = 0;
        int rc = collect_slot(d, ln.first + i, out ? out + (size_t)i * cap_per_frame : nullptr, cap_per_frame, &n);
        if (counts) counts[i] = (int32_t)n;
        if (rc && !first_err) { first_err = rc; first_msg = g_err; }
    }
    if (first_err) return fail(first_err, first_msg);
    return LM_OK;
}

// The lists of the last completed match on slots [first_slot, first_slot + n_slots) once more (they stay in the slots' result blocks
// until the next upload or match): what a caller does after LM_ERR_OVERFLOW told it the capacity it needs -- no second pass over the GPU.
int lm_match_collect(lm_detector* d, int first_slot, int n_slots, lm_match_t* out, size_t cap_per_frame, int32_t* counts) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && first_slot < ln.first + ln.n && ln.first < first_slot + n_slots) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    for (int i = 0; i < n_slots; ++i)
        if (!d->slots[first_slot + i].matched) return fail(LM_ERR_INVALID, "slot " + std::to_string(first_slot + i) + " holds no completed match");
    return collect_range(d, first_slot, n_slots, out, cap_per_frame, counts);
}

// ---- f1: colour check of many matches of one resident frame (HighLevelLinemod.cpp:113-135,159-161,424-434) ------
static int ensure_hulls(lm_detector* d) {
    if (!d->hulls_dirty) return LM_OK;
    HIP_TRY(hipDeviceSynchronize());
    hipFree(d->d_hull_class_base); hipFree(d->d_hull_off); hipFree(d->d_hull_xy);
    d->d_hull_class_base = d->d_hull_off = nullptr; d->d_hull_xy = nullptr;
    lmh::HullTable ht;
    lmh::build_hull_table(d->bank, d->cfg.num_modalities, ht);
    for (size_t t = 0; t + 1 < ht.hull_off.size(); ++t)
        if (ht.hull_off[t + 1] - ht.hull_off[t] > LM_HULL_MAX) return fail(LM_ERR_INVALID, "template hull with more than 128 vertices");
    int rc;
    if ((rc = upload_vec(&d->d_hull_class_base, ht.class_base))) return rc;
    if ((rc = upload_vec(&d->d_hull_off, ht.hull_off))) return rc;
    if ((rc = upload_vec(&d->d_hull_xy, ht.hull_xy))) return rc;
    if (!d->d_hsv_div) {
        // cv::cvtColor's 8-bit RGB2HSV tables: sdiv_table[i] = round((255 << 12) / i), hdiv_table180[i] = round((180 << 12) / (6 i))
        std::vector<int> tab(512, 0);
        for (int i = 1; i < 256; ++i) {
            tab[(size_t)i] = (int)std::lrint((255 << 12) / (1.0 * i));
            tab[256 + (size_t)i] = (int)std::lrint((180 << 12) / (6.0 * i));
        }
        if ((rc = upload_vec(&d->d_hsv_div, tab))) return rc;
    }
    d->hulls_dirty = false;
    return LM_OK;
}

// The colour check's own stream and buffers (r05): nothing of it touches a lane, so it runs while other lanes match other slots.
static int ensure_colour_check(lm_detector* d, size_t n) {
    if (!d->cc_stream) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
        // highest priority: a handful of short launches that the host waits for must not queue behind a lane's long kernels
        if (hipStreamCreateWithPriority(&d->cc_stream, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipStreamCreateWithFlags(&d->cc_stream, hipStreamNonBlocking));
        }
    }
    if (!d->cc_done) HIP_TRY(hipEventCreateWithFlags(&d->cc_done, hipEventDisableTiming));
    if (!d->dc_done) HIP_TRY(hipEventCreateWithFlags(&d->dc_done, hipEventDisableTiming));
    if (n > d->cc_cap) {
        const size_t cap = std::max<size_t>(align_up(n, 4096), 16384);
        const size_t bytes = cap * (sizeof(lm_match_t) + sizeof(int) + 2 * sizeof(long long));
        HIP_TRY(hipStreamSynchronize(d->cc_stream));
        u8* dev = nullptr; u8* host = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), bytes));
        if (hipHostMalloc(reinterpret_cast<void**>(&host), bytes) != hipSuccess) { (void)hipFree(dev); return fail(LM_ERR_HIP, "hipHostMalloc of the colour check's buffers failed"); }
        (void)hipFree(d->cc_dev); if (d->cc_host) (void)hipHostFree(d->cc_host);
        d->cc_dev = dev; d->cc_host = host; d->cc_cap = cap;
    }
    return LM_OK;
}

// slot_of: per match the slot its frame is resident in, or nullptr = all in `one_slot`.
static int colour_check_enqueue(lm_detector* d, const int32_t* slot_of, int one_slot, const double lower_hsv[3], const double upper_hsv[3],
                                const lm_match_t* matches, size_t n) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (d->cc_inflight) return fail(LM_ERR_INVALID, "a colour check is in flight: call lm_color_check_end first");
    if (!lower_hsv || !upper_hsv || (n && !matches)) return fail(LM_ERR_INVALID, "null argument");
    const int S = (int)d->slots.size();
    int s_lo = S, s_hi = -1;
    std::vector<char> used((size_t)S, 0);
    if (!slot_of) {
        if ((rc = check_slots(d, one_slot, 1))) return rc;
        used[(size_t)one_slot] = 1; s_lo = s_hi = one_slot;
    } else {
        for (size_t i = 0; i < n; ++i) {
            if (slot_of[i] < 0 || slot_of[i] >= S) return fail(LM_ERR_INVALID, "slot out of range");
            used[(size_t)slot_of[i]] = 1; s_lo = std::min(s_lo, slot_of[i]); s_hi = std::max(s_hi, slot_of[i]);
        }
    }
    if (n == 0) return LM_OK;
    for (int sl = s_lo; sl <= s_hi; ++sl) {
        if (!used[(size_t)sl]) continue;
        if (!d->slots[(size_t)sl].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
        for (const lm_detector::Lane& ln : d->lanes)
            if (ln.busy && sl >= ln.first && sl < ln.first + ln.n) return fail(LM_ERR_INVALID, "slot belongs to a match in flight: call lm_match_end first");
    }
    if (d->hulls_dirty && any_lane_busy(d)) return fail(LM_ERR_INVALID, "the bank changed while a lane has a match in flight: call lm_match_end first");
    if ((rc = ensure_hulls(d))) return rc;
    const int nc = (int)d->bank.classes.size();
    for (size_t i = 0; i < n; ++i) {
        const lm_match_t& m = matches[i];
        if (m.class_idx < 0 || m.class_idx >= nc || m.template_id < 0 || m.template_id >= (int)d->bank.classes[(size_t)m.class_idx].pyramids.size())
            return fail(LM_ERR_INVALID, "match " + std::to_string(i) + " names a template the bank does not hold");
    }
    if ((rc = ensure_colour_check(d, n))) return rc;
    hipStream_t st = d->cc_stream;
    // the frames' uploads (copy streams) must have landed before the mask kernel reads them
    for (int sl = s_lo; sl <= s_hi; ++sl) {
        const Slot& s = d->slots[(size_t)sl];
        if (used[(size_t)sl] && s.up_seq > d->up_seq_done[s.up_stream]) HIP_TRY(hipStreamWaitEvent(st, s.ev_up, 0));
    }
    const size_t off_slot = d->cc_cap * sizeof(lm_match_t), off_out = off_slot + d->cc_cap * sizeof(int);
    std::memcpy(d->cc_host, matches, n * sizeof(lm_match_t));
    if (slot_of) { int* hs = reinterpret_cast<int*>(d->cc_host + off_slot); for (size_t i = 0; i < n; ++i) hs[i] = slot_of[i] - s_lo; }
    HIP_TRY(hipMemcpyAsync(d->cc_dev, d->cc_host, n * sizeof(lm_match_t), hipMemcpyHostToDevice, st));
    if (slot_of) HIP_TRY(hipMemcpyAsync(d->cc_dev + off_slot, d->cc_host + off_slot, n * sizeof(int), hipMemcpyHostToDevice, st));
    LmHsvRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = (int)std::lrint(lower_hsv[k]); rg.hi[k] = (int)std::lrint(upper_hsv[k]); }
    // ONE mask launch for the slots [s_lo, s_hi] (a slot in between that the list does not name costs a mask nobody reads) -- unless
    // every named slot's mask was prepared for this very range beside its match (lm_color_mask_prepare)
    u32* mask = reinterpret_cast<u32*>(d->frame_arena + (size_t)s_lo * d->frame_stride + d->off_cmask);
    bool prepared = true;
    for (int sl = s_lo; sl <= s_hi && prepared; ++sl) {
        const Slot& s = d->slots[(size_t)sl];
        if (!used[(size_t)sl]) continue;
        prepared = s.mask_ready;
        for (int k = 0; k < 3 && prepared; ++k) prepared = s.mask_lo[k] == rg.lo[k] && s.mask_hi[k] == rg.hi[k];
    }
    if (!prepared) {
        lmk_hsv_mask(st, d->bgr(s_lo, 0), d->cfg.width, d->cfg.height, rg, d->d_hsv_div, mask, d->cmask_wpr, d->frame_stride, d->frame_stride, s_hi - s_lo + 1);
        for (int sl = s_lo; sl <= s_hi; ++sl) d->slots[(size_t)sl].mask_ready = false;      // (overwritten for this call's range; not recorded as prepared)
    } else {
        // the masks were written on a lane's stream (lm_color_mask_prepare): the hull kernel waits for that launch, whether or not the lane's match was
        // collected in between
        bool waited[LM_NLANES] = {};
        for (int sl = s_lo; sl <= s_hi; ++sl) {
            const Slot& sm = d->slots[(size_t)sl];
            if (!used[(size_t)sl] || sm.mask_lane < 0 || sm.mask_lane >= LM_NLANES || waited[sm.mask_lane] || !d->mask_done[sm.mask_lane]) continue;
            HIP_TRY(hipStreamWaitEvent(st, d->mask_done[sm.mask_lane], 0));
            waited[sm.mask_lane] = true;
        }
    }
    LmHullArgs a;
    a.matches = reinterpret_cast<const LmOutMatch*>(d->cc_dev); a.n = (u32)n;
    a.class_base = d->d_hull_class_base; a.hull_off = d->d_hull_off; a.hull_xy = d->d_hull_xy;
    a.mask = mask; a.wpr = d->cmask_wpr; a.w = d->cfg.width; a.h = d->cfg.height;
    a.match_slot = slot_of ? reinterpret_cast<const int*>(d->cc_dev + off_slot) : nullptr;
    a.mask_slot_words = d->frame_stride / 4;
    a.out = reinterpret_cast<long long*>(d->cc_dev + off_out);
    if (!lmk_hull_counts(st, a)) {
        (void)hipStreamSynchronize(st);
        return fail(LM_ERR_INVALID, "frame too tall for the GPU colour check (more than 4992 rows): use the host colour check");
    }
    HIP_TRY(hipMemcpyAsync(d->cc_host + off_out, a.out, n * 2 * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(d->cc_done, st));
    d->cc_pending = n; d->cc_inflight = true; d->cc_lo = s_lo; d->cc_hi = s_hi;
    return LM_OK;
}

static int colour_check_finish(lm_detector* d, int64_t* in_hull, int64_t* in_both) {
    if (!d || !d->cc_inflight) return fail(LM_ERR_INVALID, "no colour check in flight");
    const size_t n = d->cc_pending;
    d->cc_inflight = false; d->cc_pending = 0;
    if (n && (!in_hull || !in_both)) { (void)hipStreamSynchronize(d->cc_stream); return fail(LM_ERR_INVALID, "null argument"); }
    HIP_TRY(hipSetDevice(d->cfg.device));
    HIP_TRY(hipEventSynchronize(d->cc_done));          // (the event behind this check's last copy: depth counts enqueued behind it are not waited for, r06)
    HIP_TRY(hipGetLastError());
    const size_t off_out = d->cc_cap * sizeof(lm_match_t) + d->cc_cap * sizeof(int);
    const long long* out = reinterpret_cast<const long long*>(d->cc_host + off_out);
    for (size_t i = 0; i < n; ++i) { in_hull[i] = out[2 * i]; in_both[i] = out[2 * i + 1]; }
    return LM_OK;
}

static int colour_check(lm_detector* d, const int32_t* slot_of, int one_slot, const double lower_hsv[3], const double upper_hsv[3],
                        const lm_match_t* matches, size_t n, int64_t* in_hull, int64_t* in_both) {
    if (n && (!in_hull || !in_both)) return fail(LM_ERR_INVALID, "null argument");
    int rc;
    if ((rc = colour_check_enqueue(d, slot_of, one_slot, lower_hsv, upper_hsv, matches, n))) return rc;
    if (!d->cc_inflight) return LM_OK;       // n == 0
    return colour_check_finish(d, in_hull, in_both);
}

// The colour masks of slots [first_slot, first_slot + n_slots) for one HSV range, enqueued on `lane`'s stream AHEAD of the match that
// the caller begins on that lane next (lm_match_begin*): when the lane has been collected the masks are there, and a colour check of
// those slots for the same range skips its mask launch -- only the hull launch is left between lm_match_end and the counts.
int lm_color_mask_prepare(lm_detector* d, int lane, int first_slot, int n_slots, const double lower_hsv[3], const double upper_hsv[3]) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (lane < 0 || lane >= LM_NLANES) return fail(LM_ERR_INVALID, "lane out of range (0 .. 3)");
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if (!lower_hsv || !upper_hsv || n_slots <= 0) return fail(LM_ERR_INVALID, "bad argument");
    if (d->lanes[lane].busy) return fail(LM_ERR_INVALID, "lane is busy: prepare the masks before lm_match_begin");
    for (const lm_detector::Lane& ln : d->lanes)
        if (ln.busy && first_slot < ln.first + ln.n && ln.first < first_slot + n_slots) return fail(LM_ERR_INVALID, "slot belongs to a match in flight");
    for (int i = 0; i < n_slots; ++i)
        if (!d->slots[first_slot + i].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot " + std::to_string(first_slot + i));
    if (d->hulls_dirty && any_lane_busy(d)) return fail(LM_ERR_INVALID, "the bank changed while a lane has a match in flight: call lm_match_end first");
    if ((rc = ensure_hulls(d))) return rc;                 // (also uploads the HSV division tables)
    if ((rc = ensure_lane(d, lane))) return rc;
    LmHsvRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = (int)std::lrint(lower_hsv[k]); rg.hi[k] = (int)std::lrint(upper_hsv[k]); }
    activate_lane(d, lane);
    rc = enqueue_upload_wait(d, first_slot, n_slots);
    if (!rc) {
        u32* mask = reinterpret_cast<u32*>(d->frame_arena + (size_t)first_slot * d->frame_stride + d->off_cmask);
        lmk_hsv_mask(d->stream, d->bgr(first_slot, 0), d->cfg.width, d->cfg.height, rg, d->d_hsv_div, mask, d->cmask_wpr, d->frame_stride, d->frame_stride, n_slots);
        if (!d->mask_done[lane]) HIP_TRY(hipEventCreateWithFlags(&d->mask_done[lane], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(d->mask_done[lane], d->stream));
        for (int i = 0; i < n_slots; ++i) {
            Slot& s = d->slots[first_slot + i];
            s.mask_ready = true; s.mask_lane = lane;
            for (int k = 0; k < 3; ++k) { s.mask_lo[k] = rg.lo[k]; s.mask_hi[k] = rg.hi[k]; }
        }
    }
    activate_lane(d, 0);
    return rc;
}

int lm_color_check_counts(lm_detector* d, int slot, const double lower_hsv[3], const double upper_hsv[3],
                          const lm_match_t* matches, size_t n, int64_t* in_hull, int64_t* in_both) {
    return colour_check(d, nullptr, slot, lower_hsv, upper_hsv, matches, n, in_hull, in_both);
}

int lm_color_check_begin_slots(lm_detector* d, const int32_t* slot_of_match, const double lower_hsv[3], const double upper_hsv[3],
                               const lm_match_t* matches, size_t n) {
    if (n && !slot_of_match) return fail(LM_ERR_INVALID, "null argument");
    if (n == 0) { if (d) { if (d->cc_inflight) return fail(LM_ERR_INVALID, "a colour check is in flight: call lm_color_check_end first"); d->cc_inflight = true; d->cc_pending = 0; } return d ? LM_OK : fail(LM_ERR_INVALID, "null detector"); }
    return colour_check_enqueue(d, slot_of_match, 0, lower_hsv, upper_hsv, matches, n);
}

int lm_color_check_end(lm_detector* d, int64_t* in_hull, int64_t* in_both) {
    if (d && d->cc_inflight && d->cc_pending == 0) { d->cc_inflight = false; return LM_OK; }     // an empty list was begun: nothing was enqueued
    return colour_check_finish(d, in_hull, in_both);
}

int lm_color_check_counts_slots(lm_detector* d, const int32_t* slot_of_match, const double lower_hsv[3], const double upper_hsv[3],
                                const lm_match_t* matches, size_t n, int64_t* in_hull, int64_t* in_both) {
    if (n && !slot_of_match) return fail(LM_ERR_INVALID, "null argument");
    return colour_check(d, slot_of_match, 0, lower_hsv, upper_hsv, matches, n, in_hull, in_both);
}

// ---- r06: the depth check's counts for a batch of queries (include/linemod_hip.h lm_depth_counts_begin) -----------------------------------
int lm_depth_counts_begin(lm_detector* d, const lm_depth_query* q, size_t n) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (d->dc_inflight) return fail(LM_ERR_INVALID, "depth counts are in flight: call lm_depth_counts_end first");
    if (n && !q) return fail(LM_ERR_INVALID, "null argument");
    if (d->cfg.num_modalities < 2) return fail(LM_ERR_INVALID, "the detector keeps no depth frame on the device (no depth modality)");
    static_assert(sizeof(lm_depth_query) == sizeof(LmDepthQuery), "lm_depth_query layout");
    const int S = (int)d->slots.size(), W = d->cfg.width, H = d->cfg.height;
    std::vector<char> used((size_t)S, 0);
    for (size_t i = 0; i < n; ++i) {
        const lm_depth_query& e = q[i];
        if (e.slot < 0 || e.slot >= S) return fail(LM_ERR_INVALID, "slot out of range");
        if (e.x0 < 0 || e.y0 < 0 || e.x1 > W || e.y1 > H || e.x1 < e.x0 || e.y1 < e.y0) return fail(LM_ERR_INVALID, "query " + std::to_string(i) + ": crop outside the frame");
        used[(size_t)e.slot] = 1;
    }
    for (int sl = 0; sl < S; ++sl) {
        if (!used[(size_t)sl]) continue;
        if (!d->slots[(size_t)sl].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    }
    d->dc_pending = 0; d->dc_inflight = true;
    if (n == 0) return LM_OK;
    if ((rc = ensure_colour_check(d, 0))) { d->dc_inflight = false; return rc; }      // (the stream)
    if (n > d->dc_cap) {
        const size_t cap = std::max<size_t>(align_up(n, 4096), 16384);
        const size_t bytes = cap * (sizeof(LmDepthQuery) + 2 * sizeof(u32));
        u8* dev = nullptr; u8* host = nullptr;
        if (hipStreamSynchronize(d->cc_stream) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&dev), bytes) != hipSuccess) { d->dc_inflight = false; (void)hipGetLastError(); return fail(LM_ERR_HIP, "allocation of the depth counts' buffers failed"); }
        if (hipHostMalloc(reinterpret_cast<void**>(&host), bytes) != hipSuccess) { (void)hipFree(dev); d->dc_inflight = false; (void)hipGetLastError(); return fail(LM_ERR_HIP, "hipHostMalloc of the depth counts' buffers failed"); }
        (void)hipFree(d->dc_dev); if (d->dc_host) (void)hipHostFree(d->dc_host);
        d->dc_dev = dev; d->dc_host = host; d->dc_cap = cap;
    }
    hipStream_t st = d->cc_stream;
    auto bail = [&](hipError_t e) { d->dc_inflight = false; (void)hipStreamSynchronize(st); return fail(LM_ERR_HIP, hipGetErrorString(e)); };
    // the frames' uploads (copy streams) must have landed before the kernel reads them
    for (int sl = 0; sl < S; ++sl) {
        const Slot& s = d->slots[(size_t)sl];
        if (used[(size_t)sl] && s.up_seq > d->up_seq_done[s.up_stream]) { const hipError_t e = hipStreamWaitEvent(st, s.ev_up, 0); if (e != hipSuccess) return bail(e); }
    }
    const size_t off_out = d->dc_cap * sizeof(LmDepthQuery);
    std::memcpy(d->dc_host, q, n * sizeof(LmDepthQuery));
    hipError_t e = hipMemcpyAsync(d->dc_dev, d->dc_host, n * sizeof(LmDepthQuery), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return bail(e);
    LmDepthArgs a;
    a.depth = d->depth(0); a.slot_stride = d->frame_stride; a.w = W; a.h = H;
    a.q = reinterpret_cast<const LmDepthQuery*>(d->dc_dev); a.n = (u32)n;
    a.out = reinterpret_cast<u32*>(d->dc_dev + off_out);
    lmk_depth_counts(st, a);
    e = hipMemcpyAsync(d->dc_host + off_out, a.out, n * 2 * sizeof(u32), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return bail(e);
    e = hipEventRecord(d->dc_done, st);
    if (e != hipSuccess) return bail(e);
    d->dc_pending = n;
    d->dc_lo = S; d->dc_hi = -1;
    for (int sl = 0; sl < S; ++sl) if (used[(size_t)sl]) { d->dc_lo = std::min(d->dc_lo, sl); d->dc_hi = std::max(d->dc_hi, sl); }
    return LM_OK;
}

int lm_depth_counts_end(lm_detector* d, uint32_t* below, uint32_t* inside) {
    if (!d || !d->dc_inflight) return fail(LM_ERR_INVALID, "no depth counts in flight");
    const size_t n = d->dc_pending;
    d->dc_inflight = false; d->dc_pending = 0;
    if (n == 0) return LM_OK;
    if (!below || !inside) { (void)hipStreamSynchronize(d->cc_stream); return fail(LM_ERR_INVALID, "null argument"); }
    HIP_TRY(hipSetDevice(d->cfg.device));
    HIP_TRY(hipEventSynchronize(d->dc_done));
    HIP_TRY(hipGetLastError());
    const u32* out = reinterpret_cast<const u32*>(d->dc_host + d->dc_cap * sizeof(LmDepthQuery));
    for (size_t i = 0; i < n; ++i) { below[i] = out[2 * i]; inside[i] = out[2 * i + 1]; }
    return LM_OK;
}

// ---- multi-GPU exchange: RCCL all-gather of the per-shard lists ---------------------------------
static void free_gather(lm_detector* d) {
    for (auto& g : d->gather) {
        hipFree(g.d_cnt); hipFree(g.d_rec); hipFree(g.d_all_cnt); hipFree(g.d_all_rec);
        if (g.h_all_cnt) hipHostFree(g.h_all_cnt);
        if (g.h_all_rec) hipHostFree(g.h_all_rec);
        g = lm_detector::Gather();
    }
    hipFree(d->d_red); d->d_red = nullptr;
}

int lm_comm_init(lm_detector* d, int rank, int world, const char* addr, int port, int recs_per_frame_cap) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (d->comm[0]) return fail(LM_ERR_INVALID, "communicator already initialised");
    if (world < 1 || rank < 0 || rank >= world) return fail(LM_ERR_INVALID, "bad rank / world size");
    if (recs_per_frame_cap <= 0) recs_per_frame_cap = 256;
    if (recs_per_frame_cap > LM_SORT_CAP) recs_per_frame_cap = LM_SORT_CAP;
    // Buffers first, communicators last: a failure on the way leaves NOTHING behind (no communicator without its
    // buffers -- lm_match_begin_gathered keys on comm[0] -- and the call can simply be repeated).
    const size_t S = d->slots.size(), R = (size_t)world, cap = (size_t)recs_per_frame_cap * S;
    auto alloc_all = [&]() -> int {
        for (auto& g : d->gather) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_cnt), (S + 1) * sizeof(int)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_rec), cap * sizeof(LmOutMatch)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_all_cnt), R * (S + 1) * sizeof(int)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_all_rec), R * cap * sizeof(LmOutMatch)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g.h_all_cnt), R * (S + 1) * sizeof(int)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g.h_all_rec), R * cap * sizeof(LmOutMatch)));
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_red), 64 * sizeof(double)));   // [0, 32) send | [32, 64) receive
        return LM_OK;
    };
    if ((rc = alloc_all())) { const std::string msg = g_err; free_gather(d); return fail(rc, msg); }
    // ONE rendezvous for the ids of all lanes' communicators (rank 0 draws them), then the ncclCommInitRank calls in
    // lane order on every rank.
    LmComm* cs[LM_NLANES] = {};
    unsigned char ids[LM_NLANES][LM_NCCL_ID_BYTES] = {};
    std::string err;
    bool ok = true;
    for (int l = 0; l < LM_NLANES && ok; ++l) {
        cs[l] = new LmComm();
        ok = cs[l]->load(err) && (rank != 0 || cs[l]->unique_id(ids[l], err));
    }
    if (ok) ok = lm_tcp_broadcast(rank, world, addr ? addr : "127.0.0.1", port, 120, ids, sizeof(ids), err);
    for (int l = 0; l < LM_NLANES && ok; ++l) ok = cs[l]->init_rank(rank, world, ids[l], err);
    if (!ok) {
        for (auto& c : cs) delete c;
        free_gather(d);
        return fail(LM_ERR_HIP, err);
    }
    for (int l = 0; l < LM_NLANES; ++l) d->comm[l] = cs[l];
    d->comm_recs_per_frame = recs_per_frame_cap;
    return LM_OK;
}

int lm_rendezvous_broadcast(int rank, int world, const char* addr, int port, void* buf, size_t n, int timeout_s) {
    if (!buf || world < 1 || rank < 0 || rank >= world) return fail(LM_ERR_INVALID, "bad argument");
    std::string err;
    if (!lm_tcp_broadcast(rank, world, addr ? addr : "127.0.0.1", port, timeout_s > 0 ? timeout_s : 60, buf, n, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}

int lm_comm_destroy(lm_detector* d) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (d->comm[0]) {
        hipSetDevice(d->cfg.device);
        hipDeviceSynchronize();
        for (auto& c : d->comm) { delete c; c = nullptr; }
        free_gather(d);
    }
    return LM_OK;
}

int lm_comm_info(const lm_detector* d, int* rank, int* world) {
    if (!d || !d->comm[0]) return fail(LM_ERR_INVALID, "no communicator");
    if (rank) *rank = d->comm[0]->rank;
    if (world) *world = d->comm[0]->world;
    return LM_OK;
}

// element-wise maximum over the ranks of n <= 32 doubles; returns when every rank's value has arrived
int lm_comm_max(lm_detector* d, double* v, int n) {
    if (!d || !d->comm[0] || !v || n < 1 || n > 32) return fail(LM_ERR_INVALID, "bad argument");
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    HIP_TRY(hipSetDevice(d->cfg.device));
    std::string err;
    HIP_TRY(hipMemcpyAsync(d->d_red, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, d->stream));
    if (!d->comm[0]->all_reduce_max_f64(d->d_red, d->d_red + 32, (size_t)n, d->stream, err)) return fail(LM_ERR_HIP, err);
    HIP_TRY(hipMemcpyAsync(v, d->d_red + 32, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LM_OK;
}

// every rank's device is idle and every rank has reached this call
int lm_comm_barrier(lm_detector* d) {
    if (!d || !d->comm[0]) return fail(LM_ERR_INVALID, "no communicator");
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    HIP_TRY(hipSetDevice(d->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    double one = 1.0;
    int rc = lm_comm_max(d, &one, 1);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return LM_OK;
}

// behind k_sort_unique on the active lane's stream: pack the lane's sorted lists, all-gather their lengths and the
// packed records (fixed capacity per rank, so no host round trip sits between the two collectives), copy both to
// pinned host memory.
static int enqueue_gather(lm_detector* d, int lane, int first, int n) {
    lm_detector::Gather& g = d->gather[lane];
    LmComm* comm = d->comm[lane];
    const size_t R = (size_t)comm->world;
    g.cap_lane = (u32)d->comm_recs_per_frame * (u32)n;
    LmPackArgs pa;
    pa.hdr = reinterpret_cast<const LmDevHeader*>(d->aux(first, d->off_hdr));
    pa.out = reinterpret_cast<const LmOutMatch*>(d->aux(first, d->off_out));
    pa.aux_slot_stride = d->aux_stride;
    pa.nslots = n; pa.cap_total = g.cap_lane; pa.cnt = g.d_cnt; pa.rec = g.d_rec;
    lmk_pack_lists(d->stream, pa);
    std::string err;
    const size_t cb = (size_t)(n + 1) * sizeof(int), rb = (size_t)g.cap_lane * sizeof(LmOutMatch);
    if (!comm->all_gather(g.d_cnt, g.d_all_cnt, cb, d->stream, err)) return fail(LM_ERR_HIP, err);
    if (!comm->all_gather(g.d_rec, g.d_all_rec, rb, d->stream, err)) return fail(LM_ERR_HIP, err);
    // only the lengths come to the host here: lm_match_end_gathered then fetches, per rank, exactly the records of the frames
    // THIS rank merges (a contiguous piece of every rank's packed run) -- with R ranks 1 / R of the real records instead of
    // R x the gather capacity over the PCIe link every lane-step
    HIP_TRY(hipMemcpyAsync(g.h_all_cnt, g.d_all_cnt, R * cb, hipMemcpyDeviceToHost, d->stream));
    if (d->profiling) HIP_TRY(hipEventRecord(d->ev[5], d->stream));   // exchange span = ev[4] (behind the sort) -> ev[5]
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

// The sized second exchange of the gathered path: some rank's lists did not fit the fixed-capacity gather, or a frame was
// left to the host sort (> LM_SORT_CAP matches).  The single-GPU path returns such lists (collect_slot), so the sharded
// one must too (the reference consumes ALL matches, HighLevelLinemod.cpp:206-253).  Every rank: collect its own lists
// on the host (host sort where needed), all-gather the exact per-frame counts, all-gather the packed records in buffers
// sized to the largest rank, merge the owned frames.  Synchronous, on the lane's own communicator and stream; this is
// the slow path of low thresholds, not of the benchmark.
static int gather_fallback(lm_detector* d, int lane, int first, int n, int f0, int f1, lm_match_t* out, size_t cap,
                           int32_t* counts, size_t* n_out) {
    lm_detector::Gather& g = d->gather[lane];
    LmComm* comm = d->comm[lane];
    const size_t R = (size_t)comm->world;
    hipStream_t st = d->lanes[lane].stream ? d->lanes[lane].stream : d->stream;
    if (lane == 0) st = d->stream;
    // 1. this rank's lists, exact
    std::vector<lm_match_t> mine;
    std::vector<int32_t> my_cnt((size_t)n + 1, 0);
    int local_rc = LM_OK;
    std::string local_msg;
    for (int i = 0; i < n; ++i) {
        size_t k = 0;
        int rc = collect_slot(d, first + i, nullptr, 0, &k);           // length (runs the host sort for host-sorted frames)
        if (!rc) {
            const size_t at = mine.size();
            mine.resize(at + k);
            rc = collect_slot(d, first + i, mine.data() + at, k, &k);
        }
        if (rc && !local_rc) { local_rc = rc; local_msg = g_err; }
        my_cnt[(size_t)i] = (int32_t)k;
    }
    my_cnt[(size_t)n] = local_rc ? 4 : 0;                               // status travels with the counts: all ranks agree
    // 2. exact counts of every rank
    std::string err;
    const size_t cb = (size_t)(n + 1) * sizeof(int);
    HIP_TRY(hipMemcpyAsync(g.d_cnt, my_cnt.data(), cb, hipMemcpyHostToDevice, st));
    if (!comm->all_gather(g.d_cnt, g.d_all_cnt, cb, st, err)) return fail(LM_ERR_HIP, err);
    HIP_TRY(hipMemcpyAsync(g.h_all_cnt, g.d_all_cnt, R * cb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int32_t> cnt(R * (size_t)n);
    uint64_t max_total64 = 1;
    {
        int st_all = 0, bad = -1, pf0 = 0, pf1 = 0;
        const int prc = lm_gather_plan(g.h_all_cnt, (int)R, n, comm->rank, &st_all, &bad, &pf0, &pf1, cnt.data(), nullptr, nullptr);
        if (prc) return prc;
        if (st_all) {
            if (local_rc) return fail(local_rc, local_msg);
            return fail(LM_ERR_OVERFLOW, "rank " + std::to_string(bad) + " could not deliver its match lists");
        }
        (void)lm_gather_max_total(cnt.data(), (int)R, n, &max_total64);
    }
    const size_t max_total = (size_t)max_total64;
    // 3. records, in buffers sized to the largest rank
    // A rank-local failure here (an allocation on a nearly full device, a failed copy) must not leave the other ranks blocked in
    // the sized all-gather: every rank reports whether it is ready, the flags are all-gathered, and all ranks go on
    // or give up TOGETHER.
    LmOutMatch *d_send = nullptr, *d_recv = nullptr;
    std::vector<lm_match_t> all;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_send), max_total * sizeof(LmOutMatch));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_recv), R * max_total * sizeof(LmOutMatch));
    bool host_ok = true;
    try { all.resize(R * max_total); } catch (const std::bad_alloc&) { host_ok = false; }
    if (he == hipSuccess && host_ok && !mine.empty())
        he = hipMemcpyAsync(d_send, mine.data(), mine.size() * sizeof(lm_match_t), hipMemcpyHostToDevice, st);
    int32_t ready = (he == hipSuccess && host_ok) ? 0 : 1;
    std::vector<int32_t> ready_all(R, 0);
    bool ok = true;
    hipError_t fe = hipMemcpyAsync(g.d_cnt, &ready, sizeof(ready), hipMemcpyHostToDevice, st);
    if (fe == hipSuccess) ok = comm->all_gather(g.d_cnt, g.d_all_cnt, sizeof(int32_t), st, err);
    if (fe == hipSuccess && ok) fe = hipMemcpyAsync(ready_all.data(), g.d_all_cnt, R * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (fe == hipSuccess && ok) fe = hipStreamSynchronize(st);
    int not_ready = -1;
    for (size_t r = 0; r < R; ++r) if (ready_all[r] && not_ready < 0) not_ready = (int)r;
    if (!ok || fe != hipSuccess || not_ready >= 0) {
        (void)hipFree(d_send); (void)hipFree(d_recv);
        if (!ok) return fail(LM_ERR_HIP, err);
        if (fe != hipSuccess) return fail(LM_ERR_HIP, std::string("sized exchange (readiness): ") + hipGetErrorString(fe));
        if (he != hipSuccess) return fail(LM_ERR_HIP, std::string("sized exchange buffers: ") + hipGetErrorString(he));
        if (!host_ok) return fail(LM_ERR_HIP, "sized exchange: host buffer allocation failed");
        return fail(LM_ERR_HIP, "rank " + std::to_string(not_ready) + " could not set up the sized exchange; all ranks gave up together");
    }
    ok = comm->all_gather(d_send, d_recv, max_total * sizeof(LmOutMatch), st, err);
    if (ok) he = hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(lm_match_t), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess && ok) he = hipStreamSynchronize(st);
    (void)hipFree(d_send); (void)hipFree(d_recv);
    if (!ok) return fail(LM_ERR_HIP, err);
    if (he != hipSuccess) return fail(LM_ERR_HIP, std::string("sized exchange: ") + hipGetErrorString(he));
    // 4. merge the frames this rank owns
    return lm_merge_frames(all.data(), max_total, cnt.data(), (int)R, n, f0, f1, out, cap, counts, n_out);
}

int lm_match_end_gathered(lm_detector* d, int lane, lm_match_t* out, size_t cap, int32_t* counts, int* first_frame,
                          int* n_frames, size_t* n_out) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (lane < 0 || lane >= LM_NLANES) return fail(LM_ERR_INVALID, "lane out of range (0 .. 3)");
    lm_detector::Lane& ln = d->lanes[lane];
    lm_detector::Gather& g = d->gather[lane];
    if (!ln.busy || !g.active) return fail(LM_ERR_INVALID, "lane has no gathered match in flight");
    HIP_TRY(hipSetDevice(d->cfg.device));
    activate_lane(d, lane);
    const int wrc = wait_lane_done(d, ln);
    if (!wrc && ln.timed) account_profile(d, ln.n, ln.classes, true);
    activate_lane(d, 0);
    ln.busy = false; g.active = false;
    if (wrc) return wrc;
    const int n = ln.n, R = d->comm[0]->world, rank = d->comm[0]->rank;
    const int f0 = (int)((long long)n * rank / R), f1 = (int)((long long)n * (rank + 1) / R);
    if (first_frame) *first_frame = f0;
    if (n_frames) *n_frames = f1 - f0;
    for (int i = 0; i < n; ++i) note_sort_length(d, d->host_block(ln.first + i)->hdr.match_count);
    // The status words every rank gathered are identical on all ranks, so all ranks take the same branch below (the
    // fallback holds collectives): lm_gather_plan (lm_host.cpp, host-only and unit-tested at R = 2, 3, 8) reads them.
    int status = 0, bad_rank = -1, pf0 = 0, pf1 = 0;
    std::vector<int32_t> cnt((size_t)R * n);
    std::vector<uint64_t> piece_start((size_t)R), piece_len((size_t)R);
    {
        const int prc = lm_gather_plan(g.h_all_cnt, R, n, rank, &status, &bad_rank, &pf0, &pf1, cnt.data(), piece_start.data(), piece_len.data());
        if (prc) return prc;
    }
    if (status & 4) {
        if (bad_rank == rank)      // this shard's own capacity overflow: same message as the ungathered path
            for (int i = 0; i < n; ++i) {
                const LmHeader h = d->host_block(ln.first + i)->hdr;
                if (h.cand_count > d->max_cand || h.match_count > d->max_match) { size_t dummy; return collect_slot(d, ln.first + i, nullptr, 0, &dummy); }
            }
        return fail(LM_ERR_OVERFLOW, "rank " + std::to_string(bad_rank) + " overflowed its candidate / match capacity (raise lm_config.max_candidates / max_matches)");
    }
    if (status == 0) {
        // the owned frames' records of every rank: frames are packed in order, so they are ONE contiguous piece per rank
        activate_lane(d, lane);
        int crc = LM_OK;
        for (int r = 0; r < R && !crc; ++r) {
            const size_t start = (size_t)piece_start[(size_t)r], len = (size_t)piece_len[(size_t)r];
            if (start + len > (size_t)g.cap_lane) { crc = fail(LM_ERR_INVALID, "gathered counts exceed the gather capacity"); break; }
            if (!len) continue;
            const size_t at = (size_t)r * g.cap_lane + start;
            if (hipMemcpyAsync(g.h_all_rec + at, g.d_all_rec + at, len * sizeof(LmOutMatch), hipMemcpyDeviceToHost, d->stream) != hipSuccess)
                crc = fail(LM_ERR_HIP, "D2H of the gathered records failed");
        }
        if (!crc) crc = wait_stream(d);
        activate_lane(d, 0);
        if (crc) return crc;
        return lm_merge_frames(reinterpret_cast<const lm_match_t*>(g.h_all_rec), g.cap_lane, cnt.data(), R, n, f0, f1, out, cap, counts, n_out);
    }
    d->prof_exch_fallbacks += 1;
    return gather_fallback(d, lane, ln.first, n, f0, f1, out, cap, counts, n_out);
}

// 8e bookkeeping of the gathered path, host-only (lm_match_end_gathered and its sized fallback call it; tests/test_dist.py drives
// it at R = 2, 3, 8 on synthetic gathered buffers).  all_cnt: what the all-gather of the lengths delivers, R runs of n + 1 ints
// -- cnt[i] = records of frame i in that rank's packed run, [n] = the rank's status word (bit 0 lists did not fit the fixed
// capacity, bit 1 a frame was left to the host sort, bit 2 the shard overflowed its own capacity).  Out: the OR of the status
// words, the first rank with bit 2 set (or -1), the frames [f0, f1) rank `rank` merges, counts as [R][n], and per rank the
// piece of its packed run that holds exactly the owned frames (start, len in records; frames are packed in order).
int lm_gather_plan(const int32_t* all_cnt, int n_ranks, int n_frames, int rank, int* status, int* bad_rank, int* f0, int* f1,
                   int32_t* counts, uint64_t* piece_start, uint64_t* piece_len) {
    if (!all_cnt || n_ranks < 1 || n_frames < 0 || rank < 0 || rank >= n_ranks) return fail(LM_ERR_INVALID, "bad argument");
    const int n = n_frames, R = n_ranks;
    const int lo = (int)((long long)n * rank / R), hi = (int)((long long)n * (rank + 1) / R);
    int st = 0, bad = -1;
    for (int r = 0; r < R; ++r) {
        const int32_t* c = all_cnt + (size_t)r * (size_t)(n + 1);
        st |= c[n];
        if ((c[n] & 4) && bad < 0) bad = r;
        uint64_t start = 0, len = 0;
        for (int i = 0; i < n; ++i) {
            if (c[i] < 0) return fail(LM_ERR_INVALID, "negative count in the gathered lengths");
            if (counts) counts[(size_t)r * n + i] = c[i];
            if (i < lo) start += (uint64_t)c[i];
            else if (i < hi) len += (uint64_t)c[i];
        }
        if (piece_start) piece_start[r] = start;
        if (piece_len) piece_len[r] = len;
    }
    if (status) *status = st;
    if (bad_rank) *bad_rank = bad;
    if (f0) *f0 = lo;
    if (f1) *f1 = hi;
    return LM_OK;
}

// Records of the largest rank's packed run (at least 1): the per-rank buffer size of the sized second exchange.
int lm_gather_max_total(const int32_t* counts, int n_ranks, int n_frames, uint64_t* max_total) {
    if (!counts || !max_total || n_ranks < 1 || n_frames < 0) return fail(LM_ERR_INVALID, "bad argument");
    uint64_t best = 1;
    for (int r = 0; r < n_ranks; ++r) {
        uint64_t tot = 0;
        for (int i = 0; i < n_frames; ++i) {
            if (counts[(size_t)r * n_frames + i] < 0) return fail(LM_ERR_INVALID, "negative count");
            tot += (uint64_t)counts[(size_t)r * n_frames + i];
        }
        best = std::max(best, tot);
    }
    *max_total = best;
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
    if (out && !all.empty() && cap) std::memcpy(out, all.data(), std::min(all.size(), cap) * sizeof(lm_match_t));
    if (all.size() > cap && out) return fail(LM_ERR_OVERFLOW, "output buffer too small");
    return LM_OK;
}

// Per-frame lists at a fixed stride -> one contiguous run (what travels in the shard all-gather).
int lm_pack_matches(const lm_match_t* recs, size_t stride, const int32_t* counts, int n_frames, lm_match_t* out,
                    size_t cap, size_t* n_out) {
    if (!recs || !counts || n_frames < 0) return fail(LM_ERR_INVALID, "null argument");
    size_t total = 0;
    for (int i = 0; i < n_frames; ++i) {
        if (counts[i] < 0 || (size_t)counts[i] > stride) return fail(LM_ERR_INVALID, "count exceeds stride");
        total += (size_t)counts[i];
    }
    if (n_out) *n_out = total;
    if (!out) return LM_OK;
    if (total > cap) return fail(LM_ERR_OVERFLOW, "output buffer too small");
    size_t pos = 0;
    for (int i = 0; i < n_frames; ++i) {
        std::memcpy(out + pos, recs + (size_t)i * stride, (size_t)counts[i] * sizeof(lm_match_t));
        pos += (size_t)counts[i];
    }
    return LM_OK;
}

// The merge step of a whole batch after the all-gather: rank r's packed run starts at packed + r * rank_stride
// and holds its frames back to back (counts[r * n_frames + i] records for frame i).  Frame i of the output is the
// R-way merge + adjacent-unique of the R sorted lists (pairwise std::merge), frames are spread over a few host
// threads; output packed the same way with out_counts[i].
int lm_merge_batch(const lm_match_t* packed, size_t rank_stride, const int32_t* counts, int n_ranks, int n_frames,
                   lm_match_t* out, size_t cap, int32_t* out_counts, size_t* n_out) {
    return lm_merge_frames(packed, rank_stride, counts, n_ranks, n_frames, 0, n_frames, out, cap, out_counts, n_out);
}

// Only the frames [frame_lo, frame_hi) of the batch: the ranks share the merge work by frame (rank r merges the
// frames it owns; every rank still holds the gathered lists of all frames).  out_counts has frame_hi - frame_lo entries.
int lm_merge_frames(const lm_match_t* packed, size_t rank_stride, const int32_t* counts, int n_ranks, int n_frames,
                    int frame_lo, int frame_hi, lm_match_t* out, size_t cap, int32_t* out_counts, size_t* n_out) {
    if (!packed || !counts || !out_counts || n_ranks <= 0 || n_frames < 0) return fail(LM_ERR_INVALID, "bad argument");
    if (frame_lo < 0 || frame_hi < frame_lo || frame_hi > n_frames) return fail(LM_ERR_INVALID, "bad frame range");
    const size_t R = (size_t)n_ranks, F = (size_t)n_frames;
    const size_t lo = (size_t)frame_lo, hi = (size_t)frame_hi, Fo = hi - lo;
    std::vector<size_t> start(R * F);      // start of (rank, frame) inside the rank's run
    std::vector<size_t> bound(Fo + 1, 0);  // upper bound of the merged output of the owned frames before unique
    for (size_t r = 0; r < R; ++r) {
        size_t pos = 0;
        for (size_t i = 0; i < F; ++i) {
            const int32_t c = counts[r * F + i];
            if (c < 0) return fail(LM_ERR_INVALID, "negative count");
            start[r * F + i] = pos;
            pos += (size_t)c;
            if (i >= lo && i < hi) bound[i - lo + 1] += (size_t)c;
        }
        if (pos > rank_stride) return fail(LM_ERR_INVALID, "counts exceed rank_stride");
    }
    for (size_t i = 0; i < Fo; ++i) bound[i + 1] += bound[i];
    std::vector<lm_match_t> tmp(bound[Fo]);
    auto work = [&](size_t a0, size_t a1) {
        std::vector<lm_match_t> a, b;
        for (size_t k = a0; k < a1; ++k) {
            const size_t i = lo + k;
            a.clear();
            for (size_t r = 0; r < R; ++r) {
                const lm_match_t* src = packed + r * rank_stride + start[r * F + i];
                const size_t c = (size_t)counts[r * F + i];
                b.resize(a.size() + c);
                std::merge(a.begin(), a.end(), src, src + c, b.begin(), lmh::match_less);
                a.swap(b);
            }
            a.erase(std::unique(a.begin(), a.end(), lmh::match_eq), a.end());
            std::copy(a.begin(), a.end(), tmp.begin() + (ptrdiff_t)bound[k]);
            out_counts[k] = (int32_t)a.size();
        }
    };
    const size_t nthreads = std::min<size_t>(bound[Fo] >= 50000 ? 4 : 1, std::max(1u, std::thread::hardware_concurrency()));
    if (nthreads <= 1) work(0, Fo);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthreads; ++t) th.emplace_back(work, Fo * t / nthreads, Fo * (t + 1) / nthreads);
        for (auto& x : th) x.join();
    }
    size_t total = 0;
    for (size_t i = 0; i < Fo; ++i) total += (size_t)out_counts[i];
    if (n_out) *n_out = total;
    if (out) {
        if (total > cap) return fail(LM_ERR_OVERFLOW, "output buffer too small");
        size_t pos = 0;
        for (size_t i = 0; i < Fo; ++i) {
            std::memcpy(out + pos, tmp.data() + bound[i], (size_t)out_counts[i] * sizeof(lm_match_t));
            pos += (size_t)out_counts[i];
        }
    }
    return LM_OK;
}

int lm_save_bank(const lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::save_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}
int lm_load_bank(lm_detector* d, const char* path) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::load_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    d->bank_dirty = true; d->hulls_dirty = true;
    return LM_OK;
}

int lm_save_yaml(const lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmy::save_templates_yaml(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}
int lm_load_yaml(lm_detector* d, const char* path) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmy::load_templates_yaml(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    d->bank_dirty = true; d->hulls_dirty = true;
    for (Slot& sl : d->slots) sl.prepared = false;   // the file's modality parameters (thresholds) replaced the detector's: a3-a10 results are stale
    if (d->cfg.num_modalities == 2 && d->normal_lut_substitute) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            std::fprintf(stderr, "liblinemod_hip: warning: %s holds DepthNormal templates, but the built-in NORMAL_LUT is a "
                                 "substitute for OpenCV's normal_lut.i (SURVEY.md A.4): a bank WRITTEN BY OpenCV will be scored "
                                 "against differently quantised normals.  Install the real table with lm_set_normal_lut, or "
                                 "regenerate the bank with this library.\n", path);
        }
    }
    return LM_OK;
}

// Top-level scalars / number lists of a cv::FileStorage YAML file (linemod_settings.yml, models/<name>.yml,
// benchmark/pose0.yml): the host glue reads its settings through these.
// Returns the status code (LM_ERR_IO: unreadable / unparsable file; LM_ERR_INVALID: no such key) and the node.
static int yaml_top(const char* path, const char* key, lmy::Node& root, const lmy::Node** out) {
    std::string text, err;
    *out = nullptr;
    if (!lmy::read_text_file(path, text, err)) return fail(LM_ERR_IO, err);
    if (!lmy::parse(text, root, err)) return fail(LM_ERR_IO, std::string(path) + ": " + err);
    const lmy::Node* n = root.get(key);
    if (!n) return fail(LM_ERR_INVALID, std::string("no key '") + key + "' in " + path);
    *out = n;
    return LM_OK;
}
int lm_yaml_numbers(const char* path, const char* key, double* out, size_t cap, size_t* n_out) {
    if (!path || !key) return fail(LM_ERR_INVALID, "null argument");
    lmy::Node root;
    const lmy::Node* n = nullptr;
    int rc;
    if ((rc = yaml_top(path, key, root, &n))) return rc;
    if (n->kind == lmy::Node::Map && n->get("data")) n = n->get("data");   // !!opencv-matrix
    std::vector<double> v;
    double d;
    if (n->kind == lmy::Node::Nums) v = n->nums;
    else if (n->number(&d)) v.push_back(d);
    else return fail(LM_ERR_INVALID, std::string("'") + key + "' is not numeric");
    if (n_out) *n_out = v.size();
    if (out) {
        if (cap < v.size()) return fail(LM_ERR_INVALID, "buffer too small");
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    }
    return LM_OK;
}
int lm_yaml_string(const char* path, const char* key, char* out, size_t cap) {
    if (!path || !key || !out || !cap) return fail(LM_ERR_INVALID, "null argument");
    lmy::Node root;
    const lmy::Node* n = nullptr;
    int rc;
    if ((rc = yaml_top(path, key, root, &n))) return rc;
    if (n->kind != lmy::Node::Scalar) return fail(LM_ERR_INVALID, std::string("'") + key + "' is not a scalar");
    if (n->scalar.size() + 1 > cap) return fail(LM_ERR_INVALID, "buffer too small");
    std::memcpy(out, n->scalar.c_str(), n->scalar.size() + 1);
    return LM_OK;
}

// ---- stage hooks ---------------------------------------------------------------------------------
int lm_stage_color_quantize(lm_detector* d, const uint8_t* bgr, int w, int h, float weak_threshold, uint8_t* quantized,
                            float* magnitude) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (!bgr || !quantized || w <= 0 || h <= 0) return fail(LM_ERR_INVALID, "bad argument");
    size_t px = (size_t)w * h;
    size_t o_q = align_up(px * 3 + 256, 256), o_m = o_q + align_up(px, 256), o_s = o_m + align_up(px * 4, 256);
    if ((rc = ensure_scratch(d, o_s + lmk_color_scratch_bytes(w, h)))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->stream;
    HIP_TRY(hipMemcpyAsync(base, bgr, px * 3, hipMemcpyHostToDevice, st));
    lmk_color_quantize(st, base, w, h, weak_threshold, base + o_q, magnitude ? reinterpret_cast<float*>(base + o_m) : nullptr,
                       base + o_s, 0, 1);
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
    hipStream_t st = d->stream;
    HIP_TRY(hipMemcpyAsync(base, bgr, px * 3, hipMemcpyHostToDevice, st));
    lmk_pyrdown(st, base, w, h, base + o_o, 0, 1);
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
    size_t o_q = align_up(px * 2, 256), o_s = o_q + align_up(px, 256);
    if ((rc = ensure_scratch(d, o_s + px))) return rc;
    u8* base = static_cast<u8*>(d->d_scratch);
    hipStream_t st = d->stream;
    HIP_TRY(hipMemcpyAsync(base, depth, px * 2, hipMemcpyHostToDevice, st));
    lmk_depth_quantize(st, reinterpret_cast<u16*>(base), w, h, d->cfg.distance_threshold, d->cfg.difference_threshold,
                       d->d_normal_lut, normal_lut_onehot(d), base + o_q, base + o_s, 0, 1);
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
    hipStream_t st = d->stream;
    HIP_TRY(hipMemcpyAsync(base, quantized, px, hipMemcpyHostToDevice, st));
    lmk_linear_memories(st, base, w, 0, 0, w, h, T, d->d_resp_tab, base + o_l, (u32)px, 0, 0, 1);  // dense: ori_stride = T*T*W*H
    HIP_TRY(hipMemcpyAsync(out, base + o_l, 8 * px, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return LM_OK;
}

int lm_prepare_slot(lm_detector* d, int slot) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    if (!d->slots[slot].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    if ((rc = enqueue_upload_wait(d, slot, 1))) return rc;
    enqueue_preprocess(d, slot, 1);
    d->cnt_preprocess_frames += 1;
    if ((rc = wait_stream(d))) return rc;
    HIP_TRY(hipGetLastError());
    d->slots[slot].prepared = true;
    return LM_OK;
}

int lm_debug_read(lm_detector* d, int slot, int what, int level, int modality, uint8_t* out, size_t cap, size_t* size_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    if (level < 0 || level >= d->cfg.pyramid_levels || modality < 0 || modality >= d->cfg.num_modalities)
        return fail(LM_ERR_INVALID, "level/modality out of range");
    HIP_TRY(hipStreamSynchronize(d->stream));
    const LmLevelGeom& g = d->geom[level];
    if (what == 0) {
        size_t n = (size_t)g.w * g.h;
        if (size_out) *size_out = n;
        if (modality == 1 && level > 0) {  // materialise the NN pyramid of the depth modality on demand
            enqueue_depth_pyramid(d, slot, 1);
            HIP_TRY(hipStreamSynchronize(d->stream));
        }
        if (out) HIP_TRY(hipMemcpy(out, d->quant(slot, level, modality), std::min(n, cap), hipMemcpyDeviceToHost));
        return LM_OK;
    }
    if (what == 1) {  // spread linear memory [memory][pos] (refinement levels only)
        size_t blk = (size_t)g.T * g.T * g.wh;
        if (!g.spread_only) return fail(LM_ERR_INVALID, "the lowest level keeps response memories, not the spread memory");
        if (size_out) *size_out = blk;
        if (out) {
            if (cap < blk) return fail(LM_ERR_INVALID, "buffer too small");
            HIP_TRY(hipMemcpy(out, d->lm(slot, level) + (size_t)modality * g.mod_stride, blk, hipMemcpyDeviceToHost));
        }
        return LM_OK;
    }
    if (what == 2) {
        size_t blk = (size_t)g.T * g.T * g.wh;
        size_t n = 8 * blk;
        if (size_out) *size_out = n;
        if (out) {
            if (cap < n) return fail(LM_ERR_INVALID, "buffer too small");
            if (g.spread_only || (level == d->cfg.pyramid_levels - 1 && d->slots[slot].spread_low)) {
                // refinement levels hold the spread memory (and so does the scanned level of a slot prepared for the bit-plane scan alone);
                // expand it with the response LUT here (debug path)
                std::vector<u8> sp(blk);
                HIP_TRY(hipMemcpy(sp.data(), d->lm(slot, level) + (size_t)modality * g.mod_stride, blk, hipMemcpyDeviceToHost));
                for (int o = 0; o < 8; ++o)
                    for (size_t i = 0; i < blk; ++i)
                        out[o * blk + i] = std::max(d->sim_lut[32 * o + (sp[i] & 15)], d->sim_lut[32 * o + 16 + (sp[i] >> 4)]);
            } else if (g.nibble) {
                std::vector<u8> pk(blk / 2);
                for (int o = 0; o < 8; ++o) {
                    HIP_TRY(hipMemcpy(pk.data(), d->lm(slot, level) + (size_t)modality * g.mod_stride + (size_t)o * g.ori_stride,
                                      blk / 2, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < blk / 2; ++i) { out[o * blk + 2 * i] = pk[i] & 15; out[o * blk + 2 * i + 1] = pk[i] >> 4; }
                }
            } else {
                for (int o = 0; o < 8; ++o)
                    HIP_TRY(hipMemcpy(out + o * blk, d->lm(slot, level) + (size_t)modality * g.mod_stride + (size_t)o * g.ori_stride, blk,
                                      hipMemcpyDeviceToHost));
            }
        }
        return LM_OK;
    }
    return fail(LM_ERR_INVALID, "unknown buffer id");
}

int lm_stage_scan(lm_detector* d, int slot, float threshold, int class_idx, int32_t* out, size_t cap_records, size_t* n_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    ItemRange r;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, threshold))) return rc;
    {
        LmScanArgs sa = make_scan_args(d, slot, r);
        if ((rc = check_scan_args(d, slot, sa))) return rc;
        lmk_scan(d->stream, sa, d->scan_variant, 1);
        d->last_scan1_lanes = sa.lds_form ? 1000 + sa.R : sa.L1;
        scan_launched(d, sa);
    }
    LmDevHeader h;
    HIP_TRY(hipMemcpyAsync(&h, d->aux(slot, d->off_hdr), sizeof(h), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemset(d->aux(slot, d->off_hdr), 0, sizeof(LmDevHeader)));  // re-arm the counters ourselves
    if (h.cand_count > d->max_cand) return fail(LM_ERR_OVERFLOW, "candidate buffer overflow");
    std::vector<LmCand> cand(h.cand_count);
    if (h.cand_count) HIP_TRY(hipMemcpy(cand.data(), d->aux(slot, d->off_cand), cand.size() * sizeof(LmCand), hipMemcpyDeviceToHost));
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
    if ((rc = check_slots(d, slot, 1))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (iters <= 0) return fail(LM_ERR_INVALID, "iters must be positive");
    ItemRange r;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, threshold))) return rc;
    LmScanArgs a = make_scan_args(d, slot, r);
    if ((rc = check_scan_args(d, slot, a))) return rc;
    a.cand_cap = 0;  // timing only: count candidates, store none (the list would overflow across iterations)
    for (int i = 0; i < 3; ++i) { lmk_scan(d->stream, a, variant, 1); scan_launched(d, a); }
    HIP_TRY(hipEventRecord(d->ev[0], d->stream));
    for (int i = 0; i < iters; ++i) { lmk_scan(d->stream, a, variant, 1); scan_launched(d, a); }
    HIP_TRY(hipEventRecord(d->ev[1], d->stream));
    HIP_TRY(hipMemsetAsync(d->aux(slot, d->off_hdr), 0, sizeof(LmDevHeader), d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1]));
    if (avg_us_out) *avg_us_out = (double)ms * 1000.0 / iters;
    if (algorithmic_bytes_out) {
        double b = 0;
        if (class_idx < 0) for (double v : d->hb.class_alg_bytes) b += v;
        else b = d->hb.class_alg_bytes[class_idx];
        *algorithmic_bytes_out = b;
    }
    return LM_OK;
}

// The scan kernel alone over a BATCH of prepared slots (one launch = n_slots frames, as a lane-step launches it), candidates
// counted but not stored.  `variant` as lm_set_scan_variant; 8 | 64 = exhaustive scan WITHOUT the shift-undo instructions
// (wrong sums -- a timing experiment only).
int lm_time_scan_batch(lm_detector* d, int first_slot, int n_slots, float threshold, int class_idx, int iters, int variant,
                       double* avg_us_out) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, first_slot, n_slots))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (iters <= 0 || n_slots <= 0) return fail(LM_ERR_INVALID, "iters and n_slots must be positive");
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    for (int i = 0; i < n_slots; ++i)
        if (!d->slots[first_slot + i].prepared) return fail(LM_ERR_INVALID, "slot " + std::to_string(first_slot + i) + " is not prepared");
    ItemRange r;
    if ((rc = item_range(d, class_idx, &r))) return rc;
    if ((rc = enqueue_threshold(d, threshold))) return rc;
    LmScanArgs a = make_scan_args(d, first_slot, r, n_slots);
    if ((rc = check_scan_args(d, first_slot, a))) return rc;
    a.cand_cap = 0;
    for (int i = 0; i < 2; ++i) { lmk_scan(d->stream, a, variant, n_slots); scan_launched(d, a); }
    HIP_TRY(hipEventRecord(d->ev[0], d->stream));
    for (int i = 0; i < iters; ++i) { lmk_scan(d->stream, a, variant, n_slots); scan_launched(d, a); }
    HIP_TRY(hipEventRecord(d->ev[1], d->stream));
    for (int i = 0; i < n_slots; ++i) HIP_TRY(hipMemsetAsync(d->aux(first_slot + i, d->off_hdr), 0, sizeof(LmDevHeader), d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1]));
    if (avg_us_out) *avg_us_out = (double)ms * 1000.0 / iters;
    return LM_OK;
}

// Self-test of k_dnormal's float tail (lm_kernels.hip dn_rcp / dn_sqrt): every float of the tail's domain through the short
// sequences and through the compiler's correctly rounded 1.0f / x and sqrtf (__builtin_sqrtf: v_sqrt_f32 + its +-1 ulp fix-up) on
// this device; out[0] / out[1] = floats that differ, out[2] = floats on which the bare v_sqrt_f32 differs (information), out[3..5] = the same
// for the longer sequences used before (v_rcp + six steps; v_sqrt + fix-up) and for v_sqrt + one v_rsq step, out[6..7] = 0.
int lm_selftest_float_tail(lm_detector* d, uint64_t out[8]) {
    int rc;
    if (!out) return fail(LM_ERR_INVALID, "null argument");
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    unsigned long long* dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), 8 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), d->stream);
    unsigned long long host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) { lmk_selftest_float_tail(d->stream, dev); e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, d->stream); }
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(LM_ERR_HIP, hipGetErrorString(e));
    for (int k = 0; k < 8; ++k) out[k] = host[k];
    return LM_OK;
}

int lm_time_stages(lm_detector* d, int slot, float threshold, int class_idx, int iters, double out_us[4]) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = check_slots(d, slot, 1))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    if (iters <= 0 || !out_us) return fail(LM_ERR_INVALID, "bad argument");
    if (!d->slots[slot].has_frame) return fail(LM_ERR_INVALID, "no frame uploaded to slot");
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if ((rc = enqueue_match(d, slot, 1, threshold, class_idx, true))) return rc;
        HIP_TRY(hipStreamSynchronize(d->stream));
        for (int k = 0; k < 4; ++k) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev[k], d->ev[k + 1]));
            acc[k] += (double)ms * 1000.0;
        }
    }
    for (int k = 0; k < 4; ++k) out_us[k] = acc[k] / iters;
    return LM_OK;
}

int lm_last_counts(lm_detector* d, int slot, uint32_t* candidates, uint32_t* matches_before_unique) {
    if (!d || !d->dev_ready || slot < 0 || slot >= (int)d->slots.size()) return fail(LM_ERR_INVALID, "bad argument");
    const LmHeader& h = d->host_block(slot)->hdr;
    if (candidates) *candidates = h.cand_count;
    if (matches_before_unique) *matches_before_unique = h.match_count;
    return LM_OK;
}

int lm_set_profiling(lm_detector* d, int enable) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    d->profiling = enable != 0;
    for (double& v : d->prof_us) v = 0;
    d->prof_scan_bytes = 0; d->prof_launches = 0; d->prof_frames = 0;
    d->prof_exch_us = 0; d->prof_exch_launches = 0; d->prof_exch_fallbacks = 0;
    d->cnt_preprocess_frames = d->cnt_scan_launches = d->cnt_refine_launches = d->cnt_sort_launches = 0;
    return LM_OK;
}

int lm_get_profile(lm_detector* d, double stage_us[4], double* scan_algorithmic_bytes, int64_t* launches, int64_t* frames) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (stage_us) for (int k = 0; k < 4; ++k) stage_us[k] = d->prof_us[k];
    if (scan_algorithmic_bytes) *scan_algorithmic_bytes = d->prof_scan_bytes;
    if (launches) *launches = d->prof_launches;
    if (frames) *frames = d->prof_frames;
    return LM_OK;
}

int lm_scan_load_bytes(lm_detector* d, int class_idx, double* bytes_per_frame) {
    int rc;
    if (!d || !bytes_per_frame) return fail(LM_ERR_INVALID, "null argument");
    if ((rc = ready_for_compute(d))) return rc;
    if ((rc = ensure_bank(d))) return rc;
    const int nc = (int)d->hb.class_load_bytes.size();
    if (class_idx >= nc || class_idx < -1) return fail(LM_ERR_INVALID, "class index out of range");
    double b = 0;
    if (class_idx < 0) for (double v : d->hb.class_load_bytes) b += v;
    else b = d->hb.class_load_bytes[class_idx];
    *bytes_per_frame = b;
    return LM_OK;
}

int lm_set_scan_stats(lm_detector* d, int enable) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(d->d_scan_stat, 0, 4096 * sizeof(unsigned long long)));
    d->scan_stats = enable != 0;
    return LM_OK;
}

int lm_get_scan_stats(lm_detector* d, uint64_t* features_loaded, uint64_t* features_unpruned) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    std::vector<unsigned long long> h(4096);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), d->d_scan_stat, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long a = 0, b = 0;
    for (int i = 0; i < 1024; ++i) { a += h[4 * i]; b += h[4 * i + 1]; }
    if (features_loaded) *features_loaded = a;
    if (features_unpruned) *features_unpruned = b;
    return LM_OK;
}

int lm_get_scan_lane_stats(lm_detector* d, uint64_t* lane_loads_issued, uint64_t* lane_loads_unpruned) {
    int rc;
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    std::vector<unsigned long long> h(4096);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), d->d_scan_stat, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long a = 0, b = 0;
    for (int i = 0; i < 1024; ++i) { a += h[4 * i + 2]; b += h[4 * i + 1]; }
    if (lane_loads_issued) *lane_loads_issued = a;
    if (lane_loads_unpruned) *lane_loads_unpruned = 64ull * b;
    return LM_OK;
}

int lm_get_scan_form_stats(lm_detector* d, int64_t out[4]) {
    int rc;
    if (!out) return fail(LM_ERR_INVALID, "null argument");
    if ((rc = ready_for_compute(d))) return rc;
    if (any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    std::vector<unsigned long long> h(4096);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), d->d_scan_stat, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long sv = 0;
    for (int i = 0; i < 1024; ++i) sv += h[4 * i + 3];
    out[0] = d->cnt_scan1_launches; out[1] = d->cnt_scan_launches; out[2] = (int64_t)sv; out[3] = d->last_scan1_lanes;
    return LM_OK;
}

int lm_get_exchange_profile(lm_detector* d, double* exchange_us, int64_t* launches, int64_t* fallbacks) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (exchange_us) *exchange_us = d->prof_exch_us;
    if (launches) *launches = d->prof_exch_launches;
    if (fallbacks) *fallbacks = d->prof_exch_fallbacks;
    return LM_OK;
}

int lm_get_stage_counts(lm_detector* d, int64_t out[4]) {
    if (!d || !out) return fail(LM_ERR_INVALID, "null argument");
    out[0] = d->cnt_preprocess_frames; out[1] = d->cnt_scan_launches; out[2] = d->cnt_refine_launches; out[3] = d->cnt_sort_launches;
    return LM_OK;
}

int lm_device_pci_bus_id(lm_detector* d, char* out, size_t cap) {
    int rc;
    if (!out || cap < 16) return fail(LM_ERR_INVALID, "buffer too small");
    if ((rc = ready_for_compute(d))) return rc;
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)cap, d->cfg.device));
    return LM_OK;
}

// Only variants whose lists are the default's may be set on the product path: bits 6 and 7 skip work (no shift-undo / no exact
// sums of the survivors) and exist for lm_time_scan* alone, which take their variant as an argument and store no candidates.
int lm_set_scan_variant(lm_detector* d, int variant) {
    if (!d) return fail(LM_ERR_INVALID, "null detector");
    if (variant < 0 || (variant & ~LM_SCAN_VARIANT_SETTABLE))
        return fail(LM_ERR_INVALID, "scan variant " + std::to_string(variant) + " changes the match lists (bits 6 / 7 are timing experiments of lm_time_scan* only) or is unknown");
    d->scan_variant = variant;
    return LM_OK;
}

}  // extern "C"
