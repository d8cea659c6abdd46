// lm_detector_impl.h -- what the sources of the detector's host side share: the detector's state (struct lm_detector, the opaque handle of
// include/linemod_hip.h), a frame slot's bookkeeping, the error channel of the C ABI and the internal functions one source calls in another.
//   lm_detector.hip         device state, bank upload, the per-batch launch sequence (a3-a15), lanes, uploads, create / templates / match
//   lm_detector_post.hip    f1: colour check (hulls, HSV masks) and the depth check's counts on the GPU
//   lm_detector_gather.hip  8e: RCCL communicator, the gathered match, the host side of the exchange (plans, merges)
//   lm_detector_io.hip      f2: bank / YAML persistence
//   lm_detector_debug.hip   stage hooks, timing calls, counters and statistics
// Internal names live in namespace lmd (external linkage between these sources, nothing of it is part of the ABI).
#pragma once
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

namespace lmd {

extern thread_local std::string g_err;           // what lm_last_error() returns on this thread (defined in lm_detector.hip)
int fail(int code, const std::string& msg);   // sets lm_last_error() of the calling thread and returns `code`

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

}  // namespace lmd
using namespace lmd;

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

namespace lmd {

template <typename T>
int upload_vec(T** dptr, const std::vector<T>& v) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return LM_OK;
}

struct ItemRange { int lo, n; int t_lo, t_hi; };   // items of the nibble / byte scan, and the bank-local templates they belong to


// The device sort's split form pays for lists longer than one chunk, and a launch lasts as long as its longest list: the score says
// whether any of the last 4096 collected frames had such a list.
inline void note_sort_length(lm_detector* d, u32 match_count) {
    const int is_long = match_count > LM_SORT_CHUNK && match_count <= LM_SORT_CAP;
    d->sort_long_score = is_long ? 4096 : std::max(d->sort_long_score - 1, 0);
}

// defined in lm_detector.hip
bool any_lane_busy(const lm_detector* d);
int ensure_device(lm_detector* d);
int ensure_luts(lm_detector* d);
int ensure_bank(lm_detector* d);
int check_slots(lm_detector* d, int first, int n);
bool normal_lut_onehot(lm_detector* d);
void enqueue_depth_pyramid(lm_detector* d, int first, int n);
void enqueue_preprocess(lm_detector* d, int first, int n);
int item_range(lm_detector* d, int class_idx, ItemRange* r);
LmScanArgs make_scan_args(lm_detector* d, int first, ItemRange r, int nslots = 1);
int check_scan_args(const lm_detector* d, int first, const LmScanArgs& a);
void scan_launched(lm_detector* d, LmScanArgs& a);
int enqueue_threshold(lm_detector* d, float threshold);
int enqueue_upload_wait(lm_detector* d, int first, int n);
int enqueue_match(lm_detector* d, int first, int n, float threshold, std::vector<int>& classes, bool timed = false, bool prepared = false);
int enqueue_match(lm_detector* d, int first, int n, float threshold, int class_idx, bool timed = false);
int collect_slot(lm_detector* d, int slot, lm_match_t* out, size_t cap, size_t* n_out);
int ready_for_compute(lm_detector* d);
int ensure_scratch(lm_detector* d, size_t bytes);
void activate_lane(lm_detector* d, int l);
int ensure_lane(lm_detector* d, int l);
void account_profile(lm_detector* d, int n, const std::vector<int>& classes, bool gathered = false);
int wait_stream(lm_detector* d);
int wait_lane_done(lm_detector* d, lm_detector::Lane& ln);
// defined in lm_detector_gather.hip
void free_gather(lm_detector* d);
int enqueue_gather(lm_detector* d, int lane, int first, int n);

}  // namespace lmd
