// lm_detector.hip -- host side of liblinemod_hip.so: the C ABI of include/linemod_hip.h, the
// template bank (host copy + device upload), resident frame slots and the per-batch launch sequence.
//
// Mirrors cv::linemod::Detector as the reference uses it (/root/reference/src/HighLevelLinemod.cpp:
// 33-34,41-42 ctor; :93 addTemplate; :152 match; :115,181 getTemplates; :55,60,65 class queries).
// There is no CPU fallback: every compute entry point needs a HIP device and fails with
// LM_ERR_NO_DEVICE otherwise.
#include "lm_detector_impl.h"

namespace lmd {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

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
        // k_dnormal writes the label's RANK CODE (lm_dev_depth.h, a5 streaming form): looked up directly from a second table
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

LmScanArgs make_scan_args(lm_detector* d, int first, ItemRange r, int nslots) {
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
            // (on the lane's OWN stream: the lanes' streams are non-blocking, a hipMemset on the null stream is not ordered against them -- it could land after the
            // lane's first k_scan1 had started counting its survivors, and k_scan1_exact then summed fewer than were queued: the rare lost match of a lane's FIRST
            // bit-plane scan, tests/test_gpu_fullsize.py::test_config3_batch_bit_plane_scan_at_stated_size, about once in ten runs of the suite)
            else if (hipMemsetAsync(q, 0, 16 * sizeof(unsigned long long), d->stream) != hipSuccess) { hipFree(q); q = nullptr; (void)hipGetLastError(); }
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
int enqueue_match(lm_detector* d, int first, int n, float threshold, std::vector<int>& classes, bool timed,
                  bool prepared) {
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
int enqueue_match(lm_detector* d, int first, int n, float threshold, int class_idx, bool timed) {
    std::vector<int> classes(1, class_idx);
    return enqueue_match(d, first, n, threshold, classes, timed);
}

inline bool key_less(const u64* a, const u64* b) { return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]); }

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

void account_profile(lm_detector* d, int n, const std::vector<int>& classes, bool gathered) {
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

}  // namespace lmd

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
        case LM_TUNE_SURVIVOR_QUEUE:
            if (value < 64 || value > (1 << 24)) break;
            if (d->dev_ready) {       // no lane is busy (checked above); the queues' re-arming stores ran inside the matches that have ended
                HIP_TRY(hipSetDevice(d->cfg.device));
                HIP_TRY(hipDeviceSynchronize());
                for (int l = 0; l < LM_NLANES; ++l) { if (d->d_surv[l]) hipFree(d->d_surv[l]); d->d_surv[l] = nullptr; d->surv_set[l] = 0; }
            }
            d->surv_cap = ((u32)value + 7u) & ~7u;
            return LM_OK;
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

}  // extern "C"
