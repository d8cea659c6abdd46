// lm_detector_debug.hip -- stage hooks (one stage of the path in isolation, for the parity tests), timing calls (lm_time_scan*, lm_time_stages), the live
// profile, scan statistics and the A/B switches that are not LM_TUNE_* keys.  C ABI: lm_stage_*, lm_prepare_slot, lm_debug_read, lm_time_*, lm_get_*, lm_set_scan_*.
#include "lm_detector_impl.h"

extern "C" {

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
    HIP_TRY(hipMemsetAsync(d->aux(slot, d->off_hdr), 0, sizeof(LmDevHeader), d->stream));  // re-arm the counters ourselves (on the stream the scan runs on: the null stream is not ordered against it)
    HIP_TRY(hipStreamSynchronize(d->stream));
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

// Self-test of k_dnormal's float tail (lm_dev_depth.h dn_rcp / dn_sqrt): every float of the tail's domain through the short
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
    HIP_TRY(hipDeviceSynchronize());     // (the lanes' streams are non-blocking: the counters are zero before any later scan starts)
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
