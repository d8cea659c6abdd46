// lm_detector_post.hip -- f1 on the GPU: the colour check of many matches of resident frames (HighLevelLinemod.cpp:113-135,159-161,424-434: hulls of
// every template, HSV in-range masks, fill counts) and the depth check's counts (HighLevelLinemod.cpp:336-349,437-457), each on its own stream
// beside the match lanes.  C ABI: lm_color_mask_prepare, lm_color_check_*, lm_depth_counts_begin / _end.
#include "lm_detector_impl.h"

extern "C" {

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


}  // extern "C"
