// lm_detector_gather.hip -- the multi-GPU exchange: RCCL communicator per lane, the gathered match (k_pack_lists + 2 x ncclAllGather per
// lane-step, the sized collective fallback), and the host side of the exchange -- plans, merges, packing -- which is also what the CPU multi-process
// tests drive.  C ABI: lm_comm_*, lm_rendezvous_broadcast, lm_match_end_gathered, lm_gather_*, lm_merge_*, lm_pack_matches.
#include "lm_detector_impl.h"

namespace lmd {

void free_gather(lm_detector* d) {
    for (auto& g : d->gather) {
        hipFree(g.d_cnt); hipFree(g.d_rec); hipFree(g.d_all_cnt); hipFree(g.d_all_rec);
        if (g.h_all_cnt) hipHostFree(g.h_all_cnt);
        if (g.h_all_rec) hipHostFree(g.h_all_rec);
        g = lm_detector::Gather();
    }
    hipFree(d->d_red); d->d_red = nullptr;
}

// behind k_sort_unique on the active lane's stream: pack the lane's sorted lists, all-gather their lengths and the
// packed records (fixed capacity per rank, so no host round trip sits between the two collectives), copy both to
// pinned host memory.
int enqueue_gather(lm_detector* d, int lane, int first, int n) {
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

}  // namespace lmd

extern "C" {

// ---- multi-GPU exchange: RCCL all-gather of the per-shard lists ---------------------------------

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


}  // extern "C"
