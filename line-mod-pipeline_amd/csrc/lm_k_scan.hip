// lm_k_scan.hip -- a11-a13 of the LINE-MOD match path for gfx950 (CDNA4, wave64): the similarity scan of the lowest pyramid level fused with the
// threshold scan -- the HOT kernels.  k_scan (byte responses), k_scan4 (nibble responses, exact pruning), k_scan1 + k_scan1_exact (bit-plane
// miss counting, bit-sliced carry-save counters on v_bitop3_b32), k_scanl (the bit-plane scan with a frame's planes in LDS), and lmk_scan.
// Integer work only; the candidate lists of all forms are identical record for record (tests/test_gpu_scan_planes.py, tests/test_gpu_fullsize.py).
#include "lm_dev.h"

namespace {

template <int UNROLL, bool XCD_MAP>
__global__ __launch_bounds__(256) void k_scan(LmScanArgs a) {
    const int lane = threadIdx.x & 63;
    u32 slot, wg;
    if (XCD_MAP) {
        const u32 G = (u32)a.wgs_per_slot, B = (u32)a.nslots;
        const u32 b = blockIdx.x, x = b & 7u, k = b >> 3;
        if ((B & 7u) == 0) { slot = x + 8u * (k / G); wg = k % G; }          // XCD x owns slots = x mod 8
        else { const u32 r = 8u / B; slot = x % B; wg = k * r + x / B; }     // B in {1,2,4}: r XCDs share a slot
        if (wg >= G || slot >= B) return;
    } else {
        slot = blockIdx.z; wg = blockIdx.x;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)((wg * 256u + threadIdx.x) >> 6));
    if (wave >= a.n_items) return;
    const u32 ti = a.item_t[a.item_lo + wave];
    const u32 chunk = a.item_chunk[a.item_lo + wave];
    const int P = a.scan_P[ti];
    const int n = a.scan_n[ti] & 0xFF;
    const int thr = a.raw_thr_by_n[n];
    const u32 j0 = chunk * LM_SCAN_CHUNK + (u32)lane * 16u;
    // buffer addressing: descriptor base = this wave's chunk, voffset = the lane's 16 bytes, soffset = feature
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u8*>(a.lm + (size_t)slot * a.lm_slot_stride + (size_t)(chunk * LM_SCAN_CHUNK)), 0, 0x7FFFFFFF, 0x00020000);
    const u32 lane_off = (u32)lane * 16u;
    LmDevHeader* hdr = reinterpret_cast<LmDevHeader*>(reinterpret_cast<u8*>(a.hdr) + (size_t)slot * a.aux_slot_stride);
    LmCand* cand = reinterpret_cast<LmCand*>(reinterpret_cast<u8*>(a.cand) + (size_t)slot * a.aux_slot_stride);

    u32 tl[4] = {0, 0, 0, 0}, th[4] = {0, 0, 0, 0};  // u16 pairs: bytes {0,2} and {1,3} of each dword
    for (int m = 0; m < a.M; ++m) {
        const u32* offs = a.scan_off + ((size_t)ti * a.M + m) * a.fpad;
        u32x4 acc = {0, 0, 0, 0};
        for (int f = 0; f < a.fpad; f += UNROLL) {
            u32x4 v[UNROLL];
            u32 sh[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const u32 o = offs[f + k];
                sh[k] = o & 3u;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, o & ~3u, 0);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const u32 nx = next_lane(v[k][0]);
                acc[0] += __builtin_amdgcn_alignbyte(v[k][1], v[k][0], sh[k]);
                acc[1] += __builtin_amdgcn_alignbyte(v[k][2], v[k][1], sh[k]);
                acc[2] += __builtin_amdgcn_alignbyte(v[k][3], v[k][2], sh[k]);
                acc[3] += __builtin_amdgcn_alignbyte(nx, v[k][3], sh[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tl[k] += acc[k] & 0x00FF00FFu;
            th[k] += (acc[k] >> 8) & 0x00FF00FFu;
        }
    }
    // threshold scan: strict >, positions >= P hold 0 upstream (never a candidate since thr >= 0)
    u32 hit = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int s0 = (int)(tl[k] & 0xFFFF), s1 = (int)(th[k] & 0xFFFF), s2 = (int)(tl[k] >> 16), s3 = (int)(th[k] >> 16);
        hit |= (s0 > thr ? 1u : 0u) << (4 * k) | (s1 > thr ? 1u : 0u) << (4 * k + 1) |
               (s2 > thr ? 1u : 0u) << (4 * k + 2) | (s3 > thr ? 1u : 0u) << (4 * k + 3);
    }
    int valid = P - (int)j0;  // number of valid positions in this lane's 16
    if (lane == 63) valid = 0;  // lane 63 only supplies lane 62's spill-over bytes
    if (valid <= 0) hit = 0;
    else if (valid < 16) hit &= (1u << valid) - 1u;
    if (!__any(hit != 0)) return;
    const int offset = a.T / 2 + (a.T % 2 - 1);
    while (hit) {
        int b = __ffs(hit) - 1;
        hit &= hit - 1;
        int k = b >> 2, bb = b & 3;
        int raw = (bb == 0) ? (int)(tl[k] & 0xFFFF) : (bb == 1) ? (int)(th[k] & 0xFFFF) : (bb == 2) ? (int)(tl[k] >> 16) : (int)(th[k] >> 16);
        int j = (int)j0 + b;
        int r = j / a.W, c = j - r * a.W;
        u32 slot = atomicAdd(&hdr->cand_count, 1u);
        if (slot < a.cand_cap) {
            LmCand cd;
            cd.ti = ti;
            cd.x = c * a.T + offset;
            cd.y = r * a.T + offset;
            cd.sim = __fadd_rn(__fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * n)), 0.5f);
            cand[slot] = cd;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a11-a13, nibble form of the hot kernel (LmLevelGeom::nibble): responses are <= 4, so the lowest
// level stores two positions per byte.  The vector L1 spends one cycle per 4 lanes of a load whatever
// the width per lane, so every lane loads 16 bytes = 32 positions, and one wave carries one work item
// (template, chunk of LM_SCAN4_CHUNK positions) for TWO FRAMES: lanes 0..31 scan slot 2j, lanes 32..63
// slot 2j + 1 (P is about 1000 for a 640x480 frame at T = 8: one item per template).  The template is the
// same for both halves, so the feature offsets, the dword-aligned load address (buffer soffset) and the
// 0..7 nibble shift are wave-uniform scalars: a feature costs no VALU instruction for addressing.  The
// shift is undone with v_alignbit_b32, the 33rd.. nibble coming from the next lane by DPP -- lane 31
// receives the other frame's dword there, which only reaches positions >= 1017 of the chunk, hence 1016
// positions per item.  Three features are added nibble-wise (3 * 4 = 12 < 16), then split into even / odd
// positions and added byte-wise (63 * 4 = 252).
// ------------------------------------------------------------------------------------------------
// Exact pruning (PRUNE): a position becomes a candidate only if its total exceeds the raw threshold, and a feature adds
// at most 4.  So once, for EVERY position a wave holds (1016 positions of two frames), partial sum + 4 x (in-bounds
// features still to come) <= threshold, none of them can become a candidate and the wave stops loading.  The test is
// made after every block of FB features, from the first block at which even a partial sum of 0 would be out of reach
// (a scalar compare), and at the end of every modality's list: a packed-u16 max over the lane's 32 partial sums, one
// compare, one ballot -- wave-uniform, so no lane ever diverges.  (The L1, not the vector ALU, bounds this kernel: the
// tests ride in its shadow.  r02: tests only at the middle and the end of a list kept 54 % / 59 % of the loads of
// configs 2 / 3, every block 50 % / 43 %.)  The candidate list is identical with and without
// it (tests/test_gpu_match.py::test_scan_pruning_is_exact); at threshold 80 most templates stop after half their
// features.  a.stat (optional): [0] += features loaded, [1] += features an unpruned scan would load, per wave.
//
// Per-lane pruning (PRUNE == 2, the default; r03).  The same test answers per LANE: a lane none of whose 32 positions can
// still reach the threshold is dead for the rest of the item.  Dead lanes leave the exec mask of the feature blocks that
// follow (the vector L1 spends its cycles per quad of ACTIVE lanes of a load, so a wave whose survivors are the few
// lanes around a real match costs a fraction of a full wave-load); a live lane's right neighbour stays in (it supplies
// the spill-over dword of the shift), dead lanes can never emit (their true totals are below the threshold whatever
// their stale registers hold: the hit mask is cleared for them), and the wave stops once no lane is alive -- the
// wave-level rule of PRUNE == 1 is the special case "all lanes dead".  a.stat[2] counts the lane-loads really issued.
// NOSHIFT (measurement only, WRONG sums): the loaded dwords are added as they are, without the v_alignbit / DPP shift-undo --
// the upper bound of what pre-shifted copies of the linear memories could save in vector instructions, at no cost in
// footprint (lm_time_scan_batch, scan variant 8 | 64).
template <int FB, bool XCD_MAP, int PRUNE, bool NOSHIFT = false>
__global__ __launch_bounds__(256) void k_scan4(LmScanArgs a) {
    const int lane = threadIdx.x & 63;
    const u32 npairs = ((u32)a.nslots + 1u) >> 1;
    u32 pair, wg;
    if (XCD_MAP) xcd_slot_tile((u32)a.wgs_per_slot, npairs, pair, wg);
    else { pair = blockIdx.z; wg = blockIdx.x; }
    if (pair >= npairs) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)((wg * 256u + threadIdx.x) >> 6));
    if (wave >= a.n_items) return;
    const u32 ti = a.item_t[a.item_lo + wave];
    const u32 chunk = a.item_chunk[a.item_lo + wave];
    const int cnt = a.scan_n[ti];                      // n | features of modality 0 << 8 | of 1 << 16
    const int n = cnt & 0xFF;
    const int thr = a.raw_thr_by_n[n];
    const bool hi = lane >= 32;
    const u32 slot0 = 2u * pair;
    const bool have = !hi || slot0 + 1u < (u32)a.nslots;            // an odd slot count leaves the last upper half idle
    const u32 slot = slot0 + ((hi && have) ? 1u : 0u);
    const int P = have ? a.scan_P[ti] : 0;
    const u32 j0 = chunk * LM_SCAN4_CHUNK + (u32)(lane & 31) * 32u;  // first position of this lane
    // buffer addressing: descriptor = arena of slot 2j, voffset = the lane's 16 bytes inside the chunk (+ one slot
    // stride for the upper half), soffset = the feature's dword-aligned byte offset
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u8*>(a.lm + (size_t)slot0 * a.lm_slot_stride), 0, 0x7FFFFFFF, 0x00020000);
    const u32 lane_base = (j0 >> 1) + ((slot != slot0) ? (u32)a.lm_slot_stride : 0u);

    // u16 pairs: position 8k + i of the lane lives in t[k][i & 3], half i >> 2
    u32 t[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int f_in_all = 0;                                      // in-bounds features of all modalities
    for (int m = 0; m < a.M; ++m) f_in_all += (cnt >> (8 + 8 * m)) & 0xFF;
    int f_done = 0;                                        // features loaded so far (all modalities)
    // (Pruning the two frames of a wave separately -- the half whose frame is out of reach leaves the exec mask of the
    // loads -- was measured in r02: 3-6 % fewer loads, 4-8 % MORE time; the test is per wave.)
    bool pruned = false;
    unsigned long long alive = ~0ull;                      // PRUNE == 2: lanes with a position still in reach
    bool act = true;                                       // this lane loads (alive, or the right neighbour of an alive lane)
    u32 lane_loads = 0;                                    // statistics: lane-loads issued (PRUNE == 2; otherwise 64 per feature loaded)
    for (int m = 0; m < a.M && !pruned; ++m) {
        const u32* offs = a.scan_off + ((size_t)ti * a.M + m) * a.fpad;
        u32 bl[4] = {0, 0, 0, 0}, bh[4] = {0, 0, 0, 0};   // byte lanes: even / odd positions of dword k
#define LM_SCAN4_BLOCK(NF)                                                                       \
        {                                                                                        \
        if (PRUNE == 2) lane_loads += (u32)(NF) * (u32)__popcll(alive | (alive << 1));   /* scalar, unconditional: no branch in front of the loads */ \
        if (PRUNE != 2 || act) {                                                                 \
            u32x4 v[NF];                                                                         \
            u32 sh[NF];                                                                          \
            _Pragma("unroll") for (int k = 0; k < NF; ++k) {                                     \
                const u32 o = offs[f + k];                                                       \
                sh[k] = (o & 7u) << 2;                                                           \
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_base, (o >> 3) << 2, 0); \
            }                                                                                    \
            _Pragma("unroll") for (int g3 = 0; g3 < NF; g3 += 3) {                               \
                u32 nb[4] = {0, 0, 0, 0};                                                        \
                _Pragma("unroll") for (int k = g3; k < (g3 + 3 < NF ? g3 + 3 : NF); ++k) {       \
                    if (NOSHIFT) { nb[0] += v[k][0]; nb[1] += v[k][1]; nb[2] += v[k][2]; nb[3] += v[k][3]; continue; } \
                    const u32 nx = next_lane(v[k][0]);                                           \
                    nb[0] += __builtin_amdgcn_alignbit(v[k][1], v[k][0], sh[k]);                 \
                    nb[1] += __builtin_amdgcn_alignbit(v[k][2], v[k][1], sh[k]);                 \
                    nb[2] += __builtin_amdgcn_alignbit(v[k][3], v[k][2], sh[k]);                 \
                    nb[3] += __builtin_amdgcn_alignbit(nx, v[k][3], sh[k]);                      \
                }                                                                                \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                  \
                    bl[q] += nb[q] & 0x0F0F0F0Fu; bh[q] += (nb[q] >> 4) & 0x0F0F0F0Fu;           \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        }
        const int F = (cnt >> (8 + 8 * m)) & 0xFF;        // in-bounds features of this modality
        // first block boundary at or past the middle of the list: the mid-list test
        int f = 0;
        for (; f + FB <= F; f += FB) {
            LM_SCAN4_BLOCK(FB)
            // (a scalar test first: while even a partial sum of 0 could still reach the threshold, nothing can be pruned)
            if (PRUNE && f + FB < F && 4 * (f_in_all - f_done - (f + FB)) <= thr) {
                const int rem = f_in_all - f_done - (f + FB);                                   // features still to come
                if (m == 0 && F <= 31) {
                    // first modality, byte sums <= 124: "some byte > B" for B = thr - 4 rem in 0 .. 127 is a carry into bit 7
                    // of byte + (127 - B), no compare per position (B > 124: nothing can reach it; B < 0 was excluded above)
                    const int B = thr - 4 * rem;
                    unsigned long long left = 0;
                    if (B <= 124) {
                        const u32 K = (u32)(127 - B) * 0x01010101u;
                        const u32 y = (bl[0] + K) | (bl[1] + K) | (bl[2] + K) | (bl[3] + K) | (bh[0] + K) | (bh[1] + K) | (bh[2] + K) | (bh[3] + K);
                        left = __ballot((y & 0x80808080u) != 0u) & alive;
                    }
                    if (!left) { f_done += f + FB; pruned = true; break; }
                    if (PRUNE == 2) { alive = left; act = (((left | (left << 1)) >> lane) & 1ull) != 0; }
                    continue;
                }
                // largest partial sum of the lane: t (earlier modalities) + this modality's byte lanes
                u32 mx = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 s0 = t[k][0] + (bl[k] & 0x00FF00FFu), s2 = t[k][2] + ((bl[k] >> 8) & 0x00FF00FFu);
                    const u32 s1 = t[k][1] + (bh[k] & 0x00FF00FFu), s3 = t[k][3] + ((bh[k] >> 8) & 0x00FF00FFu);
                    mx = pk_max_u16(mx, pk_max_u16(pk_max_u16(s0, s1), pk_max_u16(s2, s3)));
                }
                const int best = (int)max(mx & 0xFFFFu, mx >> 16);
                const unsigned long long left = __ballot(best + 4 * (f_in_all - f_done - (f + FB)) > thr) & alive;
                if (!left) { f_done += f + FB; pruned = true; break; }
                if (PRUNE == 2) { alive = left; act = (((left | (left << 1)) >> lane) & 1ull) != 0; }
            }
        }
        if (pruned) break;
        if (FB > 6 && f + 6 <= F) { LM_SCAN4_BLOCK(6) f += 6; }
        if (FB > 3 && f + 3 <= F) { LM_SCAN4_BLOCK(3) f += 3; }
        if (F - f == 2) LM_SCAN4_BLOCK(2)
        else if (F - f == 1) LM_SCAN4_BLOCK(1)
#undef LM_SCAN4_BLOCK
        f_done += F;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k][0] += bl[k] & 0x00FF00FFu; t[k][2] += (bl[k] >> 8) & 0x00FF00FFu;
            t[k][1] += bh[k] & 0x00FF00FFu; t[k][3] += (bh[k] >> 8) & 0x00FF00FFu;
        }
        if (PRUNE && m + 1 < a.M) {                        // end of a modality's list, more modalities to come
            u32 mx = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) mx = pk_max_u16(mx, pk_max_u16(pk_max_u16(t[k][0], t[k][1]), pk_max_u16(t[k][2], t[k][3])));
            const int best = (int)max(mx & 0xFFFFu, mx >> 16);
            const unsigned long long left = __ballot(best + 4 * (f_in_all - f_done) > thr) & alive;
            if (!left) pruned = true;
            else if (PRUNE == 2) { alive = left; act = (((left | (left << 1)) >> lane) & 1ull) != 0; }
        }
    }
    if (a.stat && lane == 0) {
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u)], (unsigned long long)f_done);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 1], (unsigned long long)f_in_all);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 2], PRUNE == 2 ? (unsigned long long)lane_loads : 64ull * (unsigned long long)f_done);
    }
    if (pruned) return;
    u32 hit = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sv = (i >> 2) ? (int)(t[k][i & 3] >> 16) : (int)(t[k][i & 3] & 0xFFFF);
            hit |= (sv > thr ? 1u : 0u) << (8 * k + i);
        }
    // positions of this item: [chunk * CHUNK, min(P, (chunk + 1) * CHUNK))
    const int lim = min(P, (int)((chunk + 1u) * LM_SCAN4_CHUNK));
    const int valid = lim - (int)j0;
    if (valid <= 0) hit = 0;
    else if (valid < 32) hit &= (1u << valid) - 1u;
    if (PRUNE == 2 && !((alive >> lane) & 1ull)) hit = 0;   // a dead lane's registers are stale; its true totals cannot reach the threshold
    if (!__any(hit != 0)) return;
    LmDevHeader* hdr = reinterpret_cast<LmDevHeader*>(reinterpret_cast<u8*>(a.hdr) + (size_t)slot * a.aux_slot_stride);
    LmCand* cand = reinterpret_cast<LmCand*>(reinterpret_cast<u8*>(a.cand) + (size_t)slot * a.aux_slot_stride);
    const int offset = a.T / 2 + (a.T % 2 - 1);
    // one reservation per lane: its hits go into the list back to back, in position order, so neighbouring lattice
    // positions of one template -- which refine onto overlapping patches -- sit next to each other in the list (and
    // are refined by waves of one workgroup at the same time: their patch lines are then L1 hits)
    u32 pos = hit ? atomicAdd(&hdr->cand_count, (u32)__popc(hit)) : 0u;
    for (; hit; ++pos) {
        const int b = __ffs(hit) - 1;
        hit &= hit - 1;
        u32 tv = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) tv = ((b >> 3) == k && (b & 3) == q) ? t[k][q] : tv;
        const int raw = (b & 4) ? (int)(tv >> 16) : (int)(tv & 0xFFFF);
        const int j = (int)j0 + b;
        const int r = j / a.W, c = j - r * a.W;
        if (pos < a.cand_cap) {
            LmCand cd;
            cd.ti = ti;
            cd.x = c * a.T + offset;
            cd.y = r * a.T + offset;
            cd.sim = __fadd_rn(__fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * n)), 0.5f);
            cand[pos] = cd;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a11-a13, bit-plane form of the hot kernel (r05; LmScanArgs::L1 != 0).  k_scan4 adds every response of every feature at every
// position -- 4 bits a position, 14 vector instructions per feature and 32 positions -- although all the threshold scan wants
// to know is WHERE the sum exceeds the threshold.  A response is 4 only where the feature's orientation itself is present; every
// other response is at most 4 - delta (delta = 1 with upstream's table, whose responses are 4, 3, 2, 1, 0 by angular distance).  So a
// position that has MISSED (response below 4) more than m_max = (4 F - threshold - 1) / delta of a template's F in-bounds
// features cannot exceed the threshold whatever the missed responses were.  This kernel only counts misses:
//   * the producer (d_lm_fast, MODE 2 + planes) keeps, next to the nibble memories, one BIT per position and orientation --
//     1 = a miss -- in the same linear order; a lane's 16-byte load is 128 positions, lanes 0 .. L-1 of a frame cover a chunk of
//     128 L - 31 positions (the last 31 positions of a lane's window need the next lane's first dword: v_mov_dpp), and a wave
//     carries the same work item (template, chunk) for G = 64 / L FRAMES: feature offset and bit shift are wave-uniform
//     scalars, the shift is undone by v_alignbit_b32 (5 instructions per feature and 128 positions);
//   * the misses are counted bit-sliced: c[b] holds bit b of the counters of 32 positions; eight features enter per round
//     through a carry-save tree (7 full adders = 2 v_bitop3_b32 each, 4 half adders, one OR: 23 instructions per dword, 2.9 per
//     feature) -- 16.5 vector instructions per feature and 128 positions with the shift-undo, against 57 in k_scan4;
//   * the counters start at 127 - m_max, so bit 7 says "more than m_max misses": the dead flag of a position (sticky: the
//     counter cannot pass 255).  Invalid positions start dead.  After every round a lane whose four flag dwords are all ones is
//     dead and leaves the exec mask of the loads (per-lane pruning as in k_scan4), and the wave stops when no lane is left;
//   * the positions alive after the last feature are a superset of the candidates (the bound is exact when no response was 0).
//     Their exact sums come from the nibble memories, the wave working on one survivor at a time: lane f adds feature f's
//     response, a DPP reduction gives the sum, and only sums above the threshold are emitted -- the candidate list is the one
//     k_scan4 writes (tests/test_gpu_match.py, the fuzzer).
// a.stat: [0] += features loaded, [1] += features an unpruned scan would load, [2] += lane-loads issued, [3] += survivors.
// ------------------------------------------------------------------------------------------------
// full adder of three bit vectors on gfx950's three-input truth-table instruction: sum = a ^ b ^ c (0x96), carry = majority (0xE8)
__device__ __forceinline__ void bs_fa(u32& c, u32 a, u32 b, u32& cy) {
    cy = __builtin_amdgcn_bitop3_b32(c, a, b, 0xE8);
    c = __builtin_amdgcn_bitop3_b32(c, a, b, 0x96);
}
// TOP: the counters' flag bit (7: they count to 127; 5 (r06, k_scanl): to 31 -- two half adders fewer per round and dword)
template <int LEV, int NIN, int TOP = 7>
__device__ __forceinline__ void bs_level(u32 (&c)[8], const u32 (&in)[8]) {
    if constexpr (NIN == 0) {
        return;
    } else if constexpr (LEV == TOP) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) c[TOP] |= in[i];
    } else {
        u32 out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int NP = NIN / 2;
#pragma unroll
        for (int i = 0; i < NP; ++i) bs_fa(c[LEV], in[2 * i], in[2 * i + 1], out[i]);
        if constexpr (NIN & 1) { out[NP] = c[LEV] & in[NIN - 1]; c[LEV] ^= in[NIN - 1]; }
        bs_level<LEV + 1, NP + (NIN & 1), TOP>(c, out);
    }
}
template <int NF>
__device__ __forceinline__ void s1_load(const __amdgpu_buffer_rsrc_t rsrc, u32 lane_base, const u32* __restrict__ offs, u32 cbase,
                                        bool act, u32x4 (&v)[8], u32 (&sh)[8]) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const u32 o = offs[k] + cbase;
        sh[k] = o & 31u;
        if (act) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_base, (o >> 5) << 2, 0);
    }
}
template <int NF>
__device__ __forceinline__ void s1_compute(bool act, const u32x4 (&v)[8], const u32 (&sh)[8], u32 (&c)[4][8]) {
    if (!act) return;
    u32 x[4][8];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const u32 nx = next_lane(v[k][0]);
        x[0][k] = __builtin_amdgcn_alignbit(v[k][1], v[k][0], sh[k]);
        x[1][k] = __builtin_amdgcn_alignbit(v[k][2], v[k][1], sh[k]);
        x[2][k] = __builtin_amdgcn_alignbit(v[k][3], v[k][2], sh[k]);
        x[3][k] = __builtin_amdgcn_alignbit(nx, v[k][3], sh[k]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) bs_level<0, NF>(c[q], x[q]);
}
template <int NF>
__device__ __forceinline__ void s1_round(const __amdgpu_buffer_rsrc_t rsrc, u32 lane_base, const u32* __restrict__ offs, u32 cbase,
                                         bool act, u32 (&c)[4][8]) {
    u32x4 v[8];
    u32 sh[8];
    s1_load<NF>(rsrc, lane_base, offs, cbase, act, v, sh);
    s1_compute<NF>(act, v, sh, c);
}

__global__ __launch_bounds__(256) void k_scan1(LmScanArgs a) {
    const int lane = threadIdx.x & 63;
    const u32 ngroups = ((u32)a.nslots + (u32)a.G1 - 1u) / (u32)a.G1;
    u32 grp, wg;
    xcd_slot_tile((u32)a.wgs_per_slot, ngroups, grp, wg);
    if (grp >= ngroups) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)((wg * 256u + threadIdx.x) >> 6));
    if (wave >= a.n_items) return;
    const u32 ti = a.item_t[a.item_lo + wave];
    const u32 chunk = a.item_chunk[a.item_lo + wave];
    const int cnt = a.scan_n[ti];
    const int n = cnt & 0xFF;
    const int F = ((cnt >> 8) & 0xFF) + ((cnt >> 16) & 0xFF);    // in-bounds features of all modalities
    const int thr = a.raw_thr_by_n[n];
    const int K0 = 4 * F - thr - 1;                              // what the misses may cost in total
    if (K0 < 0) return;                                          // even F exact responses stay at or below the threshold
    int mmax = (int)(((u32)K0 * a.delta_rcp16) >> 16);           // K0 / delta
    if (mmax > 127) mmax = 127;
    const u32 pre = (u32)(127 - mmax);
    const int L = a.L1, CH = 128 * L - 31;
    const int fr = (int)(((u32)lane * a.L1_rcp16) >> 16), li = lane - fr * L;    // frame of the group, lane of the frame
    const u32 slot0 = grp * (u32)a.G1;
    const u32 slot = slot0 + (u32)fr;
    const bool have = fr < a.G1 && slot < (u32)a.nslots;
    const int P = a.scan_P[ti];
    const u32 cbase = chunk * (u32)CH;                                            // first position of the item
    const int j0 = (int)cbase + li * 128;                                         // first position of this lane
    int valid = have ? min(P, (int)((chunk + 1u) * (u32)CH)) - j0 : 0;
    valid = valid < 0 ? 0 : (valid > 128 ? 128 : valid);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u8*>(a.lm + (size_t)slot0 * a.lm_slot_stride), 0, 0x7FFFFFFF, 0x00020000);
    const u32 lane_base = (have ? (u32)fr * (u32)a.lm_slot_stride : 0u) + (u32)li * 16u;   // (a lane without a frame may still feed its left neighbour: any mapped address)
    u32 c[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 7; ++b) c[q][b] = ((pre >> b) & 1u) ? 0xFFFFFFFFu : 0u;
        const int vq = valid - 32 * q;
        c[q][7] = vq >= 32 ? 0u : (vq > 0 ? ~((1u << vq) - 1u) : 0xFFFFFFFFu);      // invalid positions start dead
    }
    unsigned long long alive = __ballot(valid > 0);
    if (!alive) return;
    bool act = (((alive | (alive << 1)) >> lane) & 1ull) != 0;
    const u32* offs = a.off1 + (size_t)ti * a.fpad1;
    u32 lane_loads = 0;
    int f = 0;
    bool pruned = false;
#define S1_TEST()                                                                                                   \
    {                                                                                                               \
        const unsigned long long left = __ballot((c[0][7] & c[1][7] & c[2][7] & c[3][7]) != 0xFFFFFFFFu) & alive;   \
        if (!left) pruned = true;                                                                                   \
        else { alive = left; act = (((left | (left << 1)) >> lane) & 1ull) != 0; }                                  \
    }
    // (r05, measured: the next round's loads issued before this round is counted -- two register sets, 152 VGPRs, three waves per SIMD --
    // take 162 instead of 130 us per 96-frame launch; the waves of a SIMD hide each other's loads better than a wave hides its own)
    for (; f + 8 <= F && !pruned; f += 8) {
        lane_loads += 8u * (u32)__popcll(alive | (alive << 1));
        s1_round<8>(rsrc, lane_base, offs + f, cbase, act, c);
        if (f + 8 > mmax && f + 8 < F) S1_TEST()                 // (nothing can be dead before mmax + 1 features are in)
    }
    if (!pruned) {
        if (F - f >= 4) { lane_loads += 4u * (u32)__popcll(alive | (alive << 1)); s1_round<4>(rsrc, lane_base, offs + f, cbase, act, c); f += 4; }
        if (F - f >= 2) { lane_loads += 2u * (u32)__popcll(alive | (alive << 1)); s1_round<2>(rsrc, lane_base, offs + f, cbase, act, c); f += 2; }
        if (F - f >= 1) { lane_loads += (u32)__popcll(alive | (alive << 1)); s1_round<1>(rsrc, lane_base, offs + f, cbase, act, c); f += 1; }
    }
#undef S1_TEST
    // survivors: positions never flagged (a dead or idle lane's flags are all ones)
    u32 h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = pruned ? 0u : ~c[q][7];
    unsigned long long hl = __ballot((h[0] | h[1] | h[2] | h[3]) != 0u);
    if (a.stat && lane == 0) {
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u)], (unsigned long long)f);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 1], (unsigned long long)F);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 2], (unsigned long long)lane_loads);
    }
    if (!hl || a.no_exact) return;
    if (a.surv) {
        // the survivors go to the stream's queue (k_scan1_exact, one lane per survivor); a lane whose reservation does not fit keeps its
        // hits for the loop below
        const u32 nh = (u32)(__popc(h[0]) + __popc(h[1]) + __popc(h[2]) + __popc(h[3]));
        if (nh) {
            // eight queues, one per XCD when the groups are dealt to the XCDs (group g runs on XCD g % 8): k_scan1_exact's workgroups of XCD x
            // take queue x, whose entries name the few frames that XCD has just scanned -- their nibble memories then meet in ITS L2
            const u32 cap8 = a.surv_cap >> 3, qx = grp & 7u;
            unsigned long long* qcount = a.surv + 8 * a.surv_set + qx;
            // (r06: the counter only grows.  r05 gave a reservation that did not fit back by a subtraction; a later subtraction of another wave could pull
            // the counter below entries a third wave had written in between -- found in k_scanl, which had the same code, as one candidate of 4 M lost on
            // a frame whose queues fill.  Now a reservation that does not fit entirely writes as many survivors as fit; the rest stay with the wave.)
            const unsigned long long at = atomicAdd(qcount, (unsigned long long)nh);
            u32 fit = at >= (unsigned long long)cap8 ? 0u : (u32)min((unsigned long long)nh, (unsigned long long)cap8 - at);
            if (a.stat && fit) atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 3], (unsigned long long)fit);
            unsigned long long* q = a.surv + 16 + (size_t)qx * cap8 + at;
            const unsigned long long hi = ((unsigned long long)ti << 32) | ((unsigned long long)slot << 20);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (; h[k] && fit; --fit) { const int b = __ffs((int)h[k]) - 1; *q++ = hi | (unsigned long long)(u32)(j0 + 32 * k + b); h[k] &= h[k] - 1u; }
        }
        hl = __ballot((h[0] | h[1] | h[2] | h[3]) != 0u);
        if (!hl) return;
    }
    // exact sums of the survivors from the nibble memories, one survivor at a time: lane k adds features k and k + 64
    const u32* offn = (a.exact_spread ? a.offs3 : a.offn) + (size_t)ti * a.fpad1;
    const u32 on0 = lane < F ? offn[lane] : 0u, on1 = lane + 64 < F ? offn[lane + 64] : 0u;
    const int offset = a.T / 2 + (a.T % 2 - 1);
    u32 n_surv = 0;
    while (hl) {
        const int src = __ffsll((long long)hl) - 1;
        const u32 w0 = (u32)__builtin_amdgcn_readlane((int)h[0], src), w1 = (u32)__builtin_amdgcn_readlane((int)h[1], src);
        const u32 w2 = (u32)__builtin_amdgcn_readlane((int)h[2], src), w3 = (u32)__builtin_amdgcn_readlane((int)h[3], src);
        const int q = w0 ? 0 : (w1 ? 1 : (w2 ? 2 : 3));
        const u32 wq = w0 ? w0 : (w1 ? w1 : (w2 ? w2 : w3));
        const int b = __ffs((int)wq) - 1;
        const int j = __builtin_amdgcn_readlane(j0, src) + 32 * q + b;
        const u32 sl = (u32)__builtin_amdgcn_readlane((int)slot, src);
        if (lane == src) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = (k == q) ? (h[k] & ~(1u << b)) : h[k];
        }
        const u8* nb = a.lm + (size_t)sl * a.lm_slot_stride;
        u32 v = 0;
        if (a.exact_spread) {     // (on0 / on1 are then the spread offsets with the orientation in bits 29 .. 31)
            if (lane < F) v = (u32)((a.resp_tab[nb[(on0 & 0x1FFFFFFFu) + (u32)j]] >> (8u * (on0 >> 29))) & 0xFFu);
            if (lane + 64 < F) v += (u32)((a.resp_tab[nb[(on1 & 0x1FFFFFFFu) + (u32)j]] >> (8u * (on1 >> 29))) & 0xFFu);
        } else {
        if (lane < F) { const u32 ad = on0 + (u32)j; const u32 by = nb[ad >> 1]; v = (ad & 1u) ? (by >> 4) : (by & 15u); }
        if (lane + 64 < F) { const u32 ad = on1 + (u32)j; const u32 by = nb[ad >> 1]; v += (ad & 1u) ? (by >> 4) : (by & 15u); }
        }
        const int raw = (int)wave_sum_u32(v);
        n_surv += 1;
        if (raw > thr && lane == 0) {
            LmDevHeader* hdr = reinterpret_cast<LmDevHeader*>(reinterpret_cast<u8*>(a.hdr) + (size_t)sl * a.aux_slot_stride);
            LmCand* cand = reinterpret_cast<LmCand*>(reinterpret_cast<u8*>(a.cand) + (size_t)sl * a.aux_slot_stride);
            const u32 pos = atomicAdd(&hdr->cand_count, 1u);
            if (pos < a.cand_cap) {
                const int r = j / a.W, cc = j - r * a.W;
                LmCand cd;
                cd.ti = ti;
                cd.x = cc * a.T + offset;
                cd.y = r * a.T + offset;
                cd.sim = __fadd_rn(__fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * n)), 0.5f);
                cand[pos] = cd;
            }
        }
        hl = __ballot((h[0] | h[1] | h[2] | h[3]) != 0u);
    }
    if (a.stat && lane == 0) atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 3], (unsigned long long)n_surv);
}

// Second half of the bit-plane scan: the exact sums of the queued survivors, one lane each (neighbours in the queue are neighbouring
// positions of one template: their nibble loads share lines).  Sums above the threshold become candidates exactly as k_scan4 emits them.
__global__ __launch_bounds__(256) void k_scan1_exact(LmScanArgs a) {
    __shared__ u64 tabs[256];                                          // the response table (exact_spread)
    if (a.exact_spread) { tabs[threadIdx.x] = a.resp_tab[threadIdx.x]; __syncthreads(); }
    const u32 cap8 = a.surv_cap >> 3, qx = blockIdx.x & 7u;           // workgroup b runs on XCD b % 8: queue b % 8
    const unsigned long long total = a.surv[8 * a.surv_set + qx];
    const u32 n = total < (unsigned long long)cap8 ? (u32)total : cap8;
    const unsigned long long* queue = a.surv + 16 + (size_t)qx * cap8;
    if (blockIdx.x < 8 && threadIdx.x == 0) a.surv[8 * (a.surv_set ^ 1) + blockIdx.x] = 0;      // the other counter set, for the stream's next launch (no memset between the launches)
    const int offset = a.T / 2 + (a.T % 2 - 1);
    for (u32 i = (blockIdx.x >> 3) * 256u + threadIdx.x; i < n; i += (gridDim.x >> 3) * 256u) {
        const unsigned long long e = queue[i];
        const u32 ti = (u32)(e >> 32), sl = ((u32)e) >> 20, j = (u32)e & 0xFFFFFu;
        const int cnt = a.scan_n[ti];
        const int nn = cnt & 0xFF;
        const int F = ((cnt >> 8) & 0xFF) + ((cnt >> 16) & 0xFF);
        const int thr = a.raw_thr_by_n[nn];
        int raw_out = 0;
        const u32* offn = a.offn + (size_t)ti * a.fpad1;
        const u8* nb = a.lm + (size_t)sl * a.lm_slot_stride;
        if (a.exact_spread) {
            // the level keeps ONE byte per position (the spread byte): response = table[spread byte], byte = the feature's orientation
            const u32* offs = a.offs3 + (size_t)ti * a.fpad1;
            int raws = 0;
            for (int f = 0; f < F; f += 8) {
                u32 sv[8], of[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { of[k] = offs[f + k]; sv[k] = nb[(of[k] & 0x1FFFFFFFu) + j]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) raws += (int)((tabs[sv[k]] >> (8u * (of[k] >> 29))) & 0xFFu);
            }
            raw_out = raws;
        } else {
        // batches of eight features: the lists are padded to a multiple of eight with offsets of the arena's zero block (response 0), so there is
        // no tail of single, dependent loads; the next batch's offsets are requested before this batch's responses
        int raw = 0;
        u32 on[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) on[k] = offn[k];
        for (int f = 0; f < F; f += 8) {
            u32 by[8], ad[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { ad[k] = on[k] + j; by[k] = nb[ad[k] >> 1]; }
            if (f + 8 < F) {
#pragma unroll
                for (int k = 0; k < 8; ++k) on[k] = offn[f + 8 + k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) raw += (int)((ad[k] & 1u) ? (by[k] >> 4) : (by[k] & 15u));
        }
        raw_out = raw;
        }
        const int raw = raw_out;
        if (raw > thr) {
            LmDevHeader* hdr = reinterpret_cast<LmDevHeader*>(reinterpret_cast<u8*>(a.hdr) + (size_t)sl * a.aux_slot_stride);
            LmCand* cand = reinterpret_cast<LmCand*>(reinterpret_cast<u8*>(a.cand) + (size_t)sl * a.aux_slot_stride);
            const u32 pos = atomicAdd(&hdr->cand_count, 1u);
            if (pos < a.cand_cap) {
                const int r = (int)j / a.W, cc = (int)j - r * a.W;
                LmCand cd;
                cd.ti = ti;
                cd.x = cc * a.T + offset;
                cd.y = r * a.T + offset;
                cd.sim = __fadd_rn(__fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * nn)), 0.5f);
                cand[pos] = cd;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// a11-a13, the bit-plane scan with a frame's planes in LDS (r06; LmScanArgs::lds_form).  k_scan1 on 640 x 480 frames is bound by the
// L2 -> L1 line rate, not by vector issue: the 8-9 lanes of a frame read 128-144 contiguous bytes of a plane at an arbitrary dword
// offset, i.e. TWO 128-byte lines for every frame and feature (0.93 of the 34.5 TB/s by the request count).  But ALL the miss planes of
// such a frame -- 8 orientations x 76 800 bits x 2 modalities = 153 600 bytes -- fit the 160 KB of LDS of one CU.  So:
//   * a workgroup of 1024 threads owns (frame, share r of R of the templates): it copies the frame's planes into LDS once (16-byte pieces,
//     the arena's pads dropped) and scans its templates from there -- no L2 -> L1 traffic at all in the loop but the feature lists;
//   * a LANE is one lane item = (template, unit of 128 positions), the items template-major, 64 consecutive items per wave: 8 templates x 8
//     units for 995 positions.  The feature's LDS address and bit shift are per lane (one table entry per feature and template, loaded
//     eight at a time: addr << 8 | shift, the unit's 16 bytes added; v_alignbit_b32 takes the entry itself as its shift operand); the five
//     dwords come by two ds_read2_b32 + one ds_read_b32 (measured, tools/microbench/lds_unaligned.hip: a dword-aligned ds_read_b128 costs 64
//     cycles per wave, this form 23) -- the fifth dword too, so a lane needs nothing from its neighbour and a chunk has no 31-position tail;
//   * counters, flag bit, per-lane pruning and the wave's stop as in k_scan1 (the bound is per lane now: the templates of a wave may differ
//     in their feature counts; the lists are padded with entries of a zero block = "no miss");
//   * second stage in the SAME launch: the survivors go to a queue in LDS (template << 15 | position); when all waves are done the
//     workgroup replaces the planes by the frame's SPREAD bytes (the same number of bytes: one per position and modality, written by
//     d_lm_fast's spread_low form) and the response table, and takes the survivors' exact sums from LDS -- 62 byte gathers per survivor cost
//     a few LDS cycles each instead of a 128-byte line lookup in the vector L1 (k_scan1_exact: 49 us per 96-frame launch of config 2).  A
//     wave whose survivors do not fit the queue takes their sums itself from the arena (k_scan1's fallback).
// The candidate lists are k_scan4's, record for record.  a.stat as k_scan1.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scanl_emit(const LmScanArgs& a, u32 sl, u32 ti, int j, int raw, int n) {
    LmDevHeader* hdr = reinterpret_cast<LmDevHeader*>(reinterpret_cast<u8*>(a.hdr) + (size_t)sl * a.aux_slot_stride);
    LmCand* cand = reinterpret_cast<LmCand*>(reinterpret_cast<u8*>(a.cand) + (size_t)sl * a.aux_slot_stride);
    const u32 pos = atomicAdd(&hdr->cand_count, 1u);
    if (pos < a.cand_cap) {
        const int offset = a.T / 2 + (a.T % 2 - 1);
        const int r = j / a.W, cc = j - r * a.W;
        LmCand cd;
        cd.ti = ti;
        cd.x = cc * a.T + offset;
        cd.y = r * a.T + offset;
        cd.sim = __fadd_rn(__fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * n)), 0.5f);
        cand[pos] = cd;
    }
}

// One lane item's rounds (k_scanl): counters of TOP + 1 bits, their flag = bit TOP, preset to 2^TOP - 1 - (misses allowed).  Returns true when the whole wave
// stopped early; flags[q] = the flag dword of positions 32 q .. 32 q + 31 otherwise; f = features counted.
typedef __attribute__((address_space(3))) u32 lds_u32_t;
template <int TOP>
__device__ __forceinline__ bool scanl_rounds(const __amdgpu_buffer_rsrc_t rs_off, u32 voff, u32 unit_add, u32 keep5, u32 pre, int valid, int Fw,
                                             int first_test, unsigned long long& alive, u32 (&flags)[4], int& f_out) {
    u32 c[4][8];
#pragma unroll
    for (int b = 0; b < TOP; ++b) {
        const u32 bit = (u32)__builtin_amdgcn_sbfe((int)pre, (u32)b, 1u);   // 0 / ~0
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q][b] = bit;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vq = valid - 32 * q;
        c[q][TOP] = vq >= 32 ? 0u : (vq > 0 ? ~((1u << vq) - 1u) : 0xFFFFFFFFu);      // invalid positions start dead
    }
    // EVERY lane reads and counts in every round, dead or not: the kernel is bound by vector issue, not by LDS cycles (measured: dropping a fifth of
    // the LDS accesses changed nothing), and exec-masked rounds cost the compiler 43 moves + 20 selects per round to merge a skipped round's
    // counters with a counted one's.  A dead lane's flags stay all ones (the flag bit is sticky), a lane without an item reads template 0's planes.
    bool pruned = false;
    int f = 0;
    // (the list entries of the NEXT round are requested before this round's LDS reads: a round's global round trip hides behind the round before it)
    u32x4 n0 = __builtin_amdgcn_raw_buffer_load_b128(rs_off, voff, 0u, 0), n1 = __builtin_amdgcn_raw_buffer_load_b128(rs_off, voff, 16u, 0);
    for (; f < Fw && !pruned; f += 8) {
        {
            const u32x4 e0 = n0, e1 = n1;
            // (past the last round: the same entries once more -- a branch here would bring the merges back)
            const u32 nf = (u32)(f + 8 < Fw ? f + 8 : f) * 4u;
            n0 = __builtin_amdgcn_raw_buffer_load_b128(rs_off, voff, nf, 0);
            n1 = __builtin_amdgcn_raw_buffer_load_b128(rs_off, voff, nf + 16u, 0);
            const u32 e[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
            u32 v[8][5];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // (an LDS address as an integer: unit_add carries the image's LDS base, so the entry's sum IS the address -- through the generic pointer
                // the compiler added the base, a literal 0, once more per feature)
                const lds_u32_t* p = (const lds_u32_t*)(unsigned long)((e[k] + unit_add) >> 8);
                v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2]; v[k][3] = p[3];
            }
            // the fifth dword is the next lane's first (the next unit of the same template) -- a DPP move instead of a fifth LDS access --
            // except in a template's LAST unit, whose neighbour belongs to another template: there it counts as "no miss"
            // for every feature.  That only weakens the bound of the unit's last positions (a few more survivors; the second stage decides).
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k][4] = next_lane(v[k][0]) & keep5;       // (an AND, not a select: the DPP move must run on the last units' lanes too -- they are its sources)
            u32 x[4][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q][k] = __builtin_amdgcn_alignbit(v[k][q + 1], v[k][q], e[k]);   // (the instruction takes bits 4..0 of the entry)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) bs_level<0, 8, TOP>(c[q], x[q]);
        }
        if (f + 8 > first_test && f + 8 < Fw) {
            const unsigned long long left = __ballot((c[0][TOP] & c[1][TOP] & c[2][TOP] & c[3][TOP]) != 0xFFFFFFFFu) & alive;
            if (!left) pruned = true;
            else alive = left;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) flags[q] = c[q][TOP];
    f_out = f;
    return pruned;
}

__device__ __forceinline__ u32 tabs_at(const u8* lds, u32 img, u32 sv, u32 ori) { return lds[img + sv * 8u + ori]; }
__global__ __launch_bounds__(1024) void k_scanl(LmScanArgs a) {
    extern __shared__ u32x4 scanl_lds[];
    u8* lds = reinterpret_cast<u8*>(scanl_lds);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 slot, r;
    {
        const u32 R = (u32)a.R, B = (u32)a.nslots, b = blockIdx.x;
        if ((B & 7u) == 0) { const u32 x = b & 7u, k = b >> 3; slot = x + 8u * (k / R); r = k - (k / R) * R; }     // a frame's workgroups on ONE XCD: its planes come from HBM once
        else { slot = b / R; r = b - slot * R; }
    }
    if (slot >= (u32)a.nslots) return;
    const u8* arena = a.lm + (size_t)slot * a.lm_slot_stride;
    const u32 IMG = (u32)a.M * 8u * a.pb;                    // bytes of the planes = of the spread bytes
    // LDS behind the image: [tbl_bytes: zeros, later the response table][16: queue header][512: raw thresholds by feature count][queue]
    u32* qcount = reinterpret_cast<u32*>(lds + IMG + a.tbl_bytes);
    const int* thr_tab = reinterpret_cast<const int*>(qcount + 4);
    u32* queue = qcount + 4 + 128;
    const u32 n_w = ((u32)a.n_litems + 63u) >> 6;
    // share r owns the wave items r + R j.  A wave takes j = wave, then whatever comes next from a counter in LDS, ONE item ahead (its 16-byte record --
    // item, feature counts, positions -- is requested as soon as the item is taken and arrives while the current item is counted): the waves of a workgroup
    // finish within one item of each other (measured: static strides left a wave waiting 18 % of the workgroup's time at the barrier before the second
    // stage, taking items two ahead 21 %).  The first item is requested before the planes are copied.
    const unsigned long long tm0 = __builtin_readcyclecounter();
    const u32 n_share = n_w > r ? (n_w - r + (u32)a.R - 1u) / (u32)a.R : 0u;
    const u32x4 no_item = {0xFFFFFFFFu, 0u, 0u, 0u};
    auto item_at = [&](u32 j) -> u32x4 {
        const u32 idx = (r + (u32)a.R * j) * 64u + (u32)lane;
        return (j < n_share && idx < (u32)a.n_litems) ? reinterpret_cast<const u32x4*>(a.litem)[(size_t)a.litem_lo + idx] : no_item;
    };
    u32 j_cur = (u32)wave;
    u32x4 rec_cur = item_at(j_cur);
    // ---- the frame's planes -> LDS: wave w copies plane w % (8 M) (16-byte pieces, the loads of a whole pass in flight before the stores)
    {
        const u32 per = a.pb >> 4, np = (u32)a.M * 8u, wpp = 16u / np;       // pieces per plane; planes; waves per plane (M = 1: 2, M = 2: 1)
        const u32 pl = (u32)wave % np, part = (u32)wave / np, m = pl >> 3, o = pl & 7u;
        const u8* src = arena + (size_t)m * a.mod_stride + a.planes_off + (size_t)o * a.plane_ori;
        u32x4* dst = scanl_lds + (size_t)pl * per;
        const u32 stride = 64u * wpp;
        if (part < wpp && a.dbg != 3) {
            // (a frame's workgroups run on one XCD at the same time: each starts its copy at another place, so that they do not queue at one L2 channel)
            const u32 rot = (r * per) / (u32)a.R;
            for (u32 k0 = (u32)lane + 64u * part; k0 < per; k0 += 8u * stride) {
                u32x4 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const u32 k = k0 + (u32)q * stride, kk = k + rot < per ? k + rot : k + rot - per; t[q] = k < per ? *reinterpret_cast<const u32x4*>(src + 16u * kk) : u32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { const u32 k = k0 + (u32)q * stride, kk = k + rot < per ? k + rot : k + rot - per; if (k < per) dst[kk] = t[q]; }
            }
        }
        for (u32 i = (u32)tid; i < (a.tbl_bytes >> 2); i += 1024u) reinterpret_cast<u32*>(lds + IMG)[i] = 0u;     // the zero block of the padded list entries
        if (tid < 128) const_cast<int*>(thr_tab)[tid] = a.raw_thr_by_n[tid];
        if (tid == 0) { qcount[0] = 0u; qcount[1] = 16u; }                   // (queue length; next item of the share that no wave has taken)
    }
    __syncthreads();
    const unsigned long long tm1 = __builtin_readcyclecounter();
    const __amdgpu_buffer_rsrc_t rs_off = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(a.offl), 0, 0x7FFFFFFF, 0x00020000);
    const u32 lds_base = (u32)(unsigned long)(__attribute__((address_space(3))) u32x4*)scanl_lds;      // LDS address of the image (0 unless the kernel gets static LDS one day)
    unsigned long long st_f = 0, st_F = 0, st_l = 0, st_s = 0;
    for (; j_cur < n_share; ) {
        // this item's record is in registers; the next item is taken now and its record arrives during the loop
        const u32 it = rec_cur[0];
        const int cnt = (int)rec_cur[1], P = (int)rec_cur[2];
        j_cur = (u32)__builtin_amdgcn_readfirstlane((int)(lane == 0 ? atomicAdd(qcount + 1, 1u) : 0u));
        rec_cur = item_at(j_cur);
        const bool has = it != 0xFFFFFFFFu;
        const u32 ti = has ? it >> 8 : 0u, unit = has ? it & 255u : 0u;
        const int n = cnt & 0xFF;
        const int F = ((cnt >> 8) & 0xFF) + ((cnt >> 16) & 0xFF);
        const int thr = thr_tab[n & 127];
        const int K0 = 4 * F - thr - 1;                              // what the misses may cost in total
        int mmax = K0 >= 0 ? (int)(((u32)K0 * a.delta_rcp16) >> 16) : 0;
        if (mmax > 127) mmax = 127;
        const u32 pre = (u32)(127 - mmax);
        const int j0 = (int)unit * 128;
        int valid = (has && K0 >= 0) ? P - j0 : 0;
        valid = valid < 0 ? 0 : (valid > 128 ? 128 : valid);
        unsigned long long alive = __ballot(valid > 0);
        if (!alive) continue;
        // (wave-uniform, and told so: the round loop's bounds and the lists' scalar offsets hang on them)
        const int Fw = __builtin_amdgcn_readfirstlane((int)wave_max_u32(valid > 0 ? (u32)F : 0u));
        const int mm_hi = __builtin_amdgcn_readfirstlane((int)wave_max_u32(valid > 0 ? (u32)mmax : 0u));
        const int first_test = 127 - __builtin_amdgcn_readfirstlane((int)wave_max_u32(valid > 0 ? pre : 0u));            // the smallest miss budget of the wave: nothing dies before
        const u32 voff = ti * (u32)a.fpad1 * 4u;
        const u32 unit_add = (unit * 16u + lds_base) << 8;
        u32 keep5 = (j0 + 128 >= P || lane == 63) ? 0u : 0xFFFFFFFFu;
        asm volatile("" : "+v"(keep5));            // (a plain register to the optimiser: `& keep5` stays an AND, which takes the DPP move as its operand -- v_and_b32_dpp -- instead of a select behind a v_mov_b32_dpp)
        u32 flags[4];
        bool pruned;
        int f;
        // counters of 6 bits (flag = bit 5) when no template of the wave may miss more than 31 features -- the usual case: 24 at threshold 80 with 62
        // features -- two half adders fewer per round and dword; 8 bits otherwise
        if (mm_hi <= 31) pruned = scanl_rounds<5>(rs_off, voff, unit_add, keep5, (u32)(31 - mmax), valid, Fw, first_test, alive, flags, f);
        else pruned = scanl_rounds<7>(rs_off, voff, unit_add, keep5, pre, valid, Fw, first_test, alive, flags, f);
        st_l += (unsigned long long)f * 64ull;                                   // (every lane reads in every round)
        st_f += (unsigned long long)(f < Fw ? f : Fw); st_F += (unsigned long long)Fw;
        // survivors: positions never flagged (a dead or idle lane's flags are all ones)
        u32 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = (pruned || !((alive >> lane) & 1ull)) ? 0u : ~flags[q];
        const u32 nh = (u32)(__popc(h[0]) + __popc(h[1]) + __popc(h[2]) + __popc(h[3]));
        if (nh && !a.no_exact) {
            // The counter only grows: a reservation that does not fit (entirely) fills the queue's last entries with as many of its survivors as fit and
            // leaves the rest to the wave's own sums below -- every entry below min(counter, capacity) is written.  (r06, first form: a reservation that
            // did not fit was given BACK by an atomic subtract; another wave's later subtract could then pull the counter below entries a third wave had
            // written in between -- one candidate of 4 M lost, on the one frame whose workgroups fill their queues: tools/stress_batch_parity.py.)
            const u32 at = atomicAdd(qcount, nh);
            u32 fit = at >= a.queue_cap ? 0u : min(nh, a.queue_cap - at);
            st_s += fit;
            u32* q = queue + at;
            const u32 hi = ti << LM_SCANL_POS_BITS;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (; h[k] && fit; --fit) { const int b = __ffs((int)h[k]) - 1; *q++ = hi | (u32)(j0 + 32 * k + b); h[k] &= h[k] - 1u; }
        }
        unsigned long long hl = a.no_exact ? 0ull : __ballot((h[0] | h[1] | h[2] | h[3]) != 0u);
        while (hl) {
            // queue full: the exact sum of one survivor at a time from the arena's spread bytes, lane k adds features k and k + 64 (k_scan1's fallback)
            const int src = __ffsll((long long)hl) - 1;
            const u32 w0 = (u32)__builtin_amdgcn_readlane((int)h[0], src), w1 = (u32)__builtin_amdgcn_readlane((int)h[1], src);
            const u32 w2 = (u32)__builtin_amdgcn_readlane((int)h[2], src), w3 = (u32)__builtin_amdgcn_readlane((int)h[3], src);
            const int q = w0 ? 0 : (w1 ? 1 : (w2 ? 2 : 3));
            const u32 wq = w0 ? w0 : (w1 ? w1 : (w2 ? w2 : w3));
            const int b = __ffs((int)wq) - 1;
            const int j = __builtin_amdgcn_readlane(j0, src) + 32 * q + b;
            const u32 sti = (u32)__builtin_amdgcn_readlane((int)ti, src);
            const int sF = __builtin_amdgcn_readlane(F, src), sn = __builtin_amdgcn_readlane(n, src), sthr = __builtin_amdgcn_readlane(thr, src);
            if (lane == src) {
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = (k == q) ? (h[k] & ~(1u << b)) : h[k];
            }
            const u32* o3 = a.offs3 + (size_t)sti * a.fpad1;
            u32 vv = 0;
            if (lane < sF) { const u32 on = o3[lane]; vv = (u32)((a.resp_tab[arena[(on & 0x1FFFFFFFu) + (u32)j]] >> (8u * (on >> 29))) & 0xFFu); }
            if (lane + 64 < sF) { const u32 on = o3[lane + 64]; vv += (u32)((a.resp_tab[arena[(on & 0x1FFFFFFFu) + (u32)j]] >> (8u * (on >> 29))) & 0xFFu); }
            const int raw = (int)wave_sum_u32(vv);
            st_s += (lane == 0) ? 1u : 0u;
            if (raw > sthr && lane == 0) scanl_emit(a, slot, sti, j, raw, sn);
            hl = __ballot((h[0] | h[1] | h[2] | h[3]) != 0u);
        }
    }
    const unsigned long long tm2 = __builtin_readcyclecounter();
    if (a.stat && a.dbg != 7) {
        // per-lane partial counts of the survivors, per-wave counts of the rest (lane 0 holds them)
        unsigned long long sv = st_s;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) sv += (unsigned long long)__shfl_xor((long long)sv, sft, 64);
        if (lane == 0) {
            atomicAdd(&a.stat[4 * (blockIdx.x & 1023u)], st_f);
            atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 1], st_F);
            atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 2], st_l);
            atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 3], sv);
        }
    }
    if (a.no_exact) return;
    // ---- second stage: the frame's spread bytes take the planes' place, the survivors' exact sums come from LDS
    __syncthreads();
    const unsigned long long tm3 = __builtin_readcyclecounter();
    const u32 qn = min(*qcount, a.queue_cap);
    if (qn == 0) return;                                          // (workgroup-uniform)
    // a LANE takes a survivor.  All 64 entries of its template's list are requested at once (sixteen 16-byte loads in flight) and BEFORE the spread
    // bytes are copied: the one global round trip of the stage hides behind the copy; the 2 x 62 byte gathers per survivor are LDS accesses.  (A wave per
    // survivor -- lane k takes feature k, a DPP sum -- measured 30 % slower: 20 dependent chains of LDS reads and cross-lane sums per wave.)
    const u32 F_MASK = 0x1FFFFFFFu;
    u32x4 ent[16];
    u32 e_cur = 0xFFFFFFFFu;
    int cn_cur = 0;
    auto request = [&](u32 i, int half) {
        e_cur = i < qn ? queue[i] : 0xFFFFFFFFu;
        const u32 ti = e_cur >> LM_SCANL_POS_BITS;
        if (e_cur != 0xFFFFFFFFu) {
            cn_cur = a.scan_n[ti];
            const u32x4* of = reinterpret_cast<const u32x4*>(a.offsl + (size_t)ti * a.fpad1) + 16 * half;
#pragma unroll
            for (int q = 0; q < 16; ++q) ent[q] = (64 * half + 4 * q < a.fpad1) ? of[q] : u32x4{0, 0, 0, 0};
        }
    };
    auto partial = [&](int F, u32 j, int half) -> int {
        int raw = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const u32 o = ent[q][c4];
                if (64 * half + 4 * q + c4 < F) raw += (int)tabs_at(lds, IMG, (u32)lds[(o & F_MASK) + j], o >> 29);
            }
        }
        return raw;
    };
    u32 i0 = (u32)tid;
    request(i0, 0);
    {
        const u32 per = (a.pb * 8u) >> 4;                          // 16-byte pieces of a modality's spread bytes (T*T*wh, at the start of its block)
        for (int m = 0; m < (a.dbg == 2 ? 0 : a.M); ++m) {
            const u8* src = arena + (size_t)m * a.mod_stride;
            u32x4* dst = scanl_lds + (size_t)m * per;
            const u32 rot = (r * per) / (u32)a.R;
            for (u32 k0 = (u32)tid; k0 < per; k0 += 4u * 1024u) {
                u32x4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const u32 k = k0 + (u32)q * 1024u, kk = k + rot < per ? k + rot : k + rot - per; t[q] = k < per ? *reinterpret_cast<const u32x4*>(src + 16u * kk) : u32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { const u32 k = k0 + (u32)q * 1024u, kk = k + rot < per ? k + rot : k + rot - per; if (k < per) dst[kk] = t[q]; }
            }
        }
        if (tid < 256) reinterpret_cast<u64*>(lds + IMG)[tid] = a.resp_tab[tid];
    }
    __syncthreads();
    const unsigned long long tm4 = __builtin_readcyclecounter();
    if (a.dbg == 1) return;
    for (; i0 < qn; ) {
        const u32 e = e_cur;
        const u32 ti = e >> LM_SCANL_POS_BITS, j = e & ((1u << LM_SCANL_POS_BITS) - 1u);
        const int n = cn_cur & 0xFF;
        const int F = ((cn_cur >> 8) & 0xFF) + ((cn_cur >> 16) & 0xFF);
        int raw = partial(F, j, 0);
        if (F > 64) { request(i0, 1); raw += partial(F, j, 1); }       // (more than 64 in-bounds features: the second half of the list)
        if (raw > thr_tab[n & 127]) scanl_emit(a, slot, ti, (int)j, raw, n);
        i0 += 1024u;
        if (i0 < qn) request(i0, 0);
    }
    if (a.stat && a.dbg == 7 && lane == 0) {
        // timing experiment: reference-clock ticks of the phases, per wave: planes copy, counting, wait for the other waves, spread copy, exact sums
        const unsigned long long tm5 = __builtin_readcyclecounter();
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u)], tm1 - tm0);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 1], tm2 - tm1);
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 2], (tm3 - tm2) + ((tm4 - tm3) << 32));
        atomicAdd(&a.stat[4 * (blockIdx.x & 1023u) + 3], tm5 - tm4);
    }
}


}  // namespace

// ================================================================================================
// launchers
// ================================================================================================
void lmk_scan(hipStream_t s, const LmScanArgs& a_in, int variant, int nslots) {
    if (a_in.n_items <= 0) return;
    LmScanArgs a = a_in;
    a.nslots = nslots;
    const int G = (a.n_items + 3) / 4;               // one wave per work item
    a.wgs_per_slot = G;
    if (a.lds_form) {
        // k_scanl: one 1024-thread workgroup = (frame, share of the templates), the frame's planes in ALL of the CU's LDS
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_scanl), hipFuncAttributeMaxDynamicSharedMemorySize, LM_SCANL_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); }
            raised = true;
        }
        a.no_exact = (variant & 128) ? 1 : 0;
        a.dbg = (variant >> 9) & 7;
        hipLaunchKernelGGL(k_scanl, dim3((unsigned)(nslots * a.R), 1, 1), dim3(1024), LM_SCANL_LDS_BYTES, s, a);
        return;
    }
    if (a.L1) {
        // k_scan1: a wave scans its item for a GROUP of G1 slots (XCD affinity per group when the group count allows)
        const int ngroups = (nslots + a.G1 - 1) / a.G1;
        a.no_exact = (variant & 128) ? 1 : 0;
        // (the two measurement variants do not run k_scan1_exact, which re-arms the other counter set for the stream's next launch: re-arm both here)
        if (a.surv && (a.no_exact || (variant & 256))) (void)hipMemsetAsync(a.surv, 0, 16 * sizeof(unsigned long long), s);
        if (variant & 256) a.surv = nullptr;             // A/B: the waves take their survivors' exact sums themselves
        hipLaunchKernelGGL(k_scan1, dim3((unsigned)(G * ngroups), 1, 1), dim3(256), 0, s, a);
        if (a.surv && !a.no_exact) hipLaunchKernelGGL(k_scan1_exact, dim3(1024), dim3(256), 0, s, a);
        return;
    }
    if (a.nibble) {
        // k_scan4: a wave scans its item for a PAIR of slots; 1-D grid with XCD affinity per pair
        const int npairs = (nslots + 1) / 2;
        dim3 grid((unsigned)(G * npairs), 1, 1);
#define SCAN4_LAUNCH(FB)                                                                              \
    do { if (variant & 8) hipLaunchKernelGGL((k_scan4<FB, true, 0>), grid, dim3(256), 0, s, a);       \
         else if ((variant & 16) || (a.M < 2 && !(variant & 32))) hipLaunchKernelGGL((k_scan4<FB, true, 1>), grid, dim3(256), 0, s, a); \
         else hipLaunchKernelGGL((k_scan4<FB, true, 2>), grid, dim3(256), 0, s, a); } while (0)
        // variant bits 0-1: features per load block (0: 6, 1: 12, 2: 3); bit 3: no pruning (the plain exhaustive scan);
        // bit 4: wave-level pruning only (r02's rule); bit 5: per-lane pruning whatever the modality count.  Default:
        // per-lane pruning for two modalities (r03, config 2: 181.6 -> 172.9 us per 96-frame launch, 35 % of the lane-loads
        // instead of 50 %), wave-level for one (config 3: the per-lane form measured 455 against 413 us per 128-frame launch:
        // with 31 features per template few lanes die long before their wave does, and the masked loads still pull the
        // same lines)
        const int fb = variant & 3;
        if ((variant & 64) && (variant & 8)) {   // measurement only (wrong sums): exhaustive scan without the shift-undo
            hipLaunchKernelGGL((k_scan4<6, true, 0, true>), grid, dim3(256), 0, s, a);
            return;
        }
        if (fb == 1) SCAN4_LAUNCH(12); else if (fb == 2) SCAN4_LAUNCH(3); else SCAN4_LAUNCH(6);
#undef SCAN4_LAUNCH
        return;
    }
    // variant bits 0-1: feature-loop unroll (0: 8 loads in flight, 1: 4, 2: 2); bit 2: plain (slot = grid.z)
    // mapping instead of the XCD-aware one
    const bool xcd = !(variant & 4) && (nslots == 1 || nslots == 2 || nslots == 4 || (nslots % 8) == 0);
    dim3 grid = xcd ? dim3(((nslots % 8) == 0) ? (unsigned)(G * nslots) : 8u * (unsigned)((G + 8 / nslots - 1) / (8 / nslots)))
                    : dim3(G, 1, nslots);
    const int u = variant & 3;
#define SCAN_LAUNCH(U)                                                                      \
    do { if (xcd) hipLaunchKernelGGL((k_scan<U, true>), grid, dim3(256), 0, s, a);  \
         else hipLaunchKernelGGL((k_scan<U, false>), grid, dim3(256), 0, s, a); } while (0)
    if (u == 1) SCAN_LAUNCH(4); else if (u == 2) SCAN_LAUNCH(2); else SCAN_LAUNCH(8);
#undef SCAN_LAUNCH
}
