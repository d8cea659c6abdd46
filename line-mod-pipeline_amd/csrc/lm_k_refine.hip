// lm_k_refine.hip -- a14-a15 of the LINE-MOD match path for gfx950 (CDNA4, wave64): k_refine_plan + k_refine (similarityLocal over the 16 x 16
// patch, first-max argmax, rescore, threshold filter), k_emit_unrefined, k_sort_unique + k_merge_unique (rank / bitonic sort in LDS +
// adjacent-unique, total order of SURVEY.md A.9), k_pack_lists (a lane's sorted lists packed for the all-gather, 8e), and their launchers.
#include "lm_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a14  One wave per candidate: lane l holds the 4 patch positions (row l/4, cols 4(l%4)..+3) of the
// 16x16 patch.  Refinement levels keep SPREAD linear memories (1 byte per position instead of 8
// response bytes: the whole level stays L2-resident and a patch load touches 1/8 of the lines); the
// response max(LUT_lo[v & 15], LUT_hi[v >> 4]) comes from an 8 x 256-byte table in LDS (one ds_read_u8
// per position; upstream does the two 16-entry nibble lookups with pshufb).
// The modality's feature records are loaded one per lane, bounds-checked in parallel (features
// shifted out of the frame read the arena's zero block: spread 0 -> response 0), then broadcast
// with v_readlane so the patch loads (one unaligned dword per lane per feature) issue back to back.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void emit_key(const LmRefineArgs& a, LmDevHeader* hdr, u64* keys, u32 ti, int x, int y,
                                         float sim) {
    u32 slot = atomicAdd(&hdr->match_count, 1u);
    if (slot < a.match_cap) {
        u32 sb = __float_as_uint(sim);
        u64 hi = ((u64)(~sb) << 32) | (u32)a.t_global[ti];
        u64 lo = ((u64)(u32)a.t_class[ti] << 48) | ((u64)((u32)(y + 0x800000) & 0xFFFFFFu) << 24) |
                 (u64)((u32)(x + 0x800000) & 0xFFFFFFu);
        keys[2 * (size_t)slot] = hi;
        keys[2 * (size_t)slot + 1] = lo;
    }
}

// Slot -> XCD plan for k_refine.  A slot's candidates should stay on ONE XCD (its spread memories live in that L2), but the
// candidate counts differ a lot between frames (30 .. 2700), and the launch lasts as long as its busiest XCD
// (fixed round-robin: max / mean = 1.35 on the bench workload).  One workgroup ranks the slots by candidate
// count (rank sort in LDS) and deals them, heaviest first, to the least loaded of the eight XCD lists.
// r04: the unit that is dealt is a PIECE of a slot's list.  A slot whose list is longer than 1 / 24 of all candidates of the
// launch is cut into pieces of that size (even, so that list neighbours stay pairs), which go to different XCDs: with few
// frames per launch (config 5: eight, one of them holding 62 % of the candidates) the heaviest slot no longer runs on 1 / 8 of
// the chip while the rest idles; with many frames (config 2: 96) at most the one or two heaviest slots are cut.  At most
// nslots + 24 pieces, cap = nslots / 8 + 8 per list.
// The dealing is sequential by nature; it runs on eight lanes of one wave, lane x holding list x's load
// and length, the least loaded list found by a three-step butterfly minimum over (load, list) keys -- about 6 us
// instead of the 30 us of round 1's single thread, which sat on the critical path of every lane-step.  (Dealing in snake
// order of the rank is fully parallel but balances the skewed counts worse: k_refine 254 instead of 215 us.)
// plan layout: [8][cap] slot numbers, [8] list lengths, [8][cap + 1] running sums of the piece lengths along every list,
// [8][cap] first list entry of every piece.
#define RP_MAXP 1056   // pieces: nslots (<= 1016) + 24, rounded up
__global__ __launch_bounds__(1024) void k_refine_plan(const LmDevHeader* __restrict__ hdr0, size_t aux_slot_stride,
                                                       int nslots, u32 cand_cap, int cap, u32* __restrict__ plan) {
    __shared__ u32 cnt[RP_MAXP], pslot[RP_MAXP], poff[RP_MAXP];        // pieces: length, slot, first entry
    __shared__ u32 sorted_cnt[RP_MAXP], sorted_piece[RP_MAXP];
    __shared__ u32 pl[8 * 136 * 3 + 16 + 8];   // [8][cap] slots | [8] lengths | [8][cap + 1] running sums | [8][cap] first entries, cap <= 136
    __shared__ u32 wsum[16], wbase[17];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u32 c = 0;
    if (tid < nslots) {
        c = slot_ptr_s(hdr0, aux_slot_stride, (u32)tid)->cand_count;
        if (c > cand_cap) c = cand_cap;
    }
    // all candidates of the launch -> piece size
    u32 t = c;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += (u32)__shfl_xor((int)t, o, 64);
    if (lane == 0) wsum[wv] = t;
    __syncthreads();
    u32 total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += wsum[k];
    const u32 P = max(64u, ((total + 23u) / 24u + 1u) & ~1u);
    const u32 np = tid < nslots ? max(1u, (c + P - 1u) / P) : 0u;         // pieces of this slot (an empty list is one empty piece)
    // exclusive scan of np over the workgroup: where this slot's pieces go
    u32 inc = np;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if (lane >= o) inc += v; }
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (tid == 0) { u32 acc = 0; for (int k = 0; k < 16; ++k) { wbase[k] = acc; acc += wsum[k]; } wbase[16] = acc; }
    __syncthreads();
    const u32 NP = wbase[16];
    const u32 pbase = wbase[wv] + inc - np;
    for (u32 k = 0; k < np; ++k) { cnt[pbase + k] = min(P, c - k * P); pslot[pbase + k] = (u32)tid; poff[pbase + k] = k * P; }
    if (np == 1 && c == 0) cnt[pbase] = 0;
    __syncthreads();
    for (u32 e = (u32)tid; e < NP; e += 1024u) {
        const u32 ce = cnt[e];
        u32 rank = 0;
#pragma unroll 8
        for (u32 j = 0; j < NP; ++j) {
            const u32 cj = cnt[j];
            rank += (cj > ce || (cj == ce && j < e)) ? 1u : 0u;
        }
        sorted_cnt[rank] = ce; sorted_piece[rank] = e;
    }
    __syncthreads();
    if (tid < 64) {   // one wave; lanes 8.. mirror lanes 0..7 (x = lane & 7) so that the butterfly needs no masking
        const u32 x = (u32)tid & 7u;
        u32 load = 0, len = 0;
        // the ranked list in registers (lane l holds ranks l, l + 64, ...): an iteration reads its entry with
        // v_readlane instead of waiting for LDS, and the minimum goes through DPP, not through the LDS crossbar
        u32 rc[2], rs[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { rc[q] = sorted_cnt[(tid + 64 * q) % RP_MAXP]; rs[q] = sorted_piece[(tid + 64 * q) % RP_MAXP]; }
        for (u32 r = 0; r < NP; ++r) {
            u32 sc, ss;
            if (r < 128) {
                const int src = (int)(r & 63u);
                sc = (u32)__builtin_amdgcn_readlane((int)(r < 64 ? rc[0] : rc[1]), src);
                ss = (u32)__builtin_amdgcn_readlane((int)(r < 64 ? rs[0] : rs[1]), src);
            } else {
                sc = sorted_cnt[r]; ss = sorted_piece[r];
            }
            u32 key = (int)len < cap ? ((load << 3) | x) : 0xFFFFFFFFu;      // loads stay below 2^29
            key = min(key, (u32)__builtin_amdgcn_mov_dpp((int)key, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
            key = min(key, (u32)__builtin_amdgcn_mov_dpp((int)key, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
            key = min(key, (u32)__builtin_amdgcn_mov_dpp((int)key, 0x141, 0xf, 0xf, true));   // row_half_mirror: lane i <-> 7 - i
            if ((key & 7u) == x) {
                if (tid < 8) pl[x * cap + len] = ss;      // (the piece; turned into slot + first entry below)
                len += 1; load += sc + 8u;                                   // + a little per piece
            }
        }
        if (tid < 8) pl[8 * cap + x] = len;
    }
    __syncthreads();
    // running sums of the piece lengths along every XCD's list (k_refine's queue): [8][cap + 1] behind the lengths; then the
    // pieces' first entries, and the piece numbers replaced by their slots
    if (tid < 8) {
        u32 acc = 0;
        u32* pre = pl + 8 * cap + 8 + tid * (cap + 1);
        u32* off = pl + 8 * cap + 8 + 8 * (cap + 1) + tid * cap;
        const u32 n = pl[8 * cap + tid];
        for (u32 k = 0; k < n; ++k) {
            const u32 pc = pl[tid * cap + k];
            pre[k] = acc; acc += cnt[pc];
            off[k] = poff[pc];
            pl[tid * cap + k] = pslot[pc];
        }
        for (u32 k = n; k <= (u32)cap; ++k) pre[k] = acc;
        for (u32 k = n; k < (u32)cap; ++k) off[k] = 0;
    }
    __syncthreads();
    for (int i = tid; i < 8 * cap + 8 + 8 * (cap + 1) + 8 * cap; i += 1024) plan[i] = pl[i];
}

#define PRUNE_REFINE true
#ifndef PRUNE_REFINE_PAIR
#define PRUNE_REFINE_PAIR false   // refine_pair's own pruning: measured, costs more than it saves (see refine_pair)
#endif
// The inner loop of the refinement, per (feature, candidate): one patch load (4 positions per lane) and four table lookups.
// rf_patch: the dword-aligned 8-byte load + v_alignbyte that stands for a byte-misaligned dword load.  `se` is wave-uniform (a
//   feature's offset + the candidate's shift); with a row pitch W that is a multiple of 4 (W4: 640 / 5, 1280 / 2, ...) the lane's
//   own offset is one too, so the aligned base and the byte shift are SCALAR: one vector add per load instead of add + and + and
//   (v_alignbyte_b32 reads only bits 1:0 of its shift operand, so the general form passes the address itself).
// (Tried r03: the four lookups as ds_read_u8_d16 / _d16_hi pairs that pack (r0 | r1 << 16) in the load itself, -2 of 11 vector
// instructions per feature and candidate: same time, alone and beside other lanes -- the kernel waits for the L1's 16 cycles per
// patch load, not for the ALU -- and the compiler cannot see hand-issued LDS reads, so the lookups stay plain C.)
template <bool W4>
__device__ __forceinline__ u32 rf_patch(const u8* __restrict__ lm, u32 se, u32 lane_off) {
    if (W4) {
        const u32x2 d = ld8a4(lm + ((se & ~3u) + lane_off));
        return __builtin_amdgcn_alignbyte(d[1], d[0], se);
    }
    const u32 t = se + lane_off;
    const u32x2 d = ld8a4(lm + (t & ~3u));
    return __builtin_amdgcn_alignbyte(d[1], d[0], t);
}
// (r06: the four responses stay four registers and are added as they are -- v_add3_u32 takes two features' responses at once, so a position costs half an
// instruction per feature; packing them into u16 pairs first (r02-r05) cost two v_lshl_or_b32 per feature and candidate to save one add)
__device__ __forceinline__ void rf_lookup(const u8* __restrict__ tab, u32 v, u32 (&r)[4]) {
    r[0] = tab[v & 0xFFu]; r[1] = tab[(v >> 8) & 0xFFu]; r[2] = tab[(v >> 16) & 0xFFu]; r[3] = tab[v >> 24];
}
// One candidate of one slot: similarityLocal over the 16 x 16 patch, first-max argmax, rescore, threshold filter.
template <bool LAST, bool W4>
__device__ __forceinline__ void refine_one(const LmRefineArgs& a, u32 slot, u32 i, const u8 (*resp)[256], int lane) {
    LmDevHeader* hdr = slot_ptr_s(a.hdr, a.aux_slot_stride, slot);
    LmCand* cand = slot_ptr_s(a.cand, a.aux_slot_stride, slot);
    u64* keys = slot_ptr_s(a.keys, a.aux_slot_stride, slot);
    const u8* lm = a.lm + (size_t)slot * a.lm_slot_stride;
    const int T = a.g.T, W = a.g.W;
    const int border = 8 * T;
    const int offset = T / 2 + (T % 2 - 1);
    const u32 lane_off = (u32)((lane >> 2) * W + (lane & 3) * 4);
    LmCand c = cand[i];
    u32 ti = (u32)__builtin_amdgcn_readfirstlane((int)c.ti);
    if (ti == LM_DROPPED) return;
    int cx = __builtin_amdgcn_readfirstlane(c.x), cy = __builtin_amdgcn_readfirstlane(c.y);
    const LmRefMeta mt = a.meta[ti];
    int max_x = a.g.w - mt.width - border, max_y = a.g.h - mt.height - border;
    int x = cx * 2 + 1, y = cy * 2 + 1;
    x = x > border ? x : border; y = y > border ? y : border;
    x = x < max_x ? x : max_x;  y = y < max_y ? y : max_y;
    int bx = x / T - 8, by = y / T - 8;
    int off_x = bx * T, off_y = by * T;
    const u32 shift = (u32)(by * W + bx);   // two's complement: feature offset + shift >= 0 for kept features
    u32 s[4] = {0, 0, 0, 0};   // the sums of this lane's patch positions 0 .. 3 (<= 126 * 4)
    if (a.stat && lane == 0) atomicAdd(&a.stat[0], 1ull);
    // Exact pruning, as in the scan: the candidate survives only if its best patch position reaches `threshold`, and a
    // feature adds at most 4.  Every 16 features the wave takes the maximum partial sum of the patch; once even
    // (maximum + 4 x features to come) * 100 / (4 n) < threshold -- the very float expression of the final test, which
    // is monotone in the score -- the candidate is dropped without loading the rest.  Most candidates that chance
    // produced on the coarse level die here after a quarter of their features.
    int f_left = mt.nfeat_total;
    const float denom = (float)(4 * mt.nfeat_total);
    bool dead = false;
    for (int m = 0; m < a.M && !dead; ++m) {
        // (selects, not mt.count[m]: a runtime index into the struct copy would put it into scratch memory)
        const int cnt = (int)(m == 0 ? mt.count[0] : mt.count[1]);
        const u32 fstart = m == 0 ? mt.start[0] : mt.start[1];
        LmRefFeat ft;
        ft.off = 0; ft.x = 0; ft.y = 0;
        if (lane < cnt) ft = a.feats[fstart + lane];
        int fx = ft.x + off_x, fy = ft.y + off_y;
        bool ok = (lane < cnt) && fx >= 0 && fy >= 0 && fx < a.g.w && fy < a.g.h;
        const u32 eff = ok ? (ft.off & 0x1FFFFFFFu) + shift : a.g.zero_off;
        const u32 lab = ft.off >> 29;
        for (int f = 0; f < cnt; f += 8) {
            u32 v[8], q[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rf_patch<W4>(lm, (u32)__builtin_amdgcn_readlane((int)eff, f + k), lane_off);
#pragma unroll
            for (int k = 0; k < 8; ++k) rf_lookup(resp[(u32)__builtin_amdgcn_readlane((int)lab, f + k)], v[k], q[k]);
#pragma unroll
            for (int k = 0; k < 8; k += 2) { s[0] += q[k][0] + q[k + 1][0]; s[1] += q[k][1] + q[k + 1][1]; s[2] += q[k][2] + q[k + 1][2]; s[3] += q[k][3] + q[k + 1][3]; }
            f_left -= min(8, cnt - f);
            if (PRUNE_REFINE && (f & 8) && f_left > 0) {       // every second batch of eight
                const u32 best_now = wave_max_u32(max(max(s[0], s[1]), max(s[2], s[3])));
                const float reach = __fdiv_rn(__fmul_rn((float)(int)(best_now + 4u * (u32)f_left), 100.f), denom);
                if (reach < a.threshold) { dead = true; break; }
            }
        }
    }
    if (dead) {
        if (lane == 0) cand[i].ti = LM_DROPPED;
        if (a.stat && lane == 0) atomicAdd(&a.stat[4], 1ull);
        return;
    }
    // first maximum in row-major order: key = score << 8 | (255 - index)
    u32 idx0 = (u32)lane * 4u;
    u32 k0 = (s[0] << 8) | (255u - idx0);
    u32 k1 = (s[1] << 8) | (254u - idx0);
    u32 k2 = (s[2] << 8) | (253u - idx0);
    u32 k3 = (s[3] << 8) | (252u - idx0);
    u32 k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
    u32 key = wave_max_u32(k01 > k23 ? k01 : k23);
    int best = (int)(key >> 8);
    int best_r = -1, best_c = -1;
    if (best > 0) { int idx = 255 - (int)(key & 255u); best_r = idx >> 4; best_c = idx & 15; }
    int nx = (bx + best_c) * T + offset, ny = (by + best_r) * T + offset;
    float sim = __fdiv_rn(__fmul_rn((float)best, 100.f), (float)(4 * mt.nfeat_total));
    if (lane == 0) {
        if (sim < a.threshold) {
            if (a.stat) atomicAdd(&a.stat[5], 1ull);
            cand[i].ti = LM_DROPPED;
        } else if (LAST) {
            emit_key(a, hdr, keys, ti, nx, ny, sim);
        } else {
            LmCand o; o.ti = ti; o.x = nx; o.y = ny; o.sim = sim;
            cand[i] = o;
        }
    }
}

// Two neighbouring list entries in lock step.  The scan leaves a lane's hits -- neighbouring lattice positions of one
// template -- next to each other in the list; their 16 x 16 patches overlap by about 80 %, i.e. they pull the SAME
// lines.  When both entries name the same template the wave issues the two patch loads of every feature back to back,
// so the second one hits the line the first has just requested: one L2 request instead of two (the kernel is bound by
// the L2 lines a patch pulls).  Different templates (or a dropped entry): one after the other, as before.
#ifndef RP_BATCH
#define RP_BATCH 8   // features per load batch of refine_pair (x 2 candidates = loads in flight per wave)
#endif
static_assert(RP_BATCH % 2 == 0, "the responses are added two features at a time (v_add3_u32)");
template <bool LAST, bool W4>
__device__ __forceinline__ void refine_pair(const LmRefineArgs& a, u32 slot, u32 i, const u8 (*resp)[256], int lane) {
    LmCand* cand = slot_ptr_s(a.cand, a.aux_slot_stride, slot);
    const u32 tiA = (u32)__builtin_amdgcn_readfirstlane((int)cand[i].ti);
    const u32 tiB = (u32)__builtin_amdgcn_readfirstlane((int)cand[i + 1].ti);
    if (tiA != tiB || tiA == LM_DROPPED) {
        refine_one<LAST, W4>(a, slot, i, resp, lane);
        refine_one<LAST, W4>(a, slot, i + 1, resp, lane);
        return;
    }
    LmDevHeader* hdr = slot_ptr_s(a.hdr, a.aux_slot_stride, slot);
    u64* keys = slot_ptr_s(a.keys, a.aux_slot_stride, slot);
    const u8* lm = a.lm + (size_t)slot * a.lm_slot_stride;
    const int T = a.g.T, W = a.g.W;
    const int border = 8 * T;
    const int offset = T / 2 + (T % 2 - 1);
    const u32 lane_off = (u32)((lane >> 2) * W + (lane & 3) * 4);
    const u32 ti = tiA;
    const LmRefMeta mt = a.meta[ti];
    const int max_x = a.g.w - mt.width - border, max_y = a.g.h - mt.height - border;
    int bx[2], by[2], off_x[2], off_y[2];
    u32 shift[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const LmCand cd = cand[i + c];
        const int cx = __builtin_amdgcn_readfirstlane(cd.x), cy = __builtin_amdgcn_readfirstlane(cd.y);
        int x = cx * 2 + 1, y = cy * 2 + 1;
        x = x > border ? x : border; y = y > border ? y : border;
        x = x < max_x ? x : max_x;  y = y < max_y ? y : max_y;
        bx[c] = x / T - 8; by[c] = y / T - 8;
        off_x[c] = bx[c] * T; off_y[c] = by[c] * T;
        shift[c] = (u32)(by[c] * W + bx[c]);
    }
    u32 s[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (a.stat && lane == 0) atomicAdd(&a.stat[1], 2ull);
    const int dcol = bx[1] - bx[0];                                   // wave-uniform
    const bool same_rows = by[0] == by[1] && dcol >= 0 && dcol <= 4;
    // r06: refine_one's exact pruning for the pair: every 16 features the wave takes each candidate's best partial sum of its
    // patch; a candidate whose (best + 4 x features to come) * 100 / (4 n) stays below the threshold -- the final test's own float expression --
    // is out.  Both out: the pair stops.  One out: the other is finished alone (refine_one from its first feature: at most the features loaded so
    // far are read twice).  MEASURED AND COMPILED OUT (PRUNE_REFINE_PAIR): list neighbours are neighbouring positions of a template that matches
    // there -- 8.1 % of the paired candidates die by this test, both of a pair in 2.6 % of the pairs, and they die on their last features; the test
    // costs 5.5 M vector wave-instructions per 96-frame launch (35.8 -> 41.3 M) for the same duration alone on the chip, and the three-lane step,
    // which is bound by vector issue, loses 0.6 % (config 2) / 1.2 % (config 5) to it (tools/ab_refine_pair_prune.sh, profiles/r06_ab_experiments.log
    // section 6).  Build with -DPRUNE_REFINE_PAIR=true to count the pairs again (LM_REFINE_STAT).
    int f_left = mt.nfeat_total;
    const float denom = (float)(4 * mt.nfeat_total);
    for (int m = 0; m < a.M; ++m) {
        // (selects, not mt.count[m]: a runtime index into the struct copy would put it into scratch memory)
        const int cnt = (int)(m == 0 ? mt.count[0] : mt.count[1]);
        const u32 fstart = m == 0 ? mt.start[0] : mt.start[1];
        LmRefFeat ft;
        ft.off = 0; ft.x = 0; ft.y = 0;
        if (lane < cnt) ft = a.feats[fstart + lane];
        u32 eff[2];
        bool ok2[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int fx = ft.x + off_x[c], fy = ft.y + off_y[c];
            const bool ok = (lane < cnt) && fx >= 0 && fy >= 0 && fx < a.g.w && fy < a.g.h;
            ok2[c] = ok;
            eff[c] = ok ? (ft.off & 0x1FFFFFFFu) + shift[c] : a.g.zero_off;
        }
        const u32 lab = ft.off >> 29;
        // ONE load for both patches (r03): the two entries are neighbouring lattice positions of one template, i.e. the second
        // patch is the first moved 0 .. 4 columns to the right in the same rows, and every feature lies inside the image for both.
        // A lane then takes 12 bytes from the first patch's aligned address and cuts both its dwords out of them -- half the
        // wave-loads (the L1 spends 16 cycles on each, whatever it returns), one more select pair per feature.
        const bool share = W4 && same_rows && __all(lane >= cnt || (ok2[0] && ok2[1]));
        for (int f = 0; f < cnt; f += RP_BATCH) {
            u32 v[2][RP_BATCH], q[2][RP_BATCH][4];
            if (share) {
#pragma unroll
                for (int k = 0; k < RP_BATCH; ++k) {
                    const u32 se = (u32)__builtin_amdgcn_readlane((int)eff[0], f + k);
                    u32 d0, d1, d2;
                    ld12a4(lm + ((se & ~3u) + lane_off), d0, d1, d2);
                    const u32 o = (se & 3u) + (u32)dcol;          // byte offset of the second patch's dword in the 12 bytes: 0 .. 7
                    const bool up = o >= 4u;                      // wave-uniform
                    v[0][k] = __builtin_amdgcn_alignbyte(d1, d0, se);
                    v[1][k] = __builtin_amdgcn_alignbyte(up ? d2 : d1, up ? d1 : d0, o);
                }
            } else {
#pragma unroll
            for (int k = 0; k < RP_BATCH; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) v[c][k] = rf_patch<W4>(lm, (u32)__builtin_amdgcn_readlane((int)eff[c], f + k), lane_off);
            }
#pragma unroll
            for (int k = 0; k < RP_BATCH; ++k) {
                const u8* tab = resp[(u32)__builtin_amdgcn_readlane((int)lab, f + k)];
#pragma unroll
                for (int c = 0; c < 2; ++c) rf_lookup(tab, v[c][k], q[c][k]);
            }
#pragma unroll
            for (int k = 0; k < RP_BATCH; k += 2)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) s[c][p4] += q[c][k][p4] + q[c][k + 1][p4];
            f_left -= min(RP_BATCH, cnt - f);
            if (PRUNE_REFINE_PAIR && (f & RP_BATCH) && f_left > 0) {       // every second batch
                bool out[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const u32 best_now = wave_max_u32(max(max(s[c][0], s[c][1]), max(s[c][2], s[c][3])));
                    out[c] = __fdiv_rn(__fmul_rn((float)(int)(best_now + 4u * (u32)f_left), 100.f), denom) < a.threshold;
                }
                if (out[0] || out[1]) {
                    if (a.stat && lane == 0) { atomicAdd(&a.stat[2], (unsigned long long)(out[0] + out[1])); if (out[0] && out[1]) atomicAdd(&a.stat[3], 1ull); }
                    if (out[0] && lane == 0) cand[i].ti = LM_DROPPED;
                    if (out[1] && lane == 0) cand[i + 1].ti = LM_DROPPED;
                    if (!out[0]) refine_one<LAST, W4>(a, slot, i, resp, lane);
                    if (!out[1]) refine_one<LAST, W4>(a, slot, i + 1, resp, lane);
                    return;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const u32 idx0 = (u32)lane * 4u;
        const u32 k0 = (s[c][0] << 8) | (255u - idx0), k1 = (s[c][1] << 8) | (254u - idx0);
        const u32 k2 = (s[c][2] << 8) | (253u - idx0), k3 = (s[c][3] << 8) | (252u - idx0);
        const u32 k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
        const u32 key = wave_max_u32(k01 > k23 ? k01 : k23);
        const int best = (int)(key >> 8);
        int best_r = -1, best_c = -1;
        if (best > 0) { const int idx = 255 - (int)(key & 255u); best_r = idx >> 4; best_c = idx & 15; }
        const int nx = (bx[c] + best_c) * T + offset, ny = (by[c] + best_r) * T + offset;
        const float sim = __fdiv_rn(__fmul_rn((float)best, 100.f), (float)(4 * mt.nfeat_total));
        if (lane == 0) {
            if (sim < a.threshold) {
                if (a.stat) atomicAdd(&a.stat[5], 1ull);
                cand[i + c].ti = LM_DROPPED;
            } else if (LAST) {
                emit_key(a, hdr, keys, ti, nx, ny, sim);
            } else {
                LmCand o; o.ti = ti; o.x = nx; o.y = ny; o.sim = sim;
                cand[i + c] = o;
            }
        }
    }
}

// Work distribution.  Every slot must stay on ONE XCD (its spread memories live in that L2), and candidate counts
// differ a lot between frames (30 .. 1500).  With a plan (k_refine_plan: eight balanced slot lists + the running sums of
// their candidate counts) an XCD's workgroups form ONE queue over all candidates of the XCD's slots: wave w takes the
// candidates w, w + waves, ... of the concatenated lists, so no wave idles while another slot of the XCD still has
// work.  (Measured r02: the same 1.9 us per frame as a fixed share of workgroups per slot -- the kernel is bound by the
// L2 lines a 16 x 16 patch pulls, 16 lines for 256 useful bytes, not by idle waves; kept because it cannot lose.)
template <bool LAST, bool W4>
__global__ __launch_bounds__(256) void k_refine(LmRefineArgs a) {
    // response of orientation o to spread byte v = max(LUT_lo[o][v & 15], LUT_hi[o][v >> 4]): 8 x 256 bytes in LDS,
    // built once per workgroup; a feature then costs one ds_read_u8 per position instead of two 16-entry
    // v_perm lookups and a byte max (the kernel was VALU-bound on those)
    __shared__ u8 resp[8][256];
    const int lane = threadIdx.x & 63;
    u32 slot = 0, tile = 0, total = 0;
    const u32* xs = nullptr; const u32* xpre = nullptr; const u32* xoff = nullptr;
    u32 xlen = 0;
    if (a.plan) {   // block b runs on XCD b % 8 (see xcd_slot_tile)
        const u32 x = blockIdx.x & 7u;
        tile = blockIdx.x >> 3;
        xs = a.plan + (size_t)x * a.plan_cap;
        xlen = a.plan[(size_t)8 * a.plan_cap + x];
        xpre = a.plan + (size_t)8 * a.plan_cap + 8 + (size_t)x * (a.plan_cap + 1);
        xoff = a.plan + (size_t)8 * a.plan_cap + 8 + (size_t)8 * (a.plan_cap + 1) + (size_t)x * a.plan_cap;     // first list entry of every piece
        total = xpre[xlen];
        if (tile * 8u >= total) return;
    } else {
        xcd_slot_tile((u32)a.blocks_per_slot, (u32)a.nslots, slot, tile);
        total = slot_ptr_s(a.hdr, a.aux_slot_stride, slot)->cand_count;
        if (total > a.cand_cap) total = a.cand_cap;
        if (tile * (total <= (u32)a.blocks_per_slot * 4u ? 4u : 8u) >= total) return;   // nothing to do for this workgroup (a wave takes one or two entries): leave before building the table
    }
    {
        const u8* sl = reinterpret_cast<const u8*>(a.sim_lut);   // [ori][lo 16 B | hi 16 B]
        const int v = threadIdx.x;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const u8 lo = sl[32 * o + (v & 15)], hi = sl[32 * o + 16 + (v >> 4)];
            resp[o][v] = lo > hi ? lo : hi;
        }
    }
    __syncthreads();
    const u32 wave0 = (u32)__builtin_amdgcn_readfirstlane((int)((tile * 256u + threadIdx.x) >> 6));
    const u32 nwaves = (u32)a.blocks_per_slot * 4u;
    // a wave takes list entries two at a time (refine_pair); an odd entry at the end of a slot's list goes alone
    if (a.plan) {
        u32 idx = 0;
        for (u32 g = 2u * wave0; g < total; g += 2u * nwaves) {
            while (idx + 1 < xlen && xpre[idx + 1] <= g) ++idx;      // g only grows: the list position moves forward
            const u32 i = g - xpre[idx] + xoff[idx];
            if (g + 1 < xpre[idx + 1]) {
                refine_pair<LAST, W4>(a, xs[idx], i, resp, lane);
            } else {
                refine_one<LAST, W4>(a, xs[idx], i, resp, lane);
                if (g + 1 < total) {                                  // the second entry opens the next slot's list
                    u32 idx2 = idx;
                    while (idx2 + 1 < xlen && xpre[idx2 + 1] <= g + 1) ++idx2;
                    refine_one<LAST, W4>(a, xs[idx2], g + 1 - xpre[idx2] + xoff[idx2], resp, lane);
                }
            }
        }
    } else if (total <= nwaves) {
        // few frames: the list fits one round of waves, an entry per wave finishes sooner than pairs on half of them
        if (wave0 < total) refine_one<LAST, W4>(a, slot, wave0, resp, lane);
    } else {
        for (u32 i = 2u * wave0; i < total; i += 2u * nwaves) {
            if (i + 1 < total) refine_pair<LAST, W4>(a, slot, i, resp, lane);
            else refine_one<LAST, W4>(a, slot, i, resp, lane);
        }
    }
}

__global__ __launch_bounds__(256) void k_emit_unrefined(LmRefineArgs a) {
    LmDevHeader* hdr = slot_ptr(a.hdr, a.aux_slot_stride);
    LmCand* cand = slot_ptr(a.cand, a.aux_slot_stride);
    u64* keys = slot_ptr(a.keys, a.aux_slot_stride);
    u32 n = hdr->cand_count;
    if (n > a.cand_cap) n = a.cand_cap;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        LmCand c = cand[i];
        if (c.ti != LM_DROPPED) emit_key(a, hdr, keys, c.ti, c.x, c.y, c.sim);
    }
}

// ------------------------------------------------------------------------------------------------
// a15  sort + unique, one workgroup per frame slot.  Keys are (hi, lo) u64 pairs whose ascending
// order is the total order of SURVEY.md A.9; equality for std::unique is (x, y, similarity, class)
// = (lo, hi >> 32).  n <= 256: rank sort (teams of threads count the keys before each key: two
// barriers in total); n <= LM_SORT_CAP: bitonic network; above that the host sorts the keys.
// The kernel also publishes the header to host-mapped memory together with the first
// LM_INLINE_MATCHES records and re-arms the device counters for the next frame.
// ------------------------------------------------------------------------------------------------
// bitonic network over hi[0, N) / lo[0, N) in LDS, N a power of two, 1024 threads.  A wave's 64 pair-threads touch only "their"
// 128 consecutive elements while j <= 64, and a wave's LDS operations execute in order: those stages need no workgroup
// barrier, only the stages with j > 64 do (10 of the 66 stages of a 2048-element sort).
__device__ __forceinline__ void sort_bitonic_lds(u64* hi, u64* lo, u32 N, int tid) {
    bool local_pending = false;   // stages since the last barrier were wave-local
    for (u32 k = 2; k <= N; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            if (j > 64 && local_pending) { __syncthreads(); local_pending = false; }
            for (u32 t = tid; t < (N >> 1); t += 1024) {
                u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                u32 p = i | j;
                bool up = (i & k) == 0;
                u64 ah = hi[i], al = lo[i], bh = hi[p], bl = lo[p];
                bool gt = ah > bh || (ah == bh && al > bl);
                if (gt == up) { hi[i] = bh; lo[i] = bl; hi[p] = ah; lo[p] = al; }
            }
            if (j > 64) __syncthreads();
            else { __builtin_amdgcn_wave_barrier(); local_pending = true; }
        }
    __syncthreads();
}

// adjacent-unique + compaction of the sorted keys hi / lo [0, n) in LDS (block-wide exclusive scan of keep flags, chunks of 1024),
// records to `out` and the first LM_INLINE_MATCHES to the host-mapped block, header published, list length left for k_pack_lists
__device__ __forceinline__ void sort_unique_publish(const u64* hi, const u64* lo, u32 n, u32* wave_tot, LmDevHeader* hdr, LmOutMatch* out,
                                                    LmHostBlock* hb, u32 cand_count, u32 match_count, int tid) {
    u32 base = 0;
    for (u32 c0 = 0; c0 < n; c0 += 1024) {
        u32 i = c0 + tid;
        u32 keep = 0;
        if (i < n) keep = (i == 0) || !(lo[i] == lo[i - 1] && (hi[i] >> 32) == (hi[i - 1] >> 32));
        unsigned long long bal = __ballot(keep);
        int lane = tid & 63, wv = tid >> 6;
        u32 pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        u32 woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { u32 t = wave_tot[k]; if (k < wv) woff += t; tot += t; }
        if (keep) {
            u64 h = hi[i], l = lo[i];
            LmOutMatch m;
            m.similarity = __uint_as_float(~(u32)(h >> 32));
            m.template_id = (int)(u32)h;
            m.class_idx = (int)(l >> 48);
            m.y = (int)((l >> 24) & 0xFFFFFFu) - 0x800000;
            m.x = (int)(l & 0xFFFFFFu) - 0x800000;
            u32 pos = base + woff + pre;
            out[pos] = m;
            if (pos < LM_INLINE_MATCHES) hb->rec[pos] = m;
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) {
        hb->hdr.cand_count = cand_count; hb->hdr.match_count = match_count;
        hb->hdr.out_count = base; hb->hdr.sorted_on_device = 1;
        hdr->pad[0] = base;              // length of the sorted list in `out` (k_pack_lists)
    }
}

// Split form (r04, a.split != 0): a list longer than LM_SORT_CHUNK keys is sorted as chunks of LM_SORT_CHUNK by the
// workgroups blockIdx.y = 0 .. LM_SORT_CAP / LM_SORT_CHUNK - 1 of its frame (in place, in `keys`), and k_merge_unique -- the next
// launch -- ranks every key against the other chunks by binary search, which replaces the last (and longest) phases of the
// network and spreads the rest over four CUs per frame: a lane-step of few frames (config 5: 8) otherwise sorts on 8 of 256
// CUs.  Lists of up to LM_SORT_CHUNK keys, overflowed ones and those left to the host are finished here by workgroup 0 as in
// the plain form, and k_merge_unique leaves them alone (it finds the counters re-armed).
__global__ __launch_bounds__(1024) void k_sort_unique(LmSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* hi = reinterpret_cast<u64*>(smem);
    u64* lo = hi + LM_SORT_CAP;
    __shared__ u32 wave_tot[16];
    LmDevHeader* hdr = slot_ptr(a.hdr, a.aux_slot_stride);
    u64* keys = slot_ptr(a.keys, a.aux_slot_stride);
    LmOutMatch* out = slot_ptr(a.out, a.aux_slot_stride);
    LmHostBlock* hb = reinterpret_cast<LmHostBlock*>(reinterpret_cast<u8*>(a.host) + (size_t)blockIdx.z * a.host_slot_stride);
    const int tid = threadIdx.x;
    const u32 chunk = blockIdx.y;
    const u32 cand_count = hdr->cand_count, match_count = hdr->match_count;
    u32 n = match_count;
    if (n > a.match_cap) n = a.match_cap;
    __syncthreads();  // everyone has read the counters
    const bool unsortable = n > LM_SORT_CAP || cand_count > a.cand_cap || match_count > a.match_cap;
    if (a.split && !unsortable && n > LM_SORT_CHUNK) {
        // this workgroup's chunk, sorted in place; counters stay armed for k_merge_unique
        const u32 c0 = chunk * LM_SORT_CHUNK;
        if (c0 >= n) return;
        const u32 cnt = min(n - c0, (u32)LM_SORT_CHUNK);
        u32 N = 64;
        while (N < cnt) N <<= 1;
        for (u32 i = tid; i < N; i += 1024) {
            hi[i] = i < cnt ? keys[2 * (size_t)(c0 + i)] : ~0ull;
            lo[i] = i < cnt ? keys[2 * (size_t)(c0 + i) + 1] : ~0ull;
        }
        __syncthreads();
        sort_bitonic_lds(hi, lo, N, tid);
        for (u32 i = tid; i < cnt; i += 1024) { keys[2 * (size_t)(c0 + i)] = hi[i]; keys[2 * (size_t)(c0 + i) + 1] = lo[i]; }
        return;
    }
    if (chunk != 0) return;
    if (tid == 0) { hdr->cand_count = 0; hdr->match_count = 0; }
    if (unsortable) {
        if (tid == 0) {
            hb->hdr.cand_count = cand_count; hb->hdr.match_count = match_count;
            hb->hdr.out_count = 0; hb->hdr.sorted_on_device = 0;
            // no device-side list for k_pack_lists: ...FF = left to the host sort, ...FE = capacity overflow of this shard
            hdr->pad[0] = (cand_count > a.cand_cap || match_count > a.match_cap) ? 0xFFFFFFFEu : 0xFFFFFFFFu;
        }
        return;
    }
    if (n <= 256) {
        // small lists: rank sort (O(n^2) compares, two barriers) with every thread busy; above 256 keys the bitonic
        // network below does an order of magnitude less work (n = 1000: 1 M compares against 55 stages x 512).
        // rank sort with every thread busy: the 1024 threads form 1024 / n2 teams (n2 = n rounded up to a power of
        // two), thread t of team p counts the keys of the p-th slice of the list that sort before key t; the partial
        // ranks meet in LDS.  The slice loop is unrolled so that its LDS reads are in flight together.
        __shared__ u32 rk[1024];
        u64 mh = ~0ull, ml = ~0ull;
        if ((u32)tid < n) { mh = keys[2 * (size_t)tid]; ml = keys[2 * (size_t)tid + 1]; }
        hi[tid] = mh; lo[tid] = ml;
        rk[tid] = 0;
        __syncthreads();
        u32 n2 = 64;
        while (n2 < n) n2 <<= 1;
        const u32 teams = 1024u / n2, e = (u32)tid & (n2 - 1u), team = (u32)tid / n2;
        const u64 eh = hi[e], el = lo[e];
        const u32 j0 = (u32)((u64)n * team / teams), j1 = (u32)((u64)n * (team + 1u) / teams);
        u32 rank = 0;
        if (e < n) {
#pragma unroll 8
            for (u32 j = j0; j < j1; ++j) {
                const u64 h = hi[j], l = lo[j];
                const bool before = h < eh || (h == eh && (l < el || (l == el && j < e)));
                rank += before ? 1u : 0u;
            }
            if (teams > 1) atomicAdd(&rk[e], rank);
            else rk[e] = rank;
        }
        __syncthreads();
        const u32 myrank = rk[tid];
        __syncthreads();
        if ((u32)tid < n) { hi[myrank] = mh; lo[myrank] = ml; }
        __syncthreads();
    } else {
        u32 N = 1;
        while (N < n) N <<= 1;
        for (u32 i = tid; i < N; i += 1024) {
            hi[i] = i < n ? keys[2 * (size_t)i] : ~0ull;
            lo[i] = i < n ? keys[2 * (size_t)i + 1] : ~0ull;
        }
        __syncthreads();
        sort_bitonic_lds(hi, lo, N, tid);
    }
    sort_unique_publish(hi, lo, n, wave_tot, hdr, out, hb, cand_count, match_count, tid);
}

// Second launch of the split form: one workgroup per frame.  The frame's chunks (sorted by k_sort_unique) come into LDS, every key
// finds its place in the whole list -- own index in its chunk + the number of keys of every other chunk that go before it (keys
// of an earlier chunk win ties: the ranks are a permutation) -- the keys are scattered to their ranks, then the same unique +
// compaction + publication as the plain form.  Frames k_sort_unique finished itself show up here with their counters re-armed.
__global__ __launch_bounds__(1024) void k_merge_unique(LmSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* hi = reinterpret_cast<u64*>(smem);
    u64* lo = hi + LM_SORT_CAP;
    __shared__ u32 wave_tot[16];
    LmDevHeader* hdr = slot_ptr(a.hdr, a.aux_slot_stride);
    const u64* keys = slot_ptr(a.keys, a.aux_slot_stride);
    LmOutMatch* out = slot_ptr(a.out, a.aux_slot_stride);
    LmHostBlock* hb = reinterpret_cast<LmHostBlock*>(reinterpret_cast<u8*>(a.host) + (size_t)blockIdx.z * a.host_slot_stride);
    const int tid = threadIdx.x;
    const u32 cand_count = hdr->cand_count, match_count = hdr->match_count;
    const u32 n = match_count;
    __syncthreads();  // everyone has read the counters
    if (n <= LM_SORT_CHUNK || n > LM_SORT_CAP || cand_count > a.cand_cap || match_count > a.match_cap) return;   // finished by k_sort_unique
    if (tid == 0) { hdr->cand_count = 0; hdr->match_count = 0; }
    for (u32 i = tid; i < n; i += 1024) { hi[i] = keys[2 * (size_t)i]; lo[i] = keys[2 * (size_t)i + 1]; }
    __syncthreads();
    constexpr int PER = LM_SORT_CAP / 1024;
    u64 eh[PER], el[PER];
    u32 rank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const u32 i = (u32)tid + 1024u * (u32)k;
        rank[k] = 0xFFFFFFFFu;
        if (i >= n) continue;
        eh[k] = hi[i]; el[k] = lo[i];
        const u32 c = i / LM_SORT_CHUNK;
        u32 r = i - c * LM_SORT_CHUNK;
        for (u32 b = 0; b < n; b += LM_SORT_CHUNK) {
            if (b == c * LM_SORT_CHUNK) continue;
            const u32 len = min(n - b, (u32)LM_SORT_CHUNK);
            const bool earlier = b < c * LM_SORT_CHUNK;      // its equal keys go before this one
            u32 pos = 0;                                       // keys of the chunk that go before (eh, el)
#pragma unroll
            for (u32 step = LM_SORT_CHUNK; step >= 1; step >>= 1) {
                const u32 q = pos + step;
                if (q <= len) {
                    const u64 h = hi[b + q - 1], l = lo[b + q - 1];
                    const bool before = h < eh[k] || (h == eh[k] && (earlier ? l <= el[k] : l < el[k]));
                    pos = before ? q : pos;
                }
            }
            r += pos;
        }
        rank[k] = r;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (rank[k] != 0xFFFFFFFFu) { hi[rank[k]] = eh[k]; lo[rank[k]] = el[k]; }
    __syncthreads();
    sort_unique_publish(hi, lo, n, wave_tot, hdr, out, hb, cand_count, match_count, tid);
}

// ------------------------------------------------------------------------------------------------
// 8e  Packs the sorted lists of `nslots` frames back to back (what a rank contributes to the all-gather): workgroup i
// adds up the lengths of the lists before its own and copies list i behind them.  cnt[i] = length of list i,
// cnt[nslots] = status (0 ok, bit 0: the lists do not fit cap_total records, bit 1: a list was left to the host sort,
// bit 2: a slot overflowed its candidate / match capacity).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_lists(LmPackArgs a) {
    __shared__ u32 part[4];
    const int tid = threadIdx.x, slot = blockIdx.x;
    u32 s = 0;
    for (int j = tid; j < slot; j += 256) {
        const u32 c = slot_ptr_s(a.hdr, a.aux_slot_stride, (u32)j)->pad[0];
        s += c >= 0xFFFFFFFEu ? 0u : c;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += (u32)__shfl_xor((int)s, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const u32 prefix = part[0] + part[1] + part[2] + part[3];
    u32 mine = slot_ptr_s(a.hdr, a.aux_slot_stride, (u32)slot)->pad[0];
    if (mine >= 0xFFFFFFFEu) { if (tid == 0) atomicOr(reinterpret_cast<u32*>(a.cnt + a.nslots), mine == 0xFFFFFFFEu ? 4u : 2u); mine = 0; }
    if (tid == 0) a.cnt[slot] = (int)mine;
    if (prefix + mine > a.cap_total) { if (tid == 0) atomicOr(reinterpret_cast<u32*>(a.cnt + a.nslots), 1u); return; }
    const u32* src = reinterpret_cast<const u32*>(slot_ptr_s(a.out, a.aux_slot_stride, (u32)slot));
    u32* dst = reinterpret_cast<u32*>(a.rec) + (size_t)prefix * 5;
    for (u32 i = tid; i < mine * 5u; i += 256) dst[i] = src[i];
}


// ------------------------------------------------------------------------------------------------
// f1  Colour check of the reference's post-processing, batched.
//   k_hsv_mask     cv::cvtColor(BGR2HSV, 8-bit) + cv::inRange (HighLevelLinemod.cpp:159-161): one bit per pixel.
//   k_hull_counts  per match: templateMask (:113-135) = fillPoly of the convex hull of the template's level-0 features
//                  moved to the match position, then the two countNonZero of colorCheck (:424-434): pixels in the
//                  hull, and pixels in the hull whose colour bit is set.  One wave per match.  fillPoly draws the
//                  polygon's interior AND its outline: per image row the filled pixels are one run -- from the
//                  leftmost to the rightmost of {exact row/polygon intersection, the outline's 8-connected line
//                  pixels on that row} -- so a lane per row suffices: the outline runs come from the lanes walking
//                  one edge each (LDS atomic min / max per row), the intersection from the same double-precision
//                  formula as the host restatement (host/PostProcess.cpp hull_counts), which is the checker.
// ------------------------------------------------------------------------------------------------

}  // namespace

// ================================================================================================
// launchers
// ================================================================================================
void lmk_refine_plan(hipStream_t s, const LmRefineArgs& a, int nslots, u32* plan, int plan_cap) {
    hipLaunchKernelGGL(k_refine_plan, dim3(1), dim3(1024), 0, s, a.hdr, a.aux_slot_stride, nslots, a.cand_cap, plan_cap, plan);
}

void lmk_refine(hipStream_t s, const LmRefineArgs& a_in, bool last, int nslots) {
    LmRefineArgs a = a_in;
    // persistent waves stride over the slot's candidate list; with the XCD-affine mapping one slot runs on one
    // XCD (32 CUs x 32 waves), so 256 blocks = 1024 waves per slot fill it
    a.blocks_per_slot = 256; a.nslots = nslots;
    // with a plan the 256 workgroups of XCD x (8 per CU) are one queue over the candidates of the slots on its list
    dim3 grid(a.plan ? (unsigned)(8 * a.blocks_per_slot) : (unsigned)(a.blocks_per_slot * nslots), 1, 1);
    const bool w4 = (a.g.W & 3) == 0;     // the patch rows' pitch: scalar alignment arithmetic in rf_patch
    if (last) { if (w4) hipLaunchKernelGGL((k_refine<true, true>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((k_refine<true, false>), grid, dim3(256), 0, s, a); }
    else { if (w4) hipLaunchKernelGGL((k_refine<false, true>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((k_refine<false, false>), grid, dim3(256), 0, s, a); }
}

void lmk_emit_unrefined(hipStream_t s, const LmRefineArgs& a, int nslots) {
    hipLaunchKernelGGL(k_emit_unrefined, dim3(64, 1, nslots), dim3(256), 0, s, a);
}

void lmk_pack_lists(hipStream_t s, const LmPackArgs& a) {
    (void)hipMemsetAsync(a.cnt + a.nslots, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_pack_lists, dim3((unsigned)a.nslots), dim3(256), 0, s, a);
}

void lmk_sort_unique(hipStream_t s, const LmSortArgs& a, int nslots) {
    size_t shmem = (size_t)LM_SORT_CAP * 16;
    hipLaunchKernelGGL(k_sort_unique, dim3(1, a.split ? LM_SORT_CAP / LM_SORT_CHUNK : 1, nslots), dim3(1024), shmem, s, a);
    if (a.split) hipLaunchKernelGGL(k_merge_unique, dim3(1, 1, nslots), dim3(1024), shmem, s, a);
}
