// lm_dev_memories.h -- a6-a10: NN pyrDown read + spread(T) + response LUT + linearize, in the byte, nibble, spread-byte and
// miss-plane forms (k_linear_memories, k_lm_fast, k_lm_spread2, k_lm_spread5).  Included by lm_k_preprocess.hip only.
#pragma once
#include "lm_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a6-a10  One workgroup per (band of T image rows, segment of `seg` memory columns).
//   LDS: response table (256 x u64: byte o = response of orientation o to spread value v),
//        (2T-1) source rows of the segment (+T-1 halo columns), their horizontal OR.
// Thread unit = (row-in-band j, column phase c0, four consecutive memory columns) so each of the 8
// orientation stores is one aligned dword and consecutive lanes write consecutive dwords.
// ------------------------------------------------------------------------------------------------
#define LMK_MAX_LOADS 8
// SPREAD_ONLY = false: the 8 response linear memories (lowest pyramid level, read by the scan).
// SPREAD_ONLY = true : one "spread linear memory" holding the spread byte itself (levels that are
//   only refined at): 1/8 of the bytes; k_refine applies the response LUT in registers.
template <int SRC_SHIFT, bool SPREAD_ONLY>
__global__ __launch_bounds__(256) void k_linear_memories(const u8* __restrict__ q0, int qpitch, int w, int h, int T,
                                                          int seg, const u64* __restrict__ resp_tab,
                                                          u8* __restrict__ lm0, u32 ori_stride, size_t q_slot_stride,
                                                          size_t lm_slot_stride) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u8* q = slot_ptr(q0, q_slot_stride);
    u8* lm = slot_ptr(lm0, lm_slot_stride);
    const int W = w / T;
    const u32 wh = (u32)W * (u32)(h / T);
    const int rows = 2 * T - 1;
    const int pitch = (seg * T + T + 3) & ~3;
    u64* tab = reinterpret_cast<u64*>(smem);
    u8* qs = smem + 2048;
    u8* ho = qs + rows * pitch;
    const int tid = threadIdx.x;
    const int band = blockIdx.y;
    const int col0 = blockIdx.x * seg;                       // first memory column of this segment
    const int ncols = (W - col0) < seg ? (W - col0) : seg;   // memory columns in this segment
    const int px0 = col0 * T, npx = ncols * T;
    const int y0 = band * T;
    const int lw = npx + T - 1;                              // source columns needed (halo to the right)

    tab[tid] = resp_tab[tid];
    {   // source rows: up to LMK_MAX_LOADS independent byte loads per thread
        const int total = rows * lw;
        u8 v[LMK_MAX_LOADS];
#pragma unroll
        for (int k = 0; k < LMK_MAX_LOADS; ++k) {
            int i = tid + k * 256;
            int yy = i / lw, xx = i - yy * lw;
            int gy = y0 + yy, gx = px0 + xx;
            u8 val = 0;
            if (i < total && gy < h && gx < w)
                val = SRC_SHIFT ? q[(size_t)(2 * gy) * qpitch + 2 * gx] : q[(size_t)gy * qpitch + gx];
            v[k] = val;
        }
#pragma unroll
        for (int k = 0; k < LMK_MAX_LOADS; ++k) {
            int i = tid + k * 256;
            if (i < total) { int yy = i / lw; qs[yy * pitch + (i - yy * lw)] = v[k]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < rows * npx; i += 256) {
        int yy = i / npx, xx = i - yy * npx;
        const u8* p = qs + yy * pitch + xx;
        u8 v = 0;
        for (int c = 0; c < T; ++c) v |= p[c];
        ho[yy * pitch + xx] = v;
    }
    __syncthreads();
    if ((W & 3) == 0 && (seg & 3) == 0) {
        const int C4 = ncols >> 2;
        const int units = T * T * C4;
        for (int u = tid; u < units; u += 256) {
            int k4 = u % C4, g = u / C4;
            int j = g / T, c0 = g - j * T;
            u8 sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int x = (4 * k4 + i) * T + c0;
                const u8* p = ho + j * pitch + x;
                u8 acc = 0;
                for (int r = 0; r < T; ++r) acc |= p[r * pitch];
                sv[i] = acc;
            }
            u8* dst = lm + (size_t)g * wh + (size_t)band * W + col0 + 4 * k4;
            if (SPREAD_ONLY) {
                *reinterpret_cast<u32*>(dst) = (u32)sv[0] | ((u32)sv[1] << 8) | ((u32)sv[2] << 16) | ((u32)sv[3] << 24);
            } else {
                u64 e0 = tab[sv[0]], e1 = tab[sv[1]], e2 = tab[sv[2]], e3 = tab[sv[3]];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    u32 v = (u32)((e0 >> (8 * o)) & 0xFF) | ((u32)((e1 >> (8 * o)) & 0xFF) << 8) |
                            ((u32)((e2 >> (8 * o)) & 0xFF) << 16) | ((u32)((e3 >> (8 * o)) & 0xFF) << 24);
                    *reinterpret_cast<u32*>(dst + (size_t)o * ori_stride) = v;
                }
            }
        }
    } else {
        const int units = T * T * ncols;
        for (int u = tid; u < units; u += 256) {
            int k = u % ncols, g = u / ncols;
            int j = g / T, c0 = g - j * T;
            const u8* p = ho + j * pitch + k * T + c0;
            u8 sv = 0;
            for (int r = 0; r < T; ++r) sv |= p[r * pitch];
            u8* dst = lm + (size_t)g * wh + (size_t)band * W + col0 + k;
            if (SPREAD_ONLY) {
                dst[0] = sv;
            } else {
                u64 e = tab[sv];
#pragma unroll
                for (int o = 0; o < 8; ++o) dst[(size_t)o * ori_stride] = (u8)(e >> (8 * o));
            }
        }
    }
}

// 8 x 8 bit transpose: in, byte i of (x | y << 32) = row i, bit o = column o; out, byte o holds bit i = in(i, o).  The three
// swap steps of the classic recursive transpose (2 x 2 blocks of 1, 2 and 4 bits), on two dwords.
__device__ __host__ __forceinline__ void bit_transpose8(u32& x, u32& y) {
    u32 t;
    t = (y ^ (y >> 7)) & 0x00AA00AAu; y = y ^ t ^ (t << 7);
    t = (x ^ (x >> 7)) & 0x00AA00AAu; x = x ^ t ^ (t << 7);
    t = (y ^ (y >> 14)) & 0x0000CCCCu; y = y ^ t ^ (t << 14);
    t = (x ^ (x >> 14)) & 0x0000CCCCu; x = x ^ t ^ (t << 14);
    t = (y & 0xF0F0F0F0u) | ((x >> 4) & 0x0F0F0F0Fu);
    x = ((y << 4) & 0xF0F0F0F0u) | (x & 0x0F0F0F0Fu);
    y = t;
}

// ------------------------------------------------------------------------------------------------
// a6-a10, fast path: T and the segment width are compile-time (the reference's T = 2, 5, 8 plus 4), so
// every index division is by a constant, and the separable OR runs on dwords (4 pixels per op, byte
// shifts by v_alignbyte_b32).  Needs w, W and the source pitch to be multiples of 4; everything else
// goes through the generic k_linear_memories above.
// ------------------------------------------------------------------------------------------------
// MODE 0: 8 response memories, one byte per position; 1: one spread memory; 2: 8 response memories packed
// two positions per byte (responses are <= 4; position 2k in the low nibble of byte k) for k_scan4.
template <int T, int SEG, int SRC_SHIFT, int MODE>
__device__ __forceinline__ void d_lm_fast(const u32 vblock, const u8* __restrict__ q0, int qpitch, int w, int h,
                                                  const u64* __restrict__ resp_tab, u8* __restrict__ lm0,
                                                  u32 ori_stride, size_t q_slot_stride, size_t lm_slot_stride,
                                                  int nseg, int nslots, u32 plane_ori = 0) {
    constexpr int ROWS = 2 * T - 1;
    constexpr int TW = SEG * T;                    // pixels per segment
    constexpr int NDW = (TW + T - 1 + 3) / 4;      // source dwords per row including the right halo
    constexpr int PD = NDW + 2;                    // LDS pitch in dwords: two zero dwords for the funnel reads
    constexpr int NLOAD = (ROWS * PD + 255) / 256;
    constexpr bool SPREAD_ONLY = MODE == 1;
    __shared__ u64 tab[SPREAD_ONLY ? 1 : 256];
    __shared__ u8 tabu[MODE == 2 ? 256 : 1];       // MODE 2 with planes: orientations whose response to a spread byte is BELOW 4 (behind resp_tab)
    __shared__ u32 qs[ROWS][PD];
    __shared__ u32 ho[ROWS][PD];
    __shared__ u32 sp[T][PD];
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)(nseg * (h / T)), (u32)nslots, slot, tile);
    const u8* q = slot_ptr_s(q0, q_slot_stride, slot);
    u8* lm = slot_ptr_s(lm0, lm_slot_stride, slot);
    const int tid = threadIdx.x;
    const int W = w / T;
    const u32 wh = (u32)W * (u32)(h / T);
    const int band = (int)(tile / (u32)nseg);
    const int col0 = (int)(tile - (u32)band * (u32)nseg) * SEG;
    const int ncols = (W - col0) < SEG ? (W - col0) : SEG;
    const int px0 = col0 * T, y0 = band * T;

    if (!SPREAD_ONLY) tab[tid] = resp_tab[tid];
    if (MODE == 2 && plane_ori) tabu[tid] = reinterpret_cast<const u8*>(resp_tab + 256)[tid];
    {
        u32 v[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int i = tid + k * 256;
            const int r = i / PD, c = i - r * PD;
            const int gy = y0 + r, gx = px0 + 4 * c;
            u32 val = 0;
            if (i < ROWS * PD && c < NDW && gy < h && gx < w) {
                if (SRC_SHIFT) {   // NN half-size read: pixels (2gy, 2gx .. 2gx+6 step 2)
                    const u32* sp2 = reinterpret_cast<const u32*>(q + (size_t)(2 * gy) * qpitch + 2 * gx);
                    val = __builtin_amdgcn_perm(sp2[1], sp2[0], 0x06040200u);
                } else {
                    val = *reinterpret_cast<const u32*>(q + (size_t)gy * qpitch + gx);
                }
            }
            v[k] = val;
        }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int i = tid + k * 256;
            if (i < ROWS * PD) { const int r = i / PD; qs[r][i - r * PD] = v[k]; }
        }
    }
    __syncthreads();
    // horizontal OR over T pixels, 4 pixels per thread-op
    for (int i = tid; i < ROWS * NDW; i += 256) {
        const int r = i / NDW, c = i - r * NDW;
        const u32 d0 = qs[r][c], d1 = qs[r][c + 1], d2 = qs[r][c + 2];
        u32 hor = d0;
#pragma unroll
        for (int k = 1; k < T; ++k) {
            if (k < 4) hor |= __builtin_amdgcn_alignbyte(d1, d0, (u32)k);
            else if (k == 4) hor |= d1;
            else hor |= __builtin_amdgcn_alignbyte(d2, d1, (u32)(k - 4));
        }
        ho[r][c] = hor;
    }
    __syncthreads();
    // vertical OR over T rows
    for (int i = tid; i < T * NDW; i += 256) {
        const int j = i / NDW, c = i - j * NDW;
        u32 v = 0;
#pragma unroll
        for (int r = 0; r < T; ++r) v |= ho[j + r][c];
        sp[j][c] = v;
    }
    __syncthreads();
    if (MODE == 2) {
        // linearize, nibble-packed: unit = (row-in-band j, column phase c0, UW consecutive memory columns).  UW = 8: one dword per orientation
        // (and one byte per miss plane); k_lm_fast<8, 80, ..> -- launched only for rows of whole 80-column segments -- takes UW = 16 (r05): an
        // 8-byte store per orientation and a 2-byte store per plane: the kernel is bound by the number of its scattered stores, not by their
        // bytes.  (A 16-column unit on a segment whose last part is 8 columns wide would write into the next row: the fuzzer's catch.)
        constexpr int UW = (T == 8 && SEG == 80) ? 16 : 8;     // (only the launch that guarantees whole 80-column segments and the stores' alignment)
        constexpr int NH = UW / 8;
        constexpr int CU = SEG / UW;
        // r05, bit 31 of plane_ori: the level's response memories are NOT written -- the bit-plane scan reads the planes, and its second stage takes
        // the few exact sums it needs from ONE byte per position (the spread byte, linearised like a refinement level's memory at the start
        // of the modality's block) through the response table: a 16-byte store per 16 positions instead of eight 8-byte stores
        const bool spread_low = (plane_ori >> 31) != 0;
        plane_ori &= 0x7FFFFFFFu;
        for (int u = tid; u < T * T * CU; u += 256) {
            const int ku = u % CU, g = u / CU;
            const int j = g / T, c0 = g - j * T;
            if (UW * ku >= ncols) continue;
            const u8* row = reinterpret_cast<const u8*>(&sp[j][0]);
            u32 nd[NH][8], pb[NH][8];        // per half of the unit: the nibble dword / the miss byte of every orientation
            u32 sb[NH][2];                   // ... / its eight spread bytes
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int k8 = NH * ku + h;
                if (spread_low) {
                    sb[h][0] = sb[h][1] = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        sb[h][0] |= (u32)row[(8 * k8 + i) * T + c0] << (8 * i);
                        sb[h][1] |= (u32)row[(8 * k8 + 4 + i) * T + c0] << (8 * i);
                    }
                }
#pragma unroll
                for (int o = 0; o < 8; ++o) nd[h][o] = 0;
                if (!spread_low) {
                    u64 e[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)   // byte o of e[i] = response o of columns 2i (low nibble) and 2i+1
                        e[i] = tab[row[(8 * k8 + 2 * i) * T + c0]] | (tab[row[(8 * k8 + 2 * i + 1) * T + c0]] << 4);
                    const u32 a0 = (u32)e[0], a1 = (u32)e[1], a2 = (u32)e[2], a3 = (u32)e[3];
                    const u32 b0 = (u32)(e[0] >> 32), b1 = (u32)(e[1] >> 32), b2 = (u32)(e[2] >> 32), b3 = (u32)(e[3] >> 32);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const u32 sel = (u32)o | ((u32)(o + 4) << 8);
                        nd[h][o] = (__builtin_amdgcn_perm(a1, a0, sel) & 0xFFFFu) | (__builtin_amdgcn_perm(a3, a2, sel) << 16);
                        nd[h][o + 4] = (__builtin_amdgcn_perm(b1, b0, sel) & 0xFFFFu) | (__builtin_amdgcn_perm(b3, b2, sel) << 16);
                    }
                }
                if (plane_ori) {
                    // r05, k_scan1's bit planes: per orientation one BIT per position, set where the response is below 4 (a "miss");
                    // 8 columns are one byte of each of the 8 planes -- an 8 x 8 bit transpose of the 8 miss masks
                    u32 x = 0, y = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        x |= (u32)tabu[row[(8 * k8 + i) * T + c0]] << (8 * i);
                        y |= (u32)tabu[row[(8 * k8 + 4 + i) * T + c0]] << (8 * i);
                    }
                    bit_transpose8(x, y);
#pragma unroll
                    for (int o = 0; o < 4; ++o) { pb[h][o] = (x >> (8 * o)) & 0xFFu; pb[h][o + 4] = (y >> (8 * o)) & 0xFFu; }
                }
            }
            const size_t pos = (size_t)g * wh + (size_t)band * W + col0 + UW * ku;
            u8* dst = lm + (pos >> 1);
            u8* pl = lm + 8 * (size_t)ori_stride + (pos >> 3);
            if (spread_low) {
                if (NH == 2) *reinterpret_cast<u32x4*>(lm + pos) = u32x4{sb[0][0], sb[0][1], sb[NH - 1][0], sb[NH - 1][1]};
                else *reinterpret_cast<u32x2*>(lm + pos) = u32x2{sb[0][0], sb[0][1]};
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                if (!spread_low) {
                    if (NH == 2) *reinterpret_cast<u32x2*>(dst + (size_t)o * ori_stride) = u32x2{nd[0][o], nd[NH - 1][o]};
                    else *reinterpret_cast<u32*>(dst + (size_t)o * ori_stride) = nd[0][o];
                }
                if (plane_ori) {
                    if (NH == 2) *reinterpret_cast<unsigned short*>(pl + (size_t)o * plane_ori) = (unsigned short)(pb[0][o] | (pb[NH - 1][o] << 8));
                    else pl[(size_t)o * plane_ori] = (u8)pb[0][o];
                }
            }
        }
        return;
    }
    // linearize: unit = (row-in-band j, column phase c0, 4 consecutive memory columns)
    constexpr int C4 = SEG / 4;
    for (int u = tid; u < T * T * C4; u += 256) {
        const int k4 = u % C4, g = u / C4;
        const int j = g / T, c0 = g - j * T;
        if (4 * k4 >= ncols) continue;
        const u8* row = reinterpret_cast<const u8*>(&sp[j][0]);
        const u32 s0 = row[(4 * k4 + 0) * T + c0], s1 = row[(4 * k4 + 1) * T + c0];
        const u32 s2 = row[(4 * k4 + 2) * T + c0], s3 = row[(4 * k4 + 3) * T + c0];
        u8* dst = lm + (size_t)g * wh + (size_t)band * W + col0 + 4 * k4;
        if (SPREAD_ONLY) {
            *reinterpret_cast<u32*>(dst) = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
        } else {
            const u64 e0 = tab[s0], e1 = tab[s1], e2 = tab[s2], e3 = tab[s3];
            const u32 a0 = (u32)e0, a1 = (u32)e1, a2 = (u32)e2, a3 = (u32)e3;
            const u32 b0 = (u32)(e0 >> 32), b1 = (u32)(e1 >> 32), b2 = (u32)(e2 >> 32), b3 = (u32)(e3 >> 32);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                // byte o of a0,a1 | byte o of a2,a3: perm selectors pick src1 bytes as 0..3, src0 bytes as 4..7
                const u32 sel = (u32)o | ((u32)(o + 4) << 8);
                u32 lo = __builtin_amdgcn_perm(a1, a0, sel) & 0xFFFFu;
                u32 hi = __builtin_amdgcn_perm(a3, a2, sel) << 16;
                *reinterpret_cast<u32*>(dst + (size_t)o * ori_stride) = lo | hi;
                u32 lo2 = __builtin_amdgcn_perm(b1, b0, sel) & 0xFFFFu;
                u32 hi2 = __builtin_amdgcn_perm(b3, b2, sel) << 16;
                *reinterpret_cast<u32*>(dst + (size_t)(o + 4) * ori_stride) = lo2 | hi2;
            }
        }
    }
}
template <int T, int SEG, int SRC_SHIFT, int MODE>
__global__ __launch_bounds__(256) void k_lm_fast(const u8* __restrict__ q0, int qpitch, int w, int h,
                                                  const u64* __restrict__ resp_tab, u8* __restrict__ lm0,
                                                  u32 ori_stride, size_t q_slot_stride, size_t lm_slot_stride,
                                                  int nseg, int nslots, u32 plane_ori) {
    d_lm_fast<T, SEG, SRC_SHIFT, MODE>(blockIdx.x, q0, qpitch, w, h, resp_tab, lm0, ori_stride, q_slot_stride, lm_slot_stride, nseg, nslots, plane_ori);
}

// ------------------------------------------------------------------------------------------------
// a6-a10 for T = 2, spread memory only (level 0 of the colour-only configuration, T = {2, 8}): a streaming pass.
// spread(y, x) = OR of the 2 x 2 block at (y, x); memory g = (y % 2) * 2 + x % 2 holds it at (y / 2) * W + x / 2.
// One lane = 32 pixels of two rows: three source rows (32 B + one dword of halo each) give 2 x 32 spread bytes, split
// into even / odd x (v_perm) = one 16-byte store per row and memory.  The tiled k_lm_fast<2, ..> needs 2400 tiny
// workgroups with three barriers each per 1280 x 960 frame (224 us per 128 frames); this moves the same 2.4 MB in
// a fraction of that.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_lm_spread2(const u32 vblock, const u8* __restrict__ q0, int qpitch, int w, int h, u8* __restrict__ lm0,
                                                     size_t q_slot_stride, size_t lm_slot_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* q = slot_ptr_s(q0, q_slot_stride, slot);
    u8* lm = slot_ptr_s(lm0, lm_slot_stride, slot);
    const int ng = w >> 5, W = w >> 1;
    const u32 wh = (u32)W * (u32)(h >> 1);
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int yp = gid / ng, g = gid - yp * ng;       // row pair, 32-pixel group
    const int y = 2 * yp;
    if (y >= h) return;
    u32 R[3][9];                                      // rows y, y + 1, y + 2: 32 bytes + the next dword
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const bool in = y + j < h;                    // below the image: zeros (the OR is one-sided, in-bounds only)
        const u8* row = q + (size_t)min(y + j, h - 1) * qpitch + 32 * g;
        const u32x4 a = ld16(row), b = ld16(row + 16);
        const u32 nx = g + 1 < ng ? *reinterpret_cast<const u32*>(row + 32) : 0u;
        R[j][0] = in ? a[0] : 0u; R[j][1] = in ? a[1] : 0u; R[j][2] = in ? a[2] : 0u; R[j][3] = in ? a[3] : 0u;
        R[j][4] = in ? b[0] : 0u; R[j][5] = in ? b[1] : 0u; R[j][6] = in ? b[2] : 0u; R[j][7] = in ? b[3] : 0u;
        R[j][8] = in ? nx : 0u;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        u32 sp[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const u32 v0 = R[r][d] | R[r + 1][d], v1 = R[r][d + 1] | R[r + 1][d + 1];
            sp[d] = v0 | __builtin_amdgcn_alignbyte(v1, v0, 1u);        // x and x + 1
        }
        // even / odd x of 8 dwords -> 4 + 4 dwords
        u32 ev[4], od[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ev[k] = __builtin_amdgcn_perm(sp[2 * k + 1], sp[2 * k], 0x06040200u);
            od[k] = __builtin_amdgcn_perm(sp[2 * k + 1], sp[2 * k], 0x07050301u);
        }
        if (y + r < h) {
            u8* base = lm + (size_t)(2 * r) * wh + (size_t)yp * W + 16 * g;   // memory (r, 0); (r, 1) is wh further
            st16(base, u32x4{ev[0], ev[1], ev[2], ev[3]});
            st16(base + wh, u32x4{od[0], od[1], od[2], od[3]});
        }
    }
}
__global__ __launch_bounds__(256) void k_lm_spread2(const u8* __restrict__ q0, int qpitch, int w, int h, u8* __restrict__ lm0,
                                                     size_t q_slot_stride, size_t lm_slot_stride, int gblocks, int nslots) {
    d_lm_spread2(blockIdx.x, q0, qpitch, w, h, lm0, q_slot_stride, lm_slot_stride, gblocks, nslots);
}

// ------------------------------------------------------------------------------------------------
// a11-a13  HOT KERNEL.  One wave per (template, chunk of 1008 positions): lane l < 63 owns positions
// [16 l, 16 l + 16) of the chunk.  For every feature the wave reads 1 KiB contiguous from the
// feature's linear memory at a wave-uniform byte offset (scalar-loaded from the bank), rounded down
// to a dword so the load runs at full rate; the 0..3 byte shift is undone in registers
// (v_alignbyte_b32 with the scalar shift; the 17th..19th byte comes from the next lane by DPP
// wave_shl:1, which is why lane 63 only feeds lane 62).  The realigned dwords are added byte-wise:
// four u32 adds carry sixteen u8 lanes, 63 features x 4 = 252 never overflows a byte.
// Modalities are then widened to u16 and summed (a12) and compared with the raw threshold (a13)
// without ever materialising the similarity map.  Feature lists are padded to a multiple of 8 with
// offsets into the arena's zero block so the inner loop has no tail.
// ------------------------------------------------------------------------------------------------
// Workgroup -> (slot, items) mapping.  XCD_MAP: a 1-D grid whose block b is assumed to run on XCD b % 8
// (observed round-robin dispatch; only speed depends on it): each XCD then works on one frame slot at a
// time, so its 4 MB L2 holds that frame's 1.2 MB of linear memories + the bank instead of all slots'.
// ------------------------------------------------------------------------------------------------
// a6-a10 for T = 5, spread memory only (level 0 of the RGB-D configuration), batches: a streaming pass in registers.
// spread(y, x) = OR of the 5 x 5 block at (y, x); memory g = (y % 5) * 5 + x % 5 holds it at (y / 5) * W + x / 5.
// One lane = (band of 5 output rows, 8 positions of every one of the 25 memories): nine source rows of 48 bytes
// (40 pixels + 4 of halo, rounded up to dwords), horizontal OR-5 on dwords (v_alignbyte), vertical OR-5 over the last
// five rows, then for every x phase the 8 bytes at stride 5 by v_perm.  The tiled k_lm_fast<5, ..> goes through LDS
// with byte reads and index divisions: 26 vector instructions per pixel against about 5 here.
// Needs W % 8 == 0 (so a row ends after 40 or 48 of a lane's source bytes) and dword-aligned rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 gather4_stride5(const u32 (&V)[11], int b0) {
    // bytes b0, b0 + 5, b0 + 10, b0 + 15 of the 44-byte row V (v_perm selectors: 0..3 = second operand, 4..7 = first)
    const int a0 = b0, a1 = b0 + 5, a2 = b0 + 10, a3 = b0 + 15;
    const u32 lo = __builtin_amdgcn_perm(V[a1 >> 2], V[a0 >> 2], (u32)(a0 & 3) | ((u32)(4 + (a1 & 3)) << 8) | 0x0c0c0000u);
    const u32 hi = __builtin_amdgcn_perm(V[a3 >> 2], V[a2 >> 2], (u32)(a2 & 3) | ((u32)(4 + (a3 & 3)) << 8) | 0x0c0c0000u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
__device__ __forceinline__ void d_lm_spread5(const u32 vblock, const u8* __restrict__ q0, int qpitch, int w, int h, u8* __restrict__ lm0,
                                             size_t q_slot_stride, size_t lm_slot_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* q = slot_ptr_s(q0, q_slot_stride, slot);
    u8* lm = slot_ptr_s(lm0, lm_slot_stride, slot);
    const int W = w / 5, HB = h / 5, ng = W >> 3;
    const u32 wh = (u32)W * (u32)HB;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int band = gid / ng, g = gid - band * ng;
    if (band >= HB) return;
    const int x0 = 40 * g, y0 = 5 * band;
    const bool full = x0 + 48 <= w;                       // else the row ends after 40 of the lane's bytes
    u32 Hr[9][11];                                        // horizontal OR-5 of source rows y0 .. y0 + 8 (static indices)
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        u32 R[12];
#pragma unroll
        for (int d = 0; d < 12; ++d) R[d] = 0;
        if (y0 + r < h) {
            const u8* row = q + (size_t)(y0 + r) * qpitch + x0;
            const u32x4 a = ld16a4(row), b = ld16a4(row + 16);
            R[0] = a[0]; R[1] = a[1]; R[2] = a[2]; R[3] = a[3]; R[4] = b[0]; R[5] = b[1]; R[6] = b[2]; R[7] = b[3];
            if (full) { const u32x4 c = ld16a4(row + 32); R[8] = c[0]; R[9] = c[1]; R[10] = c[2]; R[11] = c[3]; }
            else { const u32x2 c = ld8a4(row + 32); R[8] = c[0]; R[9] = c[1]; }
        }
#pragma unroll
        for (int d = 0; d < 11; ++d)
            Hr[r][d] = R[d] | __builtin_amdgcn_alignbyte(R[d + 1], R[d], 1u) | __builtin_amdgcn_alignbyte(R[d + 1], R[d], 2u) |
                       __builtin_amdgcn_alignbyte(R[d + 1], R[d], 3u) | R[d + 1];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        u32 V[11];
#pragma unroll
        for (int d = 0; d < 11; ++d) V[d] = Hr[j][d] | Hr[j + 1][d] | Hr[j + 2][d] | Hr[j + 3][d] | Hr[j + 4][d];
        u8* dst = lm + (size_t)(j * 5) * wh + (size_t)band * W + 8 * g;
#pragma unroll
        for (int c0 = 0; c0 < 5; ++c0)
            *reinterpret_cast<u32x2*>(dst + (size_t)c0 * wh) = u32x2{gather4_stride5(V, c0), gather4_stride5(V, c0 + 20)};
    }
}
__global__ __launch_bounds__(256) void k_lm_spread5(const u8* __restrict__ q0, int qpitch, int w, int h, u8* __restrict__ lm0,
                                                     size_t q_slot_stride, size_t lm_slot_stride, int gblocks, int nslots) {
    d_lm_spread5(blockIdx.x, q0, qpitch, w, h, lm0, q_slot_stride, lm_slot_stride, gblocks, nslots);
}


}  // namespace
