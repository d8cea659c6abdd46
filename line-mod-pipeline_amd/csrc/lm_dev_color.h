// lm_dev_color.h -- a3 / a4, the colour modality: cv::pyrDown of the BGR source, GaussianBlur 7x7 (one-shot, row-walking,
// and on the matrix cores), Sobel + max-channel + orientation + 3x3 vote (fused row-walking k_cgrad / k_cgrad_levels, streaming k_corient +
// k_cvote, LDS-tiled k_color_quantize).  Kernels and their device functions; included by lm_k_preprocess.hip only (the level-fused
// kernels there -- k_phase, k_bphase, k_bsplit -- call the device functions of all three modality headers).
// All arithmetic is integer / byte except the fastAtan2 polynomial, which uses explicit round-to-nearest intrinsics in the oracle's operation
// order: bit-identical to oracle/linemod_oracle.cpp.
#pragma once
#include "lm_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a4  cv::pyrDown, CV_8UC3: 5x5 [1 4 6 4 1]^2, BORDER_REFLECT_101, (sum + 128) >> 8
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyrdown(const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                                  int dw, int dh, size_t slot_stride) {
    const u8* src = slot_ptr(src0, slot_stride);
    u8* dst = slot_ptr(dst0, slot_stride);
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int K[5] = {1, 4, 6, 4, 1};
    int xs[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) xs[i] = refl101(2 * x + i - 2, sw) * 3;
    int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const u8* row = src + (size_t)refl101(2 * y + j - 2, sh) * sw * 3;
        int r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const u8* p = row + xs[i];
            r0 += K[i] * p[0]; r1 += K[i] * p[1]; r2 += K[i] * p[2];
        }
        s0 += K[j] * r0; s1 += K[j] * r1; s2 += K[j] * r2;
    }
    u8* o = dst + ((size_t)y * dw + x) * 3;
    o[0] = (u8)((s0 + 128) >> 8); o[1] = (u8)((s1 + 128) >> 8); o[2] = (u8)((s2 + 128) >> 8);
}

// cv::pyrDown, one lane = 8 output pixels (sw % 16 == 0): the 20 source pixels 16g-2 .. 16g+17 are 60 bytes
// at byte 10 of five aligned 16-byte blocks starting at 48 g - 16.  Vertical 1 4 6 4 1 first, on packed
// bytes (u16 pairs, sums <= 4080), then the horizontal taps on the extracted 16-bit sums.
__device__ __forceinline__ void d_pyrdown8(const u32 vblock, const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                                   int dw, int dh, size_t slot_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* src = slot_ptr_s(src0, slot_stride, slot);
    u8* dst = slot_ptr_s(dst0, slot_stride, slot);
    const int ng = dw >> 3;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int y = gid / ng, g = gid - y * ng;
    if (y >= dh) return;
    const u32 K[5] = {1, 4, 6, 4, 1};
    u32 ev[20], od[20];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const u8* row = src + (size_t)refl101(2 * y + j - 2, sh) * sw * 3 + 48 * g - 16;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            // block 0 only holds pixels 16g-2, 16g-1 and block 4 only 16g+16, 16g+17: replaced below at the row ends
            const bool ok = !(k == 0 && g == 0) && !(k == 4 && g == ng - 1);
            const u32x4 d = ok ? *reinterpret_cast<const u32x4*>(row + 16 * k) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 e = d[q] & 0x00FF00FFu, o = (d[q] >> 8) & 0x00FF00FFu;
                if (j == 0) { ev[4 * k + q] = e; od[4 * k + q] = o; }
                else { ev[4 * k + q] += K[j] * e; od[4 * k + q] += K[j] * o; }
            }
        }
    }
    int V[19][3];   // window pixel i = source x 16g - 2 + i
#pragma unroll
    for (int i = 0; i < 19; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int bb = 10 + 3 * i + c;
            const u32 d = (bb & 1) ? od[bb >> 2] : ev[bb >> 2];
            V[i][c] = (int)((bb & 2) ? (d >> 16) : (d & 0xFFFFu));
        }
    if (g == 0) {           // BORDER_REFLECT_101: x = -2 -> 2, -1 -> 1
#pragma unroll
        for (int c = 0; c < 3; ++c) { V[0][c] = V[4][c]; V[1][c] = V[3][c]; }
    }
    if (g == ng - 1) {      // x = sw -> sw - 2
#pragma unroll
        for (int c = 0; c < 3; ++c) V[18][c] = V[16][c];
    }
    u32 o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sum = V[2 * p][c] + 4 * V[2 * p + 1][c] + 6 * V[2 * p + 2][c] + 4 * V[2 * p + 3][c] + V[2 * p + 4][c];
            const int bi = 3 * p + c;
            o[bi >> 2] |= (u32)((sum + 128) >> 8) << (8 * (bi & 3));
        }
    u32x2* out = reinterpret_cast<u32x2*>(dst + ((size_t)y * dw + 8 * g) * 3);
    out[0] = u32x2{o[0], o[1]}; out[1] = u32x2{o[2], o[3]}; out[2] = u32x2{o[4], o[5]};
}
__global__ __launch_bounds__(256) void k_pyrdown8(const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                                   int dw, int dh, size_t slot_stride, int gblocks, int nslots) {
    d_pyrdown8(blockIdx.x, src0, sw, sh, dst0, dw, dh, slot_stride, gblocks, nslots);
}

// cv::pyrDown for batches (r03): a lane owns 16 source pixels (48 bytes = three aligned blocks -> 8 output pixels = 24 bytes)
// for a strip of PD_STRIP output rows and walks down one output row at a time.  k_pyrdown8 loads 5 rows x 5 blocks for every
// 8 output pixels and unpacks all of it (29 vector instructions per output byte); here
//   * the source rows live in a ring of interleaved row PAIRS (as in k_cblur_sh): output row y needs rows 2y-2 .. 2y+2 = two
//     pairs and the first row of the third, output row y+1 re-uses two of the three, so a step loads two new rows;
//   * the vertical taps 1 4 6 4 are ONE v_dot4 per byte column on the 4 x 4 byte transpose of the two pairs, the fifth row's
//     byte is a second v_dot4 with a one-byte selector that accumulates onto it;
//   * the two source pixels to the left and the one to the right that the horizontal taps need are column sums of the
//     ADJACENT lanes (DPP wave_shr / wave_shl; lanes 0 and 63 of a wave only feed, 62 (strip, segment) pairs per wave,
//     strip-major); a lane at a row end takes BORDER_REFLECT_101 from its own sums.
// Same integers as k_pyrdown8 / k_pyrdown: vertical sums <= 4080, (sum + 128) >> 8.
#define PD_STRIP 16
template <int STRIP>
__device__ __forceinline__ void d_pyrdown16_st(const u32 slot, const u32 tile, const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                               int dw, int dh, size_t slot_stride) {
    const u8* src = slot_ptr_s(src0, slot_stride, slot);
    u8* dst = slot_ptr_s(dst0, slot_stride, slot);
    const int ng = sw >> 4, total = ng * ((dh + STRIP - 1) / STRIP);
    const int lane = (int)(threadIdx.x & 63u);
    const int f0 = ((int)tile * 4 + (int)(threadIdx.x >> 6)) * 62 - 1;    // pair of lane 0 (a feeder)
    if (f0 + 1 >= total) return;
    const bool writer = lane >= 1 && lane <= 62 && f0 + lane < total;
    const int f = clampi(f0 + lane, 0, total - 1);
    const int strip = f / ng, g = f - strip * ng;
    const int y0 = strip * STRIP, y1 = min(y0 + STRIP, dh);
    const bool first = g == 0, last = g == ng - 1;
    const bool edge_wave = __any(first || last);
    const u32 pitch = (u32)sw * 3u, so = 48u * (u32)g;
    const u32 W1464 = 1u | (4u << 8) | (6u << 16) | (4u << 24);
    auto row_off = [&](int r) { return (u32)refl101(r, sh) * pitch + so; };
    u32 A[12][2], B[12][2], C0[12];     // pairs (2y-2, 2y-1), (2y, 2y+1) interleaved; row 2y+2 raw
    {
        const u8* p0 = src + row_off(2 * y0 - 2); const u8* p1 = src + row_off(2 * y0 - 1);
        const u8* p2 = src + row_off(2 * y0);     const u8* p3 = src + row_off(2 * y0 + 1);
        const u8* p4 = src + row_off(2 * y0 + 2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const u32x4 r0 = ld16(p0 + 16 * k), r1 = ld16(p1 + 16 * k), r2 = ld16(p2 + 16 * k), r3 = ld16(p3 + 16 * k), r4 = ld16(p4 + 16 * k);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                A[4 * k + q][0] = __builtin_amdgcn_perm(r1[q], r0[q], 0x05010400u); A[4 * k + q][1] = __builtin_amdgcn_perm(r1[q], r0[q], 0x07030602u);
                B[4 * k + q][0] = __builtin_amdgcn_perm(r3[q], r2[q], 0x05010400u); B[4 * k + q][1] = __builtin_amdgcn_perm(r3[q], r2[q], 0x07030602u);
                C0[4 * k + q] = r4[q];
            }
        }
    }
#pragma unroll 1
    for (int y = y0;; ++y) {
        const bool more = y + 1 < y1;
        // the two rows the next step adds (2y+3, 2y+4) are requested before this step's arithmetic
        u32x4 n3[3], n4[3];
        {
            const u8* p3 = src + row_off(2 * y + 3); const u8* p4 = src + row_off(2 * y + 4);
#pragma unroll
            for (int k = 0; k < 3; ++k) { n3[k] = ld16(p3 + 16 * k); n4[k] = ld16(p4 + 16 * k); }
        }
        // vertical 1 4 6 4 1 per byte column
        u32 V[48];
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            const u32 T0 = __builtin_amdgcn_perm(B[d][0], A[d][0], 0x05040100u), T1 = __builtin_amdgcn_perm(B[d][0], A[d][0], 0x07060302u);
            const u32 T2 = __builtin_amdgcn_perm(B[d][1], A[d][1], 0x05040100u), T3 = __builtin_amdgcn_perm(B[d][1], A[d][1], 0x07060302u);
            V[4 * d + 0] = __builtin_amdgcn_udot4(C0[d], 0x00000001u, __builtin_amdgcn_udot4(T0, W1464, 0u, false), false);
            V[4 * d + 1] = __builtin_amdgcn_udot4(C0[d], 0x00000100u, __builtin_amdgcn_udot4(T1, W1464, 0u, false), false);
            V[4 * d + 2] = __builtin_amdgcn_udot4(C0[d], 0x00010000u, __builtin_amdgcn_udot4(T2, W1464, 0u, false), false);
            V[4 * d + 3] = __builtin_amdgcn_udot4(C0[d], 0x01000000u, __builtin_amdgcn_udot4(T3, W1464, 0u, false), false);
        }
        // E[i + 6] = column sum of the lane's byte i, i = -6 .. 50: two pixels from the left neighbour, one from the right
        u32 EL[6], ER[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) EL[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)V[42 + k], 0x138, 0xf, 0xf, true);   // lane - 1
#pragma unroll
        for (int k = 0; k < 3; ++k) ER[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)V[k], 0x130, 0xf, 0xf, true);        // lane + 1
        if (edge_wave) {
            // BORDER_REFLECT_101: pixel -2 -> 2 (bytes 6 .. 8), -1 -> 1 (bytes 3 .. 5); pixel sw -> sw - 2 (the lane's pixel 14: bytes 42 .. 44)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                EL[c] = first ? V[6 + c] : EL[c];
                EL[3 + c] = first ? V[3 + c] : EL[3 + c];
                ER[c] = last ? V[42 + c] : ER[c];
            }
        }
        u32 o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int i = 6 * p + c;                                     // centre byte of output pixel p, channel c
                const u32 a = i - 6 >= 0 ? V[i - 6] : EL[i], b = i - 3 >= 0 ? V[i - 3] : EL[i + 3];
                const u32 e = i + 6 < 48 ? V[i + 6] : ER[i + 6 - 48], dd = V[i + 3];
                const u32 sum = a + e + 4u * (b + dd) + 6u * V[i] + 128u;
                const int bi = 3 * p + c;
                o[bi >> 2] |= (sum >> 8) << (8 * (bi & 3));
            }
        if (writer) {
            u32x2* out = reinterpret_cast<u32x2*>(dst + ((size_t)y * dw + 8 * g) * 3);
            out[0] = u32x2{o[0], o[1]}; out[1] = u32x2{o[2], o[3]}; out[2] = u32x2{o[4], o[5]};
        }
        if (!more) return;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 4 * k + q;
                A[d][0] = B[d][0]; A[d][1] = B[d][1];
                B[d][0] = __builtin_amdgcn_perm(n3[k][q], C0[d], 0x05010400u); B[d][1] = __builtin_amdgcn_perm(n3[k][q], C0[d], 0x07030602u);
                C0[d] = n4[k][q];
            }
    }
}
template <int STRIP>
__device__ __forceinline__ void d_pyrdown16(const u32 vblock, const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                            int dw, int dh, size_t slot_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    d_pyrdown16_st<STRIP>(slot, tile, src0, sw, sh, dst0, dw, dh, slot_stride);
}
template <int STRIP>
__global__ __launch_bounds__(256, 2) void k_pyrdown16(const u8* __restrict__ src0, int sw, int sh, u8* __restrict__ dst0,
                                                      int dw, int dh, size_t slot_stride, int gblocks, int nslots) {
    d_pyrdown16<STRIP>(blockIdx.x, src0, sw, sh, dst0, dw, dh, slot_stride, gblocks, nslots);
}

// ------------------------------------------------------------------------------------------------
// a3  ColorGradient::process.  One 32x8 output tile per 256-thread workgroup (1200 workgroups at
// 640x480 so several are resident per CU and hide each other's barriers).  The 7x7 blur (+-3), the
// Sobel (+-1) and the vote (+-1) need a 5-pixel halo, all staged through LDS.
// ------------------------------------------------------------------------------------------------
#define CT_W 32
#define CT_H 8
#define RAW_W (CT_W + 10)   // 42 px = 126 B per row
#define RAW_H (CT_H + 10)   // 18
#define RAW_PITCH 128
#define SM_W (CT_W + 4)     // 36
#define SM_H (CT_H + 4)     // 12
#define Q_W (CT_W + 2)      // 34
#define Q_H (CT_H + 2)      // 10

// cv::fastAtan2 polynomial in degrees -- same operation order as oracle fast_atan2_deg().
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180.0 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180.0 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180.0 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180.0 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    // c = min / (max + eps) and ONE polynomial: the two branches of the reference (ay / (ax + eps) when ax >= ay, else
    // ax / (ay + eps)) are this same quotient, so the if-converted code need not carry two correctly rounded divisions
    const bool steep = !(ax >= ay);
    const float mn = steep ? ax : ay, mx = steep ? ay : ax;
    const float c = __fdiv_rn(mn, __fadd_rn(mx, eps));
    const float c2 = __fmul_rn(c, c);
    float a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (steep) a = __fsub_rn(90.f, a);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__global__ __launch_bounds__(256) void k_color_quantize(const u8* __restrict__ bgr0, int w, int h, float thr2,
                                                         u8* __restrict__ quant0, float* __restrict__ mag0,
                                                         size_t slot_stride) {
    __shared__ __attribute__((aligned(16))) u8 raw[RAW_H][RAW_PITCH];
    __shared__ u16 hb[RAW_H][SM_W * 3];
    __shared__ u8 sm[SM_H][SM_W * 3 + 4];
    __shared__ u8 qn[Q_H][Q_W + 2];
    const u8* bgr = slot_ptr(bgr0, slot_stride);
    u8* quant = slot_ptr(quant0, slot_stride);
    float* mag = mag0 ? slot_ptr(mag0, slot_stride) : nullptr;
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * CT_W, oy = blockIdx.y * CT_H;

    // ---- raw tile (replicate-clamped coordinates)
    const bool interior = (ox >= 5) && (ox + CT_W + 6 < w) && (oy >= 5) && (oy + CT_H + 5 <= h);
    if (interior) {
        // 18 rows x 32 dwords (126 B used): all loads issued before the LDS stores
        const u8* base = bgr + ((size_t)(oy - 5) * w + (ox - 5)) * 3;
        u32 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int i = tid + k * 256;
            int r = i >> 5, c = i & 31;
            v[k] = (i < RAW_H * 32) ? ld4u(base + (size_t)r * w * 3 + 4 * c) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int i = tid + k * 256;
            if (i < RAW_H * 32) reinterpret_cast<u32*>(&raw[i >> 5][0])[i & 31] = v[k];
        }
    } else {
        for (int i = tid; i < RAW_H * RAW_W; i += 256) {
            int ry = i / RAW_W, rx = i - ry * RAW_W;
            int gy = clampi(oy - 5 + ry, 0, h - 1), gx = clampi(ox - 5 + rx, 0, w - 1);
            const u8* p = bgr + ((size_t)gy * w + gx) * 3;
            raw[ry][rx * 3 + 0] = p[0]; raw[ry][rx * 3 + 1] = p[1]; raw[ry][rx * 3 + 2] = p[2];
        }
    }
    __syncthreads();
    // ---- horizontal 7-tap {8,28,56,72,56,28,8} at the CLAMPED centre column (Sobel replicates the
    // smoothed image, so smoothed(-1) must equal smoothed(0), not a blur centred outside)
    for (int i = tid; i < RAW_H * SM_W; i += 256) {
        int ry = i / SM_W, tx = i - ry * SM_W;
        int cx = (clampi(ox - 2 + tx, 0, w - 1) - (ox - 5)) * 3;
        const u8* r = &raw[ry][0];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int s = 8 * (r[cx - 9 + c] + r[cx + 9 + c]) + 28 * (r[cx - 6 + c] + r[cx + 6 + c]) +
                    56 * (r[cx - 3 + c] + r[cx + 3 + c]) + 72 * r[cx + c];
            hb[ry][tx * 3 + c] = (u16)s;
        }
    }
    __syncthreads();
    for (int i = tid; i < SM_H * SM_W; i += 256) {
        int ty = i / SM_W, tx = i - ty * SM_W;
        int cy = clampi(oy - 2 + ty, 0, h - 1) - (oy - 5);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int k = tx * 3 + c;
            u32 s = 8u * (hb[cy - 3][k] + hb[cy + 3][k]) + 28u * (hb[cy - 2][k] + hb[cy + 2][k]) +
                    56u * (hb[cy - 1][k] + hb[cy + 1][k]) + 72u * hb[cy][k];
            sm[ty][k] = (u8)((s + 32768u) >> 16);
        }
    }
    __syncthreads();
    // ---- Sobel (CV_16S) on the three channels, strongest channel, orientation, 16 -> 8 bins
    const float scale = (float)(16.0 / 360.0);
    for (int i = tid; i < Q_H * Q_W; i += 256) {
        int qy = i / Q_W, qx = i - qy * Q_W;
        int gy = oy - 1 + qy, gx = ox - 1 + qx;
        u8 out = 0;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            int bdx = 0, bdy = 0, bm = -1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int a00 = sm[qy][qx * 3 + c], a01 = sm[qy][qx * 3 + 3 + c], a02 = sm[qy][qx * 3 + 6 + c];
                int a10 = sm[qy + 1][qx * 3 + c], a12 = sm[qy + 1][qx * 3 + 6 + c];
                int a20 = sm[qy + 2][qx * 3 + c], a21 = sm[qy + 2][qx * 3 + 3 + c], a22 = sm[qy + 2][qx * 3 + 6 + c];
                int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
                int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
                int m = dx * dx + dy * dy;
                // upstream cascade: B if >= both, else G if >= both, else R  ==  first maximum wins ties
                if (m > bm) { bm = m; bdx = dx; bdy = dy; }
            }
            float ang = fast_atan2_deg((float)bdy, (float)bdx);
            float qf = rintf(__fadd_rn(__fmul_rn(ang, scale), 0.0f));
            int q = (int)qf;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            bool border = (gy == 0) | (gy == h - 1) | (gx == 0) | (gx == w - 1);
            out = border ? 0 : (u8)(q & 7);
            float fm = (float)bm;
            if (fm > thr2) out |= 0x80;
            if (mag && qy >= 1 && qy <= CT_H && qx >= 1 && qx <= CT_W) mag[(size_t)gy * w + gx] = fm;
        }
        qn[qy][qx] = out;
    }
    __syncthreads();
    // ---- 3x3 majority vote (>= 5 of 9) gated by magnitude: one output pixel per thread
    {
        int ty = tid >> 5, tx = tid & 31;
        int gy = oy + ty, gx = ox + tx;
        if (gy < h && gx < w) {
            u8 res = 0;
            if (gy >= 1 && gy <= h - 2 && gx >= 1 && gx <= w - 2 && (qn[ty + 1][tx + 1] & 0x80)) {
                u32 cnt = 0;  // eight 4-bit counters
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii) cnt += 1u << (4 * (qn[ty + j][tx + ii] & 7));
                int best = 0, idx = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    int v = (cnt >> (4 * b)) & 15;
                    if (best < v) { best = v; idx = b; }
                }
                if (best >= 5) res = (u8)(1u << idx);
            }
            quant[(size_t)gy * w + gx] = res;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a3, streaming form (used when w % 16 == 0; the fused LDS-tiled k_color_quantize above is the generic
// fallback and the reference for the arithmetic).  Three passes through scratch:
//   k_cblur    7x7 Gaussian, vertical taps on raw bytes (u16 pairs) then horizontal taps with the final
//              rounding, both in registers                                     -> S   u8 [h][3w]
//   k_corient  3x3 Sobel on S (replicate), strongest channel, fastAtan2, 16 -> 8 bins, magnitude flag
//                                                                              -> qn  u8 [h][w]
//   k_cvote    3x3 majority vote gated by the flag                             -> quant
//
// Load shape.  The vector L1 takes one cycle per 4 lanes of a load instruction whatever the width per
// lane (measured: TCP_TOTAL_CACHE_ACCESSES = 16 per wave-load + one per 128-B line crossed), so a
// byte or short load per lane runs at 1/16 .. 1/8 of the rate of a 16-byte one.  Every pass
// therefore gives a lane 16 contiguous bytes per load (global_load_dwordx4) and 8 or 16 outputs.
// Intermediates are kept small on purpose (8-bit S instead of 16-bit partial sums): with 32+ frames in
// flight they do not fit the L2s and every byte written here is fabric traffic.
// ------------------------------------------------------------------------------------------------

// a1+a2  GaussianBlur 7x7 -> S (the smoothed 8-bit image).  One lane = 16 bytes of a row x 2 rows.
// Vertical taps first, on the raw bytes: the 8 source rows of the two output rows form two groups of four; a 4x4
// byte transpose (8 v_perm_b32 per 4 columns) puts the four rows of a column into one dword and v_dot4_u32_u8
// applies four taps at once (weights {8,28,56,72,56,28,8} split over the two groups, shifted by one row for the
// second output row).  Then the horizontal taps on the 16-bit column sums of the lane's 40-byte window (bytes
// -12 .. +27 around the block: the taps of byte p are bytes p-9, p-6, ..., p+9 whatever the channel) with the final
// rounding.  The two separable passes are exact integer sums, so their order does not matter.
#define CB_ROWS 2
__device__ __forceinline__ void d_cblur(const u32 vblock, const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0,
                                                size_t in_stride, size_t tmp_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* bgr = slot_ptr_s(bgr0, in_stride, slot);
    u8* S = slot_ptr_s(s0, tmp_stride, slot);
    const int nblk = (w * 3) >> 4;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int band = gid / nblk, b = gid - band * nblk;
    const int y0 = band * CB_ROWS;
    if (y0 >= h) return;
    const size_t pitch = (size_t)w * 3;
    u32 W[8][10];                                          // source row y0 - 3 + i (BORDER_REPLICATE), window dwords
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u8* row = bgr + (size_t)clampi(y0 - 3 + i, 0, h - 1) * pitch;
        const u32x4 c = ld16(row + 16 * b);
        if (b > 0) {
            const u32x4 p = ld16(row + 16 * b - 16);
            W[i][0] = p[1]; W[i][1] = p[2]; W[i][2] = p[3];
        } else {   // bytes -12..-1 replicate pixel 0 channel-wise: [B G R B][G R B G][R B G R]
            W[i][0] = __builtin_amdgcn_perm(c[0], c[0], 0x00020100u);
            W[i][1] = __builtin_amdgcn_perm(c[0], c[0], 0x01000201u);
            W[i][2] = __builtin_amdgcn_perm(c[0], c[0], 0x02010002u);
        }
        W[i][3] = c[0]; W[i][4] = c[1]; W[i][5] = c[2]; W[i][6] = c[3];
        if (b + 1 < nblk) {
            const u32x4 n = ld16(row + 16 * b + 16);
            W[i][7] = n[0]; W[i][8] = n[1]; W[i][9] = n[2];
        } else {   // bytes 3w.. replicate the last pixel (bytes 1..3 of the last dword): [B G R B][G R B G][R B G R]
            W[i][7] = __builtin_amdgcn_perm(c[3], c[3], 0x01030201u);
            W[i][8] = __builtin_amdgcn_perm(c[3], c[3], 0x02010302u);
            W[i][9] = __builtin_amdgcn_perm(c[3], c[3], 0x03020103u);
        }
    }
    // tap weights as bytes (byte i = row i of the group): output row 0 uses rows 0..6, output row 1 rows 1..7
    const u32 wA0 = 8u | (28u << 8) | (56u << 16) | (72u << 24), wB0 = 56u | (28u << 8) | (8u << 16);
    const u32 wA1 = (8u << 8) | (28u << 16) | (56u << 24), wB1 = 72u | (56u << 8) | (28u << 16) | (8u << 24);
    u32 vb[CB_ROWS][40];                                   // column sums per window byte (<= 65280)
#pragma unroll
    for (int d = 0; d < 10; ++d) {
        u32 T[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {                      // 4 x 4 byte transpose of rows 4g .. 4g+3, columns 4d .. 4d+3
            const u32 r0 = W[4 * g][d], r1 = W[4 * g + 1][d], r2 = W[4 * g + 2][d], r3 = W[4 * g + 3][d];
            const u32 x0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), x1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const u32 z0 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), z1 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            T[g][0] = __builtin_amdgcn_perm(z0, x0, 0x05040100u); T[g][1] = __builtin_amdgcn_perm(z0, x0, 0x07060302u);
            T[g][2] = __builtin_amdgcn_perm(z1, x1, 0x05040100u); T[g][3] = __builtin_amdgcn_perm(z1, x1, 0x07060302u);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vb[0][4 * d + c] = __builtin_amdgcn_udot4(T[0][c], wA0, __builtin_amdgcn_udot4(T[1][c], wB0, 0u, false), false);
            vb[1][4 * d + c] = __builtin_amdgcn_udot4(T[0][c], wA1, __builtin_amdgcn_udot4(T[1][c], wB1, 0u, false), false);
        }
    }
    const u32 K[7] = {8, 28, 56, 72, 56, 28, 8};
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        u32 o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 packed = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // output byte 4j + q of the block = window byte 12 + 4j + q; taps at window bytes wb - 9 .. wb + 9
                u32 acc = 32768u;
#pragma unroll
                for (int t = 0; t < 7; ++t) acc += K[t] * vb[r][12 + 4 * j + q + 3 * t - 9];
                packed |= (acc >> 16) << (8 * q);
            }
            o4[j] = packed;
        }
        st16(S + (size_t)y * pitch + 16 * b, u32x4{o4[0], o4[1], o4[2], o4[3]});
    }
}
__global__ __launch_bounds__(256) void k_cblur(const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0,
                                                size_t in_stride, size_t tmp_stride, int gblocks, int nslots) {
    d_cblur(blockIdx.x, bgr0, w, h, s0, in_stride, tmp_stride, gblocks, nslots);
}

#define CBS_STRIP 16     // rows per strip of the row-walking blur (k_cblur_sh) for images of up to 640 rows
// a1+a2, sliding window with the COLUMN SUMS SHARED between neighbouring lanes (r03).  r02's sliding-window kernel (k_cblur_sw,
// deleted in r05: it lost its A/B to this one in r03 and was the default nowhere) gave every lane the
// whole 40-byte window of its 16 output bytes: it loads three blocks per row and runs the vertical pass (4 x 4 byte
// transposes + v_dot4) on ten window dwords for four output dwords -- 2.5 x the vertical work and 3 x the loads.  Here a
// lane loads and sums ONLY its own block; the nine column sums to the left and to the right of it come from the adjacent
// lanes by DPP (wave_shr / wave_shl), exactly like the neighbour labels of k_cgrad: lanes 0 and 63 of a wave only feed, a
// wave covers 62 consecutive (strip, block) pairs numbered strip-major, and a lane at a row end takes the replicated
// border bytes from its own sums instead (a wave-uniform branch: only waves that hold a row end pay the selects).
// Same arithmetic, same rounding: vertical taps {8, 28, 56, 72, 56, 28, 8} on bytes, horizontal taps on the 16-bit sums, one
// round-half-up at the end.  The ring is 4 pairs x 4 dwords (32 registers instead of 80).
__device__ __forceinline__ void cbx_pair(u32 (&pr)[4][2], const u32x4& r0, const u32x4& r1) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        pr[d][0] = __builtin_amdgcn_perm(r1[d], r0[d], 0x05010400u);
        pr[d][1] = __builtin_amdgcn_perm(r1[d], r0[d], 0x07030602u);
    }
}
template <int STRIP>
__device__ __forceinline__ void d_cblur_sh_st(const u32 slot, const u32 tile, const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0,
                                              size_t in_stride, size_t tmp_stride) {
    const u8* bgr = slot_ptr_s(bgr0, in_stride, slot);
    u8* S = slot_ptr_s(s0, tmp_stride, slot);
    const int nblk = (w * 3) >> 4, total = nblk * ((h + STRIP - 1) / STRIP);
    const int lane = (int)(threadIdx.x & 63u);
    const int f0 = ((int)tile * 4 + (int)(threadIdx.x >> 6)) * 62 - 1;    // pair of lane 0 (a feeder)
    if (f0 + 1 >= total) return;                                           // whole wave past the end
    const bool writer = lane >= 1 && lane <= 62 && f0 + lane < total;
    const int f = clampi(f0 + lane, 0, total - 1);
    const int strip = f / nblk, b = f - strip * nblk;
    const int y0 = strip * STRIP, y1 = min(y0 + STRIP, h);
    const bool first = b == 0, last = b == nblk - 1;
    const bool edge_wave = __any(first || last);
    const u32 pitch = (u32)w * 3u, bo = 16u * (u32)b;
    const u32 wA0 = 8u | (28u << 8) | (56u << 16) | (72u << 24), wB0 = 56u | (28u << 8) | (8u << 16);
    const u32 wA1 = (8u << 8) | (28u << 16) | (56u << 24), wB1 = 72u | (56u << 8) | (28u << 16) | (8u << 24);
    u32 ring[4][4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // the first window: source rows y0 - 3 .. y0 + 4
        const u32x4 r0 = ld16(bgr + ((u32)clampi(y0 - 3 + 2 * k, 0, h - 1) * pitch + bo));
        const u32x4 r1 = ld16(bgr + ((u32)clampi(y0 - 2 + 2 * k, 0, h - 1) * pitch + bo));
        cbx_pair(ring[k], r0, r1);
    }
    u32x4 n0 = ld16(bgr + ((u32)clampi(y0 + 5, 0, h - 1) * pitch + bo));   // the pair of the next step
    u32x4 n1 = ld16(bgr + ((u32)clampi(y0 + 6, 0, h - 1) * pitch + bo));
    // one step = two output rows; the ring's four pairs keep their registers and the step's code names them by (B + k) & 3, B = the
    // step number mod 4 known at compile time: the loop body is written four times (no moves between steps)
    int y = y0;
    auto step = [&](auto Bc) __attribute__((always_inline)) -> bool {
        constexpr int B = decltype(Bc)::value;
        const bool more = y + 2 < y1;
        // the pair of the step after next is requested before this step's arithmetic
        const u32x4 m0 = ld16(bgr + ((u32)clampi(y + 7, 0, h - 1) * pitch + bo));
        const u32x4 m1 = ld16(bgr + ((u32)clampi(y + 8, 0, h - 1) * pitch + bo));
        // vertical pass on the lane's own 16 byte columns: rows 4g .. 4g + 3 of one byte column per dword, two v_dot4 per output row
        u32 vb[2][16];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            u32 T[2][4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const u32 x0 = ring[(B + 2 * g) & 3][d][0], x1 = ring[(B + 2 * g) & 3][d][1], z0 = ring[(B + 2 * g + 1) & 3][d][0], z1 = ring[(B + 2 * g + 1) & 3][d][1];
                T[g][0] = __builtin_amdgcn_perm(z0, x0, 0x05040100u); T[g][1] = __builtin_amdgcn_perm(z0, x0, 0x07060302u);
                T[g][2] = __builtin_amdgcn_perm(z1, x1, 0x05040100u); T[g][3] = __builtin_amdgcn_perm(z1, x1, 0x07060302u);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                vb[0][4 * d + c] = __builtin_amdgcn_udot4(T[0][c], wA0, __builtin_amdgcn_udot4(T[1][c], wB0, 0u, false), false);
                vb[1][4 * d + c] = __builtin_amdgcn_udot4(T[0][c], wA1, __builtin_amdgcn_udot4(T[1][c], wB1, 0u, false), false);
            }
        }
        u32 o4[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // E[i + 9] = column sum of window byte i, i = -9 .. 24: nine from the left neighbour (its bytes 7 .. 15), the lane's
            // own sixteen, nine from the right neighbour (its bytes 0 .. 8)
            u32 E[34];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                E[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)vb[r][7 + k], 0x138, 0xf, 0xf, true);       // lane - 1
                E[25 + k] = (u32)__builtin_amdgcn_update_dpp(0, (int)vb[r][k], 0x130, 0xf, 0xf, true);     // lane + 1
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) E[9 + i] = vb[r][i];
            if (edge_wave) {
                // BORDER_REPLICATE: byte -9 + k of the row is channel k % 3 of pixel 0 (own bytes 0 .. 2), byte 16 + k past the row
                // end channel k % 3 of the last pixel (own bytes 13 .. 15)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    E[k] = first ? vb[r][k % 3] : E[k];
                    E[25 + k] = last ? vb[r][13 + k % 3] : E[25 + k];
                }
            }
            // horizontal taps two at a time: the column sums fit 16 bits (<= 255 * 256), so Q[i] = (E[i], E[i + 3]) packs the two taps a
            // v_dot2_u32_u16 multiplies; every Q serves three outputs (as taps 0-1, 2-3 and 4-5).  28 packs + 16 x (3 dot2 + 1 mad)
            // instead of 16 x (3 adds + 5 mads: 72 is not an inline constant)
            u32 Q[28];
#pragma unroll
            for (int i = 0; i < 28; ++i) Q[i] = E[i] | (E[i + 3] << 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 packed = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i0 = 4 * j + q;                 // output byte i0: taps at E[i0], E[i0 + 3], ..., E[i0 + 18]
                    u32 acc = mad24<8>(E[i0 + 18], 32768u);
                    acc = udot2_u16(Q[i0], 8u | (28u << 16), acc);
                    acc = udot2_u16(Q[i0 + 6], 56u | (72u << 16), acc);
                    acc = udot2_u16(Q[i0 + 12], 56u | (28u << 16), acc);
                    packed |= (acc >> 16) << (8 * q);
                }
                o4[r][j] = packed;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (writer && y + r < h) st16(S + ((u32)(y + r) * pitch + bo), u32x4{o4[r][0], o4[r][1], o4[r][2], o4[r][3]});
        if (!more) return false;
        cbx_pair(ring[B & 3], n0, n1);            /* the oldest pair's registers take the newest: no ring moves */
        n0 = m0; n1 = m1;
        y += 2;
        return true;
    };
#pragma unroll 1
    for (;;) {
        if (!step(std::integral_constant<int, 0>())) return;
        if (!step(std::integral_constant<int, 1>())) return;
        if (!step(std::integral_constant<int, 2>())) return;
        if (!step(std::integral_constant<int, 3>())) return;
    }
}
template <int STRIP>
__device__ __forceinline__ void d_cblur_sh(const u32 vblock, const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0,
                                           size_t in_stride, size_t tmp_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    d_cblur_sh_st<STRIP>(slot, tile, bgr0, w, h, s0, in_stride, tmp_stride);
}
template <int STRIP>
__global__ __launch_bounds__(256, 2) void k_cblur_sh(const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0,
                                                   size_t in_stride, size_t tmp_stride, int gblocks, int nslots) {
    d_cblur_sh<STRIP>(blockIdx.x, bgr0, w, h, s0, in_stride, tmp_stride, gblocks, nslots);
}
// ------------------------------------------------------------------------------------------------
// a1+a2 on the MATRIX CORES (r04 experiment; LM_TUNE_CBLUR_VARIANT = 4).  The 7-tap Gaussian of the 8-bit path is two
// banded-Toeplitz integer products with taps {8, 28, 56, 72, 56, 28, 8} that fit i8; v_mfma_i32_32x32x32_i8 runs beside the vector
// ALU, which is what every other kernel of the pipeline is short of.
//   horizontal:  C1[row][n] = sum_k (X[row][k] - 128) * Th[k][n]        X = raw bytes (A operand, xor 0x80), Th = the band over BYTE
//                columns (the channels interleave: tap t sits 3 (t - 3) bytes away), 32 output bytes from 64 input bytes = two
//                K-blocks.  C1 = S1 - 32768 with S1 the 8.8 row sum (taps sum to 256), so C1 fits 16 bits: hi = C1 >> 8 in
//                [-128, 126], lo = C1 & 255.
//   vertical:    the accumulator tile has its byte COLUMN on the lane and its 32 rows in the 16 registers, i.e. it is already the
//                A operand (as C1 transposed) of a product that sums over rows: Z[n][j] = sum_rho C1[rho][n] * Tv[rho][j] -- no LDS,
//                no lane movement.  hi and (lo - 128) go through the same Tv; 256 Zhi + Zlo + constants = sum_v sum_h w w x,
//                and the output byte is bits 16..23 of (that + 32768): the ONE rounding of the 8-bit path.
//                Output rows j of a step lie across the boundary of the previous and the current 32-row tile (rows -16 .. 15
//                of the current one), so a step is two K-blocks again and the walk down a strip recomputes nothing.
//   result:      lane = output row, registers = 4-byte groups of the 32 byte columns: one dword store per group.
// Which k a lane's operand bytes stand for is the same function in A and B (both are built here), so only the documented
// C/D map is relied on: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31.
// BORDER_REPLICATE: rows by clamping the row a lane loads; the 16 byte columns before / behind a row are built from the row's
// first / last pixel by v_perm (channels repeat with period 3).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// Tile shape (second form, r04): a product's 32 "rows" need not be 32 image rows -- the band is shift-invariant, so operand row
// m = 4 yy + cc stands for image row yy (of 8) and the 32-byte column chunk cc (of 4) of a 128-byte wide wave.  The four lanes of a
// load quad then read ONE row (16 bytes every 32: one 128-B line per quad instead of four -- the first form, 32 image rows per
// tile, ran at a quarter of the L1's rate and stored 8-byte pieces of 32 different lines), the vertical band becomes block-diagonal
// in cc (zeros where input and output chunk differ: the matrix cores have the time), a step is 8 new rows, and after two
// v_permlane32_swap a lane holds 16 consecutive output bytes: a store instruction writes 8 whole 128-B lines.
struct MxTab { u32 v[64][16]; };      // per lane: B0 | B1 (horizontal band, K-blocks 0 / 1) | BvP | BvC (vertical band, previous / current tile)
static constexpr int mx_tap(int t) { return t == 0 || t == 6 ? 8 : t == 1 || t == 5 ? 28 : t == 2 || t == 4 ? 56 : t == 3 ? 72 : 0; }
static constexpr int mx_w(int t) { return (t < 0 || t > 6) ? 0 : mx_tap(t); }
static constexpr MxTab mx_make_tab() {
    MxTab T{};
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 31, hh = lane >> 5;
        const int yo = n >> 2, cco = n & 3;                                  // as the vertical product's output column: (output row, chunk)
        for (int q = 0; q < 4; ++q) {
            u32 b0 = 0, b1 = 0, vp = 0, vc = 0;
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e, k = 16 * hh + j;
                const int d0 = k - 16 - n, d1 = k + 16 - n;                 // input byte minus output byte, K-block 0 / 1
                const int w0 = (d0 % 3 == 0) ? mx_w(d0 / 3 + 3) : 0, w1 = (d1 % 3 == 0) ? mx_w(d1 / 3 + 3) : 0;
                const int m = (j & 3) + 8 * (j >> 2) + 4 * hh;              // operand row of the accumulator tile this operand byte holds
                const int yy = m >> 2, cc = m & 3;
                const int wp = cc == cco ? mx_w(yy - yo - 1) : 0, wc = cc == cco ? mx_w(yy - yo + 7) : 0;
                b0 |= (u32)w0 << (8 * e); b1 |= (u32)w1 << (8 * e); vp |= (u32)wp << (8 * e); vc |= (u32)wc << (8 * e);
            }
            T.v[lane][q] = b0; T.v[lane][4 + q] = b1; T.v[lane][8 + q] = vp; T.v[lane][12 + q] = vc;
        }
    }
    return T;
}
__device__ const MxTab g_mx_tab = mx_make_tab();
#define MX_WAVE_BYTES 128   // byte columns per wave (4 chunks of 32)
__device__ __forceinline__ void permlane32_swap_lo(u32& a, u32& b) {   // a[lanes 32..63] <-> b[lanes 0..31]
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void d_cblur_mx_st(const u32 slot, const u32 tile, const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0, size_t in_stride,
                                              size_t tmp_stride, int gx, int strip_rows) {
    const u8* bgr = slot_ptr_s(bgr0, in_stride, slot);
    u8* S = slot_ptr_s(s0, tmp_stride, slot);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int yy = r >> 2, cc = r & 3;
    const int cx = (int)(tile % (u32)gx), sy = (int)(tile / (u32)gx);
    const int W3 = w * 3;
    const int c0 = (cx * 4 + wave) * MX_WAVE_BYTES;
    if (c0 >= W3) return;
    const int Y0 = sy * strip_rows, Y1 = min(Y0 + strip_rows, h);
    const u32 pitch = (u32)W3;
    i32x4 B0, B1, BP, BC;
    {
        const u32x4* tp = reinterpret_cast<const u32x4*>(g_mx_tab.v[lane]);
        const u32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
        B0 = i32x4{(int)t0[0], (int)t0[1], (int)t0[2], (int)t0[3]}; B1 = i32x4{(int)t1[0], (int)t1[1], (int)t1[2], (int)t1[3]};
        BP = i32x4{(int)t2[0], (int)t2[1], (int)t2[2], (int)t2[3]}; BC = i32x4{(int)t3[0], (int)t3[1], (int)t3[2], (int)t3[3]};
    }
    // this lane's chunk: output bytes [cch, cch + 32) of its row; operand bytes of K-block kb: cch - 16 + 32 kb + 16 hh .. + 15
    const int cch = c0 + 32 * cc;
    const bool active = cch < W3;
    const bool rep_l = active && cch == 0 && hh == 0;               // K-block 0: the 16 bytes before the row
    const bool rep_r = active && cch + 32 == W3 && hh == 1;         // K-block 1: the 16 bytes behind the row
    const bool edge_wave = __any(rep_l || rep_r);
    const u32 off0 = (u32)(!active ? 0 : rep_l ? 0 : cch - 16 + 16 * hh);
    const u32 off1 = (u32)(!active ? 0 : rep_r ? W3 - 16 : cch + 16 + 16 * hh);
    auto load_tile = [&](int yt, u32x4 (&A)[2]) {
        const u32 row = (u32)clampi(yt + yy, 0, h - 1) * pitch;
        u32x4 v0 = ld16(bgr + (row + off0)), v1 = ld16(bgr + (row + off1));
        if (edge_wave) {
            // BORDER_REPLICATE: the bytes before a row repeat the channels of pixel 0 (bytes 0, 1, 2 of the row: channel (t + 2) % 3 at
            // byte t - 16), the bytes behind it those of the last pixel (bytes 1, 2, 3 of the row's last dword: channel t % 3)
            const u32 dl = v0[0], dr = v1[3];
            const u32 l0 = __builtin_amdgcn_perm(dl, dl, 0x02010002u), l1 = __builtin_amdgcn_perm(dl, dl, 0x00020100u), l2 = __builtin_amdgcn_perm(dl, dl, 0x01000201u);
            const u32 r0 = __builtin_amdgcn_perm(dr, dr, 0x01030201u), r1 = __builtin_amdgcn_perm(dr, dr, 0x02010302u), r2 = __builtin_amdgcn_perm(dr, dr, 0x03020103u);
            v0 = rep_l ? u32x4{l0, l1, l2, l0} : v0;
            v1 = rep_r ? u32x4{r0, r1, r2, r0} : v1;
        }
        A[0] = u32x4{v0[0] ^ 0x80808080u, v0[1] ^ 0x80808080u, v0[2] ^ 0x80808080u, v0[3] ^ 0x80808080u};
        A[1] = u32x4{v1[0] ^ 0x80808080u, v1[1] ^ 0x80808080u, v1[2] ^ 0x80808080u, v1[3] ^ 0x80808080u};
    };
    // horizontal pass of one tile (8 rows x 4 chunks): the accumulator tile, packed to its hi / lo operand bytes
    auto horizontal = [&](const u32x4 (&A)[2], i32x4& Hi, i32x4& Lo) {
        i32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{(int)A[0][0], (int)A[0][1], (int)A[0][2], (int)A[0][3]}, B0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{(int)A[1][0], (int)A[1][1], (int)A[1][2], (int)A[1][3]}, B1, c, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32 r0 = (u32)c[4 * g], r1 = (u32)c[4 * g + 1], r2 = (u32)c[4 * g + 2], r3 = (u32)c[4 * g + 3];
            const u32 a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);   // (lo0, lo1, hi0, hi1)
            Lo[g] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
            Hi[g] = (int)__builtin_amdgcn_perm(a23, a01, 0x07060302u);
        }
    };
    u32x4 A[2], An[2];
    i32x4 hiP, loP, hiC, loC;
    // step m writes the rows Y0 + 8 m .. + 7: (output row of the step) yo = yy, previous tile = rows Y0 + 8 m - 4 .. + 3, current
    // tile = rows Y0 + 8 m + 4 .. + 11 (the table's taps: previous tile row yy is tap yy - yo - 1, current tile row yy tap yy - yo + 7)
    load_tile(Y0 - 4, A);
    horizontal(A, hiP, loP);
    const int steps = (Y1 - Y0 + 7) / 8;
    const int KC = 32768 + 128 * 256 + 32768 * 256;        // rounding + the two biases (lo - 128; C1 = S1 - 32768), taps sum to 256
    load_tile(Y0 + 4, A);
    const u32 so = (u32)(cch + 16 * hh);                     // after the swaps: lower lanes bytes 0 .. 15 of the chunk, upper lanes 16 .. 31
    for (int m = 0; m < steps; ++m) {
        if (m + 1 < steps) load_tile(Y0 + 8 * (m + 1) + 4, An);              // the next tile travels while this one is multiplied
        horizontal(A, hiC, loC);
        i32x16 zh = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        i32x16 zl = {KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC, KC};
        zh = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiP, BP, zh, 0, 0, 0);
        zh = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiC, BC, zh, 0, 0, 0);
        zl = __builtin_amdgcn_mfma_i32_32x32x32_i8(loP, BP, zl, 0, 0, 0);
        zl = __builtin_amdgcn_mfma_i32_32x32x32_i8(loC, BC, zl, 0, 0, 0);
        u32 D[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((u32)zh[4 * g + e] << 8) + (u32)zl[4 * g + e];
            const u32 a01 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), a23 = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);   // byte 2 of each
            D[g] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);             // bytes 8 g + 4 hh .. + 3 of the chunk
        }
        permlane32_swap_lo(D[0], D[2]);                      // lower lanes: D0 = bytes 0-3, D2 = 4-7; upper lanes: D0 = 16-19, D2 = 20-23
        permlane32_swap_lo(D[1], D[3]);                      // lower lanes: D1 = 8-11, D3 = 12-15; upper lanes: D1 = 24-27, D3 = 28-31
        const int yout = Y0 + 8 * m + yy;
        if (active && yout < Y1) st16(S + ((u32)yout * pitch + so), u32x4{D[0], D[2], D[1], D[3]});
        hiP = hiC; loP = loC;
        A[0] = An[0]; A[1] = An[1];
    }
}

__global__ __launch_bounds__(256, 2) void k_cblur_mx(const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0, size_t in_stride,
                                                     size_t tmp_stride, int gx, int gy, int strip_rows, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(blockIdx.x, (u32)(gx * gy), (u32)nslots, slot, tile);
    d_cblur_mx_st(slot, tile, bgr0, w, h, s0, in_stride, tmp_stride, gx, strip_rows);
}
// the matrix-core blur of level 0 and cv::pyrDown (k_pyrdown16's tiles, vector ALU) in ONE grid, a slot's tiles dealt out evenly like
// k_blur_pyr's: the two readers of the raw image run side by side on the slot's XCD, and the pyrDown's vector work fills the issue slots
// the matrix-core tiles leave idle
__global__ __launch_bounds__(256, 2) void k_blur_mx_pyr(const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0, u8* __restrict__ bgr1,
                                                        size_t slot_stride, int gx, int gy, int strip_rows, int g_pyr, int nslots) {
    u32 slot, tile;
    const u32 g_blur = (u32)(gx * gy), G = g_blur + (u32)g_pyr;
    xcd_slot_tile_b(blockIdx.x, G, (u32)nslots, slot, tile);
    const u32 p1 = (tile + 1u) * (u32)g_pyr / G, p0 = tile * (u32)g_pyr / G;     // pyrDown tiles among the first tile + 1 / tile
    if (p1 != p0) d_pyrdown16_st<PD_STRIP>(slot, p0, bgr0, w, h, bgr1, w >> 1, h >> 1, slot_stride);
    else d_cblur_mx_st(slot, tile - p0, bgr0, w, h, s0, slot_stride, slot_stride, gx, strip_rows);
}

// Level-0 blur AND cv::pyrDown of the same frames in ONE grid, interleaved per slot (r03; VERDICT r2 #2b "blur + pyrDown from
// one pass over the raw image", as far as it pays): both read the raw level-0 image, and launched apart they read it from
// HBM twice (k_pyrdown8: 110 MB per 96-frame launch of config 2, 590 MB per 128 frames of config 3 -- it runs at the HBM rate).
// Here a slot's tiles are [blur tiles | pyrDown tiles] back to back in the order its XCD takes them (xcd_slot_tile_b over
// the combined tile count), so the second reader finds the rows in that XCD's L2.  Both parts are of one register class.
template <int SB>
__global__ __launch_bounds__(256, 2) void k_blur_pyr(const u8* __restrict__ bgr0, int w, int h, u8* __restrict__ s0, u8* __restrict__ bgr1,
                                                     size_t slot_stride, int g_blur, int g_pyr, int nslots, int interleave) {
    u32 slot, tile;
    xcd_slot_tile_b(blockIdx.x, (u32)(g_blur + g_pyr), (u32)nslots, slot, tile);
    if (!interleave) {
        if (tile < (u32)g_blur) d_cblur_sh_st<SB>(slot, tile, bgr0, w, h, s0, slot_stride, slot_stride);
        else d_pyrdown16_st<PD_STRIP>(slot, tile - (u32)g_blur, bgr0, w, h, bgr1, w >> 1, h >> 1, slot_stride);
        return;
    }
    // r04: the two kinds of tiles of a slot are dealt out evenly (Bresenham), so that the pyrDown tile of a band of rows is dispatched
    // among the blur tiles of the same band and finds the rows in the XCD's L2 while they are hot -- back to back ([all blur | all
    // pyrDown]) the second reader came after the slot's 0.9 - 3.7 MB had left a 4 MB L2 shared with the other slots in flight.
    const u32 G = (u32)(g_blur + g_pyr);
    const u32 p1 = (tile + 1u) * (u32)g_pyr / G, p0 = tile * (u32)g_pyr / G;     // pyrDown tiles among the first tile + 1 / tile
    if (p1 != p0) d_pyrdown16_st<PD_STRIP>(slot, p0, bgr0, w, h, bgr1, w >> 1, h >> 1, slot_stride);
    else d_cblur_sh_st<SB>(slot, tile - p0, bgr0, w, h, s0, slot_stride, slot_stride);
}

// a2+a3  Sobel(S, BORDER_REPLICATE) + strongest channel + fastAtan2 + 16 -> 8 bins + magnitude flag.
// One lane = 16 pixels of a row (w % 16 == 0): rows y-1, y, y+1 of S, 48 bytes each plus the dword before
// and after.  The vertical halves VS = S(y-1) + 2 S(y) + S(y+1) and VD + 256 = S(y+1) + 256 - S(y-1) are
// formed on u16 pairs; window byte of pixel i (image x = 16g - 1 + i), channel c is 1 + 3i + c.
__device__ __forceinline__ void d_corient(const u32 vblock, const u8* __restrict__ s0, int w, int h, float thr2,
                                                  u8* __restrict__ qn0, float* __restrict__ mag0, size_t tmp_stride,
                                                  size_t mag_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* S = slot_ptr_s(s0, tmp_stride, slot);
    u8* qn = slot_ptr_s(qn0, tmp_stride, slot);
    float* mag = mag0 ? slot_ptr_s(mag0, mag_stride, slot) : nullptr;
    const int ng = w >> 4;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int y = gid / ng, g = gid - y * ng;
    if (y >= h) return;
    const size_t pitch = (size_t)w * 3;
    u32 R[3][14];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u8* row = S + (size_t)clampi(y - 1 + j, 0, h - 1) * pitch + 48 * g;
        const u32x4 a = ld16(row), b = ld16(row + 16), c = ld16(row + 32);
        R[j][0] = g > 0 ? *reinterpret_cast<const u32*>(row - 4) : 0u;
        R[j][1] = a[0]; R[j][2] = a[1]; R[j][3] = a[2]; R[j][4] = a[3];
        R[j][5] = b[0]; R[j][6] = b[1]; R[j][7] = b[2]; R[j][8] = b[3];
        R[j][9] = c[0]; R[j][10] = c[1]; R[j][11] = c[2]; R[j][12] = c[3];
        R[j][13] = g + 1 < ng ? *reinterpret_cast<const u32*>(row + 48) : 0u;
    }
    u32 vse[14], vso[14], vde[14], vdo[14];
#pragma unroll
    for (int d = 0; d < 14; ++d) {
        const u32 e0 = R[0][d] & 0x00FF00FFu, o0 = (R[0][d] >> 8) & 0x00FF00FFu;
        const u32 e1 = R[1][d] & 0x00FF00FFu, o1 = (R[1][d] >> 8) & 0x00FF00FFu;
        const u32 e2 = R[2][d] & 0x00FF00FFu, o2 = (R[2][d] >> 8) & 0x00FF00FFu;
        vse[d] = e0 + 2u * e1 + e2; vso[d] = o0 + 2u * o1 + o2;
        vde[d] = e2 + 0x01000100u - e0; vdo[d] = o2 + 0x01000100u - o0;
    }
#define LM_WIN(E, O, wb) ((int)((((wb) & 1) ? O[(wb) >> 2] : E[(wb) >> 2]) >> (((wb) & 2) ? 16 : 0)) & 0xFFFF)
    u32 out[4] = {0, 0, 0, 0};
    float fmv[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int x = 16 * g + p;
        // window pixels p (x-1), p+1 (x), p+2 (x+1); at the row ends the missing neighbour is the pixel itself
        int bdx = 0, bdy = 0, bm = -1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int sl, sr, dl, dr;
            if (p == 0) {
                sl = g == 0 ? LM_WIN(vse, vso, 1 + 3 * 1 + c) : LM_WIN(vse, vso, 1 + 3 * 0 + c);
                dl = g == 0 ? LM_WIN(vde, vdo, 1 + 3 * 1 + c) : LM_WIN(vde, vdo, 1 + 3 * 0 + c);
            } else {
                sl = LM_WIN(vse, vso, 1 + 3 * p + c);
                dl = LM_WIN(vde, vdo, 1 + 3 * p + c);
            }
            if (p == 15) {
                sr = g == ng - 1 ? LM_WIN(vse, vso, 1 + 3 * 16 + c) : LM_WIN(vse, vso, 1 + 3 * 17 + c);
                dr = g == ng - 1 ? LM_WIN(vde, vdo, 1 + 3 * 16 + c) : LM_WIN(vde, vdo, 1 + 3 * 17 + c);
            } else {
                sr = LM_WIN(vse, vso, 1 + 3 * (p + 2) + c);
                dr = LM_WIN(vde, vdo, 1 + 3 * (p + 2) + c);
            }
            const int dc = LM_WIN(vde, vdo, 1 + 3 * (p + 1) + c);
            const int dx = sr - sl;
            const int dy = dl + 2 * dc + dr - 1024;       // four biases of 256
            const int m = dx * dx + dy * dy;
            if (m > bm) { bm = m; bdx = dx; bdy = dy; }   // first maximum wins ties = upstream's >= cascade
        }
        const float scale = (float)(16.0 / 360.0);
        const float ang = fast_atan2_deg((float)bdy, (float)bdx);
        const float qf = rintf(__fadd_rn(__fmul_rn(ang, scale), 0.0f));
        int q = (int)qf;
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        const bool border = (y == 0) | (y == h - 1) | (x == 0) | (x == w - 1);
        u32 o = border ? 0u : (u32)(q & 7);
        const float fm = (float)bm;
        if (fm > thr2) o |= 0x80u;
        out[p >> 2] |= o << (8 * (p & 3));
        fmv[p] = fm;
    }
#undef LM_WIN
    st16(qn + (size_t)y * w + 16 * g, u32x4{out[0], out[1], out[2], out[3]});
    if (mag) {
        float* mo = mag + (size_t)y * w + 16 * g;
#pragma unroll
        for (int p = 0; p < 16; ++p) mo[p] = fmv[p];
    }
}
__global__ __launch_bounds__(256) void k_corient(const u8* __restrict__ s0, int w, int h, float thr2,
                                                  u8* __restrict__ qn0, float* __restrict__ mag0, size_t tmp_stride,
                                                  size_t mag_stride, int gblocks, int nslots) {
    d_corient(blockIdx.x, s0, w, h, thr2, qn0, mag0, tmp_stride, mag_stride, gblocks, nslots);
}

#define CVT_ROWS 4   // output rows per lane of k_cvote
// one lane = 16 pixels x CVT_ROWS rows (w % 16 == 0).  A pixel's label becomes a one-hot nibble counter
// (1 << 4 label); horizontal then vertical 3-sums give the eight 4-bit counts of the 3x3 window, and
// since at most one label can reach 5 of 9 votes, (cnt + 0x33333333) & 0x88888888 has at most one bit.
__device__ __forceinline__ void d_cvote(const u32 vblock, const u8* __restrict__ qn0, int w, int h, u8* __restrict__ quant0,
                                                size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* qn = slot_ptr_s(qn0, tmp_stride, slot);
    u8* quant = slot_ptr_s(quant0, out_stride, slot);
    const int ng = w >> 4;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int band = gid / ng, g = gid - band * ng;
    const int y0 = band * CVT_ROWS;
    if (y0 >= h) return;
    u32 hs[3][16];        // horizontal 3-sums of rows r-1, r, r+1 (ring)
    u32 flags[3][4];      // the rows' own bytes (bit 7 = magnitude flag)
#pragma unroll
    for (int i = 0; i < CVT_ROWS + 2; ++i) {              // image row y0 - 1 + i
        const int yy = clampi(y0 - 1 + i, 0, h - 1);      // clamped rows only feed outputs that are forced to 0
        const u8* row = qn + (size_t)yy * w + 16 * g;
        const u32x4 c = ld16(row);
        const u32 lft = g > 0 ? *reinterpret_cast<const u32*>(row - 4) : 0u;
        const u32 rgt = g + 1 < ng ? *reinterpret_cast<const u32*>(row + 16) : 0u;
        u32 oh[18];
        oh[0] = 1u << (((lft >> 24) & 7u) << 2);
#pragma unroll
        for (int k = 0; k < 16; ++k) oh[1 + k] = 1u << (((c[k >> 2] >> (8 * (k & 3))) & 7u) << 2);
        oh[17] = 1u << ((rgt & 7u) << 2);
#pragma unroll
        for (int k = 0; k < 16; ++k) hs[i % 3][k] = oh[k] + oh[k + 1] + oh[k + 2];
#pragma unroll
        for (int k = 0; k < 4; ++k) flags[i % 3][k] = c[k];
        if (i >= 2) {
            const int y = y0 + i - 2;                     // centre row = ring slot (i - 1) % 3
            if (y < h) {
                u32 o[4] = {0, 0, 0, 0};
                const bool yin = y >= 1 && y <= h - 2;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const u32 cnt = hs[0][k] + hs[1][k] + hs[2][k];
                    const u32 m = (cnt + 0x33333333u) & 0x88888888u;
                    const u32 fl = (flags[(i - 1) % 3][k >> 2] >> (8 * (k & 3))) & 0x80u;
                    const int x = 16 * g + k;
                    const bool ok = yin && x >= 1 && x <= w - 2 && fl && m;
                    const u32 res = ok ? (1u << ((u32)(__ffs((int)m) - 1) >> 2)) : 0u;
                    o[k >> 2] |= res << (8 * (k & 3));
                }
                st16(quant + (size_t)y * w + 16 * g, u32x4{o[0], o[1], o[2], o[3]});
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_cvote(const u8* __restrict__ qn0, int w, int h, u8* __restrict__ quant0,
                                                size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    d_cvote(blockIdx.x, qn0, w, h, quant0, tmp_stride, out_stride, gblocks, nslots);
}

// a2+a3 in one pass: k_corient + k_cvote for batches (S -> quant, no qn image in between).
//
// Arithmetic.  Both kernels above are bound by the vector ALU (95 + 21 instructions per pixel), not by memory, so
// this one is built around the instruction count:
//   * the three rows of S are unpacked once into even / odd byte pairs (u16 x 2 per dword) and everything up to the
//     gradient runs as packed 16-bit arithmetic on them.  The Sobel taps are +-3 BYTES apart whatever the channel, and
//     a 3-byte shift of an (even, odd) pair of registers is a register rename plus one v_alignbit;
//   * the squared magnitude of a byte position is one v_dot2_i32_i16 of its (dx, dy) pair with itself;
//   * the orientation label needs no arctangent.  For |dx|, |dy| <= 1020 (all a Sobel of 8-bit data can give) the
//     label of cv::fastAtan2 -> x 16/360 -> rint -> & 7 depends on the signs, on |dy| > |dx| and on which of three
//     intervals min / max falls into, and the interval bounds are the same in every octant: with
//         s = (1282 min > 255 max) + (1384 min > 925 max)          (255/1282 and 925/1384: the mediants of the
//         q = |dy| > |dx| ? 4 - s : s,   label = (sign(dx) != sign(dy) ? -q : q) & 7     neighbouring realised ratios)
//     the label is bit-identical to the float path for ALL 2041 x 2041 inputs (tests/test_orientation_rule.py checks
//     every pair against the oracle's float code).  The threshold flag (float)m > thr^2 is m > floor(thr^2).
// Shape.  A lane owns 16 pixels x a strip of STRIP rows and walks down one row at a time: S rows r-1, r stay unpacked
// in registers, row r+1 was requested one step earlier.  Each step yields the labels of row r as one-hot nibbles,
// their horizontal 3-sums (the two neighbour pixels come from the adjacent lanes with v_mov_b32_dpp wave_shr / wave_shl)
// and, from the 3-sums of rows r-2 .. r, the voted output row r-1.  Lanes 0 and 63 of a wave only feed their
// neighbours: a wave covers 62 consecutive (strip, segment) pairs, and the pairs are numbered strip-major, so a wave
// is always full (a neighbour from another strip only ever feeds column 0 or w-1, which is zero anyway).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_sub_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ u32 pk_add_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ u32 pk_max_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int dot2_i16(u32 p, u32 k) {   // p.lo * k.lo + p.hi * k.hi, k wave-uniform (VOP3P takes no literal)
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(p), "s"(k));
    return r;
}
__device__ __forceinline__ int dot2_self(u32 p) {   // lo * lo + hi * hi of an i16 pair
    int r;
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(p));
    return r;
}
// window dwords of one S row for a 16-pixel segment: bytes 48g - 4 .. 48g + 51 (14 dwords), unpacked
struct CgRow { u32 E[14], O[14]; };    // E[d] = bytes (4d, 4d + 2), O[d] = bytes (4d + 1, 4d + 3) of the window, as u16 pairs
struct CgRaw { u32x4 a, b, c; u32 l, r; };
__device__ __forceinline__ void cg_request(CgRaw& q, const u8* S, int y, int h, u32 pitch, u32 so, u32 lo, u32 ro) {
    const u32 row = (u32)clampi(y, 0, h - 1) * pitch;
    q.a = ld16(S + (row + so)); q.b = ld16(S + (row + so + 16u)); q.c = ld16(S + (row + so + 32u));
    q.l = *reinterpret_cast<const u32*>(S + (row + lo)); q.r = *reinterpret_cast<const u32*>(S + (row + ro));
}
__device__ __forceinline__ void cg_unpack(CgRow& u, const CgRaw& q) {
    const u32 R[14] = {q.l, q.a[0], q.a[1], q.a[2], q.a[3], q.b[0], q.b[1], q.b[2], q.b[3], q.c[0], q.c[1], q.c[2], q.c[3], q.r};
#pragma unroll
    for (int d = 0; d < 14; ++d) {
        u.E[d] = __builtin_amdgcn_perm(R[d], R[d], 0x0c020c00u);
        u.O[d] = __builtin_amdgcn_perm(R[d], R[d], 0x0c030c01u);
    }
}
// labels of one row: one-hot nibbles oh[p] = 1 << 4 label, flag word (bit 15 - p = magnitude of pixel p above the
// threshold).  lmask = 28, or 0 on the first / last image row (labels forced to 0 there); first / last: segment at
// the left / right image edge.
__device__ __forceinline__ void cg_labels(const CgRow& A, const CgRow& B, const CgRow& N, u32 lmask, bool first, bool last,
                                          int ithr, u32 (&oh)[16], u32& fw) {
    u32 vse[14], vso[14], vde[14], vdo[14];
#pragma unroll
    for (int d = 0; d < 14; ++d) {
        vse[d] = A.E[d] + N.E[d] + 2u * B.E[d]; vso[d] = A.O[d] + N.O[d] + 2u * B.O[d];      // <= 1020 per half
        vde[d] = pk_sub_i16(N.E[d], A.E[d]); vdo[d] = pk_sub_i16(N.O[d], A.O[d]);
    }
    // byte position wb = 4d + k of the window: dx = VS[wb + 3] - VS[wb - 3], dy = VD[wb - 3] + 2 VD[wb] + VD[wb + 3];
    // three bytes further / back from an even pair is (O[d].hi, O[d+1].lo) / O[d-1], from an odd pair E[d+1] / (E[d-1].hi, E[d].lo)
    u32 P[56];   // (dx, dy) i16 pair per byte position, window bytes 4 .. 51
    int M[56];
    // r05: dy = VD[wb - 3] + 2 VD[wb] + VD[wb + 3] = H[wb - 3] + H[wb] with H[wb] = VD[wb] + VD[wb + 3]: the pair sums are shared by neighbouring
    // positions -- per pair of registers 6 instead of 8 instructions (He: alignbit + add, Ho: add, dye: add, dyo: alignbit + add)
    u32 He[13], Ho[13];
#pragma unroll
    for (int d = 0; d <= 12; ++d) {
        He[d] = pk_add_i16(vde[d], __builtin_amdgcn_alignbit(vdo[d + 1], vdo[d], 16));     // even pair (4d, 4d + 2) + the positions 3 bytes further (4d + 3, 4d + 5)
        Ho[d] = pk_add_i16(vdo[d], vde[d + 1]);                                            // odd pair (4d + 1, 4d + 3) + (4d + 4, 4d + 6)
    }
#pragma unroll
    for (int d = 1; d <= 12; ++d) {
        const u32 dxe = pk_sub_i16(__builtin_amdgcn_alignbit(vso[d + 1], vso[d], 16), vso[d - 1]);
        const u32 dxo = pk_sub_i16(vse[d + 1], __builtin_amdgcn_alignbit(vse[d], vse[d - 1], 16));
        const u32 dye = pk_add_i16(Ho[d - 1], He[d]);                                      // H at (4d - 3, 4d - 1) = the odd pair d - 1
        const u32 dyo = pk_add_i16(__builtin_amdgcn_alignbit(He[d], He[d - 1], 16), Ho[d]);  // H at (4d - 2, 4d) = (even pair d - 1).hi, (even pair d).lo
        P[4 * d + 0] = __builtin_amdgcn_perm(dye, dxe, 0x05040100u); P[4 * d + 1] = __builtin_amdgcn_perm(dyo, dxo, 0x05040100u);
        P[4 * d + 2] = __builtin_amdgcn_perm(dye, dxe, 0x07060302u); P[4 * d + 3] = __builtin_amdgcn_perm(dyo, dxo, 0x07060302u);
#pragma unroll
        for (int k = 0; k < 4; ++k) M[4 * d + k] = dot2_self(P[4 * d + k]);
    }
    fw = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int wb = 4 + 3 * p;
        const int m0 = M[wb], m1 = M[wb + 1], m2 = M[wb + 2];
        const int bm = max(max(m0, m1), m2);
        const u32 p0 = P[wb], p1 = P[wb + 1], p2 = P[wb + 2];  // (values first: a ?: on the array elements selects addresses)
        u32 W = m1 >= m2 ? p1 : p2;
        W = m0 == bm ? p0 : W;                                 // first maximum wins ties = upstream's >= cascade
        const u32 Aa = pk_max_i16(W, pk_sub_i16(0u, W));       // (|dx|, |dy|) = (a, b)
        // q = how many of the four sector bounds of the first quadrant b / a exceeds: 255/1282, 925/1384, 1384/925, 1282/255 (the
        // min / max form of the header comment spelled out for both octants; equality is impossible for realisable a, b, so strict
        // and non-strict compares agree).  One v_dot2_i32_i16 per bound (k a - l b < 0), its sign bit shifted into a 4-bit word by
        // v_alignbit, one v_bcnt: 9 instructions instead of 13.
        u32 sg = 0;
        sg = __builtin_amdgcn_alignbit(sg, (u32)dot2_i16(Aa, 255u | (((u32)-1282 & 0xFFFFu) << 16)), 31);
        sg = __builtin_amdgcn_alignbit(sg, (u32)dot2_i16(Aa, 925u | (((u32)-1384 & 0xFFFFu) << 16)), 31);
        sg = __builtin_amdgcn_alignbit(sg, (u32)dot2_i16(Aa, 1384u | (((u32)-925 & 0xFFFFu) << 16)), 31);
        sg = __builtin_amdgcn_alignbit(sg, (u32)dot2_i16(Aa, 1282u | (((u32)-255 & 0xFFFFu) << 16)), 31);
        int q = __builtin_popcount(sg);
        q = ((W ^ (W >> 16)) & 0x8000u) ? -q : q;
        u32 sh = ((u32)q << 2) & lmask;
        if (p == 0) sh = first ? 0u : sh;
        if (p == 15) sh = last ? 0u : sh;
        oh[p] = 1u << sh;
        fw = __builtin_amdgcn_alignbit(fw, (u32)(ithr - bm), 31);   // fw = fw << 1 | (bm > ithr)
    }
}

#define CG_STRIP 16
template <int STRIP>
__device__ __forceinline__ void d_cgrad(const u32 vblock, const u8* __restrict__ s0, int w, int h, int ithr, u8* __restrict__ quant0,
                                        size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* S = slot_ptr_s(s0, tmp_stride, slot);
    u8* quant = slot_ptr_s(quant0, out_stride, slot);
    const int ns = w >> 4, total = ns * ((h + STRIP - 1) / STRIP);
    const int lane = (int)(threadIdx.x & 63u);
    const int f0 = ((int)tile * 4 + (int)(threadIdx.x >> 6)) * 62 - 1;   // pair of lane 0 (a feeder)
    if (f0 + 1 >= total) return;                                          // whole wave past the end
    const bool writer = lane >= 1 && lane <= 62 && f0 + lane < total;
    const int f = clampi(f0 + lane, 0, total - 1);
    const int strip = f / ns, g = f - strip * ns;
    const int y0 = strip * STRIP;
    const u32 pitch = (u32)w * 3u;
    const u32 so = 48u * (u32)g;
    const u32 lo = g > 0 ? so - 4u : so, ro = g + 1 < ns ? so + 48u : so + 44u;   // at the row ends: any valid dword (column 0 / w-1 is zero)
    const bool first = g == 0, last = g == ns - 1;
    // Rings of three: S rows (r-1, r, r+1), 3-sum rows (r-2, r-1, r), flag words.  The loop body is written three times
    // with the roles rotated by name, so that nothing is moved between steps.
    CgRow R[3];
    {
        CgRaw q0, q1;
        cg_request(q0, S, y0 - 2, h, pitch, so, lo, ro);
        cg_request(q1, S, y0 - 1, h, pitch, so, lo, ro);
        cg_unpack(R[0], q0); cg_unpack(R[1], q1);
    }
    CgRaw nx;
    cg_request(nx, S, y0, h, pitch, so, lo, ro);
    u32 H[3][16], F[3] = {0, 0, 0};
#pragma unroll
    for (int p = 0; p < 16; ++p) { H[0][p] = 0; H[1][p] = 0; H[2][p] = 0; }
    const u32 emask = 0xFFFFu & ~((first ? 0x8000u : 0u) | (last ? 1u : 0u));   // columns 0 and w-1 never vote
#define CG_STEP(K)                                                                                          \
    {                                                                                                       \
        const int r = y0 - 1 + t + (K);                        /* label row of this step */                 \
        cg_unpack(R[((K) + 2) % 3], nx);                                                                    \
        cg_request(nx, S, r + 2, h, pitch, so, lo, ro);        /* S row of the next step */                 \
        u32 oh[16];                                                                                         \
        cg_labels(R[(K) % 3], R[((K) + 1) % 3], R[((K) + 2) % 3], (r <= 0 || r >= h - 1) ? 0u : 28u, first, last, ithr, oh, F[((K) + 2) % 3]); \
        const u32 ohl = (u32)__builtin_amdgcn_update_dpp(0, (int)oh[15], 0x138, 0xf, 0xf, true);   /* lane - 1's pixel 15 */ \
        const u32 ohr = (u32)__builtin_amdgcn_update_dpp(0, (int)oh[0], 0x130, 0xf, 0xf, true);    /* lane + 1's pixel 0 */  \
        _Pragma("unroll") for (int p = 0; p < 16; ++p)                                                      \
            H[((K) + 2) % 3][p] = (p == 0 ? ohl : oh[p - 1]) + oh[p] + (p == 15 ? ohr : oh[p + 1]);         \
        const int y = r - 1;                                   /* output row: centre of label rows r-2, r-1, r */ \
        if (t + (K) >= 2 && y < h) {                                                                        \
            const u32 keep = (y >= 1 && y <= h - 2) ? (F[((K) + 1) % 3] & emask) : 0u;                      \
            u32 res[16];                                                                                    \
            _Pragma("unroll") for (int p = 0; p < 16; ++p) {                                                \
                const u32 cnt = H[0][p] + H[1][p] + H[2][p];                                                \
                const u32 m = (cnt + 0x33333333u) & 0x88888888u;   /* at most one nibble reaches 5 of 9 votes */ \
                /* no winner: ffs - 1 = -1 -> 1 << 31, whose byte 0 (all the packing below takes) is 0 */   \
                const u32 one = 1u << (((u32)(__ffs((int)m) - 1) >> 2) & 31u);                              \
                res[p] = one & (u32)__builtin_amdgcn_sbfe((int)keep, 15 - p, 1);                            \
            }                                                                                               \
            u32 o[4];                                                                                       \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                   \
                o[k] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(res[4 * k + 3], res[4 * k + 2], 0x0c0c0400u), \
                                             __builtin_amdgcn_perm(res[4 * k + 1], res[4 * k], 0x0c0c0400u), 0x05040100u); \
            if (writer) st16(quant + ((u32)y * (u32)w + 16u * (u32)g), u32x4{o[0], o[1], o[2], o[3]});      \
        }                                                                                                   \
    }
#pragma unroll 1
    for (int t = 0; t < STRIP + 2; t += 3) {
        CG_STEP(0)
        if (t + 1 >= STRIP + 2) break;
        CG_STEP(1)
        if (t + 2 >= STRIP + 2) break;
        CG_STEP(2)
    }
#undef CG_STEP
}
template <int STRIP>
__global__ __launch_bounds__(256, 2) void k_cgrad(const u8* __restrict__ s0, int w, int h, int ithr, u8* __restrict__ quant0,
                                                  size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    d_cgrad<STRIP>(blockIdx.x, s0, w, h, ithr, quant0, tmp_stride, out_stride, gblocks, nslots);
}

// r06: the orientation + vote passes of level 0 AND level 1 in ONE grid.  Level 1 of a 640 x 480 frame is 5 waves of 16-row strips (10
// of 8-row strips): a launch of its own leaves more than half of the SIMDs without a wave (k_cgrad<8>: 37 us per 96 frames at VALU busy 0.39 for a quarter of
// the pixels that cost k_cgrad<16> 68 us).  Behind level 0's workgroups in the same grid its strips fill the last round's idle SIMDs instead.  Needs the level-1
// blur BEFORE the level-0 gradient (lm_detector.hip enqueue_preprocess orders the launches so).
// S0 / S1: rows per strip of the two levels (level 0 as a launch of its own would choose; level 1: 8 -- short workgroups at the end of the grid, the tail is
// one of THEM long -- or 16 when level 1 alone brings enough waves to fill the chip)
template <int S0, int S1>
__global__ __launch_bounds__(256, 2) void k_cgrad_levels(const u8* __restrict__ s0, int w0, int h0, u8* __restrict__ q0, int g0,
                                                         const u8* __restrict__ s1, int w1, int h1, u8* __restrict__ q1, int g1,
                                                         int ithr, size_t slot_stride, int nslots) {
    const u32 e0 = (u32)g0 * (u32)nslots;
    if (blockIdx.x < e0) d_cgrad<S0>(blockIdx.x, s0, w0, h0, ithr, q0, slot_stride, slot_stride, g0, nslots);
    else d_cgrad<S1>(blockIdx.x - e0, s1, w1, h1, ithr, q1, slot_stride, slot_stride, g1, nslots);
}


}  // namespace
