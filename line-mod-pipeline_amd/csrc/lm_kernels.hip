// lm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the LINE-MOD match path.
//
// One kernel per upstream stage group:
//   k_pyrdown            a4      cv::pyrDown of the BGR source
//   k_color_quantize     a3      GaussianBlur 7x7 + Sobel + max-channel + fastAtan2 + 3x3 vote, LDS-tiled
//   k_depth_quantize     a5      bilateral normals + NORMAL_LUT + 5x5 median, LDS-tiled
//   k_linear_memories    a6-a10  NN pyrDown read + spread(T) + response LUT + linearize
//   k_scan               a11-a13 similarity scan of the lowest level fused with the threshold scan (HOT)
//   k_refine             a14     similarityLocal 16x16 + first-max argmax + rescore + threshold filter
//   k_sort_unique        a15     rank / bitonic sort in LDS + adjacent-unique (total order of SURVEY.md A.9)
//
// Every kernel takes the buffers of frame slot 0 plus the byte stride between slots and processes
// slot blockIdx.z, so a batch of resident frames is one launch per stage.
//
// All arithmetic is integer/byte except two float islands (fastAtan2 polynomial, normal
// normalisation) which use the explicit round-to-nearest intrinsics in the oracle's operation order,
// so every stage is bit-identical to oracle/linemod_oracle.cpp.  No MFMA: this is OR / LUT / u8 add.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <climits>
#include "lm_common.h"
#include "lm_kernels.h"
#include "lm_median25.h"

namespace {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U32x4U { u32x4 v; };
struct __attribute__((packed, aligned(1))) U32U { u32 v; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int refl101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}
__device__ __forceinline__ u32x4 ld16u(const u8* p) { return reinterpret_cast<const U32x4U*>(p)->v; }
// 16-byte-aligned wide accesses
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32 ld4u(const u8* p) { return reinterpret_cast<const U32U*>(p)->v; }
// dword-aligned wide loads: byte-misaligned vector loads are split by the memory pipeline and run at
// less than half rate on gfx950 (measured: 16.7 -> 7.6 us per frame for the scan), dword alignment is
// enough for full rate; the byte shift is applied in registers with v_alignbyte_b32.
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) U32x4A4 { u32x4 v; };
struct __attribute__((packed, aligned(4))) U32x2A4 { u32x2 v; };
__device__ __forceinline__ u32x4 ld16a4(const u8* p) { return reinterpret_cast<const U32x4A4*>(p)->v; }
__device__ __forceinline__ u32x2 ld8a4(const u8* p) { return reinterpret_cast<const U32x2A4*>(p)->v; }
struct __attribute__((packed, aligned(4))) U32x3A4 { u32 v[3]; };
__device__ __forceinline__ void ld12a4(const u8* p, u32& d0, u32& d1, u32& d2) { const U32x3A4 t = *reinterpret_cast<const U32x3A4*>(p); d0 = t.v[0]; d1 = t.v[1]; d2 = t.v[2]; }
// value of lane + 1 (0 for lane 63): v_mov_b32_dpp wave_shl:1 bound_ctrl:1 -- every lane is written, so the
// destination needs no initialisation
__device__ __forceinline__ u32 next_lane(u32 v) {
    return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true);
}

// a + b * K for a 24-bit unsigned b and a small constant K (v_mad_u32_u24: full rate; a plain 32-bit multiply is
// v_mul_lo_u32, quarter rate, and __umul24 makes the compiler mask both operands first).  K is an inline constant of
// the instruction (0..64); larger ones are split.
template <int K>
__device__ __forceinline__ u32 mad24(u32 b, u32 a) {
    if constexpr (K > 64) {
        return mad24<64>(b, mad24<K - 64>(b, a));
    } else {
        u32 r;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(b), "n"(K), "v"(a));
        return r;
    }
}

// p.lo * k.lo + p.hi * k.hi + c on unsigned 16-bit halves, k wave-uniform (VOP3P takes no literal: the taps live in an SGPR)
__device__ __forceinline__ u32 udot2_u16(u32 p, u32 k, u32 c) {
    u32 r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "s"(k), "v"(c));
    return r;
}

// a * b for factors that fit 24 signed bits: v_mul_i32_i24, full rate (v_mul_lo_u32 is quarter rate, and __mul24 goes
// through sign-extending shifts the compiler does not always fold)
__device__ __forceinline__ int mul_i24(int a, int b) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int mad_i24(int a, int b, int c) {   // a * b + c, factors within 24 signed bits
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// v_pk_max_u16 on two u16 pairs
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
    const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(u32, r);
}

template <typename T>
__device__ __forceinline__ T* slot_ptr(T* p, size_t slot_stride) {
    return reinterpret_cast<T*>(reinterpret_cast<u8*>(const_cast<typename std::remove_const<T>::type*>(p)) +
                                (size_t)blockIdx.z * slot_stride);
}

template <typename T>
__device__ __forceinline__ T* slot_ptr_s(T* p, size_t slot_stride, u32 slot) {
    return reinterpret_cast<T*>(reinterpret_cast<u8*>(const_cast<typename std::remove_const<T>::type*>(p)) +
                                (size_t)slot * slot_stride);
}
// 1-D grids of G tiles x B slots with XCD affinity: block b is assumed to run on XCD b % 8 (observed round-robin
// dispatch; only speed depends on it), so with B % 8 == 0 every tile of a slot runs on the same XCD and the
// slot's working set (linear memories, partially written lines) lives in ONE 4 MB L2 instead of eight.
__device__ __forceinline__ void xcd_slot_tile_b(u32 b, u32 G, u32 B, u32& slot, u32& tile) {
    if ((B & 7u) == 0) { const u32 x = b & 7u, k = b >> 3; slot = x + 8u * (k / G); tile = k - (k / G) * G; }
    else { slot = b / G; tile = b - slot * G; }
}
__device__ __forceinline__ void xcd_slot_tile(u32 G, u32 B, u32& slot, u32& tile) { xcd_slot_tile_b(blockIdx.x, G, B, slot, tile); }
// the block index at which a grid of G tiles x B slots would hold (slot, tile): lets a kernel that interleaves several parts per
// slot hand a part's tile to the part's device function, which decodes it with xcd_slot_tile_b again
__device__ __forceinline__ u32 xcd_vblock(u32 slot, u32 tile, u32 G, u32 B) {
    return (B & 7u) == 0 ? ((((slot >> 3) * G + tile) << 3) | (slot & 7u)) : slot * G + tile;
}

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

// ------------------------------------------------------------------------------------------------
// a5  DepthNormal::process -> quantizedNormals + medianBlur(5).  64x8 outputs per workgroup; the
// depth tile (+-7) and the raw normals (+-2) live in LDS.
// ------------------------------------------------------------------------------------------------
#define DT_W 64
#define DT_H 8
#define N_W (DT_W + 4)    // 68
#define N_H (DT_H + 4)    // 12
#define D_W (DT_W + 14)   // 78
#define D_H (DT_H + 14)   // 22
#define D_PITCH 80
#define D_LOADS ((D_H * D_W + 255) / 256)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define LM_CE(a, b) { us2 t_ = __builtin_elementwise_min(a, b); b = __builtin_elementwise_max(a, b); a = t_; }

__global__ __launch_bounds__(256) void k_depth_quantize(const u16* __restrict__ depth0, int w, int h, int dist_thr,
                                                         int diff_thr, const u8* __restrict__ lut,
                                                         u8* __restrict__ quant0, size_t slot_stride) {
    __shared__ u16 dt[D_H][D_PITCH];
    __shared__ u8 nt[N_H][N_W + 4];
    const u16* depth = slot_ptr(depth0, slot_stride);
    u8* quant = slot_ptr(quant0, slot_stride);
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * DT_W, oy = blockIdx.y * DT_H;
    // depth tile: 22 x 78 values, independent loads first
    {
        u16 v[D_LOADS];
#pragma unroll
        for (int k = 0; k < D_LOADS; ++k) {
            int i = tid + k * 256;
            int r = i / D_W, c = i - r * D_W;
            int gy = oy - 7 + r, gx = ox - 7 + c;
            v[k] = (i < D_H * D_W && gy >= 0 && gy < h && gx >= 0 && gx < w) ? depth[(size_t)gy * w + gx] : (u16)0;
        }
#pragma unroll
        for (int k = 0; k < D_LOADS; ++k) {
            int i = tid + k * 256;
            if (i < D_H * D_W) { int r = i / D_W; dt[r][i - r * D_W] = v[k]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < N_H * N_W; i += 256) {
        int ty = i / N_W, tx = i - ty * N_W;
        int gy = clampi(oy - 2 + ty, 0, h - 1), gx = clampi(ox - 2 + tx, 0, w - 1);  // medianBlur: BORDER_REPLICATE
        u8 out = 0;
        if (gy >= 5 && gy < h - 6 && gx >= 5 && gx < w - 6) {
            const int ly = gy - (oy - 7), lx = gx - (ox - 7);
            int d = dt[ly][lx];
            if (d < dist_thr) {
                int A0 = 0, A1 = 0, A3 = 0, b0 = 0, b1 = 0;
#pragma unroll
                for (int jj = -1; jj <= 1; ++jj)
#pragma unroll
                    for (int ii = -1; ii <= 1; ++ii) {
                        if (ii == 0 && jj == 0) continue;
                        int di = ii * 5, dj = jj * 5;
                        int delta = (int)dt[ly + dj][lx + di] - d;
                        int ad = delta < 0 ? -delta : delta;
                        int f = ad < diff_thr ? 1 : 0;
                        int fi = f * di, fj = f * dj;
                        A0 += fi * di; A1 += fi * dj; A3 += fj * dj;
                        b0 += fi * delta; b1 += fj * delta;
                    }
                // |b| <= 30 * 65535, A <= 150: the 2x2 solve fits 32 bits; the scaled normal needs 64
                int det = A0 * A3 - A1 * A1;
                int ddx = A3 * b0 - A1 * b1;
                int ddy = -A1 * b0 + A0 * b1;
                float nx = (float)(1150LL * ddx);
                float ny = (float)(1150LL * ddy);
                float nz = (float)(-(long long)det * d);
                float len = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
                if (len > 0) {
                    float inv = __fdiv_rn(1.0f, len);
                    nx = __fmul_rn(nx, inv); ny = __fmul_rn(ny, inv); nz = __fmul_rn(nz, inv);
                    int v1 = (int)__fadd_rn(__fmul_rn(nx, 10.f), 10.f);
                    int v2 = (int)__fadd_rn(__fmul_rn(ny, 10.f), 10.f);
                    int v3 = (int)__fadd_rn(__fmul_rn(nz, 20.f), 20.f);
                    int flat = v3 * 400 + v2 * 20 + v1;
                    out = (flat >= 0 && flat < 8000) ? lut[flat] : 0;
                }
            }
        }
        nt[ty][tx] = out;
    }
    __syncthreads();
    {
        // 5x5 median of two horizontally adjacent pixels per thread: a 132-exchange selection network
        // (lm_median25.h) on packed 16-bit lanes (v_pk_min_u16 / v_pk_max_u16)
        const int ty = tid >> 5, tx = (tid & 31) * 2;
        const int gy = oy + ty, gx = ox + tx;
        us2 v[25];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            unsigned short b[6];
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) b[ii] = nt[ty + j][tx + ii];
#pragma unroll
            for (int ii = 0; ii < 5; ++ii) { us2 p; p[0] = b[ii]; p[1] = b[ii + 1]; v[j * 5 + ii] = p; }
        }
        LM_MEDIAN25_NETWORK(v)
        const us2 med = v[LM_MEDIAN25_OUT];
        if (gy < h) {
            if (gx < w) quant[(size_t)gy * w + gx] = (u8)med[0];
            if (gx + 1 < w) quant[(size_t)gy * w + gx + 1] = (u8)med[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a5, streaming form (w % 8 == 0 and a NORMAL_LUT whose entries are 0 or one-hot, as upstream's is; the
// LDS-tiled k_depth_quantize above is the generic fallback and the reference for the arithmetic).
//   k_dnormal  one lane = 8 pixels of a row: the 3 x 3 taps at distance 5 come from three rows x three
//              aligned 16-byte blocks.  Writes the label's RANK CODE e, not the one-hot byte:
//              0 < 1 < 2 < 4 < ... < 128 are ranks 0..8; ranks 0..3 -> e = 8 rank, 4..7 -> 8 (rank-4) + 4,
//              8 -> 32, so that 1 << e (e < 32) is a one-hot NIBBLE counter word.
//   k_dmedian  5 x 5 median (BORDER_REPLICATE) by counting: horizontal 5-sums of the nibble words, split
//              into byte counters (ranks 0..3 | 4..7), vertical 5-sums, prefix sums by one multiply, and
//              the median is the first rank whose cumulative count reaches 13 (rank 8 if none does).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 pk_sub_u16_sat(u32 a, u32 b) {   // per half: max(a - b, 0)
    u32 r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (asm, not the vector builtins: the compiler turns min(x, 1) and 0 - f on packed halves into per-half compares and
// selects, three instructions for one)
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
    u32 r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 pk_mul_lo_u16(u32 a, u32 b) {   // per half: low 16 bits of a * b
    u32 r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 pk_sub_i16_op(u32 a, u32 b) {
    u32 r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 pk_add_i16_op(u32 a, u32 b) {
    u32 r;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a.half * b.half + c for the low (HI = false) or high halves of two packed i16 pairs
template <bool HI>
__device__ __forceinline__ int mad_i16h(u32 a, u32 b, int c) {
    int r;
    if (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ u32 hw_u16(const u32x4* A, int hw) {
    const u32 d = A[hw >> 3][(hw >> 1) & 3];
    return (hw & 1) ? (d >> 16) : (d & 0xFFFFu);
}

// float tail of a5 for one pixel: normal (1150 ddx, 1150 ddy, -det d), normalised, quantised, looked up; returns the
// label's rank code (see above), 0 for an invalid pixel.  Same operation order as the oracle.
// SMALL: |ddx * 1150| and |ddy * 1150| are known to stay below 2^31 (difference_threshold <= 249: |ddx| <= 125 * 60 *
// (threshold - 1)), so the products are exact in 32-bit integers and ONE int -> float conversion rounds them exactly like
// the double product rounded to float (both round the same exact integer to nearest even) -- two integer multiplies and
// conversions instead of six double-precision instructions per pixel.
// 1 / x and sqrt(x) of the float tail, for x = 0 or a NORMAL float whose reciprocal is normal too (here the squares' sum is 0
// or in [1, 2^82) and 1 <= len < 2^41: nx, ny, nz are integers below 2^40 in magnitude):
// the compiler's correctly rounded 1.0f / x is v_div_scale x 2, v_rcp, six fused steps, v_div_fmas, v_div_fixup -- scale
// and fixup only act on operands near the ends of the exponent range -- and its sqrtf scales denormal inputs around a
// v_sqrt_f32 and its +-1 ulp fix-up.  Without the range handling: 7 instructions instead of 11 for the reciprocal; the
// square root keeps the fix-up (v_sqrt_f32 alone is a 1-ulp instruction: r04) and drops only the scaling.
// lm_selftest_float_tail sweeps every float of the domain against the CORRECTLY ROUNDED 1.0f / x and sqrtf on the device
// (__builtin_sqrtf; NOT __fsqrt_rn, which this build maps to the bare v_sqrt_f32) -- tests/test_gpu_stages.py.
__device__ __forceinline__ float dn_rcp7(float d) {       // r03: v_rcp + six fused steps (the compiler's sequence without its range handling)
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float err0 = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(err0, r1, r1);
    const float err1 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(err1, r1, q1);
}
__device__ __forceinline__ float dn_sqrt_bare(float x) { return __builtin_amdgcn_sqrtf(x); }
// correctly rounded for x = 0 or a normal x: the hardware's root s is within 1 ulp, so the answer is s or a neighbour; with
// r(t) = x - t * s (one rounding), the root is below s iff r(s-) <= 0 and above it iff r(s+) > 0 (the compiler's own sqrtf
// fix-up without its denormal scaling).  x == 0 gives s == 0: both neighbours' residuals keep s.
__device__ __forceinline__ float dn_sqrt9(float x) {      // v_sqrt + the +-1 ulp fix-up (the compiler's sequence without its denormal scaling)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __builtin_bit_cast(float, __builtin_bit_cast(u32, s) - 1u);
    const float s_up = __builtin_bit_cast(float, __builtin_bit_cast(u32, s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    return r_up > 0.0f ? s_up : r;           // (x == 0: s = 0, s- is a NaN whose residual compares false, r(s+) = 0: stays 0)
}
// QUOT: the caller passes det / 625, ddx / 125, ddy / 125 (the packed taps' sums); with SMALL the two scalings of a component are one
// 24-bit multiply, |ddx / 125| <= 8 * 6 * 248 and 125 * 1150 = 143750 < 2^24.
// Shorter sequences (r04), adopted because the exhaustive sweep below finds NO float of the domain on which they differ from the
// correctly rounded results on this hardware (profiles/r04_float_tail_sweep.log): 3 + 5 instructions instead of 7 + 9.
// dn_sqrt(0) is a NaN here (0 * inf); the caller only asks `len > 0`, which is false for it as for 0.
__device__ __forceinline__ float dn_rcp3(float d) {          // v_rcp + ONE Newton step
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float dn_sqrt5(float x) {         // v_rsq + one coupled step: g = x y, h = y / 2, g + (x - g g) h
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = __fmul_rn(x, y), hf = __fmul_rn(0.5f, y);
    const float e = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(e, hf, g);
}
__device__ __forceinline__ float dn_sqrt4(float x) {         // v_sqrt + one step with the reciprocal root: s + (x - s s) (y / 2)
    const float s = __builtin_amdgcn_sqrtf(x), y = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, __fmul_rn(0.5f, y), s);
}
// candidate (r04): 1 / sqrt(x) rounded like 1.0f / sqrtf(x) from the SAME v_rsq the root uses -- one Newton step on the reciprocal of the
// (exact) root starting at y: no v_rcp.  Kept or dropped by the sweep below.
__device__ __forceinline__ float dn_inv_from_rsq(float x, float len) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-len, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float dn_rcp(float d) { return dn_rcp3(d); }
__device__ __forceinline__ float dn_sqrt(float x) { return dn_sqrt5(x); }
// `ok` = the pixel's depth passes the distance threshold (the row-end columns are masked by the caller, once per lane, r04).
template <bool SMALL, bool QUOT = false>
__device__ __forceinline__ u32 dn_label(int det, int ddx, int ddy, int d, bool ok, const u8* __restrict__ lut) {
    // same values as upstream's 64-bit integer products rounded once to float: |ddx| < 2^30 so the
    // double product is exact; |det * d| <= 22500 * 65535 < 2^31
    if (QUOT) { det = mul_i24(det, 625); if (!SMALL) { ddx = mul_i24(ddx, 125); ddy = mul_i24(ddy, 125); } }
    float nx = SMALL ? (float)(QUOT ? mul_i24(ddx, 143750) : ddx * 1150) : (float)((double)ddx * 1150.0);
    float ny = SMALL ? (float)(QUOT ? mul_i24(ddy, 143750) : ddy * 1150) : (float)((double)ddy * 1150.0);
    float nz = (float)(-mul_i24(det, d));
    const float len0 = dn_sqrt(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
    // all three components 0: dn_sqrt(0) is a NaN (0 * inf) and a float -> int conversion of a NaN is undefined -- the
    // select keeps every value below defined: with len = 1 the components stay 0, (v1, v2, v3) = (10, 10, 20) and the index 8210
    // is outside the table, whose entry 8000 is the code 0 the oracle's `len > 0` guard gives (so no second test at the end)
    const float len = len0 > 0.0f ? len0 : 1.0f;
    const float inv = dn_rcp(len);
    nx = __fmul_rn(nx, inv); ny = __fmul_rn(ny, inv); nz = __fmul_rn(nz, inv);
    const int v1 = (int)__fadd_rn(__fmul_rn(nx, 10.f), 10.f);
    const int v2 = (int)__fadd_rn(__fmul_rn(ny, 10.f), 10.f);
    const int v3 = (int)__fadd_rn(__fmul_rn(nz, 20.f), 20.f);
    const u32 flat = (u32)mad_i24(v3, 400, mad_i24(v2, 20, v1));   // |v| small: exact; negative = far outside as unsigned
    // the label's rank code straight from the second table (ensure_luts: 8 rank / 8 (rank - 4) + 4 / 32), whose entry 8000 is 0:
    // an index outside the table (nz == 0 gives v3 == 20) reads that instead of taking a compare and two selects
    const u32 ecode = lut[LMK_NORMAL_CODE_OFFSET + min(flat, 8000u)];
    return ok ? ecode : 0u;
}

// The same tail for the TWO pixels of a packed pair at once (r05): gfx950 multiplies, adds and fuses FP32 pairwise (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32 -- IEEE results, lane for lane the scalar instructions'), so the 20 multiply / add / fma of a pixel's tail become 10 per
// pixel; conversions, v_rsq / v_rcp, the index arithmetic and the table read stay per pixel.  Same operation order, same roundings as
// dn_label (-ffp-contract=off: nothing is fused that the scalar form does not fuse): bit-identical labels (tests/test_gpu_stages.py).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool SMALL>
__device__ __forceinline__ void dn_label2(int det0, int ddx0, int ddy0, int d0, bool ok0, int det1, int ddx1, int ddy1, int d1, bool ok1,
                                          const __amdgpu_buffer_rsrc_t lut_rsrc, u32& e0, u32& e1) {
    // QUOT form only (the packed-taps path): det / 625, ddx / 125, ddy / 125 come in
    det0 = mul_i24(det0, 625); det1 = mul_i24(det1, 625);
    if (!SMALL) { ddx0 = mul_i24(ddx0, 125); ddy0 = mul_i24(ddy0, 125); ddx1 = mul_i24(ddx1, 125); ddy1 = mul_i24(ddy1, 125); }
    f32x2 nx, ny, nz;
    nx[0] = SMALL ? (float)mul_i24(ddx0, 143750) : (float)((double)ddx0 * 1150.0);
    nx[1] = SMALL ? (float)mul_i24(ddx1, 143750) : (float)((double)ddx1 * 1150.0);
    ny[0] = SMALL ? (float)mul_i24(ddy0, 143750) : (float)((double)ddy0 * 1150.0);
    ny[1] = SMALL ? (float)mul_i24(ddy1, 143750) : (float)((double)ddy1 * 1150.0);
    nz[0] = (float)(-mul_i24(det0, d0)); nz[1] = (float)(-mul_i24(det1, d1));
    const f32x2 sq = (nx * nx + ny * ny) + nz * nz;
    // dn_sqrt5, pairwise: g = x y, h = y / 2, g + (x - g g) h
    f32x2 y; y[0] = __builtin_amdgcn_rsqf(sq[0]); y[1] = __builtin_amdgcn_rsqf(sq[1]);
    const f32x2 g = sq * y, hf = y * 0.5f;
    const f32x2 len0 = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, g, sq), hf, g);
    f32x2 len; len[0] = len0[0] > 0.0f ? len0[0] : 1.0f; len[1] = len0[1] > 0.0f ? len0[1] : 1.0f;     // (zero-length normal: see dn_label)
    // dn_rcp3, pairwise
    f32x2 r0; r0[0] = __builtin_amdgcn_rcpf(len[0]); r0[1] = __builtin_amdgcn_rcpf(len[1]);
    const f32x2 inv = __builtin_elementwise_fma(__builtin_elementwise_fma(-len, r0, (f32x2)(1.0f)), r0, r0);
    nx = nx * inv; ny = ny * inv; nz = nz * inv;
    const f32x2 t1 = nx * 10.f + 10.f, t2 = ny * 10.f + 10.f, t3 = nz * 20.f + 20.f;
    const u32 flat0 = (u32)mad_i24((int)t3[0], 400, mad_i24((int)t2[0], 20, (int)t1[0]));
    const u32 flat1 = (u32)mad_i24((int)t3[1], 400, mad_i24((int)t2[1], 20, (int)t1[1]));
    // (buffer loads: the table's base sits in a scalar resource descriptor and the index is the whole per-lane address -- no 64-bit
    // vector add per pixel; an index past the table reads its entry 8000 = code 0, see ensure_luts)
    const u32 c0 = (u32)__builtin_amdgcn_raw_buffer_load_b8(lut_rsrc, (int)(LMK_NORMAL_CODE_OFFSET + min(flat0, 8000u)), 0, 0);
    const u32 c1 = (u32)__builtin_amdgcn_raw_buffer_load_b8(lut_rsrc, (int)(LMK_NORMAL_CODE_OFFSET + min(flat1, 8000u)), 0, 0);
    e0 = ok0 ? c0 : 0u; e1 = ok1 ? c1 : 0u;
}

// every float of the tail's domain through dn_rcp / dn_sqrt and through the compiler's correctly rounded forms;
// out[2]: the bare v_sqrt_f32 against the same reference (information: how often the 1-ulp instruction is off)
__global__ __launch_bounds__(256) void k_selftest_float_tail(unsigned long long* __restrict__ out) {
    const u32 lo = 0x3F800000u, hi_rcp = (127u + 42u) << 23, hi_sqrt = (127u + 84u) << 23;   // 1.0f .. 2^42 / 2^84
    unsigned long long bad_rcp = 0, bad_sqrt = 0, bad_bare = 0, c_rcp7 = 0, c_sqrt9 = 0, c_sqrt4 = 0, c_inv = 0;
    for (u32 b = lo + blockIdx.x * 256u + threadIdx.x; b <= hi_sqrt; b += gridDim.x * 256u) {
        const float x = __builtin_bit_cast(float, b);
        if (b <= hi_rcp) bad_rcp += __builtin_bit_cast(u32, dn_rcp(x)) != __builtin_bit_cast(u32, __fdiv_rn(1.0f, x));
        const u32 want = __builtin_bit_cast(u32, __builtin_sqrtf(x));
        bad_sqrt += __builtin_bit_cast(u32, dn_sqrt(x)) != want;
        bad_bare += __builtin_bit_cast(u32, dn_sqrt_bare(x)) != want;
        if (b <= hi_rcp) c_rcp7 += __builtin_bit_cast(u32, dn_rcp7(x)) != __builtin_bit_cast(u32, __fdiv_rn(1.0f, x));
        c_sqrt9 += __builtin_bit_cast(u32, dn_sqrt9(x)) != want;
        c_sqrt4 += __builtin_bit_cast(u32, dn_sqrt4(x)) != want;
        c_inv += __builtin_bit_cast(u32, dn_inv_from_rsq(x, dn_sqrt(x))) != __builtin_bit_cast(u32, __fdiv_rn(1.0f, __builtin_sqrtf(x)));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bad_sqrt += dn_sqrt(0.0f) > 0.0f ? 1u : 0u;     // (0 or a NaN: what the caller's `len > 0` needs)
    if (bad_rcp) atomicAdd(&out[0], bad_rcp);
    if (bad_sqrt) atomicAdd(&out[1], bad_sqrt);
    if (bad_bare) atomicAdd(&out[2], bad_bare);
    if (c_rcp7) atomicAdd(&out[3], c_rcp7);
    if (c_sqrt9) atomicAdd(&out[4], c_sqrt9);
    if (c_sqrt4) atomicAdd(&out[5], c_sqrt4);
    if (c_inv) atomicAdd(&out[6], c_inv);
}

template <bool SMALL>
__device__ __forceinline__ void d_dnormal_t(const u32 vblock, const u16* __restrict__ depth0, int w, int h, int dist_thr, int diff_thr,
                                                  const u8* __restrict__ lut, u8* __restrict__ code0, size_t in_stride,
                                                  size_t tmp_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u16* depth = slot_ptr_s(depth0, in_stride, slot);
    u8* code = slot_ptr_s(code0, tmp_stride, slot);
    const int ng = w >> 3;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int y = gid / ng, g = gid - y * ng;
    if (y >= h) return;
    u32 out[2] = {0, 0};
    if (y >= 5 && y < h - 6) {
        u32x4 R[3][3];   // rows y-5, y, y+5; pixels 8g-8 .. 8g+15
        // (r04, tried and reverted: the row-end blocks loaded without branches or selects from the lane's own block (only the masked columns
        // would notice) with 32-bit offsets -- fewer instructions, k_dnormal 82.3 -> 87.1 us per 96-frame launch)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const u8* row = reinterpret_cast<const u8*>(depth + (size_t)(y + 5 * (j - 1)) * w) + 16 * g - 16;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool ok = !(k == 0 && g == 0) && !(k == 2 && g == ng - 1);
                R[j][k] = ok ? ld16(row + 16 * k) : u32x4{0, 0, 0, 0};
            }
        }
        if (diff_thr >= 0 && diff_thr <= 5461) {   // (a negative threshold gates every neighbour out: the per-pixel loop below gives f = 0 like the oracle)
            const __amdgpu_buffer_rsrc_t lut_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u8*>(lut), 0, 2 * 8000 + 16, 0x00020000);
            // PACKED taps: two pixels per instruction.  A dword of a depth row is a pixel pair, the neighbours five pixels
            // to the side are one v_alignbit away; |delta| by two saturating subtracts, the gate |delta| < diff_thr by a
            // third, and ci / cj / cx / sx / sy accumulate as i16 pairs (|sx| <= 6 (diff_thr - 1) < 2^15 needs
            // diff_thr <= 5461; larger thresholds take the per-pixel loop below).
            u32 D[3][12];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 12; ++k) D[j][k] = R[j][k >> 2][k & 3];
            const u32 THR = (u32)diff_thr * 0x00010001u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 C = D[1][4 + k];
                // f (0 / 1) and the gated delta of the eight neighbours, [jj + 1][ii + 1]; then the five sums from SHARED partial sums
                // (r04: the diagonal counts and the diagonal delta differences serve two accumulators each -- 16 adds instead of 28 --
                // and the gate is one multiply by f instead of a negate and an and)
                u32 F[3][3], G[3][3];
#pragma unroll
                for (int jj = -1; jj <= 1; ++jj)
#pragma unroll
                    for (int ii = -1; ii <= 1; ++ii) {
                        if (ii == 0 && jj == 0) continue;
                        const u32* Dr = D[jj + 1];
                        const u32 N = ii == 0 ? Dr[4 + k]
                                    : ii > 0 ? __builtin_amdgcn_alignbit(Dr[7 + k], Dr[6 + k], 16)
                                             : __builtin_amdgcn_alignbit(Dr[2 + k], Dr[1 + k], 16);
                        const u32 up = pk_sub_u16_sat(N, C), dn = pk_sub_u16_sat(C, N);      // one of them is 0
                        const u32 f = pk_min_u16(pk_sub_u16_sat(THR, up | dn), 0x00010001u);  // |delta| < diff_thr ? 1 : 0
                        F[jj + 1][ii + 1] = f;
                        G[jj + 1][ii + 1] = pk_mul_lo_u16(pk_sub_i16(up, dn), f);             // the gated delta
                    }
                const u32 fdp = F[2][2] + F[0][0], fdm = F[2][0] + F[0][2];                   // diagonals with ii jj > 0 / < 0
                const u32 cd = fdp + fdm;
                const u32 cx = pk_sub_i16(fdp, fdm);
                const u32 ci = F[1][2] + F[1][0] + cd, cj = F[2][1] + F[0][1] + cd;
                const u32 ga = pk_sub_i16(G[2][2], G[0][0]), gb = pk_sub_i16(G[0][2], G[2][0]);
                const u32 sx = pk_add_i16(pk_add_i16(pk_sub_i16(G[1][2], G[1][0]), ga), gb);   // sum of ii * gated delta
                const u32 sy = pk_sub_i16(pk_add_i16(pk_sub_i16(G[2][1], G[0][1]), ga), gb);   // sum of jj * gated delta
                const u32 ncx = pk_sub_i16(0u, cx);
                // det / 625, ddx / 125, ddy / 125 of the per-pixel loop, for the low and the high pixel of the pair; then the float tail of both at once
                const int d0 = (int)(C & 0xFFFFu), d1 = (int)(C >> 16);
                const int detq0 = mad_i16h<false>(ci, cj, mad_i16h<false>(ncx, cx, 0)), detq1 = mad_i16h<true>(ci, cj, mad_i16h<true>(ncx, cx, 0));
                const int ddxq0 = mad_i16h<false>(cj, sx, mad_i16h<false>(ncx, sy, 0)), ddxq1 = mad_i16h<true>(cj, sx, mad_i16h<true>(ncx, sy, 0));
                const int ddyq0 = mad_i16h<false>(ci, sy, mad_i16h<false>(ncx, sx, 0)), ddyq1 = mad_i16h<true>(ci, sy, mad_i16h<true>(ncx, sx, 0));
                u32 e0, e1;                                  // (valid = d < dist_thr; columns x < 5 and x >= w - 6: xmask below)
                dn_label2<SMALL>(detq0, ddxq0, ddyq0, d0, d0 < dist_thr, detq1, ddxq1, ddyq1, d1, d1 < dist_thr, lut_rsrc, e0, e1);
                out[k >> 1] |= (e0 | (e1 << 8)) << (16 * (k & 1));
            }
        } else {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // BRANCHLESS: every pixel runs the whole computation and the result is selected at the end (most pixels
            // are valid); with a branch per pixel the eight pixels' chains cannot be interleaved by the scheduler
            const int d = (int)hw_u16(R[1], 8 + p);
            const bool valid = d < dist_thr;                 // (columns x < 5 and x >= w - 6: xmask below)
            // sums over the neighbours that pass the bilateral gate |delta| < diff_thr (f = 1):
            //   ci / cj = how many with i != 0 / j != 0, cx = f(+,+) + f(-,-) - f(+,-) - f(-,+),
            //   sx / sy = sum of f delta over i = +5 minus over i = -5 / the same for j
            // so that A0 = 25 ci, A3 = 25 cj, A1 = 25 cx, b0 = 5 sx, b1 = 5 sy (the accumulators of upstream's loop)
            int ci = 0, cj = 0, cx = 0, sx = 0, sy = 0;
#pragma unroll
            for (int jj = -1; jj <= 1; ++jj)
#pragma unroll
                for (int ii = -1; ii <= 1; ++ii) {
                    if (ii == 0 && jj == 0) continue;
                    const int delta = (int)hw_u16(R[jj + 1], 8 + p + 5 * ii) - d;
                    const int ad = delta < 0 ? -delta : delta;
                    const int f = ad < diff_thr ? 1 : 0;
                    const int fd = ad < diff_thr ? delta : 0;
                    if (ii != 0) { ci += f; sx += ii * fd; }
                    if (jj != 0) { cj += f; sy += jj * fd; }
                    if (ii != 0 && jj != 0) cx += ii * jj * f;
                }
            const int A0 = 25 * ci, A3 = 25 * cj, A1 = 25 * cx, b0 = 5 * sx, b1 = 5 * sy;
            // A* <= 150 and |b*| <= 6 * 5 * 65535 < 2^23: every factor fits 24 bits, so v_mul_i32_i24 / v_mad_i32_i24
            // (full rate) give the exact 32-bit products a v_mul_lo_u32 (quarter rate) would
            const int det = mul_i24(A0, A3) - mul_i24(A1, A1);
            const int ddx = mul_i24(A3, b0) - mul_i24(A1, b1);
            const int ddy = mul_i24(A0, b1) - mul_i24(A1, b0);
            const u32 e = dn_label<false>(det, ddx, ddy, d, valid, lut);
            out[p >> 2] |= e << (8 * (p & 3));
        }
        }
    }
    // columns x < 5 and x >= w - 6 stay 0 (upstream's loop bounds): one byte mask per lane instead of two compares per pixel
    {
        const int xlo = 5 - 8 * g, xhi = (w - 6) - 8 * g;            // valid pixels of this lane: xlo <= p < xhi
        unsigned long long m = ~0ull;
        if (xlo > 0) m &= xlo >= 8 ? 0ull : (~0ull << (8 * xlo));
        if (xhi < 8) m &= xhi <= 0 ? 0ull : (~0ull >> (8 * (8 - xhi)));
        out[0] &= (u32)m; out[1] &= (u32)(m >> 32);
    }
    *reinterpret_cast<u32x2*>(code + (size_t)y * w + 8 * g) = u32x2{out[0], out[1]};
}
// (a wave-uniform branch once per wave: the two bodies differ in the float tail's first two conversions, see dn_label)
__device__ __forceinline__ void d_dnormal(const u32 vblock, const u16* __restrict__ depth0, int w, int h, int dist_thr, int diff_thr,
                                          const u8* __restrict__ lut, u8* __restrict__ code0, size_t in_stride,
                                          size_t tmp_stride, int gblocks, int nslots) {
    if (diff_thr >= 0 && diff_thr <= 249) d_dnormal_t<true>(vblock, depth0, w, h, dist_thr, diff_thr, lut, code0, in_stride, tmp_stride, gblocks, nslots);
    else d_dnormal_t<false>(vblock, depth0, w, h, dist_thr, diff_thr, lut, code0, in_stride, tmp_stride, gblocks, nslots);
}
__global__ __launch_bounds__(256) void k_dnormal(const u16* __restrict__ depth0, int w, int h, int dist_thr, int diff_thr,
                                                  const u8* __restrict__ lut, u8* __restrict__ code0, size_t in_stride,
                                                  size_t tmp_stride, int gblocks, int nslots) {
    d_dnormal(blockIdx.x, depth0, w, h, dist_thr, diff_thr, lut, code0, in_stride, tmp_stride, gblocks, nslots);
}

#define DM_ROWS 4          // output rows per lane of k_dmedian, few frames (many short waves)
#ifndef DM_ROWS_BATCH
#define DM_ROWS_BATCH 16   // batches: 20 rows of horizontal sums per 16 output rows instead of 8 per 4 (r03: 49.7 -> see DESIGN.md section 7)
#endif
template <int ROWS>
__device__ __forceinline__ void d_dmedian(const u32 vblock, const u8* __restrict__ code0, int w, int h, u8* __restrict__ quant0,
                                                  size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    u32 slot, tile;
    xcd_slot_tile_b(vblock, (u32)gblocks, (u32)nslots, slot, tile);
    const u8* code = slot_ptr_s(code0, tmp_stride, slot);
    u8* quant = slot_ptr_s(quant0, out_stride, slot);
    const int ng = w >> 3;
    const int gid = (int)(tile * 256u) + (int)threadIdx.x;
    const int band = gid / ng, g = gid - band * ng;
    const int y0 = band * ROWS;
    if (y0 >= h) return;
    u32 ringE[5][8], ringO[5][8];   // byte counters of the last five rows' horizontal sums: ranks 0..3 | 4..7
    u32 sumE[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sumO[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ROWS + 4; ++i) {                  // image row y0 - 2 + i, replicated at the borders
        const int yy = clampi(y0 - 2 + i, 0, h - 1);
        // (32-bit offsets from the slot's base: a 64-bit multiply-add per row address is four quarter-rate instructions)
        const u32 ro = (u32)yy * (u32)w + 8u * (u32)g;
        const u8* row = code + ro;
        const u32x2 c = *reinterpret_cast<const u32x2*>(row);
        u32 e[12];                                        // codes of pixels 8g-2 .. 8g+9
#pragma unroll
        for (int k = 0; k < 8; ++k) e[2 + k] = (c[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        // (no branches: the row ends load a valid dword of the row and select the replicated pixel)
        const u32 l = *reinterpret_cast<const u32*>(code + (g > 0 ? ro - 4u : ro)), r = *reinterpret_cast<const u32*>(code + (g + 1 < ng ? ro + 8u : ro + 4u));
        e[0] = g > 0 ? (l >> 16) & 0xFFu : e[2]; e[1] = g > 0 ? l >> 24 : e[2];
        e[10] = g + 1 < ng ? r & 0xFFu : e[9]; e[11] = g + 1 < ng ? (r >> 8) & 0xFFu : e[9];
        u32 oh[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) oh[k] = (1u << (e[k] & 31u)) & ~(e[k] >> 5);   // code 32 (rank 8) counts nowhere: 1 << 0 cleared
        u32 t3[10];                                       // shared partial sums: two three-operand adds per 5-sum
#pragma unroll
        for (int k = 0; k < 10; ++k) t3[k] = oh[k] + oh[k + 1] + oh[k + 2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32 hs = t3[k] + oh[k + 3] + oh[k + 4];   // nibbles <= 5
            const u32 E = hs & 0x0F0F0F0Fu, O = (hs >> 4) & 0x0F0F0F0Fu;
            if (i >= 5) { sumE[k] -= ringE[i % 5][k]; sumO[k] -= ringO[i % 5][k]; }
            ringE[i % 5][k] = E; ringO[i % 5][k] = O;
            sumE[k] += E; sumO[k] += O;
        }
        if (i >= 4) {
            const int y = y0 + i - 4;
            if (y < h) {
                u32 o[2] = {0, 0};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // x * 0x01010101 (byte prefix sums) as two shift-adds: a 32-bit multiply is quarter rate
                    const u32 e1 = sumE[k] + (sumE[k] << 8), PE = e1 + (e1 << 16);      // prefix sums of ranks 0..3
                    const u32 o0 = sumO[k] + (PE >> 24);                                 // carry the total of ranks 0..3 into byte 0 ...
                    const u32 o1 = o0 + (o0 << 8), PO = o1 + (o1 << 16);                 // ... and it propagates to every byte
                    const u32 mE = (PE + 0x73737373u) & 0x80808080u;            // byte >= 13
                    const u32 mO = (PO + 0x73737373u) & 0x80808080u;
                    // The cumulative counts never decrease, so the ranks that reached 13 are exactly those from the median rank up:
                    // with n of the eight there, the median rank is 8 - n (8 if none did) and its byte (1 << rank) >> 1 = 128 >> n
                    // (ranks 0..8 -> 0, 1, 2, 4, ..., 128).  Two v_bcnt and a shift instead of two ffs, two min and three shifts.
                    const u32 res = 128u >> (u32)(__builtin_popcount(mE) + __builtin_popcount(mO));
                    o[k >> 2] |= res << (8 * (k & 3));
                }
                *reinterpret_cast<u32x2*>(quant + ((u32)y * (u32)w + 8u * (u32)g)) = u32x2{o[0], o[1]};
            }
        }
    }
}
template <int ROWS>
__global__ __launch_bounds__(256) void k_dmedian(const u8* __restrict__ code0, int w, int h, u8* __restrict__ quant0,
                                                  size_t tmp_stride, size_t out_stride, int gblocks, int nslots) {
    d_dmedian<ROWS>(blockIdx.x, code0, w, h, quant0, tmp_stride, out_stride, gblocks, nslots);
}

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

// ------------------------------------------------------------------------------------------------
// a3-a10 of few frames: the kernels of one dependency level in ONE launch, each on its own range of the block index
// (LmPhaseArgs in lm_kernels.h).  A single frame is 14 dependent launches of 3-12 us otherwise, each with its own
// dispatch and drain; here the independent ones overlap and the chain is five launches long.
// ------------------------------------------------------------------------------------------------
struct LmPhaseGrid { u32 nb[4]; int g[4]; };   // blocks / blocks-per-slot (or segments per band for the linear memories) of the parts
template <int PH, int T0>
__global__ __launch_bounds__(256) void k_phase(LmPhaseArgs a, LmPhaseGrid pg) {
    const u32 b = blockIdx.x, e0 = pg.nb[0], e1 = e0 + pg.nb[1], e2 = e1 + pg.nb[2];
    const size_t fs = a.slot_stride;
    const int w1 = a.w >> 1, h1 = a.h >> 1;
    const size_t a3_0 = ((size_t)a.w * a.h * 3 + 255) / 256 * 256, a3_1 = ((size_t)w1 * h1 * 3 + 255) / 256 * 256;   // qn behind S
    const float thr2 = a.weak_threshold * a.weak_threshold;
    if (PH == 1) {
        if (b < e0) d_cblur(b, a.bgr0, a.w, a.h, a.cs0, fs, fs, pg.g[0], a.nslots);
        else if (b < e1) d_dnormal(b - e0, a.depth, a.w, a.h, a.dist_thr, a.diff_thr, a.normal_lut, a.ds, fs, fs, pg.g[1], a.nslots);
        else d_pyrdown8(b - e1, a.bgr0, a.w, a.h, a.bgr1, w1, h1, fs, pg.g[2], a.nslots);
    } else if (PH == 2) {
        if (b < e0) d_dmedian<DM_ROWS>(b, a.ds, a.w, a.h, a.qd0, fs, fs, pg.g[0], a.nslots);
        else if (b < e1) d_cblur(b - e0, a.bgr1, w1, h1, a.cs1, fs, fs, pg.g[1], a.nslots);
        else d_corient(b - e1, a.cs0, a.w, a.h, thr2, a.cs0 + a3_0, nullptr, fs, fs, pg.g[2], a.nslots);
    } else if (PH == 3) {
        if (b < e0) d_cvote(b, a.cs0 + a3_0, a.w, a.h, a.qc0, fs, fs, pg.g[0], a.nslots);
        else if (b < e1) d_corient(b - e0, a.cs1, w1, h1, thr2, a.cs1 + a3_1, nullptr, fs, fs, pg.g[1], a.nslots);
        else if (b < e2) d_lm_fast<5, 128, 0, 1>(b - e1, a.qd0, a.w, a.w, a.h, a.resp_tab, a.lm_d0, 0u, fs, fs, pg.g[2], a.nslots);
        else d_lm_fast<8, 40, 1, 2>(b - e2, a.qd0, a.w, w1, h1, a.resp_tab, a.lm_d1, a.ori_stride1, fs, fs, pg.g[3], a.nslots, a.plane_ori1);
    } else {
        if (b < e0) d_cvote(b, a.cs1 + a3_1, w1, h1, a.qc1, fs, fs, pg.g[0], a.nslots);
        else if (T0 == 5) d_lm_fast<5, 128, 0, 1>(b - e0, a.qc0, a.w, a.w, a.h, a.resp_tab, a.lm_c0, 0u, fs, fs, pg.g[1], a.nslots);
        else d_lm_spread2(b - e0, a.qc0, a.w, a.w, a.h, a.lm_c0, fs, fs, pg.g[1], a.nslots);     // T0 == 2 (colour only)
    }
}

// ------------------------------------------------------------------------------------------------
// a3-a10 of a BATCH as four launches (r03: "horizontal fusion of the pyramid levels").  The batch kernels of one dependency
// level share ONE grid, each on its own range of the block index, the longest-running first: the level-1 kernels and the
// other short ones (a 320 x 240 level is 10 waves per frame: 960 waves for 1024 SIMDs when launched alone, each walking
// its strip serially) fill the chip's tail instead of holding a half-empty launch of their own, and a lane-step is 4 + 4
// dependent launches instead of 11 + 4.
//   1  depth normals          | blur(level 0)           | pyrDown(level 0 -> 1)
//   2  gradient + vote(0)     | median of the normals   | blur(level 1)
//   3  gradient + vote(1)     | colour spread memory(0) | depth spread memory(0) | depth response memories(1)
//   4  colour response memories(1)                                             (plain k_lm_fast launch)
// Same device functions, same results as the kernels launched one by one (LM_TUNE_BATCH_PHASES = 0).  Every part keeps
// its XCD affinity: the parts' block counts are multiples of 8 whenever the slot count is.
// SB / SG: rows per strip of the level-0 blur / gradient kernels (16, or 32 for tall images).
// ------------------------------------------------------------------------------------------------
template <int PH, int T0, int SB, int SG>
__global__ __launch_bounds__(256, 2) void k_bphase(LmPhaseArgs a, LmPhaseGrid pg) {
    const u32 b = blockIdx.x, e0 = pg.nb[0], e1 = e0 + pg.nb[1], e2 = e1 + pg.nb[2];
    const size_t fs = a.slot_stride;
    const int w1 = a.w >> 1, h1 = a.h >> 1, n = a.nslots;
    const float thr2 = a.weak_threshold * a.weak_threshold;
    const int ithr = thr2 >= 2147483648.f ? INT_MAX : (int)floorf(thr2);    // (float)m > thr2 <=> m > floor(thr2)
    if (PH == 1) {
        if (b < e0) d_dnormal(b, a.depth, a.w, a.h, a.dist_thr, a.diff_thr, a.normal_lut, a.ds, fs, fs, pg.g[0], n);
        else if (b < e1) d_cblur_sh<SB>(b - e0, a.bgr0, a.w, a.h, a.cs0, fs, fs, pg.g[1], n);
        else d_pyrdown16<PD_STRIP>(b - e1, a.bgr0, a.w, a.h, a.bgr1, w1, h1, fs, pg.g[2], n);
    } else if (PH == 2) {
        if (b < e0) d_cgrad<SG>(b, a.cs0, a.w, a.h, ithr, a.qc0, fs, fs, pg.g[0], n);
        else if (b < e1) d_dmedian<DM_ROWS_BATCH>(b - e0, a.ds, a.w, a.h, a.qd0, fs, fs, pg.g[1], n);
        else d_cblur_sh<16>(b - e1, a.bgr1, w1, h1, a.cs1, fs, fs, pg.g[2], n);
    } else {
        if (b < e0) d_cgrad<16>(b, a.cs1, w1, h1, ithr, a.qc1, fs, fs, pg.g[0], n);
        else if (b < e1) {
            if (T0 == 5) d_lm_spread5(b - e0, a.qc0, a.w, a.w, a.h, a.lm_c0, fs, fs, pg.g[1], n);
            else d_lm_spread2(b - e0, a.qc0, a.w, a.w, a.h, a.lm_c0, fs, fs, pg.g[1], n);
        }
        else if (b < e2) d_lm_spread5(b - e1, a.qd0, a.w, a.w, a.h, a.lm_d0, fs, fs, pg.g[2], n);
        else d_lm_fast<8, 40, 1, 2>(b - e2, a.qd0, a.w, w1, h1, a.resp_tab, a.lm_d1, a.ori_stride1, fs, fs, pg.g[3], n, a.plane_ori1);
    }
}

// The RGB-D form.  A fused kernel's waves all allocate the registers of its hungriest part: with the depth kernels
// (k_dnormal: 61 VGPRs, 8 waves per SIMD when launched alone) inside the grids of the blur / gradient kernels (204 / 238
// VGPRs, 2 waves per SIMD) the level-fused launches above LOSE (r03, config 2: pre-processing 4.82 -> 4.88 us per frame
// on one lane, 145 K -> 131 K detections/s with three lanes -- the fat waves also keep the other lanes' scan waves off
// the SIMDs).  So the RGB-D pyramid fuses only kernels of one register class:
//   light  0: depth normals | pyrDown                      heavy  1: gradient + vote(0) | blur(level 1)
//   light  2: colour spread memory(0) | depth spread memory(0) | depth response memories(1)
// between the plain launches of blur(level 0), median, gradient + vote(1) and the colour response memories(1): seven
// launches instead of eleven.
template <int PART, int SG>
__global__ __launch_bounds__(256, PART == 1 ? 2 : 1) void k_bsplit(LmPhaseArgs a, LmPhaseGrid pg) {
    const u32 b = blockIdx.x, e0 = pg.nb[0], e1 = e0 + pg.nb[1];
    const size_t fs = a.slot_stride;
    const int w1 = a.w >> 1, h1 = a.h >> 1, n = a.nslots;
    if (PART == 0) {
        if (b < e0) d_dnormal(b, a.depth, a.w, a.h, a.dist_thr, a.diff_thr, a.normal_lut, a.ds, fs, fs, pg.g[0], n);
        else d_pyrdown16<PD_STRIP>(b - e0, a.bgr0, a.w, a.h, a.bgr1, w1, h1, fs, pg.g[1], n);
    } else if (PART == 1) {
        const float thr2 = a.weak_threshold * a.weak_threshold;
        const int ithr = thr2 >= 2147483648.f ? INT_MAX : (int)floorf(thr2);
        if (b < e0) d_cgrad<SG>(b, a.cs0, a.w, a.h, ithr, a.qc0, fs, fs, pg.g[0], n);
        else d_cblur_sh<16>(b - e0, a.bgr1, w1, h1, a.cs1, fs, fs, pg.g[1], n);
    } else {
        if (b < e0) d_lm_spread5(b, a.qc0, a.w, a.w, a.h, a.lm_c0, fs, fs, pg.g[0], n);
        else if (b < e1) d_lm_spread5(b - e0, a.qd0, a.w, a.w, a.h, a.lm_d0, fs, fs, pg.g[1], n);
        else d_lm_fast<8, 40, 1, 2>(b - e1, a.qd0, a.w, w1, h1, a.resp_tab, a.lm_d1, a.ori_stride1, fs, fs, pg.g[2], n, a.plane_ori1);
    }
}

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
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {   // all lanes active
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);   // row_mirror: every lane holds its row's sum
    return (u32)__builtin_amdgcn_readlane((int)v, 0) + (u32)__builtin_amdgcn_readlane((int)v, 16) +
           (u32)__builtin_amdgcn_readlane((int)v, 32) + (u32)__builtin_amdgcn_readlane((int)v, 48);
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
// a14  One wave per candidate: lane l holds the 4 patch positions (row l/4, cols 4(l%4)..+3) of the
// 16x16 patch.  Refinement levels keep SPREAD linear memories (1 byte per position instead of 8
// response bytes: the whole level stays L2-resident and a patch load touches 1/8 of the lines); the
// response max(LUT_lo[v & 15], LUT_hi[v >> 4]) comes from an 8 x 256-byte table in LDS (one ds_read_u8
// per position; upstream does the two 16-entry nibble lookups with pshufb).
// The modality's feature records are loaded one per lane, bounds-checked in parallel (features
// shifted out of the frame read the arena's zero block: spread 0 -> response 0), then broadcast
// with v_readlane so the patch loads (one unaligned dword per lane per feature) issue back to back.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        u32 o = (u32)__shfl_xor((int)v, s, 64);
        v = v > o ? v : o;
    }
    return v;
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
template <int TOP>
__device__ __forceinline__ bool scanl_rounds(const u8* lds, const __amdgpu_buffer_rsrc_t rs_off, u32 voff, u32 unit_add, u32 keep5, u32 pre, int valid, int Fw,
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
                const u32* p = reinterpret_cast<const u32*>(lds + ((e[k] + unit_add) >> 8));
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
        const u32 unit_add = (unit * 16u) << 8;
        const u32 keep5 = (j0 + 128 >= P || lane == 63) ? 0u : 0xFFFFFFFFu;
        u32 flags[4];
        bool pruned;
        int f;
        // counters of 6 bits (flag = bit 5) when no template of the wave may miss more than 31 features -- the usual case: 24 at threshold 80 with 62
        // features -- two half adders fewer per round and dword; 8 bits otherwise
        if (mm_hi <= 31) pruned = scanl_rounds<5>(lds, rs_off, voff, unit_add, keep5, (u32)(31 - mmax), valid, Fw, first_test, alive, flags, f);
        else pruned = scanl_rounds<7>(lds, rs_off, voff, unit_add, keep5, pre, valid, Fw, first_test, alive, flags, f);
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
__device__ __forceinline__ void rf_lookup(const u8* __restrict__ tab, u32 v, u32& p01, u32& p23) {
    const u32 r0 = tab[v & 0xFFu], r1 = tab[(v >> 8) & 0xFFu], r2 = tab[(v >> 16) & 0xFFu], r3 = tab[v >> 24];
    p01 = r0 | (r1 << 16); p23 = r2 | (r3 << 16);
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
    u32 s01 = 0, s23 = 0;   // u16 pairs: patch positions {0, 1} and {2, 3} of this lane (sums <= 126 * 4)
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
            u32 v[8], q01[8], q23[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rf_patch<W4>(lm, (u32)__builtin_amdgcn_readlane((int)eff, f + k), lane_off);
#pragma unroll
            for (int k = 0; k < 8; ++k) rf_lookup(resp[(u32)__builtin_amdgcn_readlane((int)lab, f + k)], v[k], q01[k], q23[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s01 += q01[k]; s23 += q23[k]; }
            f_left -= min(8, cnt - f);
            if (PRUNE_REFINE && (f & 8) && f_left > 0) {       // every second batch of eight
                const u32 mx = pk_max_u16(s01, s23);
                const u32 best_now = wave_max_u32(max(mx & 0xFFFFu, mx >> 16));
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
    u32 k0 = ((s01 & 0xFFFF) << 8) | (255u - idx0);
    u32 k1 = ((s01 >> 16) << 8) | (254u - idx0);
    u32 k2 = ((s23 & 0xFFFF) << 8) | (253u - idx0);
    u32 k3 = ((s23 >> 16) << 8) | (252u - idx0);
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
    u32 s01[2] = {0, 0}, s23[2] = {0, 0};
    if (a.stat && lane == 0) atomicAdd(&a.stat[1], 2ull);
    const int dcol = bx[1] - bx[0];                                   // wave-uniform
    const bool same_rows = by[0] == by[1] && dcol >= 0 && dcol <= 4;
    // r06: refine_one's exact pruning for the pair: every 16 features the wave takes each candidate's best partial sum of its
    // patch; a candidate whose (best + 4 x features to come) * 100 / (4 n) stays below the threshold -- the final test's own float expression --
    // is out.  Both out: the pair stops.  One out: the other is finished alone (refine_one from its first feature: at most the features loaded so
    // far are read twice, a candidate that dies does so after a quarter of its list on average).
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
            u32 v[2][RP_BATCH], q01[2][RP_BATCH], q23[2][RP_BATCH];
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
                for (int c = 0; c < 2; ++c) rf_lookup(tab, v[c][k], q01[c][k], q23[c][k]);
            }
#pragma unroll
            for (int k = 0; k < RP_BATCH; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) { s01[c] += q01[c][k]; s23[c] += q23[c][k]; }
            f_left -= min(RP_BATCH, cnt - f);
            if (PRUNE_REFINE && (f & RP_BATCH) && f_left > 0) {       // every second batch
                bool out[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const u32 mx = pk_max_u16(s01[c], s23[c]);
                    const u32 best_now = wave_max_u32(max(mx & 0xFFFFu, mx >> 16));
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
        const u32 k0 = ((s01[c] & 0xFFFF) << 8) | (255u - idx0), k1 = ((s01[c] >> 16) << 8) | (254u - idx0);
        const u32 k2 = ((s23[c] & 0xFFFF) << 8) | (253u - idx0), k3 = ((s23[c] >> 16) << 8) | (252u - idx0);
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
__global__ __launch_bounds__(256) void k_hsv_mask(const u8* __restrict__ bgr0, int w, int h, LmHsvRange rg,
                                                   const int* __restrict__ divtab, u32* __restrict__ mask0, int wpr,
                                                   size_t in_stride, size_t mask_stride) {
    const u8* bgr = slot_ptr(bgr0, in_stride);
    u32* mask = slot_ptr(mask0, mask_stride);
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= wpr * h) return;
    const int y = gid / wpr, wi = gid - y * wpr;
    const int shift = 12;
    u32 bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int x = 32 * wi + k;
        if (x >= w) break;
        const u8* p = bgr + ((size_t)y * w + x) * 3;
        const int b = p[0], g = p[1], r = p[2];
        const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
        const int diff = v - vmin;
        const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        const int sat = (diff * divtab[v] + (1 << (shift - 1))) >> shift;
        int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        hh = (hh * divtab[256 + diff] + (1 << (shift - 1))) >> shift;
        hh += hh < 0 ? 180 : 0;
        const int H = hh < 0 ? 0 : (hh > 255 ? 255 : hh);
        const bool in = H >= rg.lo[0] && H <= rg.hi[0] && sat >= rg.lo[1] && sat <= rg.hi[1] && v >= rg.lo[2] && v <= rg.hi[2];
        bits |= (in ? 1u : 0u) << k;
    }
    mask[(size_t)y * wpr + wi] = bits;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_hull_counts(LmHullArgs a) {
    extern __shared__ __attribute__((aligned(16))) int hsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* vx = hsm + wv * (2 * LM_HULL_MAX + 2 * a.h);
    int* vy = vx + LM_HULL_MAX;
    int* rowL = vy + LM_HULL_MAX;
    int* rowR = rowL + a.h;
    const u32 i = blockIdx.x * 4u + (u32)wv;
    const bool live = i < a.n;                       // dead waves still take part in the barriers
    int n = 0, mx = 0, my = 0;
    u32 off0 = 0;
    const u32* mask_row0 = a.mask;      // the colour mask of the match's frame (lists that span several slots: match_slot)
    if (live) {
        const LmOutMatch m = a.matches[i];
        mx = m.x; my = m.y;
        if (a.match_slot) mask_row0 += (size_t)a.match_slot[i] * a.mask_slot_words;
        const u32 idx = a.class_base[m.class_idx] + (u32)m.template_id;
        off0 = a.hull_off[idx];
        n = (int)(a.hull_off[idx + 1] - off0);
    }
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
    for (int k = lane; k < n; k += 64) {
        const int x = (int)a.hull_xy[2 * (off0 + k)] + mx, y = (int)a.hull_xy[2 * (off0 + k) + 1] + my;
        vx[k] = x; vy[k] = y;
        bx0 = min(bx0, x); bx1 = max(bx1, x); by0 = min(by0, y); by1 = max(by1, y);
    }
    by0 = wave_min_i32(by0); by1 = wave_max_i32(by1);
    // no hull (a wave past the list, or a template without features): an empty row range -- and no INT_MAX + lane overflow
    const int ry0 = n ? max(by0, 0) : 0, ry1 = n ? min(by1, a.h - 1) : -1;
    for (int y = ry0 + lane; y <= ry1; y += 64) { rowL[y] = INT_MAX; rowR[y] = INT_MIN; }
    __syncthreads();
    // outline: edge e joins vertex e and e + 1 (mod n); the same 8-connected line walk as the checker
    for (int e = lane; e < n && n > 1; e += 64) {
        int ax = vx[e], ay = vy[e];
        const int bx = vx[(e + 1) % n], by = vy[(e + 1) % n];
        const int dx = abs(bx - ax), dy = -abs(by - ay), sx = ax < bx ? 1 : -1, sy = ay < by ? 1 : -1;
        int err = dx + dy;
        for (;;) {
            if (ay >= ry0 && ay <= ry1) { atomicMin(&rowL[ay], ax); atomicMax(&rowR[ay], ax); }
            if (ax == bx && ay == by) break;
            const int e2 = 2 * err;
            if (e2 >= dy) { err += dy; ax += sx; }
            if (e2 <= dx) { err += dx; ay += sy; }
        }
    }
    __syncthreads();
    long long in_hull = 0, in_both = 0;
    for (int y = ry0 + lane; y <= ry1; y += 64) {
        double lo = 1e300, hi = -1e300;
        for (int e = 0; e < n; ++e) {
            const int ax = vx[e], ay = vy[e], bx = vx[(e + 1) % n], by = vy[(e + 1) % n];
            if (ay == by) {
                if (ay == y) { lo = fmin(lo, (double)min(ax, bx)); hi = fmax(hi, (double)max(ax, bx)); }
                continue;
            }
            if (y < min(ay, by) || y > max(ay, by)) continue;
            const double x = ax + (double)(y - ay) * (bx - ax) / (double)(by - ay);
            lo = fmin(lo, x); hi = fmax(hi, x);
        }
        if (n == 1) { lo = hi = vx[0]; }
        int L = rowL[y], R = rowR[y];
        if (lo <= hi) { L = min(L, (int)ceil(lo - 1e-9)); R = max(R, (int)floor(hi + 1e-9)); }
        L = max(L, 0); R = min(R, a.w - 1);
        if (L > R) continue;
        in_hull += R - L + 1;
        const u32* row = mask_row0 + (size_t)y * a.wpr;
        for (int wi = L >> 5; wi <= (R >> 5); ++wi) {
            const int b0 = wi == (L >> 5) ? (L & 31) : 0, b1 = wi == (R >> 5) ? (R & 31) : 31;
            const u32 mk = (0xFFFFFFFFu >> (31 - b1)) & (0xFFFFFFFFu << b0);
            in_both += __popc(row[wi] & mk);
        }
    }
    in_hull = wave_sum_i64(in_hull); in_both = wave_sum_i64(in_both);
    if (live && lane == 0) { a.out[2 * (size_t)i] = in_hull; a.out[2 * (size_t)i + 1] = in_both; }
}

// materialises the NN pyramid of the depth modality's quantised image (levels >= 2, stage hooks)
// r06: the by-products of the host's crop pass (PostProcess.cpp crop_depth) for a whole batch of depth checks: one wave per query, rows of the crop
// one after the other, 64 columns at a time.  t = depth <= 1 ? 65535 : depth (threshold(.., 1, 65535) inverted and added; the zeros the principal-point
// shift moved in are depths <= 1 like every hole); counted: t < lo, lo <= t <= hi.  The host then knows the verdict "outside the window" of about four
// checks in five without touching the frame (more than n / 4 values below the window, or none inside it) and runs std::nth_element for the rest.
__global__ __launch_bounds__(256) void k_depth_counts(LmDepthArgs a) {
    // one WORKGROUP per query (a crop is tens of thousands of pixels: one wave per query walked 600 dependent row segments and took hundreds of
    // microseconds, A/B r06): thread t takes the 8-pixel pieces t, t + 256, .. of the crop, eight independent 2-byte loads in flight each
    __shared__ u32 acc[2];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const u32 qi = blockIdx.x;
    if (tid < 2) acc[tid] = 0u;
    __syncthreads();
    const LmDepthQuery q = a.q[qi];
    const u16* img = reinterpret_cast<const u16*>(reinterpret_cast<const u8*>(a.depth) + (size_t)q.slot * a.slot_stride);
    const int cw = q.x1 - q.x0, ch = q.y1 - q.y0;
    const int pr = (cw + 7) >> 3;                      // pieces per row
    const int np = pr * ch;
    u32 below = 0, inside = 0;
    for (int p = tid; p < np; p += 256) {
        const int r = p / pr, c = (p - r * pr) << 3;
        const u16* src = img + (size_t)(q.y0 + r) * a.w + q.x0 + c;
        const int m = cw - c < 8 ? cw - c : 8;
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k < m ? (int)src[k] : 2;      // (a value that counts nowhere is substituted below)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = v[k] <= 1 ? 65535 : v[k];
            const bool ok = k < m;
            below += (ok && t < q.lo) ? 1u : 0u;
            inside += (ok && t >= q.lo && t <= q.hi) ? 1u : 0u;
        }
    }
    below = wave_sum_u32(below);
    inside = wave_sum_u32(inside);
    if (lane == 0) { atomicAdd(&acc[0], below); atomicAdd(&acc[1], inside); }
    __syncthreads();
    if (tid == 0) { a.out[2 * qi] = acc[0]; a.out[2 * qi + 1] = acc[1]; }
}

__global__ void k_nn_half(const u8* __restrict__ src0, int sp, u8* __restrict__ dst0, int dw, int dh,
                          size_t slot_stride) {
    const u8* src = slot_ptr(src0, slot_stride);
    u8* dst = slot_ptr(dst0, slot_stride);
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < dw && y < dh) dst[(size_t)y * dw + x] = src[(size_t)(2 * y) * sp + 2 * x];
}

}  // namespace

// ================================================================================================
// launchers
// ================================================================================================
// Kernel selection is by WORK, not by frame count (r04): the few-frame kernels (many short waves, finish sooner) and
// the batch kernels (row-walking, fewer instructions per pixel) were tuned on 640 x 480 frames, where the break-even is 16 frames.
// A call's frames count `weight` times, weight = level-0 pixels / (640 x 480) rounded down, at least 1: eight 1280 x 960 frames
// (config 5) carry the pixels of 32 VGA frames and take the batch kernels.  Set per host thread around a call's launches.
static thread_local int g_slot_weight = 1;
void lmk_set_slot_weight(int w) { g_slot_weight = w < 1 ? 1 : w; }
static inline int sel_slots(int nslots) { return nslots * g_slot_weight; }
static int g_pyrdown_variant = 0;   // 0: by batch size (k_pyrdown8 below 16 frames, the row-walking k_pyrdown16 from there), 1: k_pyrdown8, 2: k_pyrdown16
void lmk_set_pyrdown_variant(int v) { g_pyrdown_variant = v; }
void lmk_pyrdown(hipStream_t s, const u8* src, int sw, int sh, u8* dst, size_t slot_stride, int nslots) {
    int dw = sw / 2, dh = sh / 2;
    if (g_pyrdown_variant != 1 && (g_pyrdown_variant == 2 || sel_slots(nslots) >= 16) && (sw % 16) == 0 && (sh % 2) == 0 && sh >= 4 &&
        ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0 && (slot_stride % 16) == 0) {
        const int n_w = (((sw / 16) * ((dh + PD_STRIP - 1) / PD_STRIP) + 61) / 62 + 3) / 4;
        hipLaunchKernelGGL(k_pyrdown16<PD_STRIP>, dim3((unsigned)(n_w * nslots)), dim3(256), 0, s, src, sw, sh, dst, dw, dh, slot_stride, n_w, nslots);
        return;
    }
    if ((sw % 16) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0 && (slot_stride % 16) == 0) {
        const int lanes = (dw / 8) * dh;
        hipLaunchKernelGGL(k_pyrdown8, dim3((unsigned)(((lanes + 255) / 256) * nslots)), dim3(256), 0, s, src, sw, sh, dst, dw, dh, slot_stride, (lanes + 255) / 256, nslots);
        return;
    }
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, nslots);
    hipLaunchKernelGGL(k_pyrdown, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh, slot_stride);
}

void lmk_nn_half(hipStream_t s, const u8* src, int src_pitch, u8* dst, int dw, int dh, size_t slot_stride, int nslots) {
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, nslots);
    hipLaunchKernelGGL(k_nn_half, grid, dim3(256), 0, s, src, src_pitch, dst, dw, dh, slot_stride);
}

// r04: 4 = the blur on the matrix cores (k_cblur_mx; inside k_blur_mx_pyr at level 0 of a batch).  Measured (tools/ab_blur_mx.sh,
// profiles/r04_ab_experiments.log): bit-identical; alone on the chip no faster than k_cblur_sh (the launch is bound by its pyrDown tiles and
// by memory), beside the other lanes config 2 +1.5 .. 2 % (the vector ALU is what the whole pipeline is short of), config 3 -0.8 % (HBM-bound
// launch).  Hence 0 = auto takes it for batches of frames of up to 2 MB and k_cblur_sh above.
static bool mx_auto(int w, int h, int nslots);
static int g_cblur_variant = 0;   // 0: by batch size (one-shot below 16 frames, k_cblur_sh from there), 1: one-shot blur (k_cblur),
                                  // (2 was r02's sliding-window k_cblur_sw, deleted in r05,) 3: sliding window with the column sums
                                  // shared between neighbouring lanes (k_cblur_sh, r03: config 2 146.3 -> 150.7 K, config 3 81.9 -> 86.1 K
                                  // detections/s); A/B knob of tools/ and tests
void lmk_set_cblur_variant(int v) { g_cblur_variant = v; }
static int g_blur_strip = 0;   // rows per strip of the level-0 blur inside k_blur_pyr / of the matrix-core blur: 0 = by shape and batch size, 16 / 32 / 64 = forced (A/B, tests)
void lmk_set_blur_strip(int v) { g_blur_strip = v; }
// rows per strip of the matrix-core blur (one extra 8-row tile per strip for the vertical taps).  Measured r04 (profiles/r04_ab_experiments.log):
// 48 / 96 / 192 / 480 rows are within 1 % of each other on configs 2 and 3, 96 best; LM_TUNE_BLUR_STRIP forces 16 / 32 / 64.
static int mx_strip_rows() { return g_blur_strip ? g_blur_strip : 96; }
static bool mx_auto(int w, int h, int nslots) { return g_cblur_variant == 0 && sel_slots(nslots) >= 16 && (long)w * h * 3 <= 2000000L && ((w * 3) % 32) == 0; }
static int g_dmedian_variant = 0;   // 0: by batch size (4 output rows per lane below 16 frames, DM_ROWS_BATCH from there), 1 / 2: force either
void lmk_set_dmedian_variant(int v) { g_dmedian_variant = v; }
static int g_cgrad_variant = 0;   // 0: by batch size (fused k_cgrad from 16 frames), 1: k_corient + k_cvote, 2: k_cgrad, 3: k_cgrad with 32-row strips
void lmk_set_cgrad_variant(int v) { g_cgrad_variant = v; }

size_t lmk_color_scratch_bytes(int w, int h) {
    // S u8 [h][3w] | qn u8 [h][w], each 256-B aligned (also the rank-code image of the depth passes)
    size_t px = (size_t)w * h;
    return (px * 3 + 255) / 256 * 256 + (px + 255) / 256 * 256;
}

// k_blur_pyr: a slot's blur and pyrDown tiles dealt out evenly by rows (r04) instead of back to back.  Measured (tools/ab_blur_pyr.sh,
// profiles/r04_ab_experiments.log): 1280 x 960 (3.7 MB per frame, never L2-resident back to back) reads 8.64 -> 7.54 MB per frame, the launch
// 271.7 -> 260.1 us per 128 frames, config 3 +0.8 %; 640 x 480 reads 2.09 -> 1.98 MB but the launch gets 4 us LONGER (70.7 -> 74.7) and the
// headline does not move: 2 (auto) deals evenly only frames of more than 2 MB, 1 always, 0 never.
static int g_blur_pyr_interleave_mode = 2;
void lmk_set_blur_pyr_interleave(int v) { g_blur_pyr_interleave_mode = v; }
static int g_blur_pyr = 1;   // level-0 blur and cv::pyrDown of a batch in one slot-interleaved launch (k_blur_pyr); 0: two launches
void lmk_set_blur_pyr(int v) { g_blur_pyr = v; }
bool lmk_blur_pyrdown(hipStream_t s, const u8* bgr0, int w, int h, u8* scratch0, u8* bgr1, u8* quant0, size_t slot_stride, int nslots) {
    // exactly the shapes lmk_color_quantize's streaming path and k_pyrdown16 take, batches only
    if (!g_blur_pyr || sel_slots(nslots) < 16 || g_cblur_variant == 1 || g_cblur_variant == 2 || g_pyrdown_variant == 1) return false;
    if (!scratch0 || (w % 16) != 0 || (h % 2) != 0 || h < 4 || (slot_stride % 16) != 0) return false;
    if (((uintptr_t)bgr0 & 15) || ((uintptr_t)scratch0 & 15) || ((uintptr_t)quant0 & 15) || ((uintptr_t)bgr1 & 7)) return false;
    const int dh = h / 2;
    const int g_blur_pyr_interleave = g_blur_pyr_interleave_mode == 1 || (g_blur_pyr_interleave_mode == 2 && (long)w * h * 3 > 2000000L);
    auto waves4 = [](int pairs) { return ((pairs + 61) / 62 + 3) / 4; };
    const int g_pyr = waves4((w / 16) * ((dh + PD_STRIP - 1) / PD_STRIP));
    if (g_cblur_variant == 4 || mx_auto(w, h, nslots)) {
        if (((w * 3) % 32) != 0) return false;
        const int gx = (w * 3 + 4 * MX_WAVE_BYTES - 1) / (4 * MX_WAVE_BYTES);
        const int strip_rows = mx_strip_rows();
        const int gy = (h + strip_rows - 1) / strip_rows;
        hipLaunchKernelGGL(k_blur_mx_pyr, dim3((unsigned)((gx * gy + g_pyr) * nslots)), dim3(256), 0, s, bgr0, w, h, scratch0, bgr1, slot_stride, gx, gy, strip_rows, g_pyr, nslots);
        return true;
    }
    // rows per blur strip: 16, or 32 for tall images.  A strip of S rows reads and sums S + 6 (16: 1.375 x the image, 32: 1.19 x, 64:
    // 1.09 x) but taller strips measured no faster (r03, LM_TUNE_BLUR_STRIP: config 2 163.2 / 162.8 / 160.8 K detections/s at 16 /
    // 32 / 64, config 3 90.8 / 90.7 K at 32 / 64): fewer, longer waves
    if (g_blur_strip == 64) {
        const int g_blur = waves4((w * 3 / 16) * ((h + 63) / 64));
        hipLaunchKernelGGL(k_blur_pyr<64>, dim3((unsigned)((g_blur + g_pyr) * nslots)), dim3(256), 0, s, bgr0, w, h, scratch0, bgr1, slot_stride, g_blur, g_pyr, nslots, g_blur_pyr_interleave);
    } else if (g_blur_strip == 32 || (g_blur_strip == 0 && h > 640 && (long)(waves4((w * 3 / 16) * ((h + 31) / 32)) + g_pyr) * nslots >= 768)) {
        // (r04: 32-row strips only when they still give the chip three rounds of workgroups -- eight 1280 x 960 frames, config 5,
        // are 312 workgroups of 32-row strips on 512 slots)
        const int g_blur = waves4((w * 3 / 16) * ((h + 31) / 32));
        hipLaunchKernelGGL(k_blur_pyr<32>, dim3((unsigned)((g_blur + g_pyr) * nslots)), dim3(256), 0, s, bgr0, w, h, scratch0, bgr1, slot_stride, g_blur, g_pyr, nslots, g_blur_pyr_interleave);
    } else {
        const int g_blur = waves4((w * 3 / 16) * ((h + CBS_STRIP - 1) / CBS_STRIP));
        hipLaunchKernelGGL(k_blur_pyr<CBS_STRIP>, dim3((unsigned)((g_blur + g_pyr) * nslots)), dim3(256), 0, s, bgr0, w, h, scratch0, bgr1, slot_stride, g_blur, g_pyr, nslots, g_blur_pyr_interleave);
    }
    return true;
}

void lmk_color_quantize(hipStream_t s, const u8* bgr, int w, int h, float weak_threshold, u8* quant, float* mag,
                        u8* scratch, size_t slot_stride, int nslots, bool blurred) {
    const float thr2 = weak_threshold * weak_threshold;
    if (scratch && (w % 16) == 0 && ((uintptr_t)bgr & 15) == 0 && ((uintptr_t)scratch & 15) == 0 &&
        ((uintptr_t)quant & 15) == 0 && (slot_stride % 16) == 0) {
        const size_t px = (size_t)w * h, a3 = (px * 3 + 255) / 256 * 256;
        u8* S = scratch;
        u8* qn = scratch + a3;
        const int n_b = (w * 3 / 16) * ((h + CB_ROWS - 1) / CB_ROWS);         // 16-byte blocks x row bands
        const int n_o = (w / 16) * h;                                         // 16-pixel groups
        const int n_t = (w / 16) * ((h + CVT_ROWS - 1) / CVT_ROWS);           // 16-pixel groups x bands
        // few frames: the one-shot kernel's many short waves finish sooner (a single frame is 57 sliding-window waves of
        // eight dependent steps: 166 instead of 150 us per resident single-frame match); batches: the sliding window
        if (blurred) {
            // S is already in `scratch` (lmk_blur_pyrdown)
        } else if ((g_cblur_variant == 4 || mx_auto(w, h, nslots)) && ((w * 3) % 32) == 0 && h >= 1) {
            // r04 experiment: the blur on the matrix cores (k_cblur_mx); a workgroup = four waves side by side, each 128 byte columns
            // wide, walking down a strip of rows in steps of 8 (one extra tile of 8 rows per strip for the vertical taps)
            const int gx = (w * 3 + 4 * MX_WAVE_BYTES - 1) / (4 * MX_WAVE_BYTES);
            const int strip_rows = mx_strip_rows();
            const int gy = (h + strip_rows - 1) / strip_rows;
            hipLaunchKernelGGL(k_cblur_mx, dim3((unsigned)(gx * gy * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, gx, gy, strip_rows, nslots);
        } else if (g_cblur_variant == 1 || (g_cblur_variant == 0 && sel_slots(nslots) < 16)) {
            hipLaunchKernelGGL(k_cblur, dim3((unsigned)(((n_b + 255) / 256) * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, (n_b + 255) / 256, nslots);
        } else {
            // column sums shared between neighbouring lanes: 62 (strip, block) pairs per wave, four waves per workgroup
            if (h > 640) {
                const int n_w = (((w * 3 / 16) * ((h + 31) / 32) + 61) / 62 + 3) / 4;
                hipLaunchKernelGGL(k_cblur_sh<32>, dim3((unsigned)(n_w * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, n_w, nslots);
            } else {
                const int n_w = (((w * 3 / 16) * ((h + CBS_STRIP - 1) / CBS_STRIP) + 61) / 62 + 3) / 4;
                hipLaunchKernelGGL(k_cblur_sh<CBS_STRIP>, dim3((unsigned)(n_w * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, n_w, nslots);
            }
        }
        // orientation + vote: fused for batches (k_cgrad), two kernels for few frames (many short waves) and whenever the
        // caller wants the magnitude image
        if (!mag && (g_cgrad_variant >= 2 || (g_cgrad_variant == 0 && sel_slots(nslots) >= 16))) {
            const int ithr = thr2 >= 2147483648.f ? INT_MAX : (int)floorf(thr2);    // (float)m > thr2 <=> m > floor(thr2)
            // rows per strip: 16 (2 of 18 label rows are recomputed by the neighbouring strips), 8 when that would leave
            // SIMDs without a wave (a 320 x 240 level is 5 waves per frame at 16)
            auto waves = [&](int strip) { return ((w / 16) * ((h + strip - 1) / strip) + 61) / 62; };
            if (g_cgrad_variant == 3 || (h > 640 && (long)waves(32) * nslots >= 3072)) {        // tall images: 2 of 34 label rows recomputed instead of 2 of 18
                const int n_w = waves(32);
                hipLaunchKernelGGL(k_cgrad<32>, dim3((unsigned)(((n_w + 3) / 4) * nslots)), dim3(256), 0, s, S, w, h, ithr, quant, slot_stride, slot_stride, (n_w + 3) / 4, nslots);
            } else if ((long)waves(CG_STRIP) * nslots >= 1536) {
                const int n_w = waves(CG_STRIP);
                hipLaunchKernelGGL(k_cgrad<CG_STRIP>, dim3((unsigned)(((n_w + 3) / 4) * nslots)), dim3(256), 0, s, S, w, h, ithr, quant, slot_stride, slot_stride, (n_w + 3) / 4, nslots);
            } else {
                const int n_w = waves(8);
                hipLaunchKernelGGL(k_cgrad<8>, dim3((unsigned)(((n_w + 3) / 4) * nslots)), dim3(256), 0, s, S, w, h, ithr, quant, slot_stride, slot_stride, (n_w + 3) / 4, nslots);
            }
            return;
        }
        hipLaunchKernelGGL(k_corient, dim3((unsigned)(((n_o + 255) / 256) * nslots)), dim3(256), 0, s, S, w, h, thr2, qn, mag, slot_stride, slot_stride, (n_o + 255) / 256, nslots);
        hipLaunchKernelGGL(k_cvote, dim3((unsigned)(((n_t + 255) / 256) * nslots)), dim3(256), 0, s, qn, w, h, quant, slot_stride, slot_stride, (n_t + 255) / 256, nslots);
        return;
    }
    dim3 grid((w + CT_W - 1) / CT_W, (h + CT_H - 1) / CT_H, nslots);
    hipLaunchKernelGGL(k_color_quantize, grid, dim3(256), 0, s, bgr, w, h, thr2, quant, mag, slot_stride);
}

// r06: the two halves of lmk_color_quantize for batches whose level-0 and level-1 gradients share a grid.  lmk_color_blur: the level's Gaussian blur into
// `scratch` alone (false: this shape takes the fused LDS-tiled kernel, nothing launched).  lmk_cgrad_levels: orientation + vote of both levels from their blurred
// images (false: not a batch / shape not supported, nothing launched).
bool lmk_color_blur(hipStream_t s, const u8* bgr, int w, int h, u8* scratch, size_t slot_stride, int nslots) {
    if (!(scratch && (w % 16) == 0 && ((uintptr_t)bgr & 15) == 0 && ((uintptr_t)scratch & 15) == 0 && (slot_stride % 16) == 0)) return false;
    u8* S = scratch;
    if ((g_cblur_variant == 4 || mx_auto(w, h, nslots)) && ((w * 3) % 32) == 0 && h >= 1) {
        const int gx = (w * 3 + 4 * MX_WAVE_BYTES - 1) / (4 * MX_WAVE_BYTES);
        const int strip_rows = mx_strip_rows();
        const int gy = (h + strip_rows - 1) / strip_rows;
        hipLaunchKernelGGL(k_cblur_mx, dim3((unsigned)(gx * gy * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, gx, gy, strip_rows, nslots);
        return true;
    }
    if (g_cblur_variant == 1 || (g_cblur_variant == 0 && sel_slots(nslots) < 16)) return false;
    if (h > 640) {
        const int n_w = (((w * 3 / 16) * ((h + 31) / 32) + 61) / 62 + 3) / 4;
        hipLaunchKernelGGL(k_cblur_sh<32>, dim3((unsigned)(n_w * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, n_w, nslots);
    } else {
        const int n_w = (((w * 3 / 16) * ((h + CBS_STRIP - 1) / CBS_STRIP) + 61) / 62 + 3) / 4;
        hipLaunchKernelGGL(k_cblur_sh<CBS_STRIP>, dim3((unsigned)(n_w * nslots)), dim3(256), 0, s, bgr, w, h, S, slot_stride, slot_stride, n_w, nslots);
    }
    return true;
}
static int g_cgrad_levels = 1;       // LM_TUNE_CGRAD_LEVELS: 1 (default) the two levels' gradients of a batch in one grid, 0 one launch per level
void lmk_set_cgrad_levels(int v) { g_cgrad_levels = v; }
bool lmk_cgrad_levels_wanted(int w0, int h0, int nslots) {
    return g_cgrad_levels != 0 && (g_cgrad_variant == 0 || g_cgrad_variant == 2 || g_cgrad_variant == 3) && sel_slots(nslots) >= 16 && (w0 % 32) == 0 && (h0 % 2) == 0;
}
bool lmk_cgrad_levels(hipStream_t s, const u8* S0, int w0, int h0, u8* q0, const u8* S1, int w1, int h1, u8* q1, float weak_threshold, size_t slot_stride, int nslots) {
    if (!lmk_cgrad_levels_wanted(w0, h0, nslots)) return false;
    if ((((uintptr_t)S0 | (uintptr_t)S1 | (uintptr_t)q0 | (uintptr_t)q1) & 15) != 0 || (slot_stride % 16) != 0 || (w1 % 16) != 0) return false;
    const float thr2 = weak_threshold * weak_threshold;
    const int ithr = thr2 >= 2147483648.f ? INT_MAX : (int)floorf(thr2);
    auto waves = [&](int w, int h, int strip) { return ((w / 16) * ((h + strip - 1) / strip) + 61) / 62; };
    auto blocks = [&](int w, int h, int strip) { return (waves(w, h, strip) + 3) / 4; };
    // level 0's strip as lmk_color_quantize chooses it for a launch of its own; level 1: 16 rows when it alone fills the chip, 8 otherwise
    const int s0 = (g_cgrad_variant == 3 || (h0 > 640 && (long)waves(w0, h0, 32) * nslots >= 3072)) ? 32 : ((long)waves(w0, h0, CG_STRIP) * nslots >= 1536 ? CG_STRIP : 8);
    const int s1 = (long)waves(w1, h1, 16) * nslots >= 1536 ? 16 : 8;
    const int g0 = blocks(w0, h0, s0), g1 = blocks(w1, h1, s1);
    const dim3 grid((unsigned)((g0 + g1) * nslots));
#define LM_CGL(A, B) hipLaunchKernelGGL((k_cgrad_levels<A, B>), grid, dim3(256), 0, s, S0, w0, h0, q0, g0, S1, w1, h1, q1, g1, ithr, slot_stride, nslots)
    if (s0 == 32) { if (s1 == 16) LM_CGL(32, 16); else LM_CGL(32, 8); }
    else if (s0 == CG_STRIP) { if (s1 == 16) LM_CGL(CG_STRIP, 16); else LM_CGL(CG_STRIP, 8); }
    else LM_CGL(8, 8);
#undef LM_CGL
    return true;
}

void lmk_depth_quantize(hipStream_t s, const u16* depth, int w, int h, int dist_thr, int diff_thr, const u8* lut,
                        bool lut_onehot, u8* quant, u8* scratch, size_t slot_stride, int nslots) {
    if (scratch && lut_onehot && (w % 8) == 0 && ((uintptr_t)depth & 15) == 0 && ((uintptr_t)scratch & 7) == 0 &&
        ((uintptr_t)quant & 7) == 0 && (slot_stride % 16) == 0) {
        const bool dm_batch = g_dmedian_variant == 2 || (g_dmedian_variant == 0 && sel_slots(nslots) >= 16);
        const int dm_rows = dm_batch ? DM_ROWS_BATCH : DM_ROWS;
        const int n_n = (w / 8) * h, n_m = (w / 8) * ((h + dm_rows - 1) / dm_rows);
        hipLaunchKernelGGL(k_dnormal, dim3((unsigned)(((n_n + 255) / 256) * nslots)), dim3(256), 0, s, depth, w, h, dist_thr, diff_thr,
                           lut, scratch, slot_stride, slot_stride, (n_n + 255) / 256, nslots);
        if (dm_batch) hipLaunchKernelGGL(k_dmedian<DM_ROWS_BATCH>, dim3((unsigned)(((n_m + 255) / 256) * nslots)), dim3(256), 0, s, scratch, w, h, quant,
                                             slot_stride, slot_stride, (n_m + 255) / 256, nslots);
        else hipLaunchKernelGGL(k_dmedian<DM_ROWS>, dim3((unsigned)(((n_m + 255) / 256) * nslots)), dim3(256), 0, s, scratch, w, h, quant,
                                slot_stride, slot_stride, (n_m + 255) / 256, nslots);
        return;
    }
    dim3 grid((w + DT_W - 1) / DT_W, (h + DT_H - 1) / DT_H, nslots);
    hipLaunchKernelGGL(k_depth_quantize, grid, dim3(256), 0, s, depth, w, h, dist_thr, diff_thr, lut, quant,
                       slot_stride);
}

template <int T, int SEG>
static void lm_fast_launch(hipStream_t s, const u8* q, int qpitch, int src_shift, int mode, int w, int h,
                           const u64* resp_tab, u8* lm, u32 ori_stride, size_t q_slot_stride, size_t lm_slot_stride,
                           int nslots, u32 plane_ori) {
    const int W = w / T;
    const int nseg = (W + SEG - 1) / SEG;
    dim3 grid((unsigned)(nseg * (h / T) * nslots), 1, 1);
#define LMF(SH, MD)                                                                                           \
    hipLaunchKernelGGL((k_lm_fast<T, SEG, SH, MD>), grid, dim3(256), 0, s, q, qpitch, w, h, resp_tab, lm, ori_stride, \
                       q_slot_stride, lm_slot_stride, nseg, nslots, mode == 2 ? plane_ori : 0u)
    if (src_shift) { if (mode == 1) LMF(1, 1); else if (mode == 2) LMF(1, 2); else LMF(1, 0); }
    else           { if (mode == 1) LMF(0, 1); else if (mode == 2) LMF(0, 2); else LMF(0, 0); }
#undef LMF
}

bool lmk_nibble_supported(int w, int h, int T) {
    (void)h;
    const int W = w / T;
    return (T == 2 || T == 4 || T == 5 || T == 8) && (w % 4 == 0) && (W % 8 == 0);
}

void lmk_linear_memories(hipStream_t s, const u8* q, int qpitch, int src_shift, int mode, int w, int h, int T,
                         const u64* resp_tab, u8* lm, u32 ori_stride, size_t q_slot_stride, size_t lm_slot_stride,
                         int nslots, u32 plane_ori) {
    const int W = w / T;
    const bool spread_only = mode == 1;
    const bool aligned = (w % 4 == 0) && (W % 4 == 0) && (qpitch % 4 == 0) && (((uintptr_t)q & 3) == 0) &&
                         (q_slot_stride % 4 == 0);
    if (aligned) {
#define LMF_ARGS s, q, qpitch, src_shift, mode, w, h, resp_tab, lm, ori_stride, q_slot_stride, lm_slot_stride, nslots, plane_ori
        switch (T) {
            case 2:
                if (mode == 1 && !src_shift && (w % 32) == 0 && (h % 2) == 0 && (qpitch % 16) == 0 && (((uintptr_t)q & 15) == 0) &&
                    (((uintptr_t)lm & 15) == 0) && (q_slot_stride % 16) == 0 && (lm_slot_stride % 16) == 0 && (((size_t)W * (h / 2)) % 16) == 0) {
                    const int n_l = (w / 32) * (h / 2);
                    hipLaunchKernelGGL(k_lm_spread2, dim3((unsigned)(((n_l + 255) / 256) * nslots)), dim3(256), 0, s, q, qpitch, w, h, lm,
                                       q_slot_stride, lm_slot_stride, (n_l + 255) / 256, nslots);
                    return;
                }
                lm_fast_launch<2, 128>(LMF_ARGS); return;
            case 4: lm_fast_launch<4, 64>(LMF_ARGS); return;
            case 5:
                // batches: the streaming kernel (one short wave per frame and band would not fill the chip for few frames)
                if (mode == 1 && !src_shift && sel_slots(nslots) >= 16 && (W % 8) == 0 && (h % 5) == 0 && (((uintptr_t)lm & 7) == 0) &&
                    (lm_slot_stride % 8) == 0 && (((size_t)W * (h / 5)) % 8) == 0) {
                    const int n_l = (W / 8) * (h / 5);
                    hipLaunchKernelGGL(k_lm_spread5, dim3((unsigned)(((n_l + 255) / 256) * nslots)), dim3(256), 0, s, q, qpitch, w, h, lm,
                                       q_slot_stride, lm_slot_stride, (n_l + 255) / 256, nslots);
                    return;
                }
                lm_fast_launch<5, 128>(LMF_ARGS); return;
            case 8:
                // r05: whole segments of 80 columns take 16-column units (half the scattered stores); everything else 40-column segments
                if (mode == 2 && (W % 80) == 0 && (((size_t)W * (h / 8)) % 16) == 0 && (((uintptr_t)lm & 15) == 0) && (lm_slot_stride % 16) == 0 && (ori_stride % 8) == 0 && (plane_ori % 2) == 0) {
                    lm_fast_launch<8, 80>(LMF_ARGS); return;
                }
                lm_fast_launch<8, 40>(LMF_ARGS); return;
            default: break;
        }
#undef LMF_ARGS
    }
    // generic path (any T, any width)
    // segment width: about 1024 linear-memory bytes per (band, segment), a multiple of 4 columns, and
    // few enough source bytes for LMK_MAX_LOADS loads per thread
    int seg = (1024 / (T * T)) & ~3;
    if (seg < 4) seg = 4;
    while (seg > 4 && (2 * T - 1) * (seg * T + T - 1) > LMK_MAX_LOADS * 256) seg -= 4;
    if (seg > W) seg = (W + 3) & ~3;
    int nseg = (W + seg - 1) / seg;
    int pitch = (seg * T + T + 3) & ~3;
    size_t shmem = 2048 + 2 * (size_t)(2 * T - 1) * pitch;
    dim3 grid(nseg, h / T, nslots);
#define LMK_LAUNCH(SH, SP)                                                                                    \
    hipLaunchKernelGGL((k_linear_memories<SH, SP>), grid, dim3(256), shmem, s, q, qpitch, w, h, T, seg, resp_tab, lm, \
                       ori_stride, q_slot_stride, lm_slot_stride)
    if (src_shift) { if (spread_only) LMK_LAUNCH(1, true); else LMK_LAUNCH(1, false); }
    else           { if (spread_only) LMK_LAUNCH(0, true); else LMK_LAUNCH(0, false); }
#undef LMK_LAUNCH
}

bool lmk_phases_supported(const LmPhaseArgs& a, int T0, int T1, int mode0, int mode1, bool lut_onehot) {
    // exactly the shapes the streaming kernels and k_lm_fast<5, 128, ., 1> (or k_lm_spread2) / <8, 40, ., 2> take
    if ((T0 != 5 && !(T0 == 2 && !a.depth)) || T1 != 8 || mode0 != 1 || mode1 != 2) return false;
    if ((a.w % 32) != 0 || (a.h % 2) != 0 || (a.slot_stride % 16) != 0 || a.nslots < 1) return false;
    const int w1 = a.w / 2, h1 = a.h / 2;
    if (!lmk_nibble_supported(w1, h1, 8) || (w1 / 8) % 4 != 0) return false;
    auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
    if (T0 == 5 && (a.w / 5) % 4 != 0) return false;
    if (T0 == 2 && (!al(a.lm_c0, 16) || (((size_t)(a.w / 2) * (a.h / 2)) % 16) != 0)) return false;
    if (!al(a.bgr0, 16) || !al(a.bgr1, 16) || !al(a.cs0, 16) || !al(a.cs1, 16) || !al(a.qc0, 16) || !al(a.qc1, 16)) return false;
    if (a.depth && (!lut_onehot || !al(a.depth, 16) || !al(a.ds, 8) || !al(a.qd0, 8))) return false;
    return true;
}

void lmk_preprocess_phases(hipStream_t s, const LmPhaseArgs& a, int T0) {
    const int w = a.w, h = a.h, w1 = w / 2, h1 = h / 2, n = a.nslots;
    const bool dep = a.depth != nullptr;
    auto per = [](int lanes) { return (lanes + 255) / 256; };
    const int g_blur0 = per((w * 3 / 16) * ((h + CB_ROWS - 1) / CB_ROWS)), g_blur1 = per((w1 * 3 / 16) * ((h1 + CB_ROWS - 1) / CB_ROWS));
    const int g_ori0 = per((w / 16) * h), g_ori1 = per((w1 / 16) * h1);
    const int g_vote0 = per((w / 16) * ((h + CVT_ROWS - 1) / CVT_ROWS)), g_vote1 = per((w1 / 16) * ((h1 + CVT_ROWS - 1) / CVT_ROWS));
    const int g_pyr = per((w1 / 8) * h1), g_nrm = per((w / 8) * h), g_med = per((w / 8) * ((h + DM_ROWS - 1) / DM_ROWS));
    // linear memories: segments per band (k_lm_fast); for T0 = 2 the streaming kernel's blocks per slot instead
    const int seg0 = T0 == 5 ? (w / 5 + 127) / 128 : per((w / 32) * (h / 2)), seg1 = (w1 / 8 + 39) / 40;
    const u32 b_lm0 = T0 == 5 ? (u32)(seg0 * (h / 5) * n) : (u32)(seg0 * n), b_lm1 = (u32)(seg1 * (h1 / 8) * n);
    auto launch = [&](auto kern, const LmPhaseGrid& pg) {
        const u32 nb = pg.nb[0] + pg.nb[1] + pg.nb[2] + pg.nb[3];
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, a, pg);
    };
    LmPhaseGrid p1 = {{(u32)(g_blur0 * n), dep ? (u32)(g_nrm * n) : 0u, (u32)(g_pyr * n), 0u}, {g_blur0, g_nrm, g_pyr, 0}};
    launch(k_phase<1, 5>, p1);
    LmPhaseGrid p2 = {{dep ? (u32)(g_med * n) : 0u, (u32)(g_blur1 * n), (u32)(g_ori0 * n), 0u}, {g_med, g_blur1, g_ori0, 0}};
    launch(k_phase<2, 5>, p2);
    LmPhaseGrid p3 = {{(u32)(g_vote0 * n), (u32)(g_ori1 * n), dep ? b_lm0 : 0u, dep ? b_lm1 : 0u}, {g_vote0, g_ori1, seg0, seg1}};
    launch(k_phase<3, 5>, p3);
    LmPhaseGrid p4 = {{(u32)(g_vote1 * n), b_lm0, 0u, 0u}, {g_vote1, seg0, 0, 0}};
    if (T0 == 5) launch(k_phase<4, 5>, p4); else launch(k_phase<4, 2>, p4);
    hipLaunchKernelGGL((k_lm_fast<8, 40, 0, 2>), dim3(b_lm1), dim3(256), 0, s, a.qc1, w1, w1, h1, a.resp_tab, a.lm_c1, a.ori_stride1,
                       a.slot_stride, a.slot_stride, seg1, n, a.plane_ori1);
}

bool lmk_batch_phases_supported(const LmPhaseArgs& a, int T0, int T1, int mode0, int mode1, bool lut_onehot) {
    if (!lmk_phases_supported(a, T0, T1, mode0, mode1, lut_onehot)) return false;
    if (sel_slots(a.nslots) < 16 || (a.w % 32) != 0 || (a.h % 2) != 0) return false;
    auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
    if (T0 == 5) {   // k_lm_spread5's shape
        const int W = a.w / 5;
        if ((a.w % 5) || (a.h % 5) || (W % 8) || (((size_t)W * (a.h / 5)) % 8) || !al(a.lm_c0, 8) || (a.depth && !al(a.lm_d0, 8)) || (a.slot_stride % 8)) return false;
    } else if (a.depth) return false;     // T0 == 2 is the colour-only pyramid
    return true;
}

void lmk_selftest_float_tail(hipStream_t s, unsigned long long* out2) {
    hipLaunchKernelGGL(k_selftest_float_tail, dim3(8192), dim3(256), 0, s, out2);
}
void lmk_preprocess_batch_phases(hipStream_t s, const LmPhaseArgs& a, int T0) {
    const int w = a.w, h = a.h, w1 = w / 2, h1 = h / 2, n = a.nslots;
    const bool dep = a.depth != nullptr;
    const bool tall = h > 640;                                     // 32-row strips at level 0 (fewer re-read window rows)
    auto per = [](int lanes) { return (lanes + 255) / 256; };
    auto strips = [](int rows, int strip) { return (rows + strip - 1) / strip; };
    auto gwaves = [&](int ww, int hh, int strip) { return (((ww / 16) * strips(hh, strip) + 61) / 62 + 3) / 4; };   // k_cgrad: 62 useful lanes per wave, 4 waves per block
    const int sb = tall ? 32 : 16, sg = tall ? 32 : 16;
    auto bwaves = [&](int ww, int hh, int strip) { return (((ww * 3 / 16) * strips(hh, strip) + 61) / 62 + 3) / 4; };   // k_cblur_sh: 62 useful lanes per wave
    const int g_nrm = per((w / 8) * h), g_blur0 = bwaves(w, h, sb), g_pyr = (((w / 16) * strips(h1, PD_STRIP) + 61) / 62 + 3) / 4;   // k_pyrdown16
    const int g_grad0 = gwaves(w, h, sg), g_med = per((w / 8) * strips(h, DM_ROWS_BATCH)), g_blur1 = bwaves(w1, h1, 16);
    const int g_grad1 = gwaves(w1, h1, 16);
    const int g_sp = T0 == 5 ? per(((w / 5) / 8) * (h / 5)) : per((w / 32) * (h / 2));
    const int seg1 = (w1 / 8 + 39) / 40;
    const u32 b_lm1 = (u32)(seg1 * (h1 / 8) * n);
    auto launch = [&](auto kern, const LmPhaseGrid& pg) {
        const u32 nb = pg.nb[0] + pg.nb[1] + pg.nb[2] + pg.nb[3];
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, a, pg);
    };
    if (dep) {
        // RGB-D: only kernels of one register class share a grid (see k_bsplit)
        const float thr2 = a.weak_threshold * a.weak_threshold;
        const int ithr = thr2 >= 2147483648.f ? INT_MAX : (int)floorf(thr2);
        const size_t fs = a.slot_stride;
        const LmPhaseGrid l0 = {{(u32)(g_nrm * n), (u32)(g_pyr * n), 0u, 0u}, {g_nrm, g_pyr, 0, 0}};
        const LmPhaseGrid h1g = {{(u32)(g_grad0 * n), (u32)(g_blur1 * n), 0u, 0u}, {g_grad0, g_blur1, 0, 0}};
        const LmPhaseGrid l2 = {{(u32)(g_sp * n), (u32)(g_sp * n), b_lm1, 0u}, {g_sp, g_sp, seg1, 0}};
        if (lmk_blur_pyrdown(s, a.bgr0, w, h, a.cs0, a.bgr1, a.qc0, fs, n)) {
            // blur(0) and pyrDown share the slot-interleaved launch (one read of the raw image); the normals go alone
            hipLaunchKernelGGL(k_dnormal, dim3((unsigned)(g_nrm * n)), dim3(256), 0, s, a.depth, w, h, a.dist_thr, a.diff_thr, a.normal_lut, a.ds, fs, fs, g_nrm, n);
        } else {
            launch(k_bsplit<0, 16>, l0);
            if (tall) hipLaunchKernelGGL(k_cblur_sh<32>, dim3((unsigned)(g_blur0 * n)), dim3(256), 0, s, a.bgr0, w, h, a.cs0, fs, fs, g_blur0, n);
            else hipLaunchKernelGGL(k_cblur_sh<16>, dim3((unsigned)(g_blur0 * n)), dim3(256), 0, s, a.bgr0, w, h, a.cs0, fs, fs, g_blur0, n);
        }
        if (tall) launch(k_bsplit<1, 32>, h1g); else launch(k_bsplit<1, 16>, h1g);
        hipLaunchKernelGGL(k_dmedian<DM_ROWS_BATCH>, dim3((unsigned)(g_med * n)), dim3(256), 0, s, a.ds, w, h, a.qd0, fs, fs, g_med, n);
        // level 1 alone: 8-row strips when 16-row ones would leave SIMDs without a wave (as lmk_color_quantize chooses)
        const int waves16 = ((w1 / 16) * strips(h1, 16) + 61) / 62;
        if ((long)waves16 * n >= 1536) hipLaunchKernelGGL(k_cgrad<16>, dim3((unsigned)(g_grad1 * n)), dim3(256), 0, s, a.cs1, w1, h1, ithr, a.qc1, fs, fs, g_grad1, n);
        else { const int g8 = gwaves(w1, h1, 8); hipLaunchKernelGGL(k_cgrad<8>, dim3((unsigned)(g8 * n)), dim3(256), 0, s, a.cs1, w1, h1, ithr, a.qc1, fs, fs, g8, n); }
        launch(k_bsplit<2, 16>, l2);
    } else {
        const LmPhaseGrid p1 = {{0u, (u32)(g_blur0 * n), (u32)(g_pyr * n), 0u}, {g_nrm, g_blur0, g_pyr, 0}};
        const LmPhaseGrid p2 = {{(u32)(g_grad0 * n), 0u, (u32)(g_blur1 * n), 0u}, {g_grad0, g_med, g_blur1, 0}};
        const LmPhaseGrid p3 = {{(u32)(g_grad1 * n), (u32)(g_sp * n), 0u, 0u}, {g_grad1, g_sp, g_sp, seg1}};
        const bool bp = lmk_blur_pyrdown(s, a.bgr0, w, h, a.cs0, a.bgr1, a.qc0, a.slot_stride, n);   // launch 1, slot-interleaved (one read of the raw image)
        if (T0 == 5) {
            if (tall) { if (!bp) launch(k_bphase<1, 5, 32, 32>, p1); launch(k_bphase<2, 5, 32, 32>, p2); launch(k_bphase<3, 5, 32, 32>, p3); }
            else { if (!bp) launch(k_bphase<1, 5, 16, 16>, p1); launch(k_bphase<2, 5, 16, 16>, p2); launch(k_bphase<3, 5, 16, 16>, p3); }
        } else {
            if (tall) { if (!bp) launch(k_bphase<1, 2, 32, 32>, p1); launch(k_bphase<2, 2, 32, 32>, p2); launch(k_bphase<3, 2, 32, 32>, p3); }
            else { if (!bp) launch(k_bphase<1, 2, 16, 16>, p1); launch(k_bphase<2, 2, 16, 16>, p2); launch(k_bphase<3, 2, 16, 16>, p3); }
        }
    }
    const int W1 = w1 / 8;
    if ((W1 % 80) == 0 && (((size_t)W1 * (h1 / 8)) % 16) == 0 && (((uintptr_t)a.lm_c1 & 15) == 0) && (a.slot_stride % 16) == 0 && (a.ori_stride1 % 8) == 0 && (a.plane_ori1 % 2) == 0) {
        // (16-column units: half the scattered stores, see d_lm_fast MODE 2)
        const int seg80 = W1 / 80;
        hipLaunchKernelGGL((k_lm_fast<8, 80, 0, 2>), dim3((unsigned)(seg80 * (h1 / 8) * n)), dim3(256), 0, s, a.qc1, w1, w1, h1, a.resp_tab, a.lm_c1, a.ori_stride1,
                           a.slot_stride, a.slot_stride, seg80, n, a.plane_ori1);
        return;
    }
    hipLaunchKernelGGL((k_lm_fast<8, 40, 0, 2>), dim3(b_lm1), dim3(256), 0, s, a.qc1, w1, w1, h1, a.resp_tab, a.lm_c1, a.ori_stride1,
                       a.slot_stride, a.slot_stride, seg1, n, a.plane_ori1);
}

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

void lmk_hsv_mask(hipStream_t s, const u8* bgr, int w, int h, const LmHsvRange& rg, const int* divtab, u32* mask, int wpr,
                  size_t in_stride, size_t mask_stride, int nslots) {
    hipLaunchKernelGGL(k_hsv_mask, dim3((unsigned)((wpr * h + 255) / 256), 1, (unsigned)nslots), dim3(256), 0, s, bgr, w, h, rg, divtab,
                       mask, wpr, in_stride, mask_stride);
}

bool lmk_hull_counts(hipStream_t s, const LmHullArgs& a) {
    if (a.n == 0) return true;
    // per wave: the hull's vertices + one (left, right) pair per image row
    const size_t shmem = 4 * (size_t)(2 * LM_HULL_MAX + 2 * a.h) * sizeof(int);
    if (shmem > 160 * 1024) return false;          // more rows than a CU's LDS holds (h > 4992): the caller checks on the host
    if (shmem > 64 * 1024) {
        // frames taller than 1920 rows need more than the default 64 KB of dynamic LDS
        static size_t raised = 0;
        if (shmem > raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_hull_counts), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            raised = shmem;
        }
    }
    hipLaunchKernelGGL(k_hull_counts, dim3((a.n + 3) / 4), dim3(256), shmem, s, a);
    return true;
}

void lmk_depth_counts(hipStream_t s, const LmDepthArgs& a) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_depth_counts, dim3(a.n), dim3(256), 0, s, a);
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
