// lm_dev_depth.h -- a5, the depth modality: bilateral normals in closed form + NORMAL_LUT + 5x5 median (LDS-tiled
// k_depth_quantize; streaming k_dnormal -> k_dmedian; the float tail's device self-test).  Included by lm_k_preprocess.hip only.
// The normalisation is a float island in the oracle's operation order (explicit round-to-nearest intrinsics).
#pragma once
#include "lm_dev.h"
#include "lm_median25.h"

namespace {

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


}  // namespace
