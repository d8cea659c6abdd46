// lm_dev.h -- device-side helpers shared by the kernel sources (lm_k_preprocess.hip, lm_k_scan.hip, lm_k_refine.hip, lm_k_post.hip): wide and
// dword-aligned loads, DPP moves, 24-bit multiplies, packed 16-bit maxima, slot pointers, the XCD-affine block mapping, wave reductions.
// Everything is __forceinline__ in an unnamed namespace: each translation unit gets its own copy, nothing is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <climits>
#include "lm_common.h"
#include "lm_kernels.h"

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


__device__ __forceinline__ u32 wave_sum_u32(u32 v) {   // all lanes active
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);   // row_mirror: every lane holds its row's sum
    return (u32)__builtin_amdgcn_readlane((int)v, 0) + (u32)__builtin_amdgcn_readlane((int)v, 16) +
           (u32)__builtin_amdgcn_readlane((int)v, 32) + (u32)__builtin_amdgcn_readlane((int)v, 48);
}
// all lanes active; the result is wave-uniform (four DPP steps inside a row of 16 lanes, the four rows' maxima combined on the scalar unit).  Until
// round 6 this was six __shfl_xor steps = six dependent ds_bpermute_b32 round trips through the LDS crossbar per call -- k_scanl takes three maxima per
// wave item, k_refine one per pruning test and candidate.
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    u32 o;
    o = (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); v = v > o ? v : o;     // quad_perm [1,0,3,2]
    o = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); v = v > o ? v : o;     // quad_perm [2,3,0,1]
    o = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true); v = v > o ? v : o;    // row_half_mirror
    o = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true); v = v > o ? v : o;    // row_mirror: every lane holds its row's maximum
    const u32 a = (u32)__builtin_amdgcn_readlane((int)v, 0), b = (u32)__builtin_amdgcn_readlane((int)v, 16);
    const u32 c = (u32)__builtin_amdgcn_readlane((int)v, 32), d = (u32)__builtin_amdgcn_readlane((int)v, 48);
    const u32 ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

}  // namespace
