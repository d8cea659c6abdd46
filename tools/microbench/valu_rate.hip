// Scratch microbenchmark: issue cost of single vector instructions on gfx950, cycles per wave-instruction per SIMD,
// at 1, 2 and 4 waves per SIMD (16 independent accumulators per wave, so no dependency stalls).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
#define ITERS 2000
#define OPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL_(NAME, ASM, CLOB)                                                                           \
    __global__ __launch_bounds__(256) void NAME(u32* out, u32 seed) {                               \
        u32 a[16];                                                                                  \
        u32 b = seed + threadIdx.x, c = seed * 3u + 1u;                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;               \
        asm volatile("s_mov_b64 vcc, 0x55\n s_mov_b64 s[20:21], 0x33" ::: "vcc", "s20", "s21");               \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) CLOB); \
        }                                                                                           \
        u32 s = 0;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s ^= a[i];                                   \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                 \
    }
#define KERNEL(NAME, ASM) KERNEL_(NAME, ASM, )
#define KERNELV(NAME, ASM) KERNEL_(NAME, ASM, : "vcc")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_i32 %0, %0, 3, 5")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL(k_min_sdwa, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNELV(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL(k_mad_i32_i16, "v_mad_i32_i16 %0, %1, %2, %0")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(k_or_b32, "v_or_b32 %0, %0, %1")
KERNEL(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_ashrrev, "v_ashrrev_i32 %0, 3, %0")
KERNEL(k_lshrrev, "v_lshrrev_b32 %0, 3, %0")
KERNELV(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[20:21]")
KERNELV(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNELV(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %1")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(k_and_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL(k_mov_dpp_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
__global__ __launch_bounds__(256) void k_pk_fma_f32(u32* out, u32 seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8]; f2 b = {1.0001f, 0.9999f}, c = {1e-3f, 2e-3f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)(seed + i), (float)threadIdx.x};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    if (s == 1.2345f) out[threadIdx.x] = (u32)s;
}
struct Entry { const char* name; void (*fn)(u32*, u32); };
int main() {
    u32* out; hipMalloc(&out, 4096);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double ghz = pr.clockRate / 1e6;
    printf("device %s, %d CUs, clock %.2f GHz (cycles below assume it)\n", pr.gcnArchName, pr.multiProcessorCount, ghz);
#define E(N) {#N, N}
    std::vector<Entry> es = {E(k_add_u32), E(k_and_b32), E(k_mov_b32), E(k_lshlrev), E(k_add3), E(k_perm), E(k_alignbit), E(k_pk_sub_i16),
        E(k_pk_add_u16), E(k_pk_max_i16), E(k_pk_mad_u16), E(k_dot2_i16), E(k_dot4_u8), E(k_mad_u24), E(k_mul_u24), E(k_mad_i24), E(k_mul_lo),
        E(k_max3), E(k_bfe), E(k_lshl_or), E(k_min_sdwa), E(k_sub_sdwa), E(k_bitop3), E(k_ffbl), E(k_mad_i32_i16), E(k_mov_dpp), E(k_cmp),
        E(k_sub_u32), E(k_or_b32), E(k_xor_b32), E(k_max_i32), E(k_min_u32), E(k_lshl_add), E(k_and_or), E(k_or3), E(k_ashrrev), E(k_lshrrev), E(k_cndmask_vcc), E(k_cndmask_s), E(k_addc), E(k_add_co), E(k_mul_i24), E(k_cvt_i32_f32), E(k_cvt_f32_ubyte0), E(k_fmac_f32), E(k_mad_u32_u16), E(k_sad_u8), E(k_pk_mul_lo_u16), E(k_pk_lshlrev_b16), E(k_add_u16), E(k_pk_add_f16), E(k_pk_fma_f16), E(k_alignbyte), E(k_and_sdwa), E(k_add_sdwa), E(k_mov_dpp_shr), E(k_add_dpp), E(k_fma_f32), E(k_add_f32), E(k_mul_f32), E(k_cvt_f32_i32), E(k_rcp_f32), E(k_sqrt_f32), E(k_pk_fma_f32)};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-16s %10s %10s %10s   (cycles per wave-instruction per SIMD)\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (auto& e : es) {
        printf("%-16s", e.name + 2);
        for (int wps : {1, 2, 4}) {
            const int blocks = pr.multiProcessorCount * wps;     // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)ITERS * 16 * wps;
            printf(" %10.2f", ms * 1e-3 * ghz * 1e9 / instr);
        }
        printf("\n");
    }
    return 0;
}
