// r06 scratch microbenchmark: LDS read throughput for the access pattern of a bit-plane scan whose planes live in LDS:
// groups of 8 lanes read 8 consecutive 16-byte pieces starting at a DWORD-aligned (not 16-byte-aligned) group base, 8 groups per wave at
// unrelated bases.  Forms: ds_read_b128 at 16-byte-aligned addresses (the reference), ds_read_b128 at dword-aligned addresses, the same plus
// a ds_read_b32 of the 5th dword, and 2 x ds_read2_b32 + ds_read_b32.  One 1024-thread workgroup per CU, 150 KB of LDS.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip ; run: ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 512
#define LDS_BYTES 153600

template <int FORM>
__global__ __launch_bounds__(1024) void k_lds(u32* out, u32 seed, int check) {
    extern __shared__ u32 lds[];
    for (u32 i = threadIdx.x; i < LDS_BYTES / 4; i += 1024) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, li = lane & 7, grp = (threadIdx.x >> 3);
    u32 h = grp * 0x9E3779B9u + seed;
    u32x4 acc = {0, 0, 0, 0};
    u32 acc5 = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            u32 dw = (u32)(((unsigned long long)(h >> 8) * 37000ull) >> 24);   // group base in dwords
            if (FORM == 0) dw &= ~3u;
            const u32 addr = dw * 4u + li * 16u;
            u32x4 v; u32 w = 0;
            if (FORM <= 2) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
                if (FORM == 2) asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(w) : "v"(addr));
            } else if (FORM == 4 || FORM == 5) {
                unsigned long long t0, t1;
                asm volatile("ds_read_b64 %0, %1" : "=v"(t0) : "v"(addr));
                asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(t1) : "v"(addr));
                if (FORM == 4) asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(w) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)");
                v[0] = (u32)t0; v[1] = (u32)(t0 >> 32); v[2] = (u32)t1; v[3] = (u32)(t1 >> 32);
            } else {
                u32 a0, a1, a2, a3;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(*(unsigned long long*)&v) : "v"(addr));
                unsigned long long t;
                asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(t) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(w) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)");
                v[2] = (u32)t; v[3] = (u32)(t >> 32);
                (void)a0; (void)a1; (void)a2; (void)a3;
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
            if (check) {
                // expected values from the fill pattern
                for (int q = 0; q < 4; ++q) if (v[q] != (dw + li * 4 + q) * 2654435761u + seed) atomicAdd(&out[1], 1u);
                if (FORM >= 2 && FORM != 5 && w != (dw + li * 4 + 4) * 2654435761u + seed) atomicAdd(&out[1], 1u);
            }
            acc ^= v; acc5 ^= w;
        }
    }
    u32 s = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ acc5;
    if (s == 0x12345678u) out[0] = s;
}

template <int FORM>
__global__ __launch_bounds__(1024) void k_lds_pipe(u32* out, u32 seed) {     // 8 reads in flight before one wait (as a scan round would)
    extern __shared__ u32 lds[];
    for (u32 i = threadIdx.x; i < LDS_BYTES / 4; i += 1024) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, li = lane & 7, grp = (threadIdx.x >> 3);
    u32 h = grp * 0x9E3779B9u + seed;
    u32x4 acc = {0, 0, 0, 0};
    u32 acc5 = 0;
    for (int it = 0; it < ITERS; ++it) {
        u32x4 v[8]; u32 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            u32 dw = (u32)(((unsigned long long)(h >> 8) * 37000ull) >> 24);
            if (FORM == 0) dw &= ~3u;
            const u32 addr = dw * 4u + li * 16u;
            w[k] = 0;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(addr));
            if (FORM == 2) asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(w[k]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc ^= v[k]; acc5 ^= w[k]; }
    }
    u32 s = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ acc5;
    if (s == 0x12345678u) out[0] = s;
}

int main() {
    u32* out; hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int ncu = 256;
    auto run = [&](const char* name, auto kern, int check) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 256);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), LDS_BYTES + 256, 0, out, 7u, check);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), LDS_BYTES + 256, 0, out, 7u, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        u32 h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); hipMemset(out, 0, 64);
        // per CU: 16 waves x ITERS x 8 wave-reads
        double reads = 16.0 * ITERS * 8;
        printf("%-34s %8.1f us  %6.2f ns per wave-read per CU = %5.1f cycles at 2.4 GHz   errors %u   (%s)\n", name, ms * 1e3, ms * 1e6 / reads, ms * 1e6 / reads * 2.4, h[1], hipGetErrorString(hipGetLastError()));
    };
    run("b128 16B-aligned, wait each", k_lds<0>, 1);
    run("b128 dword-aligned, wait each", k_lds<1>, 1);
    run("b128 dword-aligned + b32, wait each", k_lds<2>, 1);
    run("2 x read2_b32 + b32, wait each", k_lds<3>, 1);
    run("2 x b64 dword-aligned + b32", k_lds<4>, 1);
    run("2 x b64 dword-aligned", k_lds<5>, 1);
    auto runp = [&](const char* name, auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 256);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), LDS_BYTES + 256, 0, out, 7u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), LDS_BYTES + 256, 0, out, 7u);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double reads = 16.0 * ITERS * 8;
        printf("%-34s %8.1f us  %6.2f ns per wave-read per CU = %5.1f cycles at 2.4 GHz   (%s)\n", name, ms * 1e3, ms * 1e6 / reads, ms * 1e6 / reads * 2.4, hipGetErrorString(hipGetLastError()));
    };
    runp("b128 16B-aligned, 8 in flight", k_lds_pipe<0>);
    runp("b128 dword-aligned, 8 in flight", k_lds_pipe<1>);
    runp("b128 dword-al. + b32, 8 in flight", k_lds_pipe<2>);
    return 0;
}
