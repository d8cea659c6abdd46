// r06 scratch microbenchmark, second part: which DS read form is cheapest for k_scanl's access pattern ONCE the host has scheduled the lists (lm_host.cpp
// schedule_lds_lists)?  A wave = 8 groups of 8 lanes; group g reads 8 consecutive 16-byte pieces from a dword-aligned base D_g; a 32-lane half = 4 groups.
// Patterns of the bases: 0 = unrelated (random dwords), 1 = "scheduled": the four groups of a half have the four residues D mod 4, 2 = all residues equal.
// Forms (8 features in flight before one wait, as a scan round): 0 = 2 x ds_read2_b32 offsets (0,1) (2,3) -- what k_scanl does --, 1 = 2 x ds_read2_b32
// offsets (0,2) (1,3), 2 = 4 x ds_read_b32, 3 = ds_read_b128 at the dword-aligned address (the hardware splits it), 4 = 2 x ds_read2_b32 (0,1) (2,3) + the fifth
// dword by ds_read_b32 (the form before the DPP move).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_read_forms2 lds_read_forms2.hip ; run: ./lds_read_forms2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 512
#define LDS_BYTES 153600

template <int FORM, int PAT>
__global__ __launch_bounds__(1024) void k_forms(u32* out, u32 seed, int check) {
    extern __shared__ u32 lds[];
    for (u32 i = threadIdx.x; i < LDS_BYTES / 4; i += 1024) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, li = lane & 7, grp = (threadIdx.x >> 3);
    u32 h = grp * 0x9E3779B9u + seed;
    u32 acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        u32 v[8][5];
        u32 dws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            u32 dw = (u32)(((unsigned long long)(h >> 8) * 37000ull) >> 24);   // group base in dwords
            if (PAT == 1) dw = (dw & ~3u) | ((grp + (u32)k) & 3u);
            if (PAT == 2) dw &= ~3u;
            dws[k] = dw;
            const u32 addr = dw * 4u + li * 16u;
            v[k][4] = 0;
            if (FORM == 0 || FORM == 4) {
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(*(unsigned long long*)&v[k][0]) : "v"(addr));
                asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(*(unsigned long long*)&v[k][2]) : "v"(addr));
                if (FORM == 4) asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(v[k][4]) : "v"(addr));
            } else if (FORM == 1) {
                unsigned long long a, b;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:2" : "=v"(a) : "v"(addr));
                asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:3" : "=v"(b) : "v"(addr));
                asm volatile("" : "+v"(a), "+v"(b));
                *(unsigned long long*)&v[k][0] = a; *(unsigned long long*)&v[k][2] = b;      // (order fixed after the wait below)
            } else if (FORM == 2) {
                asm volatile("ds_read_b32 %0, %1" : "=v"(v[k][0]) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(v[k][1]) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(v[k][2]) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(v[k][3]) : "v"(addr));
            } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(*(u32x4*)&v[k][0]) : "v"(addr));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FORM == 1) { const u32 t = v[k][1]; v[k][1] = v[k][2]; v[k][2] = t; }     // (0,2)(1,3) -> 0,1,2,3
            if (check) {
                for (int q = 0; q < 4; ++q) if (v[k][q] != (dws[k] + li * 4 + q) * 2654435761u + seed) atomicAdd(&out[1], 1u);
                if (FORM == 4 && v[k][4] != (dws[k] + li * 4 + 4) * 2654435761u + seed) atomicAdd(&out[1], 1u);
            }
            acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3] ^ v[k][4];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    u32* out; hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 256);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), LDS_BYTES + 256, 0, out, 7u, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), LDS_BYTES + 256, 0, out, 7u, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        u32 h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); hipMemset(out, 0, 64);
        double reads = 16.0 * ITERS * 8;
        printf("%-58s %8.1f us  %5.1f cycles per wave-feature per CU at 2.4 GHz   errors %u (%s)\n", name, ms * 1e3, ms * 1e6 / reads * 2.4, h[1], hipGetErrorString(hipGetLastError()));
    };
    run("2 x read2 (0,1)(2,3)   bases unrelated", k_forms<0, 0>);
    run("2 x read2 (0,1)(2,3)   bases scheduled (4 residues/half)", k_forms<0, 1>);
    run("2 x read2 (0,1)(2,3)   bases all residue 0", k_forms<0, 2>);
    run("2 x read2 (0,2)(1,3)   bases unrelated", k_forms<1, 0>);
    run("2 x read2 (0,2)(1,3)   bases scheduled", k_forms<1, 1>);
    run("2 x read2 (0,2)(1,3)   bases all residue 0", k_forms<1, 2>);
    run("4 x read_b32           bases unrelated", k_forms<2, 0>);
    run("4 x read_b32           bases scheduled", k_forms<2, 1>);
    run("4 x read_b32           bases all residue 0", k_forms<2, 2>);
    run("read_b128 dword-al.    bases unrelated", k_forms<3, 0>);
    run("read_b128 dword-al.    bases scheduled", k_forms<3, 1>);
    run("read_b128 16B-aligned  (bases all residue 0)", k_forms<3, 2>);
    run("2 x read2 + b32        bases unrelated", k_forms<4, 0>);
    run("2 x read2 + b32        bases scheduled", k_forms<4, 1>);
    return 0;
}
