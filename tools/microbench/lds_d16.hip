// r06 scratch test: do ds_read_u8_d16 / ds_read_u8_d16_hi PRESERVE the other half of their destination on this chip (gfx950 runs with SRAM ECC, for which
// LLVM assumes d16 loads may clobber it)?  Each lane reads two table bytes into one register, low half then high half, and the other order.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_d16 lds_d16.hip ; run: ./lds_d16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
__global__ void k(u32* out) {
    __shared__ unsigned char tab[256];
    tab[threadIdx.x] = (unsigned char)(threadIdx.x * 7 + 3);
    __syncthreads();
    const u32 a0 = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)tab + ((threadIdx.x * 5) & 255u);
    const u32 a1 = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)tab + ((threadIdx.x * 11 + 1) & 255u);
    u32 r = 0xDEADBEEFu, s = 0xDEADBEEFu;
    asm volatile("ds_read_u8_d16 %0, %2\n\tds_read_u8_d16_hi %0, %3\n\tds_read_u8_d16_hi %1, %3\n\tds_read_u8_d16 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(r), "+v"(s) : "v"(a0), "v"(a1));
    out[2 * threadIdx.x] = r; out[2 * threadIdx.x + 1] = s;
}
int main() {
    u32* d; hipMalloc(&d, 2 * 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    u32 h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const u32 b0 = (u32)(unsigned char)(((t * 5) & 255) * 7 + 3), b1 = (u32)(unsigned char)(((t * 11 + 1) & 255) * 7 + 3);
        const u32 want = b0 | (b1 << 16);
        if (h[2 * t] != want || h[2 * t + 1] != want) { if (bad < 4) printf("lane %d: got %08x %08x want %08x\n", t, h[2 * t], h[2 * t + 1], want); ++bad; }
    }
    printf("d16 pairs: %d of 256 lanes differ (%s)\n", bad, hipGetErrorString(hipGetLastError()));
    return 0;
}
