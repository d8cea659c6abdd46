// lm_k_post.hip -- f1, the GPU side of the reference's post-processing checks for gfx950 (CDNA4, wave64): k_hsv_mask (HSV in-range bit mask of
// resident frames), k_hull_counts (convex-hull fill counts, one wave per match), k_depth_counts (the depth check's early verdicts: counts of a
// crop's depths below / inside the window of passing medians), and their launchers.
#include "lm_dev.h"

namespace {

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


}  // namespace

// ================================================================================================
// launchers
// ================================================================================================
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
