// lm_k_preprocess.hip -- a3-a10 of the LINE-MOD match path for gfx950 (CDNA4, wave64): the kernels of lm_dev_color.h, lm_dev_depth.h and
// lm_dev_memories.h, the level-fused kernels that run several of their device functions in one grid (k_phase: few frames; k_bphase / k_bsplit:
// a lone lane of a batch), and every pre-processing launcher (lmk_pyrdown .. lmk_preprocess_batch_phases) with the LM_TUNE_* state they read.
// Every kernel takes the buffers of frame slot 0 plus the byte stride between slots, so a batch of resident frames is one launch per stage.
#include "lm_dev_color.h"
#include "lm_dev_depth.h"
#include "lm_dev_memories.h"

namespace {

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
