// lm_kernels.h -- host-callable launchers of the gfx950 kernels in lm_k_preprocess.hip (a3-a10), lm_k_scan.hip (a11-a13),
// lm_k_refine.hip (a14-a15, 8e) and lm_k_post.hip (f1).
// Every launcher processes `nslots` consecutive frame slots (grid.z) whose buffers are `*_slot_stride`
// bytes apart; pass stride 0 / nslots 1 for a single set of buffers.
#pragma once
#include <hip/hip_runtime.h>
#include "lm_common.h"

// a4: cv::pyrDown on dense BGR (sw x sh) -> (sw/2 x sh/2)
void lmk_pyrdown(hipStream_t s, const u8* src, int sw, int sh, u8* dst, size_t slot_stride, int nslots);
void lmk_set_pyrdown_variant(int v);   // 0: by batch size (default), 1: one lane per 8 output pixels (k_pyrdown8), 2: row-walking k_pyrdown16
// DepthNormalPyramid::pyrDown: nearest-neighbour half-size copy of a quantised image
void lmk_nn_half(hipStream_t s, const u8* src, int src_pitch, u8* dst, int dw, int dh, size_t slot_stride, int nslots);
// a3: ColorGradient quantisation of a dense w x h BGR image; mag may be null.  `scratch` (per slot,
// lmk_color_scratch_bytes(w, h), slot_stride apart like everything else) enables the 4-pass streaming
// form when w % 4 == 0; without it (or for other widths) the fused LDS-tiled kernel runs.
size_t lmk_color_scratch_bytes(int w, int h);
void lmk_set_cgrad_variant(int v);   // 0: by batch size (default), 1: k_corient + k_cvote, 2: fused k_cgrad, 3: fused, 32-row strips
void lmk_set_cblur_variant(int v);   // 0: by batch size (default), 1: one-shot blur, 3: row walker with shared column sums, 4: matrix cores
// blurred: the level's Gaussian-blurred image S is already in `scratch` (lmk_blur_pyrdown ran): orientation + vote only.
void lmk_color_quantize(hipStream_t s, const u8* bgr, int w, int h, float weak_threshold, u8* quant, float* mag,
                        u8* scratch, size_t slot_stride, int nslots, bool blurred = false);
// Batches: the level-0 blur (into scratch0, as lmk_color_quantize would) AND cv::pyrDown level 0 -> 1 in one slot-interleaved
// launch, so that the raw image is read from HBM once.  false: shape not supported, nothing launched (the caller launches
// the two kernels itself).
bool lmk_blur_pyrdown(hipStream_t s, const u8* bgr0, int w, int h, u8* scratch0, u8* bgr1, u8* quant0, size_t slot_stride, int nslots);
// r06: a batch's level-0 and level-1 gradients in one grid (the level-1 blur must have run): see lm_dev_color.h k_cgrad_levels
bool lmk_color_blur(hipStream_t s, const u8* bgr, int w, int h, u8* scratch, size_t slot_stride, int nslots);
bool lmk_cgrad_levels_wanted(int w0, int h0, int nslots);
bool lmk_cgrad_levels(hipStream_t s, const u8* S0, int w0, int h0, u8* q0, const u8* S1, int w1, int h1, u8* q1, float weak_threshold, size_t slot_stride, int nslots);
void lmk_set_cgrad_levels(int v);
void lmk_set_blur_pyr(int v);
void lmk_set_blur_pyr_interleave(int v);
void lmk_set_slot_weight(int w);   // frames count `w` times in the few-frame / batch kernel selection of this host thread's launches (1 = 640 x 480 frames)
void lmk_set_blur_strip(int v);
// a5: DepthNormal quantisation (normals + LUT + 5x5 median)
// scratch: w*h bytes per slot (rank codes between the two streaming passes), nullptr or a NORMAL_LUT that is
// not 0 / one-hot selects the LDS-tiled fallback kernel.
void lmk_depth_quantize(hipStream_t s, const u16* depth, int w, int h, int dist_thr, int diff_thr, const u8* lut,
                        bool lut_onehot, u8* quant, u8* scratch, size_t slot_stride, int nslots);
// a6+a8+a9+a10: (optional NN half-size read of `q`) -> spread(T) -> 8 response maps -> linear memories.
// q is the quantised image to read with row pitch qpitch: src_shift 0 = this level's image, 1 = the finer
// level's image sampled at (2y, 2x).  lm points at the modality's first orientation block.
// spread_only: write one spread linear memory (refinement levels) instead of 8 response memories.
// mode 0: 8 response memories (1 byte per position); 1: one spread memory; 2: response memories packed two
// positions per byte for the nibble scan (only when lmk_nibble_supported).
bool lmk_nibble_supported(int w, int h, int T);
void lmk_linear_memories(hipStream_t s, const u8* q, int qpitch, int src_shift, int mode, int w, int h, int T,
                         const u64* resp_tab, u8* lm, u32 ori_stride, size_t q_slot_stride, size_t lm_slot_stride,
                         int nslots, u32 plane_ori = 0);

struct LmScanArgs {
    const u8* lm;            // lowest level arena of slot 0
    size_t lm_slot_stride;
    const u32* item_t;       // work items: bank-local template index
    const u32* item_chunk;   //             chunk of LM_SCAN_CHUNK positions
    int item_lo, n_items;
    int nibble;              // 1: two positions per byte (k_scan4), scan_off in nibbles, fpad % 3 == 0
    const u32* scan_off;     // [nt][M][fpad] byte offsets into the arena
    const int* scan_P;       // [nt] template_positions
    const int* scan_n;       // [nt] total number of features at the lowest level | per-modality in-bounds counts << 8, << 16
    int M, fpad;
    const int* raw_thr_by_n; // [128]
    int W, T;
    LmDevHeader* hdr;        // slot 0; aux_slot_stride apart
    LmCand* cand;
    size_t aux_slot_stride;
    u32 cand_cap;
    unsigned long long* stat; // optional [1024][4] counters: features loaded / features an unpruned scan loads / lane-loads issued / (k_scan1) survivors whose exact sums were taken
    int wgs_per_slot, nslots; // filled by lmk_scan
    // bit-plane form (k_scan1, r05), L1 != 0: item_t / item_chunk are then the items of chunks of 128 L1 - 31 positions
    int L1, G1;               // lanes per frame, frames per wave (64 / L1)
    u32 L1_rcp16;             // ceil(65536 / L1): lane / L1 = (lane * L1_rcp16) >> 16 for lane < 64
    u32 delta_rcp16;          // ceil(65536 / delta), delta = 4 - the largest response below 4 of the similarity table
    const u32* off1;          // [nt][fpad1] BIT offsets of the features' miss planes in the arena (all modalities, one list)
    const u32* offn;          // [nt][fpad1] the same features' nibble offsets (exact sums of the survivors)
    int fpad1;
    int no_exact;             // measurement only (scan variant bit 7, WRONG lists): the survivors' exact sums are skipped
    unsigned long long* surv; // survivor queues of the launch's stream, one per XCD: [16 + x * (surv_cap / 8) + i] = template << 32 | slot << 20 | position;
    u32 surv_cap;             //    k_scan1_exact takes their exact sums (null / overflow: the wave does it itself).  [8 * surv_set + x] = entries appended
    int surv_set;             //    to queue x: two sets of counters, a launch uses one and its k_scan1_exact zeroes the other for the next launch
    int exact_spread;         // 1: the slots keep ONE spread byte per position instead of the response memories (d_lm_fast, bit 31 of plane_ori): the exact
    const u32* offs3;         //    sums go through the response table.  offs3 [nt][fpad1]: orientation << 29 | byte offset of the feature's spread memory
    const u64* resp_tab;      //    [256] responses of the 8 orientations to a spread byte
    // r06, the bit-plane scan with a frame's planes in LDS (k_scanl), lds_form != 0: one 1024-thread workgroup = (frame, share of the templates)
    int lds_form, R;          // R: workgroups per frame
    const u32* offl;          // [nt][fpad1] (LDS byte address of the feature's first dword) << 8 | bit shift
    const u32* offsl;         // [nt][fpad1] orientation << 29 | offset in the LDS image of the spread bytes
    const u32* litem;         // lane items [.][4]: template << 8 | unit of 128 positions (0xFFFFFFFF: none), the template's scan_n, scan_P, 0
    int litem_lo, n_litems;   // the launch's lane items
    u32 pb;                   // bytes of one miss plane, T*T*wh / 8
    u32 mod_stride, planes_off, plane_ori;   // arena: modality block stride, offset of a block's planes (8 * ori_stride), stride between them
    u32 tbl_bytes;            // LDS behind the image: zeros during the first stage (the padded list entries of ANY unit read them: ceil(wh / 128) * 16 + 32 bytes
                              //    at least), the response table during the second; then 16 bytes of queue header and the queue
    u32 queue_cap;            // survivor entries the LDS queue holds
    int dbg;                  // timing experiments of lm_time_scan_batch only (variant bits 9..11), WRONG lists: see k_scanl
};
// a11+a12+a13: similarity scan over the lowest level fused with the threshold scan.
// variant selects the unroll depth of the feature loop (0: 8 loads in flight, 1: 4, 2: 2).
// The variants lm_set_scan_variant accepts: bits 0-5 (k_scan4's load blocks and pruning rules) and bit 8 (k_scan1's survivors summed by the
// wave itself) leave the candidate lists as they are; bits 6 / 7 skip work and belong to lm_time_scan* alone.
#define LM_SCAN_VARIANT_SETTABLE (0x3F | 0x100)
void lmk_scan(hipStream_t s, const LmScanArgs& a, int variant, int nslots);

struct LmRefineArgs {
    const u8* lm;            // arena of the level being refined at, slot 0
    size_t lm_slot_stride;
    LmLevelGeom g;
    int M;
    const LmRefMeta* meta;   // [nt] for this level
    const LmRefFeat* feats;
    const u32* sim_lut;      // SIMILARITY_LUT as 64 dwords: [ori][lo 16 B | hi 16 B]
    LmDevHeader* hdr;
    LmCand* cand;
    u64* keys;               // [match_cap][2]
    size_t aux_slot_stride;
    u32 cand_cap;
    u32 match_cap;
    float threshold;
    const int* t_global;
    const int* t_class;
    const u32* plan;         // slot -> XCD plan of k_refine_plan ([8][plan_cap] slots + [8] lengths), or nullptr
    int plan_cap;
    int blocks_per_slot, nslots;  // filled by lmk_refine
    unsigned long long* stat;     // counting experiment (LM_REFINE_STAT=1): [0] candidates refined alone, [1] in pairs, [2] pair candidates the pruning dropped, [3] pairs in which BOTH were,
                                  //   [4] single candidates the pruning dropped, [5] candidates dropped by the final test; nullptr otherwise
};
// Balanced slot -> XCD lists for lmk_refine from the slots' candidate counts (nslots <= 1024, nslots % 8 == 0).
void lmk_refine_plan(hipStream_t s, const LmRefineArgs& a, int nslots, u32* plan, int plan_cap);
// a14: similarityLocal + argmax + rescore (+ threshold filter); last=true also emits sort keys.
void lmk_refine(hipStream_t s, const LmRefineArgs& a, bool last, int nslots);
// pyramid_levels == 1: candidates become matches unrefined.
void lmk_emit_unrefined(hipStream_t s, const LmRefineArgs& a, int nslots);

struct LmSortArgs {
    LmDevHeader* hdr;
    u64* keys;               // (hi, lo) per match; the split form sorts chunks of LM_SORT_CHUNK in place
    LmOutMatch* out;         // [LM_SORT_CAP] per slot
    size_t aux_slot_stride;
    LmHostBlock* host;       // host-mapped, slot 0
    size_t host_slot_stride;
    u32 cand_cap, match_cap;
    int split;               // 1: lists longer than LM_SORT_CHUNK keys are sorted as chunks by LM_SORT_CAP / LM_SORT_CHUNK workgroups per
                             // slot and merged by a second launch (k_merge_unique); same lists either way
};
// a15: sort + adjacent-unique of up to LM_SORT_CAP keys, one workgroup per slot (split: four + one).
void lmk_sort_unique(hipStream_t s, const LmSortArgs& a, int nslots);

struct LmPackArgs {
    const LmDevHeader* hdr;  // slot 0 of the range; aux_slot_stride apart (pad[0] = length of the sorted list)
    const LmOutMatch* out;   // sorted lists, LM_SORT_CAP records per slot
    size_t aux_slot_stride;
    int nslots;
    u32 cap_total;           // records `rec` can hold
    int* cnt;                // [nslots + 1]: list lengths, then the status word
    LmOutMatch* rec;         // packed lists
};
// 8e: the sorted lists of nslots frames back to back + their lengths (a rank's contribution to the all-gather).
void lmk_pack_lists(hipStream_t s, const LmPackArgs& a);

// ---- f1: batched colour check (HighLevelLinemod.cpp:113-135,159-161,424-434) -------------------------------------
#define LM_HULL_MAX 128      // hull vertices per template (two modalities x 63 features at most = 126 points)
struct LmHsvRange { int lo[3], hi[3]; };
// one bit per pixel: 8-bit HSV of the BGR image inside [lo, hi]; divtab = sdiv_table[256] | hdiv_table180[256]
void lmk_hsv_mask(hipStream_t s, const u8* bgr, int w, int h, const LmHsvRange& rg, const int* divtab, u32* mask, int wpr,
                  size_t in_stride, size_t mask_stride, int nslots);
// a3-a10 of FEW frames as five launches instead of fourteen (single-frame latency; the default two-level RGB-D / colour
// pyramid with T = {5, 8}, or {2, 8} without depth, only).  Every launch runs the independent kernels of one dependency level side by side, each
// on its own range of the block index:
//   1  blur(level 0)            | depth normals          | pyrDown(level 0 -> 1)
//   2  median of the normals    | blur(level 1)          | orientation(level 0)
//   3  vote(level 0)            | orientation(level 1)   | depth linear memories of levels 0 and 1
//   4  vote(level 1)            | colour linear memories of level 0
//   5  colour linear memories of level 1
// the device NORMAL_LUT buffer holds the 8000-byte table and, behind it, the same table as the rank codes k_dnormal writes
#define LMK_NORMAL_CODE_OFFSET 8000
struct LmPhaseArgs {
    const u8* bgr0; u8* bgr1; const u16* depth;      // level-0 colour image, level-1 colour image (written by launch 1), depth (or null)
    u8 *cs0, *cs1, *ds;                              // scratch: colour level 0 / 1 (lmk_color_scratch_bytes each), depth (w * h)
    u8 *qc0, *qc1, *qd0;                             // quantised images
    u8 *lm_c0, *lm_c1, *lm_d0, *lm_d1;               // linear memories: modality base pointers of levels 0 and 1
    int w, h;                                        // level 0; level 1 is w / 2 x h / 2
    float weak_threshold; int dist_thr, diff_thr;
    const u8* normal_lut; const u64* resp_tab;
    u32 ori_stride1;                                 // bytes between the response memories of level 1 (nibble packed)
    u32 plane_ori1;                                  // bytes between level 1's miss-bit planes (behind the 8 response memories of a modality), 0: none
    size_t slot_stride; int nslots;
};
bool lmk_phases_supported(const LmPhaseArgs& a, int T0, int T1, int mode0, int mode1, bool lut_onehot);
void lmk_preprocess_phases(hipStream_t s, const LmPhaseArgs& a, int T0);
// The same idea for BATCHES (16+ frames, the batch kernels: sliding-window blur, fused gradient + vote, streaming spread
// memories): the kernels of one dependency level share one grid, so the short level-1 launches fill the tail of the
// long level-0 ones; four launches per lane-step instead of eleven.
bool lmk_batch_phases_supported(const LmPhaseArgs& a, int T0, int T1, int mode0, int mode1, bool lut_onehot);
void lmk_preprocess_batch_phases(hipStream_t s, const LmPhaseArgs& a, int T0);
void lmk_set_dmedian_variant(int v);
// out2[0] / out2[1] += floats of k_dnormal's tail domain on which its short reciprocal / square root differ from the compiler's
// correctly rounded ones (device counters, zeroed by the caller)
void lmk_selftest_float_tail(hipStream_t s, unsigned long long* out2);

struct LmHullArgs {
    const LmOutMatch* matches; u32 n;
    const u32* class_base;       // [n_classes] first hull of the class in hull_off
    const u32* hull_off;         // [n_templates + 1] first vertex of every template's hull
    const int16_t* hull_xy;      // vertices (x, y) relative to the template origin
    const u32* mask; int wpr;    // colour bit mask of the frame
    const int* match_slot;       // optional [n]: match i lies in the frame whose mask starts mask_slot_words * match_slot[i] words behind `mask`
    size_t mask_slot_words;
    int w, h;
    long long* out;              // [n][2]: pixels in the hull, pixels in the hull with the colour bit set
};
// false: the frame has more rows than the kernel's per-wave row table fits into LDS (nothing was launched)
bool lmk_hull_counts(hipStream_t s, const LmHullArgs& a);

// r06, the depth check's early verdicts on the GPU (medianMat, HighLevelLinemod.cpp:336-349,437-457): per query the crop [x0, x1) x [y0, y1) of a resident
// depth frame with depths <= 1 counted as 65535 -- how many values lie below `lo`, how many inside [lo, hi]
struct LmDepthQuery { int x0, y0, x1, y1; int lo, hi; int slot; int pad; };      // (same layout as lm_depth_query of the C ABI)
struct LmDepthArgs {
    const u16* depth;            // depth image of slot 0 (the resident, already translated frame)
    size_t slot_stride;          // bytes between the slots' frames
    int w, h;
    const LmDepthQuery* q; u32 n;
    u32* out;                    // [n][2]: values below lo, values in [lo, hi]
};
void lmk_depth_counts(hipStream_t s, const LmDepthArgs& a);
