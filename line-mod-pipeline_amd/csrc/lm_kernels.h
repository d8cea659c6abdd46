// lm_kernels.h -- host-callable launchers of the gfx950 kernels in lm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "lm_common.h"

// a4: cv::pyrDown on dense BGR (sw x sh) -> (sw/2 x sh/2)
void lmk_pyrdown(hipStream_t s, const u8* src, int sw, int sh, u8* dst);
// a3: ColorGradient quantisation of a dense w x h BGR image; mag may be null
void lmk_color_quantize(hipStream_t s, const u8* bgr, int w, int h, float weak_threshold, u8* quant, float* mag);
// a5: DepthNormal quantisation (normals + LUT + 5x5 median)
void lmk_depth_quantize(hipStream_t s, const u16* depth, int w, int h, int dist_thr, int diff_thr, const u8* normal_lut,
                        u8* quant);
// a6+a8+a9+a10: (optional NN half-size read of `q`) -> spread(T) -> 8 response maps -> linear memories.
// q is the quantised image to read with row pitch qpitch: src_shift 0 = this level's image, 1 = the finer
// level's image sampled at (2y, 2x).  lm points at the modality's first orientation block.
void lmk_linear_memories(hipStream_t s, const u8* q, int qpitch, int src_shift, int w, int h, int T,
                         const u64* resp_tab, u8* lm, u32 ori_stride);

struct LmScanArgs {
    const u8* lm;            // lowest level arena
    const u32* item_t;       // work items: bank-local template index
    const u32* item_chunk;   //             chunk of LM_SCAN_CHUNK positions
    int item_lo, n_items;
    const u32* scan_off;     // [nt][M][fpad] byte offsets into the arena
    const int* scan_P;       // [nt] template_positions
    const int* scan_n;       // [nt] total number of features at the lowest level
    int M, fpad;
    const int* raw_thr_by_n; // [128]
    int W, T;
    LmCand* cand;
    u32* cand_count;
    u32 cand_cap;
};
// a11+a12+a13: similarity scan over the lowest level fused with the threshold scan.
// variant selects the load strategy (0 = 16 B/lane unaligned vector loads; others see lm_kernels.hip).
void lmk_scan(hipStream_t s, const LmScanArgs& a, int variant);

struct LmRefineArgs {
    const u8* lm;            // arena of the level being refined at
    LmLevelGeom g;
    int M;
    const LmRefMeta* meta;   // [nt] for this level
    const LmRefFeat* feats;
    LmCand* cand;
    const u32* cand_count;
    u32 cand_cap;
    float threshold;
    // LAST level only: emit sort keys
    const int* t_global;
    const int* t_class;
    u64* keys;               // [match_cap][2]
    u32* match_count;
    u32 match_cap;
};
// a14: similarityLocal + argmax + rescore (+ threshold filter); last=true also emits sort keys.
void lmk_refine(hipStream_t s, const LmRefineArgs& a, bool last);
// pyramid_levels == 1: candidates become matches unrefined.
void lmk_emit_unrefined(hipStream_t s, const LmRefineArgs& a);
// a15: sort + adjacent-unique of up to LM_SORT_CAP keys in one workgroup; writes lm_match-layout records.
void lmk_sort_unique(hipStream_t s, const u64* keys, const u32* match_count, u32 match_cap, void* out_matches,
                     LmHeader* hdr);
