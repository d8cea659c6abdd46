// lm_common.h -- types shared by the HIP kernels and the host side of liblinemod_hip.so.
//
// HBM layout.  All frame slots live in two allocations with a fixed byte stride between slots, so a
// batch of resident frames is processed by one launch per stage (slot = blockIdx.z):
//   frame arena (slot stride = frame_stride), per slot, every block 256-B aligned:
//     bgr[l]      level-l BGR image, dense [h_l][w_l][3] u8 (level 0 uploaded, l>0 by k_pyrdown)
//     depth       level-0 depth, dense [h][w] u16
//     quant[l][m] quantised image of modality m at level l, dense [h_l][w_l] u8 (one-hot or 0)
//     lm[l]       linear-memory arena of level l:
//                   [modality m][orientation o][memory g = (y%T)*T + x%T][position (y/T)*W + x/T]  u8
//                 each orientation block is followed by `pad` zero bytes and the level ends with one
//                 more zero block; reads that upstream would make past an orientation's T*T x W*H
//                 cv::Mat land in those zeros (see oracle lm_read()).
//   aux arena (slot stride = aux_stride), per slot:
//     LmDevHeader counters, candidate list, sort keys, sorted output records
//   plus one host-mapped pinned LmHostBlock per slot that the sort kernel writes directly.
#pragma once
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define LM_SCAN_CHUNK 1008      // positions one wave covers in the similarity scan (63 lanes x 16 B; lane 63 feeds lane 62)
#define LM_SCAN4_CHUNK 1016     // nibble scan: positions per work item (half a wave: 32 lanes x 32 positions, the last 8 are
                                // polluted by the other half's data); two items per wave
#define LM_SCAN_FPAD 8          // feature lists are padded to a multiple of this with zero-block offsets
// r06, the bit-plane scan with the planes in LDS (k_scanl): a 1024-thread workgroup keeps ONE frame's miss planes of all modalities -- and, for the
// second stage, the frame's spread bytes in their place -- in the CU's 160 KB of LDS
#define LM_SCANL_LDS_BYTES 163840   // the whole LDS of a gfx950 CU
#define LM_SCANL_IMAGE_MAX 153600   // bytes of the planes (= of the spread bytes): M * T*T*wh = M * level pixels (640 x 480 RGB-D at level 1: exactly this)
#define LM_SCANL_TABLE_BYTES 2048   // behind the image: the response table of the second stage (a zero block during the first)
#define LM_SCANL_POS_BITS 15        // survivor entry: template << 15 | position
#define LM_SORT_CAP 4096        // matches sorted on the device (LDS); more are sorted by the host
#define LM_SORT_CHUNK 1024      // split form of the device sort: keys per chunk workgroup (LM_SORT_CAP / LM_SORT_CHUNK workgroups per frame)
#define LM_INLINE_MATCHES 2048  // records the sort kernel also writes straight into host-mapped memory
#define LM_DROPPED 0xFFFFFFFFu

struct LmLevelGeom {
    int w, h;          // quantised image size at this level
    int T;             // spread size = linear-memory stride
    int W, H;          // w/T, h/T
    int spread_only;   // 0: lowest level, 8 response memories per modality (scan); 1: refinement level,
                       //    one spread linear memory per modality (response LUT applied in k_refine)
    int nibble;        // lowest level only: response memories packed two positions per byte (position 2k in
                       //    the low nibble of byte k); strides below stay in bytes, bank offsets are in nibbles
    u32 wh;            // W*H = positions per linear memory
    u32 ori_stride;    // response arena: T*T*wh (256-aligned) + pad; unused for spread arenas
    u32 plane_ori;     // r05, nibble levels: bytes between the 8 miss-bit planes of a modality (k_scan1: one bit per position, T*T*wh / 8 bytes
                       //    (256-aligned) + pad), which follow its 8 response memories; 0: none
    u32 mod_stride;    // response arena: 8*ori_stride + 8*plane_ori; spread arena: T*T*wh (256-aligned) + pad
    u32 zero_off;      // offset (inside the level arena) of a zero block of `pad` bytes
    u32 arena_bytes;   // M*mod_stride + zero block
};

// One candidate / refined match in flight between the scan and the sort.
struct LmCand {
    u32 ti;      // bank-local template index, LM_DROPPED once filtered out
    int x, y;    // position at the level it was last refined at
    float sim;   // similarity as upstream stores it at that point
};

// Refinement feature at a level above the lowest: byte offset of the unshifted feature inside the
// level's spread arena (modality block included) in bits 0..28, its label in bits 29..31, plus its
// template coordinates for the bounds check of similarityLocal.
struct LmRefFeat {
    u32 off;
    int16_t x, y;
};

struct LmRefMeta {
    int width, height;   // tp[start].width/height: first modality's template size at this level
    int nfeat_total;     // sum over modalities of features.size() at this level
    u32 start[2];        // first LmRefFeat of modality m
    u32 count[2];
};

// Device counters of one slot; zero between frames (the sort kernel re-arms them).
struct LmDevHeader {
    u32 cand_count;      // candidates produced by the scan (may exceed capacity)
    u32 match_count;     // refined matches that passed the threshold (may exceed capacity)
    u32 pad[2];          // pad[0]: length of the sorted unique list k_sort_unique left in `out` (0xFFFFFFFF: none, the
                         //         host sorts), read by k_pack_lists
};

// Same layout as lm_match_t of the C ABI.
struct LmOutMatch { int x, y; float similarity; int template_id; int class_idx; };

struct LmHeader {
    u32 cand_count;
    u32 match_count;
    u32 out_count;         // matches after sort + unique (only when sorted on the device)
    u32 sorted_on_device;  // 0: capacity overflow or more than LM_SORT_CAP matches (host sorts the keys)
};

// Host-mapped result block of one slot, written by k_sort_unique.
struct LmHostBlock {
    LmHeader hdr;
    LmOutMatch rec[LM_INLINE_MATCHES];
};
