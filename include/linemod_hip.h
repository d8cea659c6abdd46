/*
 * linemod_hip.h -- C ABI of the MI355X-native LINE-MOD detector (liblinemod_hip.so).
 *
 * Drop-in boundary: this library replaces the object the reference holds as
 *     cv::Ptr<cv::linemod::Detector> detector;      /root/reference/include/HighLevelLinemod.h:102
 * Every entry point below names the cv::linemod::Detector call of the reference it stands in for
 * (complete list of calls crossing that seam: SURVEY.md section 8b).  Plain C types only: opaque
 * handle, caller-owned buffers with explicit strides/capacities, int status (0 = LM_OK), no
 * exceptions across the ABI; lm_last_error() returns the text of the last failure on the calling
 * thread.  All compute runs in hand-written HIP kernels for gfx950; there is NO CPU fallback --
 * every compute entry point fails with LM_ERR_NO_DEVICE when no HIP device is usable.
 *
 * Image formats (SURVEY.md 8b "Conventions"): colour = 8-bit BGR interleaved (CV_8UC3 as
 * delivered by cv::VideoCapture, detector.cpp:24), depth = uint16 millimetres (CV_16UC1,
 * detector.cpp:25-26).  Strides are in bytes.
 */
#ifndef LINEMOD_HIP_H
#define LINEMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_MAX_LEVELS 4
#define LM_MAX_FEATURES 63 /* upstream CV_Assert(features.size() <= 63): 63*4 = 252 fits a byte */

enum {
    LM_OK = 0,
    LM_ERR_INVALID = 1,    /* bad argument / violated precondition (the upstream CV_Assert cases)   */
    LM_ERR_NO_DEVICE = 2,  /* no usable HIP device: the product path has no CPU fallback            */
    LM_ERR_HIP = 3,        /* a HIP runtime call failed                                              */
    LM_ERR_OVERFLOW = 4,   /* more candidates/matches than the configured capacity    */
    LM_ERR_IO = 5,
    LM_ERR_EXTRACT = 6     /* addTemplate could not extract enough features (upstream returns -1)   */
};

/* cv::linemod::Feature {int x, y, label} */
typedef struct lm_feature { int32_t x, y, label; } lm_feature;
/* cv::linemod::Template header: {width, height, pyramid_level, features.size()} */
typedef struct lm_template_desc { int32_t width, height, pyramid_level, num_features; } lm_template_desc;
/* cv::linemod::Match {x, y, similarity, class_id, template_id}; class_id as index into lm_class_id() */
typedef struct lm_match_t { int32_t x, y; float similarity; int32_t template_id; int32_t class_idx; } lm_match_t;
typedef struct lm_rect { int32_t x, y, width, height; } lm_rect;

typedef struct lm_config {
    int32_t width, height;        /* level-0 frame size; fixed per detector (buffers are sized for it)      */
    int32_t num_modalities;       /* 1 = {ColorGradient}; 2 = {ColorGradient, DepthNormal}                   */
    int32_t pyramid_levels;       /* T_pyramid.size(); the reference uses 2                                  */
    int32_t T[LM_MAX_LEVELS];     /* {5,8} RGB-D, {2,8} colour only  (HighLevelLinemod.cpp:32,40)            */
    float   weak_threshold;       /* ColorGradient: 10                                                       */
    int32_t num_features;         /* ColorGradient: 63                                                       */
    float   strong_threshold;     /* ColorGradient: 55                                                       */
    int32_t distance_threshold;   /* DepthNormal: 2000                                                       */
    int32_t difference_threshold; /* DepthNormal: 50                                                         */
    int32_t depth_num_features;   /* DepthNormal: 63                                                         */
    int32_t extract_threshold;    /* DepthNormal: 2                                                          */
    int32_t device;               /* HIP device ordinal                                                      */
    int32_t shard_rank;           /* template-bank shard held by this detector: templates of every class     */
    int32_t shard_size;           /*   are split into shard_size contiguous template_id ranges (8e)          */
    int32_t max_candidates;       /* capacity of the device candidate buffer (0 = default 1<<18)             */
    int32_t max_matches;          /* capacity of the device match buffer     (0 = default 1<<18)             */
    int32_t frame_slots;          /* resident-frame slots for lm_match_batch (0 = default 8)                 */
    int32_t flags;                /* LM_FLAG_* (0 = defaults)                                                */
} lm_config;

/* Keep the lowest pyramid level's response memories at one byte per position (upstream's layout) instead
 * of the default two positions per byte.  Results are identical; this only selects the scan kernel. */
#define LM_FLAG_BYTE_RESPONSES 1
/* lm_match* / lm_match_end sleep on a blocking HIP event instead of spinning in hipStreamSynchronize: for hosts
 * that have fewer CPUs than busy processes (several GPUs' worth of matcher + exchange processes under a small
 * cgroup CPU quota).  Costs some wake-up latency per wait. */
#define LM_FLAG_BLOCKING_SYNC 2

typedef struct lm_detector lm_detector;

const char* lm_last_error(void);
const char* lm_version(void);

/* Fills the reference's two constructions (HighLevelLinemod.cpp:26-43): color_only=0 ->
 * {ColorGradient, DepthNormal}, T={5,8}; color_only=1 -> {ColorGradient}, T={2,8}. */
void lm_default_config(lm_config* cfg, int color_only, int width, int height);

/* cv::linemod::Detector(modalities, T_pyramid)            HighLevelLinemod.cpp:33-34,41-42 */
int  lm_create(const lm_config* cfg, lm_detector** out);
/* cv::Ptr::release()                                      HighLevelLinemod.cpp:50          */
void lm_destroy(lm_detector* det);

/* SIMILARITY_LUT / NORMAL_LUT are data parameters. */
int lm_set_similarity_lut(lm_detector* det, const uint8_t lut[256]);
int lm_set_normal_lut(lm_detector* det, const uint8_t lut[8000]);
int lm_get_similarity_lut(const lm_detector* det, uint8_t lut[256]);
int lm_get_normal_lut(const lm_detector* det, uint8_t lut[8000]);
/* 1 while the built-in NORMAL_LUT is active.  It is a documented SUBSTITUTE (azimuth of the cell centre, 8 bins), not
 * OpenCV's normal_lut.i, which could not be restated: DepthNormal labels then differ from cv::linemod's,
 * so a DepthNormal bank written by OpenCV must not be matched with it (lm_load_yaml warns on stderr).  Banks generated
 * through this library are self-consistent.  lm_set_normal_lut(det, <the 8000 bytes of normal_lut.i>) clears the flag. */
int lm_normal_lut_is_substitute(const lm_detector* det);

/* Detector::numClasses()  HighLevelLinemod.cpp:60,527 ; numTemplates() :65 ; classIds() :55,145,262,526 */
int         lm_num_classes(const lm_detector* det);
int         lm_num_templates(const lm_detector* det);                   /* total, all classes, whole bank */
int         lm_class_num_templates(const lm_detector* det, int class_idx);
const char* lm_class_id(const lm_detector* det, int class_idx);
int         lm_find_class(const lm_detector* det, const char* class_id); /* -1 if absent */
/* Detector::getT(level) :184 ; getModalities().size() :119 ; pyramidLevels() */
int lm_get_T(const lm_detector* det, int level);
int lm_num_modalities(const lm_detector* det);
int lm_pyramid_levels(const lm_detector* det);

/* Detector::readClass / bulk template upload              HighLevelLinemod.cpp:299
 * Appends n_templates pre-extracted template pyramids to class `class_id` (created if new).
 * descs: n_templates * pyramid_levels * num_modalities entries ordered [template][level*M + modality];
 * features: concatenated in the same order.  Template ids continue from the class's current count.
 * Returns the class index in *class_idx_out (may be NULL). */
int lm_add_class(lm_detector* det, const char* class_id, int n_templates, const lm_template_desc* descs,
                 const lm_feature* features, int* class_idx_out);

/* Detector::addTemplate(sources, class_id, object_mask, &bounding_box) -> template_id or -1
 *                                                         HighLevelLinemod.cpp:93-97
 * Quantisation runs on the GPU, feature selection on the host.  depth may be NULL for a colour-only
 * detector; mask may be NULL.  *template_id_out = -1 and status LM_ERR_EXTRACT when extraction fails. */
int lm_add_template(lm_detector* det, const char* class_id, const uint8_t* bgr, size_t bgr_stride,
                    const uint16_t* depth, size_t depth_stride, const uint8_t* mask, size_t mask_stride,
                    int* template_id_out, lm_rect* bbox_out);

/* Detector::getTemplates(class_id, template_id)[level*M + modality]   HighLevelLinemod.cpp:115-116,181-183
 * features may be NULL to query the count. */
int lm_get_template(const lm_detector* det, int class_idx, int template_id, int level, int modality,
                    int* width, int* height, lm_feature* features, int* num_features);

/* Detector::match(sources, threshold, matches, class_ids)             HighLevelLinemod.cpp:152
 * class_idx >= 0: that class only (the reference always passes exactly one id, :145); -1: all classes.
 * Host frame in, sorted unique matches out (total order SURVEY.md A.9: similarity desc, template_id
 * asc, class asc, y asc, x asc; adjacent-unique on (x, y, similarity, class)).  *n_out receives the
 * number of matches found; if it exceeds `cap` only the first cap are written and LM_ERR_OVERFLOW
 * is returned.  With shard_size > 1 only this shard's templates are searched (ids stay global).
 * Frames that lie inside a block from lm_host_alloc (pinned memory; e.g. a cv::Mat constructed over it) go to the
 * device without the staging copy (one transfer when the depth image directly follows the colour image). */
int lm_match(lm_detector* det, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
             float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out);

/* The same with upstream's class LIST (std::vector<String> class_ids; the reference builds a one-element list, :145):
 * the linear memories are built once and every named class is scanned against them; the list holds the matches of all
 * named classes in the total order (lm_match_t.class_idx tells them apart).  n_classes == 0 or {-1} = all classes. */
int lm_match_classes(lm_detector* det, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                     float threshold, const int32_t* class_idxs, int n_classes, lm_match_t* out, size_t cap, size_t* n_out);

/* Resident-frame path used by the benchmark and by batch-of-frames serving: upload once, match many.
 *
 * Streaming input (the reference's real call pattern is one fresh camera frame per detect() call,
 * detector.cpp:17-42 -> PoseDetection.cpp:66): uploads are ASYNCHRONOUS.  lm_upload_frame packs the (pageable,
 * strided) source into the slot's pinned staging buffer and enqueues the H2D copies on one of the detector's copy streams,
 * behind which it records the slot's "uploaded" event; it returns without waiting for the copy and without touching
 * any compute stream.  Every lm_match_slot / lm_match_batch / lm_match_begin makes its stream wait for the uploads
 * of the slots it reads (hipStreamWaitEvent), so
 *     begin(lane, A);  upload(B ...);  end(lane);  begin(lane, B);  upload(A ...);  ...
 * overlaps the transfer of step k + 1 with the compute of step k.  A slot that belongs to a match in flight cannot
 * be uploaded to (LM_ERR_INVALID).  The source buffer may be reused as soon as lm_upload_frame returns. */
int lm_upload_frame(lm_detector* det, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                    size_t depth_stride);
/* lm_upload_frame of the frame translated by (shift_x, shift_y) pixels with zeros shifted in -- the reference's translateImg on
 * both images before the match (principal-point shift: PoseDetection.cpp:54-59, 192-197) -- done while the staging buffer is
 * filled: one pass over the frame instead of a translated host copy followed by the staging copy.  Same asynchronous contract. */
int lm_upload_frame_shifted(lm_detector* det, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                            size_t depth_stride, int shift_x, int shift_y);
/* The same for a source in PINNED host memory (lm_host_alloc, hipHostMalloc, hipHostRegister): no staging copy, the
 * DMA engine reads the caller's buffer, which therefore must stay untouched until lm_upload_wait(det, slot) returns
 * or a match that covers the slot has been collected.  This is the path that reaches the PCIe rate. */
int lm_upload_frame_pinned(lm_detector* det, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                           size_t depth_stride);
/* A run of frames in ONE strided transfer: host frame i = [colour dense | depth dense] (colour only for a colour-only
 * detector) at frames + i * frame_stride (0 = densely packed), pinned, into slots [first_slot, first_slot + n_slots).
 * lm_upload_frame_pinned makes the same single copy per frame when it is handed such a contiguous pair. */
int lm_upload_frames_pinned(lm_detector* det, int first_slot, int n_slots, const uint8_t* frames, size_t frame_stride);
/* lm_upload_frame_shifted for a source in PINNED host memory (r05): the DMA engine copies the overlapping rectangle row by row
 * (hipMemcpy2DAsync) behind a memset of the destination: no staging copy and no host pass over the pixels.  Same contract as
 * lm_upload_frame_pinned.  Shifts beyond the frame size are clamped (an all-zero frame either way). */
int lm_upload_frame_pinned_shifted(lm_detector* det, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth,
                                   size_t depth_stride, int shift_x, int shift_y);
/* Staged uploads of PAGEABLE frames (r05): lm_upload_frame[_shifted] in three steps, so that a pool of host threads fills the pinned
 * staging buffers of a whole batch in parallel (the staging copy of a 1280 x 960 RGB-D frame is 6 MB: 277 us on one thread) while
 * the thread that owns the detector goes on:
 *     lm_stage_reserve(det, first, n)                    owner thread: the slots' earlier uploads have landed, their staging buffers exist
 *     lm_stage_rows(det, slot, ..., row0, row1) x many   ANY thread, disjoint row ranges of a slot concurrently: rows [row0, row1) of both
 *                                                        images, translated by (shift_x, shift_y) with zeros shifted in; host memory only
 *     lm_upload_staged(det, slot)                        owner thread, after all rows of the slot are in: the H2D copies + upload ticket
 * The source may be reused as soon as the lm_stage_rows calls covering it have returned. */
int lm_stage_reserve(lm_detector* det, int first_slot, int n_slots);
int lm_stage_rows(lm_detector* det, int slot, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                  int shift_x, int shift_y, int row0, int row1);
int lm_upload_staged(lm_detector* det, int slot);
/* Host waits until the upload of `slot` (-1: of every slot) has landed in device memory. */
int lm_upload_wait(lm_detector* det, int slot);
/* Pinned host memory for frame sources (hipHostMalloc); needs a HIP device. */
int  lm_host_alloc(size_t bytes, void** out);
void lm_host_free(void* p);
/* Pageable sources: number of pieces the staging memcpy is cut into so that it overlaps the DMA (default 1:
 * on the MI355X host every extra hipMemcpyAsync call cost more than the overlap won). */
int lm_set_stage_chunks(lm_detector* det, int chunks);
/* Launch-shape knobs (results never depend on them; tools/ and the tests flip them for A/B runs).  Keys 1, 2 and 10 (three concurrent
 * pre-processing chains for single frames; lm_match's copies on the copy streams; level-1 kernels inside the level-0 grids) lost their
 * A/B runs in r02 / r03 (DESIGN_HISTORY.md) and were removed with their code in r05: lm_set_tuning rejects them. */
/* LM_TUNE_COPY_STREAMS: copy streams the uploads are dealt to, slot -> stream round-robin (1..4, default 4: one
 *   in-order stream moved 0.6-0.9 MB images at 25.6 GB/s, several keep several DMA engines busy). */
#define LM_TUNE_COPY_STREAMS 3
/* LM_TUNE_CBLUR_VARIANT (process-wide): Gaussian blur kernel 0 = by batch size and frame size (default: one-shot below 16 frames;
 *   batches: frames of up to 2 MB on the matrix cores, larger ones the row walker), 1 = one-shot, (2 = r02's sliding window: deleted in r05, rejected,)
 *   3 = row walker with the column sums shared between neighbouring lanes (r03), 4 = the two banded products of the 8-bit
 *   Gaussian as v_mfma_i32_32x32x32_i8 (r04; rows of a multiple of 32 bytes, other shapes take 3).  Same bytes from all. */
#define LM_TUNE_CBLUR_VARIANT 4
/* LM_TUNE_CGRAD_VARIANT (process-wide): gradient orientation + 3x3 vote 0 = by batch size (default: two kernels below 16
 *   frames, the fused strip kernel from there), 1 = two kernels, 2 = fused, 3 = fused with 32-row strips (what tall
 *   images take in large batches). */
#define LM_TUNE_CGRAD_VARIANT 5
/* LM_TUNE_PHASE_MAX_SLOTS: calls of at most this many frames run the pre-processing (a3-a10) as five launches, each
 *   holding the independent kernels of one dependency level on ranges of the block index, instead of fourteen dependent
 *   ones (default 15: from 16 frames the batch kernels take over; 0 = off; two-level T = {5, 8} pyramids only, anything else takes the plain sequence). */
#define LM_TUNE_PHASE_MAX_SLOTS 6
/* LM_TUNE_BATCH_PHASES: calls of 16 or more frames run the pre-processing as launches in which batch kernels of one
 *   dependency level (and one register class) share a grid -- the level-1 kernels fill the tail of the level-0 ones: four
 *   (colour only) or seven (RGB-D) dependent launches instead of eleven.  0 = never, 1 = always, 2 (default) = when no
 *   other lane has a match in flight: measured r03, the fused launches win when a lane has the chip to itself and lose
 *   beside other lanes, whose kernels fill the tails anyway.  Two-level T = {5, 8} / {2, 8} pyramids of 32-pixel-aligned
 *   frames only; anything else takes the plain sequence.  Results never depend on it. */
#define LM_TUNE_BATCH_PHASES 7
/* LM_TUNE_PYRDOWN_VARIANT (process-wide): cv::pyrDown kernel 0 = by batch size (default: one lane per 8 output pixels below 16
 *   frames, the row-walking kernel whose column sums travel between lanes from there), 1 / 2 = force either. */
#define LM_TUNE_PYRDOWN_VARIANT 8
/* LM_TUNE_BLUR_PYR (process-wide): batches run the level-0 Gaussian blur and cv::pyrDown level 0 -> 1 as ONE launch whose
 *   blocks are interleaved per frame slot: 1 = [all blur tiles | all pyrDown tiles] of a slot back to back (r03); 2 = a slot's blur
 *   and pyrDown tiles dealt out evenly (r04), so the tiles of a band of rows run side by side and the second reader finds the
 *   rows in the L2; 3 (default) = 2 for frames of more than 2 MB, 1 below (measured: pays at 1280 x 960, not at 640 x 480);
 *   0 = two launches. */
#define LM_TUNE_BLUR_PYR 9
/* LM_TUNE_DMEDIAN_VARIANT (process-wide): 5 x 5 median of the normals' labels, output rows per lane 0 = by batch size (default:
 *   4 below 16 frames -- many short waves --, 16 from there: 20 rows of horizontal sums per 16 outputs instead of 8 per 4),
 *   1 / 2 = force either. */
#define LM_TUNE_DMEDIAN_VARIANT 11
/* LM_TUNE_BLUR_STRIP (process-wide): rows per strip of the level-0 Gaussian blur inside the blur + pyrDown launch of a batch:
 *   0 = by frame shape and batch size (default: as tall as still fills the chip), 16 / 32 / 64 = forced. */
#define LM_TUNE_BLUR_STRIP 12
/* LM_TUNE_WORK_WEIGHT: 1 (default, r04) = the few-frame / batch kernel selection of the pre-processing (the keys above that say
 *   "below 16 frames") counts a frame as (level-0 pixels / (640 x 480)) frames, at least one -- a call's WORK decides, so
 *   eight 1280 x 960 frames take the batch kernels; 0 = by frame count alone (r03).  Results never depend on it. */
#define LM_TUNE_WORK_WEIGHT 13
/* LM_TUNE_SORT_SPLIT: the device sort (a15) of lists longer than 1024 matches 0 = one workgroup per frame (r03), 1 = four chunk
 *   workgroups per frame + a merge launch, 2 (default) = the latter while the lists this detector collected lately were that
 *   long.  Same lists either way. */
#define LM_TUNE_SORT_SPLIT 14
/* LM_TUNE_SCAN_LIST_ORDER: order of a template's feature list at the scanned level: 0 = ascending linear-memory offsets (r01-r03:
 *   all features of orientation 0 first), 1 = dealt round-robin over the eight orientations, 2 = descending, 3 (default, r04) =
 *   greedy farthest-point order in (x, y, orientation): the first features sample the template's whole extent and all its
 *   orientations, so the exact pruning gives up on a work item sooner (49.7 -> 46.3 % of the feature loads on config 2).
 *   The similarity sums, and therefore every result, do not depend on it; changing it rebuilds the device bank. */
#define LM_TUNE_SCAN_LIST_ORDER 15
/* LM_TUNE_SCAN_FORM (r05): which kernel scans the lowest level when it keeps nibble-packed memories: 0 (default) = by cost -- the
 *   bit-plane scan k_scan1 (counts the features a position MISSES on one bit per position and orientation, then takes the exact
 *   sums of the few positions the miss bound leaves: k_scan1_exact) for detectors of ONE modality and calls of at least 8 frames
 *   (and a whole group of frames per wave), when it needs at least a fifth fewer waves than the nibble scan k_scan4 and the
 *   threshold is at least LM_TUNE_SCAN1_MIN_THRESHOLD (measured: +9-10 % on the
 *   colour-only 1280 x 960 workload, a loss with two modalities, where k_scan4's exact pruning stops far sooner than the miss
 *   bound); 1 = always k_scan4; 2 = k_scan1 whenever the level has planes.  The candidate lists, and therefore every result, are
 *   the same.  Under 0, a call whose own scan is k_scan1 writes the scanned level of its slots as one spread byte + the bit planes,
 *   WITHOUT the response memories: those slots are scanned by k_scan1 from then on (also by lm_match_prepared at a lower threshold),
 *   and a call that mixes them with slots prepared without planes is refused (LM_ERR_INVALID).
 *   r06: 3 = the bit-plane scan with a frame's planes in LDS (k_scanl) wherever they fit -- the scanned level's planes of all modalities
 *   within 153 600 bytes: 640 x 480 RGB-D at T = {5, 8} exactly --, k_scan1 where they do not.  One 1024-thread workgroup copies a frame's
 *   planes into its CU's LDS, counts misses from there (k_scan1 on such frames is bound by the L2 -> L1 line rate: every frame and feature
 *   is two 128-byte lines) and takes the survivors' exact sums from the frame's spread bytes, in LDS as well, in the same launch.  Under 0
 *   it is picked for calls of at least 16 frames at a threshold of at least LM_TUNE_SCAN1_MIN_THRESHOLD, for one modality or two; its
 *   slots too keep the spread byte + the planes instead of the response memories.  lm_get_scan_form_stats reports 1000 + the workgroups
 *   per frame as the "lanes per frame" of such a launch. */
#define LM_TUNE_SCAN_FORM 16
/* LM_TUNE_SCAN1_MIN_THRESHOLD (r05): similarity threshold in percent (0..100, default 50) below which LM_TUNE_SCAN_FORM 0 keeps
 *   k_scan4: the lower the threshold the more positions survive the miss bound and need their exact sums. */
#define LM_TUNE_SCAN1_MIN_THRESHOLD 17
/* LM_TUNE_CGRAD_LEVELS (r06): 1 (default) = a batch's colour-gradient orientation + vote passes of pyramid levels 0 and 1 run in ONE launch (level 1 of a
 *   640 x 480 frame is a handful of waves: behind level 0's workgroups they fill idle SIMDs instead of a launch of their own that leaves half the chip
 *   without a wave); 0 = one launch per level (r02-r05).  Same labels either way. */
#define LM_TUNE_CGRAD_LEVELS 18
/* LM_TUNE_SURVIVOR_QUEUE (r06): entries of a lane's survivor queues of the bit-plane scan k_scan1 (64 .. 16 777 216, rounded up to a multiple of 8; default
 *   1 048 576 = 131 072 per XCD queue; 8 bytes each, allocated on a lane's first bit-plane scan).  A wave whose survivors do not fit takes their exact sums itself, so
 *   the lists never depend on the value; small values exist for the tests that drive the full-queue path (concurrent reservations at the capacity, partial
 *   fits).  Setting it waits for the device and frees the queues of all lanes. */
#define LM_TUNE_SURVIVOR_QUEUE 19
int lm_set_tuning(lm_detector* det, int key, int value);
int lm_match_slot(lm_detector* det, int slot, float threshold, int class_idx, lm_match_t* out, size_t cap, size_t* n_out);
/* Matches slots [0, n_slots) back-to-back on the detector's streams; out is n_slots * cap_per_frame
 * records, counts n_slots entries. */
int lm_match_batch(lm_detector* det, int n_slots, float threshold, int class_idx, lm_match_t* out, size_t cap_per_frame,
                   int32_t* counts);
/* Detector::match(sources, threshold, matches, class_ids) with upstream's class LIST      HighLevelLinemod.cpp:145,152
 * on the resident frames of slots [first_slot, first_slot + n_slots): a3-a10 run ONCE per frame, every named class is
 * scanned against the same linear memories (one scan launch per run of neighbouring class indices), one refinement,
 * one sort.  A list holds the matches of all named classes in the total order; lm_match_t.class_idx tells them apart.
 * n_classes == 0 or {-1} = all classes; duplicates are ignored.  The per-class incremental cost is scan + refine only. */
int lm_match_batch_classes(lm_detector* det, int first_slot, int n_slots, float threshold, const int32_t* class_idxs,
                           int n_classes, lm_match_t* out, size_t cap_per_frame, int32_t* counts);
/* a11-a15 ONLY, for slots whose a3-a10 results are current: a match or lm_prepare_slot has run on the frame the slot
 * holds and neither an upload nor a LUT change has happened since (LM_ERR_INVALID otherwise -- never a silent
 * re-use of stale memories).  This is the reference's call pattern for several objects in one camera frame
 * (PoseDetection::detect per class name, PoseDetection.cpp:45-66 -> HighLevelLinemod.cpp:145-152): upload + prepare
 * once, one lm_match_prepared per class. */
int lm_match_prepared(lm_detector* det, int first_slot, int n_slots, float threshold, const int32_t* class_idxs,
                      int n_classes, lm_match_t* out, size_t cap_per_frame, int32_t* counts);

/* Asynchronous halves of lm_match_batch on one of four LANES (lane 0 = the detector's stream, lanes 1-3 = further
 * HIP streams with their own events and threshold tables; two lanes are what bench.py drives).  lm_match_begin enqueues a3-a15 for the resident frames
 * of slots [first_slot, first_slot + n_slots) and returns; lm_match_end waits for that lane and delivers the
 * lists exactly like lm_match_batch (out + i * cap_per_frame, counts[i]; i counts from first_slot).  Two lanes
 * working on disjoint slot ranges overlap each other's stages on the GPU (the scan is L1/L2-bound, the
 * preprocess passes VALU / fabric-bound) from ONE host thread:
 *     begin(0, A); begin(1, B);  loop { end(0); begin(0, A');  end(1); begin(1, B'); }
 * A lane's slots must not be uploaded to while it is busy (lm_upload_frame* refuse); uploads to the OTHER slots
 * may be issued at any time and are ordered against the lane that later reads them by per-slot events.  The
 * synchronous entry points refuse to run while a lane is busy. */
int lm_match_begin(lm_detector* det, int lane, int first_slot, int n_slots, float threshold, int class_idx);
/* lm_match_begin with a class list (see lm_match_batch_classes); collected by lm_match_end. */
int lm_match_begin_classes(lm_detector* det, int lane, int first_slot, int n_slots, float threshold,
                           const int32_t* class_idxs, int n_classes);
/* hipDeviceSynchronize() on the detector's device (what torch.cuda.synchronize() is for a torch program). */
int lm_synchronize(lm_detector* det);
int lm_match_end(lm_detector* det, int lane, lm_match_t* out, size_t cap_per_frame, int32_t* counts);

/* The lists of the last COMPLETED match on slots [first_slot, first_slot + n_slots) once more -- they stay in the slots' result blocks
 * until the next upload to / match on the slot: what a caller does after LM_ERR_OVERFLOW (counts[] held the capacity needed) instead
 * of a second pass over the GPU.  LM_ERR_INVALID when a slot holds no completed match or belongs to a match in flight. */
int lm_match_collect(lm_detector* det, int first_slot, int n_slots, lm_match_t* out, size_t cap_per_frame, int32_t* counts);

/* ---- f1: the colour check of the reference's match post-processing, batched on the GPU ----------
 * For every match of the list (any class / template of the bank, e.g. a merged multi-GPU list): templateMask =
 * fillPoly of the convex hull of the template's level-0 features moved to (match.x, match.y)
 * (HighLevelLinemod.cpp:113-135), counted once alone and once AND-ed with the colour mask = inRange(cvtColor(frame,
 * BGR2HSV), lower, upper) (:159-161): in_hull[i] and in_both[i] are the two countNonZero of colorCheck (:424-434),
 * whose verdict is in_both * 100 / in_hull > percentToPassCheck.  The frame is the one resident in `slot`.  The
 * reference builds a full-frame mask and counts the full frame once per tested match; here the colour mask is
 * one bit per pixel, built once per call, and one wave rasterises one hull. */
int lm_color_check_counts(lm_detector* det, int slot, const double lower_hsv[3], const double upper_hsv[3],
                          const lm_match_t* matches, size_t n, int64_t* in_hull, int64_t* in_both);

/* The same for a list whose matches lie in SEVERAL resident frames (slot_of_match[i] = the slot of match i's frame): one colour-mask
 * launch for the slots the list names, one hull launch for all matches, one wait -- a batch's colour checks in one call.
 * r05: both forms run on a stream and buffers of their own and only refuse slots that belong to a match in flight, so the checks of
 * batch k overlap the match of batch k + 1 on another lane. */
int lm_color_check_counts_slots(lm_detector* det, const int32_t* slot_of_match, const double lower_hsv[3], const double upper_hsv[3],
                                const lm_match_t* matches, size_t n, int64_t* in_hull, int64_t* in_both);
/* The colour masks of slots [first_slot, first_slot + n_slots) for one HSV range, enqueued on `lane`'s stream AHEAD of the match the
 * caller begins on that lane next: when the lane has been collected the masks are there, and a colour check of those slots for the
 * same range skips its mask launch (only the hull launch is left between lm_match_end and the counts).  An upload to a slot, or a
 * colour check with another range, invalidates the slot's mask. */
int lm_color_mask_prepare(lm_detector* det, int lane, int first_slot, int n_slots, const double lower_hsv[3], const double upper_hsv[3]);
/* The two halves of lm_color_check_counts_slots: begin enqueues the copies and the two launches on the colour-check stream and returns,
 * end waits and delivers the counts of the list begun (one check in flight per detector; `matches` may be reused after begin returns).
 * Between them the calling thread is free -- HighLevelLineMOD starts the first depth checks of a batch's groups meanwhile. */
int lm_color_check_begin_slots(lm_detector* det, const int32_t* slot_of_match, const double lower_hsv[3], const double upper_hsv[3],
                               const lm_match_t* matches, size_t n);
int lm_color_check_end(lm_detector* det, int64_t* in_hull, int64_t* in_both);

/* r06: the depth check's early verdicts on the GPU (the reference's medianMat + depthCheck, HighLevelLinemod.cpp:336-349,437-457).  The check wants to
 * know whether v[n / 5] after std::nth_element(v, v + n / 4) -- v the crop of the (translated) depth frame under the template's bounding box with depths
 * <= 1 replaced by 65535 -- lies inside the window [lo, hi] of medians that pass |depthDiff| < stepSize.  That element is one of the n / 4 + 1 smallest:
 * when more than n / 4 values lie below lo, or none lies inside [lo, hi], the verdict "outside" is certain without the selection.  For every query this
 * call counts, in the frame resident in slot `slot`, the values of the crop [x0, x1) x [y0, y1) (the caller clips it to the frame) below lo and inside
 * [lo, hi] -- one wave per query on the colour-check stream, beside the lanes.  The host then runs std::nth_element (it must stay the host library's: WHICH
 * of those elements comes back is implementation-defined) only for the checks no early verdict decides.  Detectors without a depth modality keep no
 * depth frame on the device: LM_ERR_INVALID.  begin / end as the colour check's halves (one depth query list in flight per detector, independent of a
 * colour check in flight). */
typedef struct lm_depth_query { int32_t x0, y0, x1, y1; int32_t lo, hi; int32_t slot; int32_t reserved; } lm_depth_query;
int lm_depth_counts_begin(lm_detector* det, const lm_depth_query* queries, size_t n);
int lm_depth_counts_end(lm_detector* det, uint32_t* below, uint32_t* inside);

/* ---- multi-GPU: template-bank shards + the ONE exchange step of the path ------------------------
 * One process per GPU; every rank creates its detector with lm_config.shard_rank / shard_size (contiguous template_id
 * ranges, global ids preserved), uploads the SAME frames and calls the same sequence of lm_match_begin_gathered /
 * lm_match_end_gathered.  lm_comm_init creates the RCCL communicators (librccl is loaded then, not before; one
 * communicator per lane so that the lanes' collectives never wait for each other); the ncclUniqueId goes from rank 0
 * to the others over TCP addr:port (one node: "127.0.0.1", e.g. MASTER_PORT + 1) -- no torch, no MPI, no second
 * process.  recs_per_frame_cap (0 = 256): average records per frame a rank may contribute to one gather; the gather
 * buffers have a fixed size of n_frames * recs_per_frame_cap records per rank so that no host round trip sits between
 * the two collectives.  More records than that -- or a frame with more than 4096 matches on some shard, which the
 * single-GPU path hands to the host sort -- is NOT an error: lm_match_end_gathered then runs a second, exactly sized
 * exchange (every rank sees the same status words, so all ranks take it together) and returns the same lists the
 * single-GPU path returns (the reference consumes ALL matches: HighLevelLinemod.cpp:206-253).  Only a shard that
 * overflows its own lm_config.max_candidates / max_matches fails (LM_ERR_OVERFLOW on every rank), as on one GPU.
 * The ids of the lanes' communicators travel in ONE rendezvous on `port`; on any failure nothing is left behind and
 * the call may be repeated. */
int lm_comm_init(lm_detector* det, int rank, int world, const char* addr, int port, int recs_per_frame_cap);
/* The rendezvous lm_comm_init uses, on its own (host only, no GPU): n bytes from rank 0's buf into every rank's buf. */
int lm_rendezvous_broadcast(int rank, int world, const char* addr, int port, void* buf, size_t n, int timeout_s);
int lm_comm_destroy(lm_detector* det);
int lm_comm_info(const lm_detector* det, int* rank, int* world);
/* lm_match_begin + behind the sort kernel, on the lane's stream: k_pack_lists (the lane's sorted lists back to back +
 * their lengths, device buffers), ncclAllGather of the lengths, ncclAllGather of the packed records, D2H of the lengths. */
int lm_match_begin_gathered(lm_detector* det, int lane, int first_slot, int n_slots, float threshold, int class_idx);
/* Waits for the lane, fetches from every rank's gathered run exactly the records of the frames THIS RANK OWNS (one
 * contiguous piece per rank: with R ranks 1 / R of the records crosses the PCIe link, not R x the gather capacity) and
 * merges them (R-way merge + adjacent-unique, lm_merge_frames): frames [n * rank / R, n * (rank + 1) / R) of the lane's n
 * frames (*first_frame, *n_frames), so neither the host work nor the D2H volume per rank grows with R.  out: the owned frames' merged lists back to back (cap records), counts[i] their lengths. */
int lm_match_end_gathered(lm_detector* det, int lane, lm_match_t* out, size_t cap, int32_t* counts, int* first_frame,
                          int* n_frames, size_t* n_out);
/* Every rank has reached this call and every rank's device is idle (ncclAllReduce + hipDeviceSynchronize): the
 * "barrier + synchronize" that brackets a timed region.  lm_comm_max: element-wise maximum of n <= 32 doubles. */
int lm_comm_barrier(lm_detector* det);
int lm_comm_max(lm_detector* det, double* values, int n);
/* "0000:c1:00.0"-style PCI bus id of the detector's device (hipDeviceGetPCIBusId): bench.py checks that the N ranks of
 * a node really sit on N different GPUs. */
int lm_device_pci_bus_id(lm_detector* det, char* out, size_t cap);

/* R-way merge of per-shard sorted lists + adjacent-unique: the step after the all-gather (8e).  Host-side. */
int lm_merge_matches(const lm_match_t* lists, const int32_t* counts, int n_lists, size_t stride, lm_match_t* out,
                     size_t cap, size_t* n_out);

/* The same for a whole batch (8e, one call per step instead of one per frame).  lm_pack_matches turns the
 * fixed-stride output of lm_match_batch into one contiguous run (what a rank sends); lm_merge_batch takes the
 * R gathered runs (rank r at packed + r * rank_stride, counts[r * n_frames + i] records for frame i) and writes
 * the merged + unique lists of all frames back to back with out_counts[i].  Host-side, multi-threaded. */
int lm_pack_matches(const lm_match_t* recs, size_t stride, const int32_t* counts, int n_frames, lm_match_t* out,
                    size_t cap, size_t* n_out);
int lm_merge_batch(const lm_match_t* packed, size_t rank_stride, const int32_t* counts, int n_ranks, int n_frames,
                   lm_match_t* out, size_t cap, int32_t* out_counts, size_t* n_out);
/* The same for the frames [frame_lo, frame_hi) only (out_counts: frame_hi - frame_lo entries): with R ranks each rank
 * merges the frames it owns, so the host work of the exchange does not grow with R. */
int lm_merge_frames(const lm_match_t* packed, size_t rank_stride, const int32_t* counts, int n_ranks, int n_frames,
                    int frame_lo, int frame_hi, lm_match_t* out, size_t cap, int32_t* out_counts, size_t* n_out);
/* The bookkeeping of lm_match_end_gathered on its own (host-only, no GPU, no RCCL: unit-testable at any world size).  all_cnt =
 * the all-gathered lengths, n_ranks runs of n_frames + 1 ints (per-frame record counts of the rank's packed run, then the rank's
 * status word: bit 0 lists exceed the fixed gather capacity, bit 1 a frame left to the host sort, bit 2 shard capacity overflow).
 * Out: OR of the status words; first rank with bit 2 (-1: none); the frames [*f0, *f1) = [n * rank / R, n * (rank + 1) / R) this
 * rank merges; counts[r * n_frames + i]; per rank the piece (start, length in records) of its packed run that holds exactly the
 * owned frames -- the only records lm_match_end_gathered copies to the host.  counts / piece_* may be NULL. */
int lm_gather_plan(const int32_t* all_cnt, int n_ranks, int n_frames, int rank, int* status, int* bad_rank, int* f0, int* f1,
                   int32_t* counts, uint64_t* piece_start, uint64_t* piece_len);
/* Records of the largest rank's packed run (at least 1): the per-rank buffer size of the sized second exchange. */
int lm_gather_max_total(const int32_t* counts, int n_ranks, int n_frames, uint64_t* max_total);

/* Template-bank persistence (Detector::write/writeClass/read/readClass, HighLevelLinemod.cpp:260,267,294,299):
 * own compact binary format, see DESIGN.md. */
int lm_save_bank(const lm_detector* det, const char* path);
int lm_load_bank(lm_detector* det, const char* path);
/* The reference's own file, "linemod_templates.yml.gz": cv::FileStorage YAML of Detector::write(fs) followed
 * by "classes" [ { Detector::writeClass } ... ] (HighLevelLinemod.cpp:256-270), gzip when the path ends in
 * .gz.  lm_load_yaml = Detector::read(fs.root()) + readClass per entry (HighLevelLinemod.cpp:292-303): the
 * file's pyramid_levels, T and modality types must equal the detector's, the modality parameters are taken
 * from the file, classes already present are left alone. */
int lm_save_yaml(const lm_detector* det, const char* path);
int lm_load_yaml(lm_detector* det, const char* path);
/* Top-level entries of a cv::FileStorage YAML file (plain or .gz): what the reference reads with fs["key"] >> x
 * from linemod_settings.yml (utility.cpp), models/<name>.yml (HighLevelLinemod.cpp:523-543) and
 * benchmark/pose0.yml.  Numbers: a scalar, a flow sequence, or the data of an !!opencv-matrix. */
int lm_yaml_numbers(const char* path, const char* key, double* out, size_t cap, size_t* n_out);
int lm_yaml_string(const char* path, const char* key, char* out, size_t cap);

/* ---- stage-level hooks (parity tests diff intermediate buffers against the oracle) ------------ */
/* a3 ColorGradient quantisation of an arbitrary w x h BGR image (dense).  magnitude may be NULL. */
int lm_stage_color_quantize(lm_detector* det, const uint8_t* bgr, int w, int h, float weak_threshold,
                            uint8_t* quantized, float* magnitude);
/* a4 cv::pyrDown of a dense BGR image -> (w/2) x (h/2) */
int lm_stage_pyrdown(lm_detector* det, const uint8_t* bgr, int w, int h, uint8_t* out);
/* a5 DepthNormal quantisation of a dense uint16 image */
int lm_stage_depth_quantize(lm_detector* det, const uint16_t* depth, int w, int h, uint8_t* quantized);
/* a8+a9+a10 spread(T) -> response maps -> linearize; out is 8 * T*T * (w/T)*(h/T) bytes [ori][memory][pos] */
int lm_stage_linear_memories(lm_detector* det, const uint8_t* quantized, int w, int h, int T, uint8_t* out);
/* Runs a3-a10 on the frame in `slot` and stops (no matching). */
int lm_prepare_slot(lm_detector* det, int slot);
/* Reads back an intermediate buffer of `slot` after lm_prepare_slot / lm_match_slot:
 * what 0 = quantized image [h][w], 2 = linear memories [ori][memory][pos] (pads stripped).
 * Returns the byte size in *size_out; copies min(size, cap). */
int lm_debug_read(lm_detector* det, int slot, int what, int level, int modality, uint8_t* out, size_t cap,
                  size_t* size_out);
/* a11-a13 only: candidates of the global scan at the lowest level as (template_id, class_idx, x, y)
 * quadruples of int32, sorted; for parity tests of the scan kernel in isolation. */
int lm_stage_scan(lm_detector* det, int slot, float threshold, int class_idx, int32_t* out, size_t cap_records,
                  size_t* n_out);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------- */
/* Times `iters` back-to-back launches of the similarity-scan kernel alone on the linear memories of
 * `slot` with HIP events on the launch stream; returns the average launch duration in microseconds
 * and the algorithmic bytes one launch reads (sum over templates/modalities of F*P). */
int lm_time_scan(lm_detector* det, int slot, float threshold, int class_idx, int iters, int variant,
                 double* avg_us_out, double* algorithmic_bytes_out);
/* The same over a batch of PREPARED slots (one launch = n_slots frames, what a lane-step launches); candidates are
 * counted, not stored.  variant 8 | 64 runs the exhaustive scan WITHOUT its shift-undo instructions -- wrong sums, a
 * timing experiment only (the upper bound of what pre-shifted copies of the linear memories could save). */
int lm_time_scan_batch(lm_detector* det, int first_slot, int n_slots, float threshold, int class_idx, int iters, int variant,
                       double* avg_us_out);
/* Self-test (no upstream counterpart): the depth-normal kernel's float tail takes 1 / len and sqrt by sequences that drop the
 * compiler's exponent-range handling; this runs every float of the tail's domain through both forms on the device, the reference
 * being the correctly rounded 1.0f / x and sqrtf.  out[0] / out[1] = number of floats whose reciprocal / square root differ
 * (0 / 0 expected); out[2] = floats on which the bare v_sqrt_f32 instruction differs (information only: the kernel does not
 * rely on it); out[3], out[4], out[5] = the same counts for the longer sequences the kernel used before (v_rcp + six fused steps;
 * v_sqrt + the +-1 ulp fix-up) and for v_sqrt + one step with v_rsq: all 0 expected; out[6] = floats on which a REJECTED candidate
 * (the reciprocal taken from the root's own v_rsq plus one Newton step) differs -- non-zero (84) by design: it shows that the sweep
 * discriminates; out[7] = 0.  The 8-word form dates from lm_version "0.3": a caller built against an older header passes 2 words. */
int lm_selftest_float_tail(lm_detector* det, uint64_t out[8]);
/* Per-stage average microseconds of the last lm_match_slot-style pipeline, measured with HIP events
 * over `iters` runs: out[0]=preprocess (a3-a10), out[1]=scan, out[2]=refine, out[3]=sort+copy. */
int lm_time_stages(lm_detector* det, int slot, float threshold, int class_idx, int iters, double out_us[4]);
/* Live profile of lm_match / lm_match_slot / lm_match_batch: when enabled every call brackets its
 * stages with HIP events on the launch stream and accumulates, per call: stage_us[0] preprocess
 * (a3-a10), [1] similarity scan (ONE kernel launch covering all frames of the call), [2] refinement,
 * [3] sort+publish; the algorithmic bytes the scan launches read; launches and frames.
 * lm_set_profiling(det, x) also resets the accumulators. */
int lm_set_profiling(lm_detector* det, int enable);
int lm_get_profile(lm_detector* det, double stage_us[4], double* scan_algorithmic_bytes, int64_t* launches,
                   int64_t* frames);
/* Accumulated with the profile above: HIP-event span of the exchange of lm_match_begin_gathered (behind the sort
 * kernel: k_pack_lists, the two ncclAllGather, the two D2H copies), the number of exchanges, and (counted with
 * profiling off too) how many lane-steps needed the sized second exchange because a rank's lists overflowed the
 * fixed gather capacity or a frame was left to the host sort (raise recs_per_frame_cap if that is the common case). */
int lm_get_exchange_profile(lm_detector* det, double* exchange_us, int64_t* exchanges, int64_t* fallbacks);
/* Work counters since lm_set_profiling: out[0] frames that went through a3-a10, out[1] scan launches, out[2]
 * refinement launches, out[3] sort launches (they count with profiling off too). */
int lm_get_stage_counts(lm_detector* det, int64_t out[4]);
/* Bytes the similarity scan's vector loads request per frame for `class_idx` (-1 = all classes): the on-chip
 * (L2 -> L1) traffic of the hot kernel, next to the algorithmic bytes lm_get_profile reports. */
int lm_scan_load_bytes(lm_detector* det, int class_idx, double* bytes_per_frame);
/* Counters of the last match on `slot`: scan candidates and refined matches before sort + unique. */
int lm_last_counts(lm_detector* det, int slot, uint32_t* candidates, uint32_t* matches_before_unique);
/* The nibble scan kernel stops loading a work item's features as soon as NO position it holds can still exceed the raw
 * threshold (partial sum + 4 x features to come <= threshold: exact, the candidate list never changes).  With the
 * statistics switched on every wave adds the (feature, work item) loads it made and the loads an exhaustive scan makes
 * to device counters (per wave pair of frames; lm_set_scan_stats also zeroes them). */
int lm_set_scan_stats(lm_detector* det, int enable);
int lm_get_scan_stats(lm_detector* det, uint64_t* features_loaded, uint64_t* features_unpruned);
/* The same at lane granularity (nibble kernel): with per-lane pruning a lane none of whose 32 positions can still reach
 * the threshold leaves the exec mask of the loads that follow.  lane_loads_issued: 16-byte lane-loads really made (a live
 * lane's right neighbour included); lane_loads_unpruned: 64 x the feature loads of an exhaustive scan. */
int lm_get_scan_lane_stats(lm_detector* det, uint64_t* lane_loads_issued, uint64_t* lane_loads_unpruned);
/* r05, the bit-plane scan (LM_TUNE_SCAN_FORM): out[0] = scan launches of lm_match* that took k_scan1 and out[1] = all of them since
 * the detector was created; out[2] = positions that survived k_scan1's miss bound and had their exact sums taken since
 * lm_set_scan_stats(1) (a superset of the candidates); out[3] = lanes per frame of the last scan launch (0: it was k_scan4). */
int lm_get_scan_form_stats(lm_detector* det, int64_t out[4]);
/* Selects the similarity-scan kernel variant used by lm_match* (0 = default; bits 0-1: features per load block;
 * bit 3 (value 8): no pruning, the plain exhaustive scan; bit 4 (value 16): wave-level pruning only, without the
 * per-lane exec masking; bit 5 (value 32): per-lane pruning also for one-modality detectors, which default to the
 * wave-level rule; see lm_k_scan.hip).  Bits 0-5 address the nibble scan k_scan4; bit 8 (value 256), the bit-plane scan k_scan1: the
 * waves take their survivors' exact sums themselves instead of queueing them for k_scan1_exact.  Every variant this call accepts gives
 * the SAME lists as variant 0.  The two timing experiments that skip work -- bit 6 (value 64, with bit 3: k_scan4 without its shift-undo)
 * and bit 7 (value 128: k_scan1 without the survivors' exact sums) -- and unknown bits are refused with LM_ERR_INVALID (r06): they exist
 * only as the `variant` argument of lm_time_scan / lm_time_scan_batch, which count candidates and store none. */
int lm_set_scan_variant(lm_detector* det, int variant);

#ifdef __cplusplus
}
#endif
#endif /* LINEMOD_HIP_H */
