/*
 * linemod_oracle.h -- CPU oracle for the LINE-MOD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar restatement of the algorithm the reference reaches through
 *   detector->match(in_imgs, detectorThreshold, matches, currentClass)
 *   (/root/reference/src/HighLevelLinemod.cpp:152)
 * and
 *   detector->addTemplate(templateImgs, in_modelName, maskRotated, &boundingBox)
 *   (/root/reference/src/HighLevelLinemod.cpp:93).
 *
 * The arithmetic itself lives in a third-party dependency that is NOT vendored in the
 * reference and NOT installed in the build image: OpenCV-contrib `rgbd` module,
 * `cv::linemod` (modules/rgbd/src/linemod.cpp, normal_lut.i), version unpinned
 * (/root/reference/CMakeLists.txt:53 `find_package(OpenCV REQUIRED)`, README.md:68
 * "OPENCV4").  The published algorithm is restated here from SURVEY.md Appendix A.
 *
 *                      *** PARITY UNPINNED ***
 * The reference holds no tests, golden vectors or known-answer fixtures for this path
 *  and neither the reference nor OpenCV can be built in this image,
 * so this oracle cannot be checked against cv::linemod outputs.  What IS pinned:
 *   - committed golden vectors produced by this oracle on the reference's own data files
 *     benchmark/img0.png + depth0.png (tests/golden/), so the oracle cannot drift silently;
 *   - known-answer tests: templates self-extracted from a frame must be found at their
 *     crop origin with similarity 100.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (line-mod-pipeline_amd/) never links or calls it.
 */
#ifndef LINEMOD_ORACLE_H
#define LINEMOD_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 4

typedef struct orc_feature { int32_t x, y, label; } orc_feature;          /* cv::linemod::Feature */
typedef struct orc_match   { int32_t x, y; float similarity; int32_t template_id; int32_t class_idx; } orc_match;
typedef struct orc_rect    { int32_t x, y, width, height; } orc_rect;

typedef struct orc_config {
    int32_t num_modalities;          /* 1 = ColorGradient, 2 = ColorGradient + DepthNormal (HighLevelLinemod.cpp:26-43) */
    int32_t pyramid_levels;          /* T_pyramid.size(), 2 in the reference                                          */
    int32_t T[ORC_MAX_LEVELS];       /* {5,8} RGB-D, {2,8} colour only (HighLevelLinemod.cpp:32,40)                    */
    float   weak_threshold;          /* ColorGradient default 10                                                       */
    int32_t num_features;            /* ColorGradient default 63                                                       */
    float   strong_threshold;        /* ColorGradient default 55                                                       */
    int32_t distance_threshold;      /* DepthNormal default 2000                                                       */
    int32_t difference_threshold;    /* DepthNormal default 50                                                         */
    int32_t depth_num_features;      /* DepthNormal default 63                                                         */
    int32_t extract_threshold;       /* DepthNormal default 2                                                          */
} orc_config;

typedef struct orc_detector orc_detector;

/* ---- tables (data parameters) ---- */
void orc_default_config(orc_config* cfg, int color_only);
void orc_default_similarity_lut(uint8_t lut[256], int variant);   /* 2 = upstream's table, SURVEY.md A.5 (the default of orc_create); 0 linear |i-j|, 1 circular */
void orc_default_normal_lut(uint8_t lut[8000]);

/* ---- stage functions (each one upstream helper; all buffers dense row-major) ---- */
void orc_gaussian7_u8c3(const uint8_t* src, int w, int h, uint8_t* dst);                       /* a3 GaussianBlur 7x7        */
void orc_sobel3_s16c3(const uint8_t* src, int w, int h, int16_t* dx, int16_t* dy);             /* a3 Sobel CV_16S            */
void orc_color_quantize(const uint8_t* bgr, int w, int h, float weak_threshold,
                        uint8_t* quantized, float* magnitude /* may be NULL */);              /* a3 quantizedOrientations   */
void orc_orientation_labels(const int32_t* dx, const int32_t* dy, size_t n, uint8_t* label);  /* a3 steps 4-5 on given gradients */
void orc_fast_atan2(const float* y, const float* x, size_t n, int variant, float* angle_degrees);   /* cv::fastAtan2 / cv::phase, either form */
int  orc_set_atan_variant(int variant);   /* bit 0: fused multiply-adds in the fastAtan2 polynomial (upstream's v_atan_f32 on an AVX2 build); returns the old value */
void orc_orientation_labels_variant(const int32_t* dx, const int32_t* dy, size_t n, int variant /* bit 0 fused, bit 1 double convertTo */,
                                    uint8_t* label, uint8_t* raw16 /* may be NULL */);
void orc_pyrdown_u8c3(const uint8_t* src, int w, int h, uint8_t* dst);                         /* a4 cv::pyrDown             */
void orc_depth_quantize(const uint16_t* depth, int w, int h, int distance_threshold,
                        int difference_threshold, const uint8_t* normal_lut, uint8_t* quantized); /* a5 quantizedNormals  */
void orc_median5_u8(const uint8_t* src, int w, int h, uint8_t* dst);                            /* a5 medianBlur(.., 5), BORDER_REPLICATE */
void orc_erode3_u8(const uint8_t* src, int w, int h, int iters, uint8_t* dst);                  /* extractTemplate: erode 3x3 */
void orc_dist_c(const uint8_t* src, int w, int h, float* dst);                                  /* extractTemplate: distanceTransform(DIST_C, 3) */
void orc_resize_nn_half(const uint8_t* src, int w, int h, uint8_t* dst);                       /* a6 NN resize               */
void orc_spread(const uint8_t* src, int w, int h, int T, uint8_t* dst);                        /* a8 spread                  */
void orc_response_maps(const uint8_t* spread, int n, const uint8_t* lut, uint8_t* maps);       /* a9 computeResponseMaps     */
void orc_linearize(const uint8_t* response, int w, int h, int T, uint8_t* linearized);         /* a10 linearize              */

/* ---- detector ---- */
orc_detector* orc_create(const orc_config* cfg);
void          orc_destroy(orc_detector* d);
void          orc_set_similarity_lut(orc_detector* d, const uint8_t lut[256]);
void          orc_set_normal_lut(orc_detector* d, const uint8_t lut[8000]);
int           orc_num_classes(const orc_detector* d);
int           orc_num_templates(const orc_detector* d);                 /* total over classes */
int           orc_class_num_templates(const orc_detector* d, int class_idx);

/* Append pre-extracted templates to a class (created if new).  descs is
 * n_templates * pyramid_levels * num_modalities entries ordered [template][level*M + modality],
 * each {width, height, pyramid_level, n_features}; features are concatenated in the same order.
 * Returns the class index, or -1 on error. */
typedef struct orc_template_desc { int32_t width, height, pyramid_level, num_features; } orc_template_desc;
int orc_add_class(orc_detector* d, const char* class_id, int n_templates,
                  const orc_template_desc* descs, const orc_feature* features);

/* Detector::addTemplate.  mask may be NULL (no mask).  Returns template id or -1. */
int orc_add_template(orc_detector* d, const char* class_id, const uint8_t* bgr, const uint16_t* depth,
                     const uint8_t* mask, int w, int h, orc_rect* bbox);

/* Read one template back. features may be NULL to query n only. */
int orc_get_template(const orc_detector* d, int class_idx, int template_id, int level, int modality,
                     int* width, int* height, orc_feature* features, int* n_features);

/* Detector::match for one class (class_idx >= 0) or all classes (-1).  Templates restricted to
 * template ids [tid_lo, tid_hi) of each class (pass 0, INT32_MAX for all).  threads<=1: serial
 * upstream loop; threads>1: OpenMP over templates (same result).  Returns number of matches
 * (may exceed cap; only cap are written) or -1 on error. */
int orc_match_frame(orc_detector* d, const uint8_t* bgr, const uint16_t* depth, int w, int h,
              float threshold, int class_idx, int tid_lo, int tid_hi, int threads,
              orc_match* out, int cap);

/* Same, but split so the bench can time only matchClass (a11-a15) on prebuilt linear memories. */
int orc_prepare_frame(orc_detector* d, const uint8_t* bgr, const uint16_t* depth, int w, int h);
int orc_match_prepared(orc_detector* d, float threshold, int class_idx, int tid_lo, int tid_hi,
                       int threads, orc_match* out, int cap);

/* a11-a13 only: candidates of the global scan of the prepared frame before refinement, (template_id, class_idx,
 * x, y) int32 quadruples sorted by (class, template, y, x).  Returns the count (may exceed cap_records). */
int orc_scan_candidates(orc_detector* d, float threshold, int class_idx, int tid_lo, int tid_hi, int threads,
                        int32_t* out, int cap_records);

/* Access intermediate buffers of the last prepared frame (for stage-by-stage parity tests).
 * what: 0 quantized, 1 spread, 2 linear memories (8 * T*T * W*H bytes, [ori][memory][pos]).
 * Returns byte size, copies at most cap bytes. */
int64_t orc_get_stage(const orc_detector* d, int what, int level, int modality, uint8_t* out, int64_t cap);

/* R-way merge of per-shard sorted match lists + adjacent-unique. */
int orc_merge(const orc_match* lists, const int32_t* counts, int n_lists, int stride, orc_match* out, int cap);

/* Process-wide: 0 = one bounds check per byte of the similarity sums (scalar), 1 = hoisted bounds check, vectorisable byte
 * adds (default).  Identical results. */
void orc_set_scan_mode(int mode);
int  orc_get_scan_mode(void);
const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
