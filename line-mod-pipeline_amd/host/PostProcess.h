// PostProcess.h -- host glue around the hot path: what the reference's
// HighLevelLineMOD does with the match list after detector->match() returns
// (/root/reference/src/HighLevelLinemod.cpp:157-175, 206-253, 336-349, 351-379, 382-515) and the
// principal-point shift of PoseDetection (/root/reference/src/PoseDetection.cpp:54-59,192-197).
//
// Plain C++17, no OpenCV / GLM: the few pieces of both that the reference uses are restated here
// (8-bit BGR->HSV + inRange, convex hull + polygon fill, nth_element quartile, lookAt / quaternion
// maths).  OpenCV and GLM are absent from the build image, so -- like the oracle -- these restatements
// are PARITY UNPINNED against the real libraries; each function says what it restates.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/linemod_hip.h"
#include "HighLevelLinemod.h"

namespace lmamd {

// HighLevelLinemod.h:130-148 `struct Template`; written raw to linemod_tempPosFile.bin (:272-284):
// glm::vec3 (12 B) | glm::qua<float> x,y,z,w (16 B) | cv::Rect (16 B) | uint16 | pad  = 48 B
struct TemplatePose {
    float translation[3];
    float quat_xyzw[4];
    int32_t bb[4];          // x, y, width, height
    uint16_t medianDepth;
    uint16_t pad;
};
static_assert(sizeof(TemplatePose) == 48, "layout of the reference's raw Template record");

// models/<name>.yml: HSV range of the object (HighLevelLinemod.cpp:523-543)
struct ModelProperties {
    double lowerColorRange[3] = {0, 0, 0};
    double upperColorRange[3] = {255, 255, 255};
};

struct PostProcessSettings {   // the TemplateGenerationSettings fields the post-processing reads
    bool onlyColorModality = false;
    uint16_t videoWidth = 640, videoHeight = 480;
    float fy = 1045.69141f;
    uint16_t stepSize = 50;
    uint16_t percentToPassCheck = 50;
    uint16_t numberWantedPoses = 1;
    float radiusThresholdNewObject = 45.f;
    float discardGroupRatio = 35.f;
    bool useDepthImprovement = true;
    float depthOffset = 30.f;
};

// ---- mini GLM ------------------------------------------------------------------------------------
struct Mat4 { float m[4][4]; };   // column-major like glm: m[col][row]
Vec3 cross(const Vec3& a, const Vec3& b);
Vec3 normalize(const Vec3& v);
float length(const Vec3& v);
Mat4 lookAt(const Vec3& eye, const Vec3& center, const Vec3& up);   // glm::lookAt (right-handed)
Mat4 mul(const Mat4& a, const Mat4& b);
Mat4 transpose(const Mat4& a);
Mat4 toMat4(const Quat& q);                                          // glm::mat4_cast
Quat toQuat(const Mat4& m);                                          // glm::quat_cast of the upper 3x3
Vec3 rotate(const Vec3& v, float angle, const Vec3& axis);           // glm::rotate (gtx/rotate_vector)

// ---- image helpers -------------------------------------------------------------------------------
// cvtColor(BGR2HSV) on 8-bit + inRange(lower, upper): 255 where all three HSV channels are in range.
void bgr2hsv_inrange(const uint8_t* bgr, int w, int h, size_t stride, const double lower[3], const double upper[3],
                     std::vector<uint8_t>& mask);
// PoseDetection::translateImg: integer shift with zero fill (warpAffine with a pure translation).
void translate_u8c3(const uint8_t* src, int w, int h, int ox, int oy, std::vector<uint8_t>& dst);
void translate_u16(const uint16_t* src, int w, int h, int ox, int oy, std::vector<uint16_t>& dst);
// HighLevelLineMOD::medianMat (:336-349): zeros -> 65535, crop, nth_element at n/4, returns element n/position.
uint16_t median_mat(const uint16_t* depth, int w, int h, Rect bb, uint8_t position, int shift_x = 0, int shift_y = 0);
// The same when only medians inside [win_lo, win_hi] matter (the depth check): false = outside, decided -- when it can be -- from the crop's
// smallest element and the number of elements below win_lo instead of the selection; true + *median = the reference's value.
bool median_mat_in_window(const uint16_t* depth, int w, int h, Rect bb, uint8_t position, int shift_x, int shift_y, int win_lo, int win_hi,
                          uint16_t* median, bool* decided_early = nullptr);
struct Pt { int x, y; };
std::vector<Pt> convex_hull(std::vector<Pt> pts);                   // cv::convexHull (hull vertices, ccw, no collinear points)
// templateMask (:113-135) restricted to the hull's bounding box: returns (pixels in hull, pixels in hull
// with colour mask set) -- the two countNonZero of colorCheck (:424-434).
void hull_counts(const std::vector<Pt>& hull, const uint8_t* color_mask, int w, int h, long* in_hull, long* in_both);

// ---- the reference's post-processing pipeline -----------------------------------------------------
struct MatchGroup { Pt position; std::vector<uint32_t> matchIndices; };   // PotentialMatch, HighLevelLinemod.h:150-160
std::vector<MatchGroup> group_similar_matches(const std::vector<lm_match_t>& matches, float radius);      // :206-229
std::vector<MatchGroup> discard_small_groups(const std::vector<MatchGroup>& groups, float ratio);          // :232-253

// calculateTemplatePose (:351-379): pose of a template rendered from `cameraPosition` with in-plane angle.
void calculate_template_pose(Vec3 cameraPosition, int16_t inplaneRot, float translation_out[3], float quat_xyzw_out[4]);

class PostProcessor {
public:
    PostProcessor(lm_detector* det, const PostProcessSettings& s) : det(det), st(s) {}
    // detectTemplate's tail (:157-175): colour mask, grouping, per-group colour/depth checks, poses.
    // gpu_slot >= 0: the (same) frame is resident in that slot of the detector and the colour checks of ALL matches of
    // the surviving groups are computed there in one batch (lm_color_check_counts: colour mask once per frame, one
    // wave per hull) instead of one full-frame fillPoly + two countNonZero per tested match; -1: on the host.
    std::vector<std::vector<ObjectPose>> run(const std::vector<lm_match_t>& matches, const uint8_t* bgr, size_t bgr_stride,
                                             const uint16_t* depth, size_t depth_stride,
                                             const std::vector<TemplatePose>& templates, const ModelProperties& props,
                                             int gpu_slot = -1);
    // where run()'s wall time went, accumulated over all PostProcessors of the thread since resetTimes() (bench.py's pose_e2e leg)
    struct Times {
        double grouping = 0, colour = 0, depth = 0, pose = 0; long colour_checks = 0, depth_checks = 0, poses = 0, groups = 0;
        long depth_decided_early = 0; // depth checks whose failing verdict was certain from the crop's minimum / count below the pass window (no nth_element)
        void add(const Times& o) { grouping += o.grouping; colour += o.colour; depth += o.depth; pose += o.pose; colour_checks += o.colour_checks;
                                   depth_checks += o.depth_checks; poses += o.poses; groups += o.groups; depth_decided_early += o.depth_decided_early; }
    };
    static Times& times();
    static void resetTimes() { times() = Times(); }
    // run() in two steps (r04): prepare = grouping + the colour verdict inputs of every match of the surviving groups (uses the
    // detector when gpu_slot >= 0: NOT thread-safe, call it from the thread that owns the detector); finish_group = the sequential
    // accept / break loop of ONE group (colour verdict, depth check, pose), which touches nothing shared and may run on any thread --
    // the groups of a frame, and of different frames and classes, are independent.  run() = prepare + finish_group in order.
    struct Prepared {
        std::vector<MatchGroup> groups;
        std::vector<int64_t> gin, gboth;          // GPU colour counts, [gpos[match index]]
        std::vector<size_t> gpos;
        std::vector<lm_match_t> todo;             // the matches whose counts are wanted, in gpos order (prepare_groups)
        std::vector<uint8_t> color_mask;          // host colour check: the frame's HSV in-range mask
        bool gpu = false, failed = false;
        // r06: the depth check's GPU counts (lm_depth_counts_begin), [gpos[match index]]: crop values below the pass window, inside it, and the crop's size
        std::vector<uint32_t> dbelow, dinside, dsize;
        bool depth_counts = false;
    };
    // tm: where the call's times go (nullptr = the calling thread's times(); pool tasks hand in their own block)
    Prepared prepare(const std::vector<lm_match_t>& matches, const uint8_t* bgr, size_t bgr_stride,
                     const std::vector<TemplatePose>& templates, const ModelProperties& props, int gpu_slot, Times* tm = nullptr);
    // prepare() in two halves for callers that batch the GPU colour counts of several classes of one frame into ONE
    // lm_color_check_counts call (same HSV range): the grouping + the list of matches whose counts are wanted (host only), then the
    // counts, in p.todo's order
    Prepared prepare_groups(const std::vector<lm_match_t>& matches, const std::vector<TemplatePose>& templates, Times* tm = nullptr);
    static void set_counts(Prepared& p, const int64_t* in_hull, const int64_t* in_both) {
        p.gin.assign(in_hull, in_hull + p.todo.size()); p.gboth.assign(in_both, in_both + p.todo.size()); p.gpu = true;
    }
    // r06: one lm_depth_query per match of p.todo, for the frame resident in `slot` (the translated frame: match coordinates as they are): the crop under
    // the template's bounding box clipped to the frame, and the window of medians that pass the depth check.  A query whose crop is empty or whose
    // window is (x1 == x0: nothing is counted, the host decides).  set_depth_counts: the counts come back in the same order.
    void depth_queries(const Prepared& p, const std::vector<TemplatePose>& templates, int slot, std::vector<lm_depth_query>& out) const;
    static void set_depth_counts(Prepared& p, const lm_depth_query* q, const uint32_t* below, const uint32_t* inside) {
        const size_t n = p.todo.size();
        p.dbelow.assign(below, below + n); p.dinside.assign(inside, inside + n); p.dsize.resize(n);
        for (size_t i = 0; i < n; ++i) p.dsize[i] = (uint32_t)(q[i].x1 - q[i].x0) * (uint32_t)(q[i].y1 - q[i].y0);
        p.depth_counts = true;
    }
    std::vector<ObjectPose> finish_group(const Prepared& p, size_t group, const std::vector<lm_match_t>& matches, const uint16_t* dense_depth,
                                         const std::vector<TemplatePose>& templates, Times* tm) const;
    // finish_group in pieces (r05): a match's two checks are pure functions of the frame and may be evaluated ahead of the sequential
    // walk, which alone decides which verdicts count (HighLevelLineMOD's streamed post-processing evaluates a group's depth checks in
    // growing waves on the pool).  colour_ok = the lookup of the GPU counts (or the host check); depth_part = the depth check (:437-457);
    // accept_range = the reference's walk over matches [from, to) of the group (v[k - from] = verdict of the k-th), true = the group has
    // its numberWantedPoses poses.  finish_group = these three, one match at a time.
    struct MatchVerdict { bool colour_ok = false, depth_done = false, depth_ok = true; int32_t tempDepth = 0; };
    bool colour_ok(const Prepared& p, uint32_t idx, const lm_match_t& m) const;
    void depth_part(const lm_match_t& m, const uint16_t* dense_depth, const std::vector<TemplatePose>& templates, MatchVerdict& v, Times* tm) const;
    // ... with the GPU's counts of match idx (p.depth_counts): the verdict "outside the window" without touching the frame when they decide it
    // (use_counts: the caller has seen the counts arrive -- HighLevelLineMOD's walks start before they do and look at a flag)
    void depth_part(const Prepared& p, uint32_t idx, const lm_match_t& m, const uint16_t* dense_depth, const std::vector<TemplatePose>& templates, MatchVerdict& v, Times* tm,
                    bool use_counts = true) const;
    bool accept_range(const Prepared& p, size_t group, const std::vector<lm_match_t>& matches, const std::vector<TemplatePose>& templates,
                      size_t from, size_t to, const MatchVerdict* v, std::vector<ObjectPose>& objPoses, Times* tm) const;
    // the two counts of colorCheck for one match, on the host (also the checker of the GPU path)
    bool color_counts(const lm_match_t& m, const std::vector<uint8_t>& color_mask, long* in_hull, long* in_both) const;
    const std::string& lastError() const { return error; }
    // The depth image handed to run() / finish_group() is the frame BEFORE a translation by (ox, oy) with zeros shifted in (the
    // reference's principal-point shift): the depth check then reads pixel (x - ox, y - oy), 0 outside -- the same values the
    // translated copy would hold, without making the copy.  Match coordinates are in the translated frame either way.
    void setDepthShift(int ox, int oy) { depth_ox = ox; depth_oy = oy; }

private:
    bool color_check(const lm_match_t& m, const std::vector<uint8_t>& color_mask) const;                    // :424-434
    std::string error;
    // decided_early (r05): the verdict "fails" came from median_mat_in_window's bounds, without the selection
    // counts (r06): {values of the crop below the window, inside it, crop size} from the GPU, or nullptr
    bool depth_check(const lm_match_t& m, const uint16_t* depth, const std::vector<TemplatePose>& t, int32_t* tempDepth, bool* decided_early = nullptr,
                     const uint32_t* counts = nullptr) const;  // :437-457
    void depth_window(const TemplatePose& tp, int* win_lo, int* win_hi) const;      // the medians that pass |depthDiff| < stepSize: [lo, hi] (empty: lo > hi)
    ObjectPose make_pose(const lm_match_t& m, const std::vector<TemplatePose>& t, int32_t tempDepth) const; // :459-515
    lm_detector* det;
    PostProcessSettings st;
    int depth_ox = 0, depth_oy = 0;
};

}  // namespace lmamd
