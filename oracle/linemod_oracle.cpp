/*
 * linemod_oracle.cpp -- CPU oracle for the LINE-MOD hot path.  TEST INFRASTRUCTURE ONLY.
 * *** PARITY UNPINNED *** -- see linemod_oracle.h for what that means and why.
 *
 * Scalar restatement of OpenCV-contrib `cv::linemod` (rgbd module, version unpinned by the
 * reference) as used behind /root/reference/src/HighLevelLinemod.cpp:26-43 (construction),
 * :93 (addTemplate) and :152 (match).  Each function names the upstream helper it restates
 *  and the reference call site that reaches it.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: the float stages must not be fused, the
 * HIP kernels use the same operation order with contraction disabled).
 */
#include "linemod_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;
// Threads the image stages may use (the match stage takes its own `threads` argument).  Upstream's
// matchClass is serial but the OpenCV imgproc primitives it calls (GaussianBlur, Sobel, pyrDown, ...)
// are internally threaded, so the all-cores CPU baseline threads these row loops as well.
int g_threads = 1;
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")
void set_err(const std::string& s) { g_err = s; }

typedef uint8_t u8;
typedef uint16_t u16;

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int reflect101(int p, int n) {  // BORDER_REFLECT_101
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

// ------------------------------------------------------------------------------------------
// A.5 tables
// ------------------------------------------------------------------------------------------
// Single-bit score of orientation `ori` against spread bit `bit`.
int bit_score(int ori, int bit, int variant) {
    int d = std::abs(ori - bit);
    if (variant == 1) d = std::min(d, 8 - d);                       // circular
    if (variant == 2 && ori >= 3) d = std::min(d, 8 - d);           // table as recalled in SURVEY.md A.5
    return std::max(0, 4 - d);
}

struct Template {
    int width = 0, height = 0, pyramid_level = 0;
    std::vector<orc_feature> features;
};
typedef std::vector<Template> TemplatePyramid;  // [level*M + modality]

struct ClassEntry {
    std::string id;
    std::vector<TemplatePyramid> pyramids;
};

struct LevelData {
    int w = 0, h = 0, T = 0;  // quantized image size at this level
    // per modality
    std::vector<std::vector<u8>> quantized, spread, lm;  // lm: [ori][T*T][W*H]
};

}  // namespace

struct orc_detector {
    orc_config cfg;
    u8 sim_lut[256];
    u8 normal_lut[8000];
    std::vector<ClassEntry> classes;
    std::vector<LevelData> levels;  // last prepared frame
    bool prepared = false;
};

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

void orc_default_config(orc_config* c, int color_only) {
    std::memset(c, 0, sizeof(*c));
    c->num_modalities = color_only ? 1 : 2;
    c->pyramid_levels = 2;
    c->T[0] = color_only ? 2 : 5;   // HighLevelLinemod.cpp:32,40
    c->T[1] = 8;
    c->weak_threshold = 10.0f;      // ColorGradient() defaults (A.2)
    c->num_features = 63;
    c->strong_threshold = 55.0f;
    c->distance_threshold = 2000;   // DepthNormal() defaults (A.3)
    c->difference_threshold = 50;
    c->depth_num_features = 63;
    c->extract_threshold = 2;
}

// SIMILARITY_LUT layout (A.5): [ori 0..7][low nibble 16 | high nibble 16]; entry = max over the
// set bits of the nibble of the single-bit score.  variant 2 (the DEFAULT of orc_create) is the table
// printed in SURVEY.md A.5, which is the one upstream's linemod.cpp ships (rows 0-2 non-circular, rows
// 3-7 circular: an upstream quirk, reproduced); 0 is the linear max(0, 4-|i-j|) table, 1 the circular one.
void orc_default_similarity_lut(uint8_t lut[256], int variant) {
    for (int ori = 0; ori < 8; ++ori)
        for (int half = 0; half < 2; ++half)
            for (int v = 0; v < 16; ++v) {
                int best = 0;
                for (int b = 0; b < 4; ++b)
                    if (v & (1 << b)) best = std::max(best, bit_score(ori, half * 4 + b, variant));
                lut[32 * ori + 16 * half + v] = (u8)best;
            }
}

// NORMAL_LUT[20][20][20] (A.4): upstream normal_lut.i is not recallable; this is OUR rule
// (documented non-OpenCV): cell centre -> azimuth of (nx,ny) -> one of 8 bins -> one-hot.
void orc_default_normal_lut(uint8_t lut[8000]) {
    const double PI = 3.14159265358979323846;
    for (int v3 = 0; v3 < 20; ++v3)
        for (int v2 = 0; v2 < 20; ++v2)
            for (int v1 = 0; v1 < 20; ++v1) {
                double nx = (v1 + 0.5 - 10.0) / 10.0, ny = (v2 + 0.5 - 10.0) / 10.0;
                double a = std::atan2(ny, nx);
                if (a < 0) a += 2 * PI;
                int bin = (int)std::floor(a / (PI / 4.0));
                if (bin > 7) bin = 7;
                lut[v3 * 400 + v2 * 20 + v1] = (u8)(1u << bin);
            }
}

// ------------------------------------------------------------------------------------------
// a3: GaussianBlur(src, 7x7, sigma 0, BORDER_REPLICATE) on CV_8UC3 (A.2 step 1).
// sigma=0,k=7 selects the fixed kernel [1/32 7/64 7/32 9/32 7/32 7/64 1/32] = {8,28,56,72,56,28,8}/256;
// the 8-bit path is 8.8 fixed point per axis, rounded half-up once at the end.
// ------------------------------------------------------------------------------------------
void orc_gaussian7_u8c3(const uint8_t* src, int w, int h, uint8_t* dst) {
    static const int K[7] = {8, 28, 56, 72, 56, 28, 8};
    std::vector<u16> tmp((size_t)w * h * 3);
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                int s = 0;
                for (int i = 0; i < 7; ++i) s += K[i] * src[((size_t)y * w + clampi(x + i - 3, 0, w - 1)) * 3 + c];
                tmp[((size_t)y * w + x) * 3 + c] = (u16)s;  // <= 255*256
            }
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                uint32_t s = 0;
                for (int j = 0; j < 7; ++j) s += (uint32_t)K[j] * tmp[((size_t)clampi(y + j - 3, 0, h - 1) * w + x) * 3 + c];
                dst[((size_t)y * w + x) * 3 + c] = (u8)((s + 32768u) >> 16);
            }
}

// a3: Sobel(smoothed, CV_16S, ksize 3, scale 1, BORDER_REPLICATE), dx and dy (A.2 step 2).
void orc_sobel3_s16c3(const uint8_t* s, int w, int h, int16_t* dx, int16_t* dy) {
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y) {
        int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1);
        for (int x = 0; x < w; ++x) {
            int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
            for (int c = 0; c < 3; ++c) {
#define P(yy, xx) ((int)s[((size_t)(yy) * w + (xx)) * 3 + c])
                int gx = (P(ym, xp) + 2 * P(y, xp) + P(yp, xp)) - (P(ym, xm) + 2 * P(y, xm) + P(yp, xm));
                int gy = (P(yp, xm) + 2 * P(yp, x) + P(yp, xp)) - (P(ym, xm) + 2 * P(ym, x) + P(ym, xp));
#undef P
                dx[((size_t)y * w + x) * 3 + c] = (int16_t)gx;
                dy[((size_t)y * w + x) * 3 + c] = (int16_t)gy;
            }
        }
    }
}

}  // extern "C"

namespace {

// cv::fastAtan2 polynomial, degrees (A.2 step 4).  Two forms of the same polynomial exist upstream
// [UPSTREAM-RECALLED, core/src/mathfuncs_core.simd.hpp]: the scalar atan_f32 (plain multiplies and adds) and the
// vector v_atan_f32 that cv::phase runs on whole rows, whose three Horner steps are v_fma -- a true fused
// multiply-add in the AVX2-dispatched build (every x86 wheel), an unfused a*b+c in the SSE2 baseline.
// g_atan_variant selects: 0 = unfused (default), 1 = fused.  The two agree on the LABEL for every gradient a 3x3
// Sobel of 8-bit data can produce (tests/test_orientation_rule.py sweeps all 2041^2; profiles/r04_atan_fma_sweep.log).
int g_atan_variant = 0;

inline float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180.0 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180.0 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180.0 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180.0 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (g_atan_variant & 1) {
        // v_atan_f32::compute: c = min / (max + eps); a = fma(fma(fma(cc, p7, p5), cc, p3), cc, p1) * c; select
        float mn = ax < ay ? ax : ay, mx = ax < ay ? ay : ax;
        c = mn / (mx + eps);
        c2 = c * c;
        a = std::fmaf(std::fmaf(std::fmaf(c2, p7, p5), c2, p3), c2, p1) * c;
        if (!(ax >= ay)) a = 90.f - a;
    } else if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// saturate_cast<uchar>(cvRound(v)) with round-half-even (A.2 step 5).
inline u8 sat_u8_rint(float v) {
    long r = lrintf(v);  // default FE_TONEAREST = half to even
    return (u8)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

extern "C" {

// a3: ColorGradient::process -> quantizedOrientations + hysteresisGradient (A.2 steps 1-6).
void orc_color_quantize(const uint8_t* bgr, int w, int h, float weak_threshold, uint8_t* quantized,
                        float* magnitude) {
    size_t n = (size_t)w * h;
    std::vector<u8> smoothed(n * 3);
    std::vector<int16_t> dx(n * 3), dy(n * 3);
    orc_gaussian7_u8c3(bgr, w, h, smoothed.data());
    orc_sobel3_s16c3(smoothed.data(), w, h, dx.data(), dy.data());

    std::vector<float> mag(n);
    std::vector<u8> q(n);
    const float scale = (float)(16.0 / 360.0);
    ORC_PAR_FOR
    for (size_t i = 0; i < n; ++i) {
        // step 3: channel with the largest dx^2+dy^2; ties B, then G, then R (>= cascade)
        int m0 = dx[3 * i] * dx[3 * i] + dy[3 * i] * dy[3 * i];
        int m1 = dx[3 * i + 1] * dx[3 * i + 1] + dy[3 * i + 1] * dy[3 * i + 1];
        int m2 = dx[3 * i + 2] * dx[3 * i + 2] + dy[3 * i + 2] * dy[3 * i + 2];
        int ch, m;
        if (m0 >= m1 && m0 >= m2) { ch = 0; m = m0; }
        else if (m1 >= m0 && m1 >= m2) { ch = 1; m = m1; }
        else { ch = 2; m = m2; }
        float fx = (float)dx[3 * i + ch], fy = (float)dy[3 * i + ch];
        mag[i] = (float)m;
        float ang = fast_atan2_deg(fy, fx);                // step 4: phase(dx, dy, degrees)
        q[i] = sat_u8_rint(ang * scale + 0.0f);            // step 5: convertTo(CV_8U, 16/360)
    }
    // step 5 cont.: zero first/last rows and columns, interior &= 7
    for (int x = 0; x < w; ++x) { q[x] = 0; q[(size_t)(h - 1) * w + x] = 0; }
    for (int y = 0; y < h; ++y) { q[(size_t)y * w] = 0; q[(size_t)y * w + w - 1] = 0; }
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) q[(size_t)y * w + x] &= 7;

    // step 6: 3x3 majority vote gated by magnitude
    const float thr = weak_threshold * weak_threshold;
    std::memset(quantized, 0, n);
    ORC_PAR_FOR
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            if (!(mag[(size_t)y * w + x] > thr)) continue;
            int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) hist[q[(size_t)(y + j) * w + x + i]]++;
            int max_votes = 0, index = -1;
            for (int b = 0; b < 8; ++b)
                if (max_votes < hist[b]) { index = b; max_votes = hist[b]; }
            if (max_votes >= 5) quantized[(size_t)y * w + x] = (u8)(1u << index);
        }
    if (magnitude) std::memcpy(magnitude, mag.data(), n * sizeof(float));
}

// The orientation label of orc_color_quantize for a batch of gradient vectors (steps 4-5: fastAtan2, x 16/360, rint,
// & 7), so a test can sweep every (dx, dy) a Sobel of 8-bit data can produce.
void orc_orientation_labels(const int32_t* dx, const int32_t* dy, size_t n, uint8_t* label) {
    const float scale = (float)(16.0 / 360.0);
    ORC_PAR_FOR
    for (size_t i = 0; i < n; ++i)
        label[i] = (u8)(sat_u8_rint(fast_atan2_deg((float)dy[i], (float)dx[i]) * scale + 0.0f) & 7);
}

// Which form of the fastAtan2 polynomial orc_color_quantize / orc_orientation_labels evaluate: bit 0 = the fused
// multiply-adds of upstream's vector code path (above).  Returns the previous value.
int orc_set_atan_variant(int variant) { int old = g_atan_variant; g_atan_variant = variant & 1; return old; }

// The polynomial's angle itself (degrees), for the pinning hook: which of the two forms an OpenCV build runs shows in the
// bits of cv::phase's output, not in the labels.
void orc_fast_atan2(const float* y, const float* x, size_t n, int variant, float* angle) {
    const int saved = orc_set_atan_variant(variant & 1);
    for (size_t i = 0; i < n; ++i) angle[i] = fast_atan2_deg(y[i], x[i]);
    orc_set_atan_variant(saved);
}

// The same labels with convertTo's working type widened to double (bit 1 of `variant`; bit 0 as above) -- a second
// recall decision the sweep retires: raw16[i] = the 16-bin value before `& 7`, may be NULL.
void orc_orientation_labels_variant(const int32_t* dx, const int32_t* dy, size_t n, int variant, uint8_t* label,
                                    uint8_t* raw16) {
    const int saved = orc_set_atan_variant(variant & 1);
    const float scale = (float)(16.0 / 360.0);
    ORC_PAR_FOR
    for (size_t i = 0; i < n; ++i) {
        float ang = fast_atan2_deg((float)dy[i], (float)dx[i]);
        u8 r;
        if (variant & 2) { long v = lrint((double)ang * (16.0 / 360.0)); r = (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
        else r = sat_u8_rint(ang * scale + 0.0f);
        if (raw16) raw16[i] = r;
        label[i] = (u8)(r & 7);
    }
    orc_set_atan_variant(saved);
}

// a4: cv::pyrDown on CV_8UC3: 5x5 [1 4 6 4 1]^2/256, BORDER_REFLECT_101, (sum+128)>>8 (A.2 pyrDown).
void orc_pyrdown_u8c3(const uint8_t* src, int w, int h, uint8_t* dst) {
    static const int K[5] = {1, 4, 6, 4, 1};
    int dw = w / 2, dh = h / 2;
    ORC_PAR_FOR
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x)
            for (int c = 0; c < 3; ++c) {
                int s = 0;
                for (int j = 0; j < 5; ++j) {
                    int sy = reflect101(2 * y + j - 2, h);
                    int rs = 0;
                    for (int i = 0; i < 5; ++i) rs += K[i] * src[((size_t)sy * w + reflect101(2 * x + i - 2, w)) * 3 + c];
                    s += K[j] * rs;
                }
                dst[((size_t)y * dw + x) * 3 + c] = (u8)((s + 128) >> 8);
            }
}

// a5: DepthNormal::process -> quantizedNormals (+accumBilateral, NORMAL_LUT, medianBlur 5) (A.3).
void orc_depth_quantize(const uint16_t* depth, int w, int h, int distance_threshold, int difference_threshold,
                        const uint8_t* normal_lut, uint8_t* quantized) {
    size_t n = (size_t)w * h;
    std::vector<u8> raw(n, 0);
    const int r = 5;
    ORC_PAR_FOR
    for (int y = r; y < h - r - 1; ++y)
        for (int x = r; x < w - r - 1; ++x) {
            long d = depth[(size_t)y * w + x];
            u8 out = 0;
            if (d < distance_threshold) {
                long A0 = 0, A1 = 0, A3 = 0, b0 = 0, b1 = 0;
                for (int jj = -1; jj <= 1; ++jj)
                    for (int ii = -1; ii <= 1; ++ii) {
                        if (ii == 0 && jj == 0) continue;
                        long i = ii * r, j = jj * r;
                        long delta = (long)depth[(size_t)(y + j) * w + (x + i)] - d;
                        long f = std::labs(delta) < difference_threshold ? 1 : 0;
                        long fi = f * i, fj = f * j;
                        A0 += fi * i; A1 += fi * j; A3 += fj * j;
                        b0 += fi * delta; b1 += fj * delta;
                    }
                long det = A0 * A3 - A1 * A1;
                long ddx = A3 * b0 - A1 * b1;
                long ddy = -A1 * b0 + A0 * b1;
                float nx = (float)(1150 * ddx);
                float ny = (float)(1150 * ddy);
                float nz = (float)(-det * d);
                float len = sqrtf(nx * nx + ny * ny + nz * nz);
                if (len > 0) {
                    float inv = 1.0f / len;
                    nx *= inv; ny *= inv; nz *= inv;
                    int v1 = (int)(nx * 10 + 10);
                    int v2 = (int)(ny * 10 + 10);
                    int v3 = (int)(nz * 20 + 20);
                    // upstream indexes NORMAL_LUT[v3][v2][v1] unchecked; we define it as a flat
                    // index into the 8000-byte table and 0 when that index is outside it
                    // (v3 == 20 happens whenever nz == 0, e.g. depth 0).
                    int flat = v3 * 400 + v2 * 20 + v1;
                    out = (flat >= 0 && flat < 8000) ? normal_lut[flat] : 0;
                }
            }
            raw[(size_t)y * w + x] = out;
        }
    orc_median5_u8(raw.data(), w, h, quantized);
}

// medianBlur(src, dst, 5) on CV_8UC1: BORDER_REPLICATE (exported on its own for tests/test_oracle_independent.py)
void orc_median5_u8(const uint8_t* raw, int w, int h, uint8_t* quantized) {
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            u8 v[25];
            int k = 0;
            for (int j = -2; j <= 2; ++j)
                for (int i = -2; i <= 2; ++i) v[k++] = raw[(size_t)clampi(y + j, 0, h - 1) * w + clampi(x + i, 0, w - 1)];
            std::nth_element(v, v + 12, v + 25);
            quantized[(size_t)y * w + x] = v[12];
        }
}

// a6: resize(..., INTER_NEAREST) to (w/2, h/2): picks src(2y, 2x).
void orc_resize_nn_half(const uint8_t* src, int w, int h, uint8_t* dst) {
    int dw = w / 2, dh = h / 2;
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) dst[(size_t)y * dw + x] = src[(size_t)(2 * y) * w + 2 * x];
}

// a8: spread + orUnaligned8u: dst(y,x) = OR_{0<=r,c<T, in bounds} src(y+r, x+c).
void orc_spread(const uint8_t* src, int w, int h, int T, uint8_t* dst) {
    std::memset(dst, 0, (size_t)w * h);
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y)   // same ORs as upstream's T*T shifted orUnaligned8u passes, regrouped per output row
        for (int r = 0; r < T && y + r < h; ++r)
            for (int c = 0; c < T; ++c)
                for (int x = 0; x + c < w; ++x) dst[(size_t)y * w + x] |= src[(size_t)(y + r) * w + x + c];
}

// a9: computeResponseMaps: maps[ori][i] = max(LUT[32 ori + lo], LUT[32 ori + 16 + hi]).
void orc_response_maps(const uint8_t* spread, int n, const uint8_t* lut, uint8_t* maps) {
    ORC_PAR_FOR
    for (int ori = 0; ori < 8; ++ori)
        for (int i = 0; i < n; ++i) {
            u8 lo = spread[i] & 15, hi = (spread[i] & 240) >> 4;
            maps[(size_t)ori * n + i] = std::max(lut[32 * ori + lo], lut[32 * ori + 16 + hi]);
        }
}

// a10: linearize: memory (y%T)*T + x%T, element (y/T)*(w/T) + x/T.
void orc_linearize(const uint8_t* resp, int w, int h, int T, uint8_t* lin) {
    int mw = w / T, mh = h / T;
    size_t idx = 0;
    for (int r0 = 0; r0 < T; ++r0)
        for (int c0 = 0; c0 < T; ++c0)
            for (int r = r0; r < mh * T; r += T)
                for (int c = c0; c < mw * T; c += T) lin[idx++] = resp[(size_t)r * w + c];
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Detector
// ------------------------------------------------------------------------------------------
namespace {

int num_features_at(const orc_config& c, int modality, int level) {
    int nf = modality == 0 ? c.num_features : c.depth_num_features;
    for (int l = 0; l < level; ++l) nf /= 2;  // pyrDown(): num_features /= 2
    return nf;
}

// Read lm[ori] (T*T*W*H bytes) at flat index; beyond the orientation's block reads 0.
// Upstream reads a contiguous T*T x (W*H) cv::Mat; crossing into the next memory row is defined
// behaviour there and is reproduced; reading past the Mat is upstream UB, defined here as 0.
inline u8 lm_read(const std::vector<u8>& lm, int ori, size_t block, size_t idx) {
    return idx < block ? lm[(size_t)ori * block + idx] : 0;
}

// 0: every byte of the similarity sums goes through lm_read (one bounds check per byte: a scalar loop); 1 (default): the
// bounds check is hoisted out of the inner loops -- the run [base, base + n) is split once into the part inside the
// orientation's block and the zeros past it -- so the compiler vectorises the byte adds the way upstream's SSE path adds
// 16 bytes per _mm_add_epi8.  Same sums either way (tests/test_oracle.py); bench.py's cpu_baseline reports both.
int g_scan_mode = 1;
}  // namespace
extern "C" void orc_set_scan_mode(int mode) { g_scan_mode = mode ? 1 : 0; }
extern "C" int orc_get_scan_mode(void) { return g_scan_mode; }
namespace {

// accessLinearMemory: returns flat index inside the orientation's block.
inline size_t lm_index(const orc_feature& f, int T, int W, int H) {
    int grid = (f.y % T) * T + (f.x % T);
    return (size_t)grid * W * H + (size_t)(f.y / T) * W + (f.x / T);
}

// a11: similarity() -- global scan at the lowest level, one modality.
void similarity(const LevelData& L, int m, const Template& t, std::vector<u8>& dst) {
    int T = L.T, W = L.w / T, H = L.h / T;
    dst.assign((size_t)W * H, 0);
    int wf = (t.width - 1) / T + 1, hf = (t.height - 1) / T + 1;
    int span_x = W - wf, span_y = H - hf;
    int P = span_y * W + span_x + 1;
    if (P > W * H) P = W * H;  // cannot happen for width,height >= 1; guards the dst buffer
    size_t block = (size_t)T * T * W * H;
    for (const orc_feature& f : t.features) {
        if (f.x < 0 || f.x >= L.w || f.y < 0 || f.y >= L.h) continue;
        size_t base = lm_index(f, T, W, H);
        if (g_scan_mode == 0) {
            for (int j = 0; j < P; ++j) dst[j] = (u8)(dst[j] + lm_read(L.lm[m], f.label, block, base + j));
        } else {
            const int n_in = (P > 0 && base < block) ? (int)std::min<size_t>((size_t)P, block - base) : 0;   // the rest reads 0
            const u8* __restrict__ src = L.lm[m].data() + (size_t)f.label * block + base;
            u8* __restrict__ out = dst.data();
            for (int j = 0; j < n_in; ++j) out[j] = (u8)(out[j] + src[j]);
        }
    }
}

// a14: similarityLocal() -- 16x16 patch at a higher-resolution level, one modality.
void similarity_local(const LevelData& L, int m, const Template& t, int cx, int cy, u8 dst[256]) {
    int T = L.T, W = L.w / T, H = L.h / T;
    std::memset(dst, 0, 256);
    int off_x = (cx / T - 8) * T, off_y = (cy / T - 8) * T;
    size_t block = (size_t)T * T * W * H;
    for (orc_feature f : t.features) {
        f.x += off_x; f.y += off_y;
        if (f.x < 0 || f.y < 0 || f.x >= L.w || f.y >= L.h) continue;
        size_t base = lm_index(f, T, W, H);
        if (g_scan_mode == 0 || base + 15 * (size_t)W + 16 > block) {
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c)
                    dst[r * 16 + c] = (u8)(dst[r * 16 + c] + lm_read(L.lm[m], f.label, block, base + (size_t)r * W + c));
        } else {   // the whole patch lies inside the block: sixteen 16-byte row adds
            const u8* __restrict__ src = L.lm[m].data() + (size_t)f.label * block + base;
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) dst[r * 16 + c] = (u8)(dst[r * 16 + c] + src[(size_t)r * W + c]);
        }
    }
}

// Total order of SURVEY.md A.9: similarity desc, template_id asc, class asc, y asc, x asc.
inline bool match_less(const orc_match& a, const orc_match& b) {
    if (a.similarity != b.similarity) return a.similarity > b.similarity;
    if (a.template_id != b.template_id) return a.template_id < b.template_id;
    if (a.class_idx != b.class_idx) return a.class_idx < b.class_idx;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
}
// Match::operator== : x, y, similarity, class_id (NOT template_id).
inline bool match_eq(const orc_match& a, const orc_match& b) {
    return a.x == b.x && a.y == b.y && a.similarity == b.similarity && a.class_idx == b.class_idx;
}

// a11 + a12 + a13: matchClass, first half -- similarity maps of the lowest level, their sum and the threshold scan.
void scan_template(const orc_detector* d, int class_idx, int template_id, const TemplatePyramid& tp, float threshold,
                   std::vector<orc_match>& cand) {
    const orc_config& c = d->cfg;
    int M = c.num_modalities, Lc = c.pyramid_levels;
    const LevelData& low = d->levels[Lc - 1];
    int lowest_start = (int)tp.size() - M;
    int lowest_T = c.T[Lc - 1];
    int W = low.w / lowest_T, H = low.h / lowest_T;

    std::vector<std::vector<u8>> sims(M);
    int num_features = 0;
    for (int i = 0; i < M; ++i) {
        const Template& t = tp[lowest_start + i];
        num_features += (int)t.features.size();
        similarity(low, i, t, sims[i]);
    }
    // a12 addSimilarities
    std::vector<u16> total((size_t)W * H);
    for (size_t k = 0; k < total.size(); ++k) {
        unsigned s = 0;
        for (int i = 0; i < M; ++i) s += sims[i][k];
        total[k] = (u16)s;
    }
    // A.7 raw threshold, float arithmetic
    int raw_threshold = (int)(2 * num_features + (threshold / 100.f) * (2 * num_features) + 0.5f);

    for (int r = 0; r < H; ++r)
        for (int cc = 0; cc < W; ++cc) {
            int raw = total[(size_t)r * W + cc];
            if (raw > raw_threshold) {
                int offset = lowest_T / 2 + (lowest_T % 2 - 1);
                orc_match mm;
                mm.x = cc * lowest_T + offset;
                mm.y = r * lowest_T + offset;
                mm.similarity = (raw * 100.f) / (4 * num_features) + 0.5f;
                mm.template_id = template_id;
                mm.class_idx = class_idx;
                cand.push_back(mm);
            }
        }
}

// a13 + a14: matchClass for one template.
void match_template(const orc_detector* d, int class_idx, int template_id, const TemplatePyramid& tp,
                    float threshold, std::vector<orc_match>& out) {
    const orc_config& c = d->cfg;
    int M = c.num_modalities, Lc = c.pyramid_levels;
    std::vector<orc_match> cand;
    scan_template(d, class_idx, template_id, tp, threshold, cand);

    for (int l = Lc - 2; l >= 0; --l) {
        const LevelData& L = d->levels[l];
        int T = c.T[l];
        int start = l * M;
        int border = 8 * T;
        int offset = T / 2 + (T % 2 - 1);
        int max_x = L.w - tp[start].width - border;
        int max_y = L.h - tp[start].height - border;
        for (orc_match& m2 : cand) {
            int x = m2.x * 2 + 1, y = m2.y * 2 + 1;
            x = std::max(x, border); y = std::max(y, border);
            x = std::min(x, max_x);  y = std::min(y, max_y);
            int numFeatures = 0;
            unsigned tot[256];
            std::memset(tot, 0, sizeof(tot));
            for (int i = 0; i < M; ++i) {
                const Template& t = tp[start + i];
                numFeatures += (int)t.features.size();
                u8 loc[256];
                similarity_local(L, i, t, x, y, loc);
                for (int k = 0; k < 256; ++k) tot[k] += loc[k];
            }
            int best_score = 0, best_r = -1, best_c = -1;
            for (int r = 0; r < 16; ++r)
                for (int cc = 0; cc < 16; ++cc) {
                    int s = (int)tot[r * 16 + cc];
                    if (s > best_score) { best_score = s; best_r = r; best_c = cc; }
                }
            m2.x = (x / T - 8 + best_c) * T + offset;
            m2.y = (y / T - 8 + best_r) * T + offset;
            m2.similarity = (best_score * 100.f) / (4 * numFeatures);
        }
        cand.erase(std::remove_if(cand.begin(), cand.end(),
                                  [threshold](const orc_match& m) { return m.similarity < threshold; }),
                   cand.end());
    }
    out.insert(out.end(), cand.begin(), cand.end());
}

int find_class(const orc_detector* d, const char* id) {
    for (size_t i = 0; i < d->classes.size(); ++i)
        if (d->classes[i].id == id) return (int)i;
    return -1;
}

bool check_dims(const orc_detector* d, int w, int h) {
    const orc_config& c = d->cfg;
    int lw = w, lh = h;
    for (int l = 0; l < c.pyramid_levels; ++l) {
        if (l > 0) { lw /= 2; lh /= 2; }
        int T = c.T[l];
        if (T <= 0 || lw % T || lh % T || ((lw * lh) % 16)) {  // CV_Assert in linearize / computeResponseMaps
            set_err("frame size violates rows%T==0, cols%T==0 or (rows*cols)%16==0 at level " + std::to_string(l));
            return false;
        }
    }
    return true;
}

}  // namespace

extern "C" {

orc_detector* orc_create(const orc_config* cfg) {
    if (!cfg || cfg->num_modalities < 1 || cfg->num_modalities > 2 || cfg->pyramid_levels < 1 ||
        cfg->pyramid_levels > ORC_MAX_LEVELS) { set_err("bad config"); return nullptr; }
    orc_detector* d = new orc_detector();
    d->cfg = *cfg;
    orc_default_similarity_lut(d->sim_lut, 2);
    orc_default_normal_lut(d->normal_lut);
    return d;
}
void orc_destroy(orc_detector* d) { delete d; }
void orc_set_similarity_lut(orc_detector* d, const uint8_t lut[256]) { std::memcpy(d->sim_lut, lut, 256); d->prepared = false; }
void orc_set_normal_lut(orc_detector* d, const uint8_t lut[8000]) { std::memcpy(d->normal_lut, lut, 8000); d->prepared = false; }
int orc_num_classes(const orc_detector* d) { return (int)d->classes.size(); }
int orc_num_templates(const orc_detector* d) {
    int n = 0;
    for (const ClassEntry& c : d->classes) n += (int)c.pyramids.size();
    return n;
}
int orc_class_num_templates(const orc_detector* d, int ci) {
    if (ci < 0 || ci >= (int)d->classes.size()) return -1;
    return (int)d->classes[ci].pyramids.size();
}

int orc_add_class(orc_detector* d, const char* class_id, int n_templates, const orc_template_desc* descs,
                  const orc_feature* features) {
    int per = d->cfg.pyramid_levels * d->cfg.num_modalities;
    int ci = find_class(d, class_id);
    if (ci < 0) { d->classes.push_back(ClassEntry{class_id, {}}); ci = (int)d->classes.size() - 1; }
    size_t fo = 0;
    for (int t = 0; t < n_templates; ++t) {
        TemplatePyramid tp(per);
        for (int k = 0; k < per; ++k) {
            const orc_template_desc& ds = descs[(size_t)t * per + k];
            if (ds.num_features < 0 || ds.num_features > 63) { set_err("template with more than 63 features"); return -1; }
            tp[k].width = ds.width; tp[k].height = ds.height; tp[k].pyramid_level = ds.pyramid_level;
            tp[k].features.assign(features + fo, features + fo + ds.num_features);
            fo += ds.num_features;
        }
        d->classes[ci].pyramids.push_back(std::move(tp));
    }
    return ci;
}

int orc_get_template(const orc_detector* d, int ci, int tid, int level, int modality, int* width, int* height,
                     orc_feature* features, int* n_features) {
    if (ci < 0 || ci >= (int)d->classes.size()) return -1;
    const ClassEntry& c = d->classes[ci];
    if (tid < 0 || tid >= (int)c.pyramids.size()) return -1;
    if (level < 0 || level >= d->cfg.pyramid_levels || modality < 0 || modality >= d->cfg.num_modalities) return -1;
    const Template& t = c.pyramids[tid][level * d->cfg.num_modalities + modality];
    if (width) *width = t.width;
    if (height) *height = t.height;
    if (n_features) *n_features = (int)t.features.size();
    if (features) std::memcpy(features, t.features.data(), t.features.size() * sizeof(orc_feature));
    return 0;
}

// Detector::match, first half (a3-a10): quantise every modality, build the pyramid of linear memories.
int orc_prepare_frame(orc_detector* d, const uint8_t* bgr, const uint16_t* depth, int w, int h) {
    const orc_config& c = d->cfg;
    int M = c.num_modalities;
    if (!bgr || (M == 2 && !depth)) { set_err("sources.size() != modalities.size()"); return -1; }
    if (!check_dims(d, w, h)) return -1;
    d->levels.assign(c.pyramid_levels, LevelData());
    std::vector<u8> cur_bgr(bgr, bgr + (size_t)w * h * 3);
    std::vector<u8> cur_norm;
    int lw = w, lh = h;
    for (int l = 0; l < c.pyramid_levels; ++l) {
        LevelData& L = d->levels[l];
        if (l > 0) {  // quantizers[i]->pyrDown()
            std::vector<u8> nb((size_t)(lw / 2) * (lh / 2) * 3);
            orc_pyrdown_u8c3(cur_bgr.data(), lw, lh, nb.data());
            cur_bgr.swap(nb);
            if (M == 2) {
                std::vector<u8> nn((size_t)(lw / 2) * (lh / 2));
                orc_resize_nn_half(cur_norm.data(), lw, lh, nn.data());
                cur_norm.swap(nn);
            }
            lw /= 2; lh /= 2;
        }
        L.w = lw; L.h = lh; L.T = c.T[l];
        L.quantized.resize(M); L.spread.resize(M); L.lm.resize(M);
        size_t n = (size_t)lw * lh;
        for (int m = 0; m < M; ++m) {
            L.quantized[m].resize(n);
            if (m == 0) {
                orc_color_quantize(cur_bgr.data(), lw, lh, c.weak_threshold, L.quantized[m].data(), nullptr);
            } else {
                if (l == 0) {
                    cur_norm.resize(n);
                    orc_depth_quantize(depth, lw, lh, c.distance_threshold, c.difference_threshold, d->normal_lut,
                                       cur_norm.data());
                }
                L.quantized[m] = cur_norm;
            }
            // a7 quantize(): copy with an empty mask (match() is called without masks, HighLevelLinemod.cpp:152)
            L.spread[m].resize(n);
            orc_spread(L.quantized[m].data(), lw, lh, L.T, L.spread[m].data());
            std::vector<u8> resp(8 * n);
            orc_response_maps(L.spread[m].data(), (int)n, d->sim_lut, resp.data());
            L.lm[m].resize(8 * n);
            for (int o = 0; o < 8; ++o) orc_linearize(resp.data() + o * n, lw, lh, L.T, L.lm[m].data() + o * n);
        }
    }
    d->prepared = true;
    return 0;
}

int64_t orc_get_stage(const orc_detector* d, int what, int level, int modality, uint8_t* out, int64_t cap) {
    if (!d->prepared || level < 0 || level >= (int)d->levels.size() || modality < 0 ||
        modality >= d->cfg.num_modalities) return -1;
    const LevelData& L = d->levels[level];
    const std::vector<u8>* v = what == 0 ? &L.quantized[modality] : what == 1 ? &L.spread[modality] : &L.lm[modality];
    int64_t n = (int64_t)v->size();
    if (out && n > 0 && cap > 0) std::memcpy(out, v->data(), (size_t)std::min(n, cap));
    return n;
}

// Detector::match, second half (a11-a15).
int orc_match_prepared(orc_detector* d, float threshold, int class_idx, int tid_lo, int tid_hi, int threads,
                       orc_match* out, int cap) {
    if (!d->prepared) { set_err("no frame prepared"); return -1; }
    if (class_idx >= (int)d->classes.size()) { set_err("class index out of range"); return -1; }
    std::vector<orc_match> matches;
    int c_lo = class_idx < 0 ? 0 : class_idx, c_hi = class_idx < 0 ? (int)d->classes.size() : class_idx + 1;
    for (int ci = c_lo; ci < c_hi; ++ci) {
        const ClassEntry& ce = d->classes[ci];
        int lo = std::max(0, tid_lo), hi = std::min((int)ce.pyramids.size(), tid_hi);
        if (threads <= 1) {
            for (int t = lo; t < hi; ++t) match_template(d, ci, t, ce.pyramids[t], threshold, matches);
        } else {
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
            {
                std::vector<orc_match> local;
#pragma omp for schedule(dynamic, 8) nowait
                for (int t = lo; t < hi; ++t) match_template(d, ci, t, ce.pyramids[t], threshold, local);
#pragma omp critical
                matches.insert(matches.end(), local.begin(), local.end());
            }
#else
            for (int t = lo; t < hi; ++t) match_template(d, ci, t, ce.pyramids[t], threshold, matches);
#endif
        }
    }
    // a15: sort + unique under the total order of A.9
    std::sort(matches.begin(), matches.end(), match_less);
    matches.erase(std::unique(matches.begin(), matches.end(), match_eq), matches.end());
    int n = (int)matches.size();
    if (out && n > 0 && cap > 0) std::memcpy(out, matches.data(), sizeof(orc_match) * (size_t)std::min(n, cap));   // (memcpy from the null data() of an empty vector is undefined even for 0 bytes)
    return n;
}

// a11-a13 only: the candidates of the global scan of the prepared frame, before any refinement, as
// (template_id, class_idx, x, y) int32 quadruples sorted by (class, template, y, x) -- what the HIP scan kernel must
// hand to the refinement stage.  Returns the count (may exceed cap_records; only cap_records are written).
int orc_scan_candidates(orc_detector* d, float threshold, int class_idx, int tid_lo, int tid_hi, int threads,
                        int32_t* out, int cap_records) {
    if (!d->prepared) { set_err("no frame prepared"); return -1; }
    if (class_idx >= (int)d->classes.size()) { set_err("class index out of range"); return -1; }
    std::vector<orc_match> all;
    int c_lo = class_idx < 0 ? 0 : class_idx, c_hi = class_idx < 0 ? (int)d->classes.size() : class_idx + 1;
    for (int ci = c_lo; ci < c_hi; ++ci) {
        const ClassEntry& ce = d->classes[ci];
        int lo = std::max(0, tid_lo), hi = std::min((int)ce.pyramids.size(), tid_hi);
        std::vector<std::vector<orc_match>> per((size_t)std::max(hi - lo, 0));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 1 ? threads : 1)
#endif
        for (int t = lo; t < hi; ++t) scan_template(d, ci, t, ce.pyramids[t], threshold, per[(size_t)(t - lo)]);
        for (const auto& v : per) all.insert(all.end(), v.begin(), v.end());   // already (class, template, y, x) ordered
    }
    int n = (int)all.size();
    if (out)
        for (int i = 0; i < std::min(n, cap_records); ++i) {
            out[4 * i] = all[i].template_id; out[4 * i + 1] = all[i].class_idx;
            out[4 * i + 2] = all[i].x; out[4 * i + 3] = all[i].y;
        }
    return n;
}

int orc_match_frame(orc_detector* d, const uint8_t* bgr, const uint16_t* depth, int w, int h, float threshold,
              int class_idx, int tid_lo, int tid_hi, int threads, orc_match* out, int cap) {
    g_threads = threads > 1 ? threads : 1;
    int rc = orc_prepare_frame(d, bgr, depth, w, h);
    g_threads = 1;
    if (rc != 0) return -1;
    return orc_match_prepared(d, threshold, class_idx, tid_lo, tid_hi, threads, out, cap);
}

int orc_merge(const orc_match* lists, const int32_t* counts, int n_lists, int stride, orc_match* out, int cap) {
    std::vector<orc_match> all;
    for (int i = 0; i < n_lists; ++i) all.insert(all.end(), lists + (size_t)i * stride, lists + (size_t)i * stride + counts[i]);
    std::sort(all.begin(), all.end(), match_less);  // inputs are sorted; a sort of the concatenation is the R-way merge
    all.erase(std::unique(all.begin(), all.end(), match_eq), all.end());
    int n = (int)all.size();
    if (out && n > 0 && cap > 0) std::memcpy(out, all.data(), sizeof(orc_match) * (size_t)std::min(n, cap));
    return n;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// A.8 template extraction (Detector::addTemplate, reached from HighLevelLinemod.cpp:93)
// ------------------------------------------------------------------------------------------
namespace {

struct Candidate { orc_feature f; float score; };
inline bool cand_less(const Candidate& a, const Candidate& b) { return a.score > b.score; }

inline int get_label(int q) {
    switch (q) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3;
                 case 16: return 4; case 32: return 5; case 64: return 6; case 128: return 7; }
    return -1;
}

// QuantizedPyramid::selectScatteredFeatures
void select_scattered(const std::vector<Candidate>& cands, std::vector<orc_feature>& feats, size_t num, float distance) {
    feats.clear();
    float dsq = distance * distance;
    int i = 0;
    while (feats.size() < num) {
        const Candidate& c = cands[i];
        bool keep = true;
        for (size_t j = 0; j < feats.size() && keep; ++j) {
            const orc_feature& f = feats[j];
            keep = (c.f.x - f.x) * (c.f.x - f.x) + (c.f.y - f.y) * (c.f.y - f.y) >= dsq;
        }
        if (keep) feats.push_back(c.f);
        if (++i == (int)cands.size()) { i = 0; distance -= 1.0f; dsq = distance * distance; }
    }
}

// erode 3x3, BORDER_REPLICATE, `iters` times
void erode3(std::vector<u8>& m, int w, int h, int iters) {
    for (int it = 0; it < iters; ++it) {
        std::vector<u8> o(m.size());
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                u8 v = 255;
                for (int j = -1; j <= 1; ++j)
                    for (int i = -1; i <= 1; ++i) v = std::min(v, m[(size_t)clampi(y + j, 0, h - 1) * w + clampi(x + i, 0, w - 1)]);
                o[(size_t)y * w + x] = v;
            }
        m.swap(o);
    }
}

// distanceTransform(src, DIST_C, 3): chessboard distance to the nearest zero pixel (two-pass chamfer
// with a=b=1 is exact for the chessboard metric); pixels outside the image count as far away.
void dist_c(const std::vector<u8>& src, int w, int h, std::vector<float>& out) {
    const int BIG = INT_MAX >> 2;
    std::vector<int> d((size_t)w * h);
    for (size_t i = 0; i < d.size(); ++i) d[i] = src[i] ? BIG : 0;
    auto at = [&](int y, int x) -> int { return (y < 0 || y >= h || x < 0 || x >= w) ? BIG : d[(size_t)y * w + x]; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = d[(size_t)y * w + x];
            if (!v) continue;
            v = std::min(v, std::min(std::min(at(y - 1, x - 1), at(y - 1, x)), std::min(at(y - 1, x + 1), at(y, x - 1))) + 1);
            d[(size_t)y * w + x] = v;
        }
    for (int y = h - 1; y >= 0; --y)
        for (int x = w - 1; x >= 0; --x) {
            int v = d[(size_t)y * w + x];
            if (!v) continue;
            v = std::min(v, std::min(std::min(at(y + 1, x + 1), at(y + 1, x)), std::min(at(y + 1, x - 1), at(y, x + 1))) + 1);
            d[(size_t)y * w + x] = v;
        }
    out.resize(d.size());
    for (size_t i = 0; i < d.size(); ++i) out[i] = (float)d[i];
}

// ColorGradientPyramid::extractTemplate
bool extract_color(const std::vector<u8>& quant, const std::vector<float>& mag, const std::vector<u8>& mask, int w, int h,
                   float strong_threshold, size_t num_features, int level, Template& t) {
    std::vector<u8> local;
    bool no_mask = mask.empty();
    if (!no_mask) {
        local = mask;
        erode3(local, w, h, 1);
        for (size_t i = 0; i < local.size(); ++i) {  // subtract(mask, eroded), saturating
            int v = (int)mask[i] - (int)local[i];
            local[i] = (u8)(v < 0 ? 0 : v);
        }
    }
    std::vector<Candidate> cands;
    float thr = strong_threshold * strong_threshold;
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            size_t i = (size_t)r * w + c;
            if (no_mask || local[i]) {
                u8 q = quant[i];
                if (q > 0 && mag[i] > thr) cands.push_back(Candidate{{c, r, get_label(q)}, mag[i]});
            }
        }
    if (cands.size() < num_features) return false;
    std::stable_sort(cands.begin(), cands.end(), cand_less);
    float distance = (float)(cands.size() / num_features + 1);
    select_scattered(cands, t.features, num_features, distance);
    t.width = -1; t.height = -1; t.pyramid_level = level;
    return true;
}

// DepthNormalPyramid::extractTemplate
bool extract_depth(const std::vector<u8>& normal, const std::vector<u8>& mask, int w, int h, int extract_threshold,
                   size_t num_features, int level, Template& t) {
    std::vector<u8> local;
    bool no_mask = mask.empty();
    if (!no_mask) { local = mask; erode3(local, w, h, 2); }
    std::vector<float> dist[8];
    for (int i = 0; i < 8; ++i) {
        std::vector<u8> tmp((size_t)w * h, 0);
        for (size_t k = 0; k < tmp.size(); ++k) {
            u8 v = (no_mask || local[k]) ? (u8)(1u << i) : 0;  // temp.setTo(1<<i, local_mask)
            tmp[k] = v & normal[k];
        }
        dist_c(tmp, w, h, dist[i]);
    }
    int label_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<Candidate> cands;
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            size_t i = (size_t)r * w + c;
            if (no_mask || local[i]) {
                u8 q = normal[i];
                if (q != 0 && q != 255) {
                    int label = get_label(q);
                    if (label < 0) continue;  // not one-hot: cannot happen with a one-hot LUT
                    float dd = dist[label][i];
                    if (dd >= (float)extract_threshold) { cands.push_back(Candidate{{c, r, label}, dd}); ++label_counts[label]; }
                }
            }
        }
    if (cands.size() < num_features) return false;
    for (Candidate& c : cands) c.score /= (float)label_counts[c.f.label];
    std::stable_sort(cands.begin(), cands.end(), cand_less);
    float area = 0;
    if (no_mask) area = (float)((size_t)w * h);
    else for (u8 v : local) area += v ? 1.f : 0.f;
    float distance = sqrtf(area) / sqrtf((float)num_features) + 1.5f;
    select_scattered(cands, t.features, num_features, distance);
    t.width = -1; t.height = -1; t.pyramid_level = level;
    return true;
}

// cropTemplates
orc_rect crop_templates(TemplatePyramid& tp) {
    int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
    for (const Template& t : tp)
        for (const orc_feature& f : t.features) {
            int x = f.x << t.pyramid_level, y = f.y << t.pyramid_level;
            min_x = std::min(min_x, x); min_y = std::min(min_y, y);
            max_x = std::max(max_x, x); max_y = std::max(max_y, y);
        }
    if (min_x % 2 == 1) --min_x;
    if (min_y % 2 == 1) --min_y;
    for (Template& t : tp) {
        t.width = (max_x - min_x) >> t.pyramid_level;
        t.height = (max_y - min_y) >> t.pyramid_level;
        int ox = min_x >> t.pyramid_level, oy = min_y >> t.pyramid_level;
        for (orc_feature& f : t.features) { f.x -= ox; f.y -= oy; }
    }
    return orc_rect{min_x, min_y, max_x - min_x, max_y - min_y};
}

}  // namespace

extern "C" int orc_add_template(orc_detector* d, const char* class_id, const uint8_t* bgr, const uint16_t* depth,
                                const uint8_t* mask, int w, int h, orc_rect* bbox) {
    const orc_config& c = d->cfg;
    int M = c.num_modalities, Lc = c.pyramid_levels;
    if (!bgr || (M == 2 && !depth)) { set_err("sources.size() != modalities.size()"); return -1; }
    TemplatePyramid tp((size_t)M * Lc);
    for (int m = 0; m < M; ++m) {
        std::vector<u8> cur_mask;
        if (mask) cur_mask.assign(mask, mask + (size_t)w * h);
        std::vector<u8> cur_bgr, cur_norm;
        if (m == 0) cur_bgr.assign(bgr, bgr + (size_t)w * h * 3);
        int lw = w, lh = h;
        for (int l = 0; l < Lc; ++l) {
            if (l > 0) {
                if (m == 0) {
                    std::vector<u8> nb((size_t)(lw / 2) * (lh / 2) * 3);
                    orc_pyrdown_u8c3(cur_bgr.data(), lw, lh, nb.data());
                    cur_bgr.swap(nb);
                } else {
                    std::vector<u8> nn((size_t)(lw / 2) * (lh / 2));
                    orc_resize_nn_half(cur_norm.data(), lw, lh, nn.data());
                    cur_norm.swap(nn);
                }
                if (!cur_mask.empty()) {
                    std::vector<u8> nm((size_t)(lw / 2) * (lh / 2));
                    orc_resize_nn_half(cur_mask.data(), lw, lh, nm.data());
                    cur_mask.swap(nm);
                }
                lw /= 2; lh /= 2;
            }
            size_t n = (size_t)lw * lh;
            bool ok;
            if (m == 0) {
                std::vector<u8> q(n);
                std::vector<float> mag(n);
                orc_color_quantize(cur_bgr.data(), lw, lh, c.weak_threshold, q.data(), mag.data());
                ok = extract_color(q, mag, cur_mask, lw, lh, c.strong_threshold, (size_t)num_features_at(c, 0, l), l,
                                   tp[(size_t)l * M + m]);
            } else {
                if (l == 0) {
                    cur_norm.resize(n);
                    orc_depth_quantize(depth, lw, lh, c.distance_threshold, c.difference_threshold, d->normal_lut,
                                       cur_norm.data());
                }
                int et = c.extract_threshold;
                for (int k = 0; k < l; ++k) et /= 2;
                ok = extract_depth(cur_norm, cur_mask, lw, lh, et, (size_t)num_features_at(c, 1, l), l, tp[(size_t)l * M + m]);
            }
            if (!ok) return -1;
        }
    }
    orc_rect bb = crop_templates(tp);
    if (bbox) *bbox = bb;
    int ci = find_class(d, class_id);
    if (ci < 0) { d->classes.push_back(ClassEntry{class_id, {}}); ci = (int)d->classes.size() - 1; }
    d->classes[ci].pyramids.push_back(std::move(tp));
    return (int)d->classes[ci].pyramids.size() - 1;
}

// ---- the two image primitives of extractTemplate on their own (tests/test_oracle_independent.py checks them, like the ones above,
// against scipy.ndimage -- an implementation this file's author did not write)
extern "C" {
void orc_erode3_u8(const uint8_t* src, int w, int h, int iters, uint8_t* dst) {
    std::vector<u8> m(src, src + (size_t)w * h);
    erode3(m, w, h, iters);
    std::memcpy(dst, m.data(), m.size());
}
void orc_dist_c(const uint8_t* src, int w, int h, float* dst) {
    std::vector<u8> m(src, src + (size_t)w * h);
    std::vector<float> o;
    dist_c(m, w, h, o);
    std::memcpy(dst, o.data(), o.size() * sizeof(float));
}
}
