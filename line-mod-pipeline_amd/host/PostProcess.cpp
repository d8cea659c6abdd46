// PostProcess.cpp -- see PostProcess.h.  Restates the reference's match post-processing
// (/root/reference/src/HighLevelLinemod.cpp) and the pieces of OpenCV / GLM it leans on.
#include "PostProcess.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lmamd {

// ==================================================================================================
// mini GLM (column-major m[col][row], right-handed)
// ==================================================================================================
Vec3 cross(const Vec3& a, const Vec3& b) { return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
float length(const Vec3& v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
static float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec3 normalize(const Vec3& v) {
    float inv = 1.0f / std::sqrt(dot(v, v));   // glm: v * inversesqrt(dot(v, v))
    return Vec3{v.x * inv, v.y * inv, v.z * inv};
}

Mat4 lookAt(const Vec3& eye, const Vec3& center, const Vec3& up) {   // glm::lookAtRH
    Vec3 f = normalize(Vec3{center.x - eye.x, center.y - eye.y, center.z - eye.z});
    Vec3 s = normalize(cross(f, up));
    Vec3 u = cross(s, f);
    Mat4 r;
    std::memset(&r, 0, sizeof(r));
    r.m[0][0] = s.x; r.m[1][0] = s.y; r.m[2][0] = s.z;
    r.m[0][1] = u.x; r.m[1][1] = u.y; r.m[2][1] = u.z;
    r.m[0][2] = -f.x; r.m[1][2] = -f.y; r.m[2][2] = -f.z;
    r.m[3][0] = -dot(s, eye); r.m[3][1] = -dot(u, eye); r.m[3][2] = dot(f, eye);
    r.m[3][3] = 1.0f;
    return r;
}

Mat4 mul(const Mat4& a, const Mat4& b) {   // (a * b)[c][r] = sum_k a[k][r] * b[c][k]
    Mat4 o;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += a.m[k][r] * b.m[c][k];
            o.m[c][r] = s;
        }
    return o;
}

Mat4 transpose(const Mat4& a) {
    Mat4 o;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) o.m[c][r] = a.m[r][c];
    return o;
}

Mat4 toMat4(const Quat& q) {   // glm::mat4_cast
    Mat4 r;
    std::memset(&r, 0, sizeof(r));
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r.m[0][0] = 1 - 2 * (qyy + qzz); r.m[0][1] = 2 * (qxy + qwz); r.m[0][2] = 2 * (qxz - qwy);
    r.m[1][0] = 2 * (qxy - qwz); r.m[1][1] = 1 - 2 * (qxx + qzz); r.m[1][2] = 2 * (qyz + qwx);
    r.m[2][0] = 2 * (qxz + qwy); r.m[2][1] = 2 * (qyz - qwx); r.m[2][2] = 1 - 2 * (qxx + qyy);
    r.m[3][3] = 1.0f;
    return r;
}

Quat toQuat(const Mat4& m4) {   // glm::quat_cast(mat3(m))
    const float (*m)[4] = m4.m;
    float fourX = m[0][0] - m[1][1] - m[2][2], fourY = m[1][1] - m[0][0] - m[2][2];
    float fourZ = m[2][2] - m[0][0] - m[1][1], fourW = m[0][0] + m[1][1] + m[2][2];
    int biggest = 0;
    float big = fourW;
    if (fourX > big) { big = fourX; biggest = 1; }
    if (fourY > big) { big = fourY; biggest = 2; }
    if (fourZ > big) { big = fourZ; biggest = 3; }
    float bv = std::sqrt(big + 1.0f) * 0.5f, mult = 0.25f / bv;
    Quat q;
    switch (biggest) {
        case 0: q.w = bv; q.x = (m[1][2] - m[2][1]) * mult; q.y = (m[2][0] - m[0][2]) * mult; q.z = (m[0][1] - m[1][0]) * mult; break;
        case 1: q.x = bv; q.w = (m[1][2] - m[2][1]) * mult; q.y = (m[0][1] + m[1][0]) * mult; q.z = (m[2][0] + m[0][2]) * mult; break;
        case 2: q.y = bv; q.w = (m[2][0] - m[0][2]) * mult; q.x = (m[0][1] + m[1][0]) * mult; q.z = (m[1][2] + m[2][1]) * mult; break;
        default: q.z = bv; q.w = (m[0][1] - m[1][0]) * mult; q.x = (m[2][0] + m[0][2]) * mult; q.y = (m[1][2] + m[2][1]) * mult; break;
    }
    return q;
}

Vec3 rotate(const Vec3& v, float angle, const Vec3& normal) {   // glm::rotate(v, angle, normal): mat3(rotate(angle, normal)) * v
    float c = std::cos(angle), s = std::sin(angle);
    Vec3 a = normalize(normal);
    Vec3 t{(1 - c) * a.x, (1 - c) * a.y, (1 - c) * a.z};
    float R[3][3];   // [col][row]
    R[0][0] = c + t.x * a.x; R[0][1] = t.x * a.y + s * a.z; R[0][2] = t.x * a.z - s * a.y;
    R[1][0] = t.y * a.x - s * a.z; R[1][1] = c + t.y * a.y; R[1][2] = t.y * a.z + s * a.x;
    R[2][0] = t.z * a.x + s * a.y; R[2][1] = t.z * a.y - s * a.x; R[2][2] = c + t.z * a.z;
    return Vec3{R[0][0] * v.x + R[1][0] * v.y + R[2][0] * v.z, R[0][1] * v.x + R[1][1] * v.y + R[2][1] * v.z,
                R[0][2] * v.x + R[1][2] * v.y + R[2][2] * v.z};
}

// ==================================================================================================
// image helpers
// ==================================================================================================
// cv::cvtColor(COLOR_BGR2HSV) for CV_8U (RGB2HSV_b: 12-bit fixed-point division tables, H in [0,180))
// followed by cv::inRange with scalar bounds (HighLevelLinemod.cpp:159-161).
void bgr2hsv_inrange(const uint8_t* bgr, int w, int h, size_t stride, const double lower[3], const double upper[3],
                     std::vector<uint8_t>& mask) {
    const int shift = 12;
    // (a function-local static with an initialiser: built once, thread-safely -- r05 calls this from the pool's threads)
    struct DivTables { int sdiv[256], hdiv[256]; DivTables() { sdiv[0] = hdiv[0] = 0; for (int i = 1; i < 256; ++i) { sdiv[i] = (int)std::lrint((255 << 12) / (1.0 * i)); hdiv[i] = (int)std::lrint((180 << 12) / (6.0 * i)); } } };
    static const DivTables tables;
    const int* sdiv = tables.sdiv;
    const int* hdiv = tables.hdiv;
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = (int)std::lrint(lower[k]); hi[k] = (int)std::lrint(upper[k]); }
    if (stride == 0) stride = (size_t)w * 3;
    mask.assign((size_t)w * h, 0);
    for (int y = 0; y < h; ++y) {
        const uint8_t* p = bgr + y * stride;
        for (int x = 0; x < w; ++x, p += 3) {
            int b = p[0], g = p[1], r = p[2];
            int v = std::max(b, std::max(g, r)), vmin = std::min(b, std::min(g, r));
            int diff = v - vmin;
            int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            int s = (diff * sdiv[v] + (1 << (shift - 1))) >> shift;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (hh * hdiv[diff] + (1 << (shift - 1))) >> shift;
            hh += hh < 0 ? 180 : 0;
            int H = hh < 0 ? 0 : (hh > 255 ? 255 : hh);
            bool in = H >= lo[0] && H <= hi[0] && s >= lo[1] && s <= hi[1] && v >= lo[2] && v <= hi[2];
            mask[(size_t)y * w + x] = in ? 255 : 0;
        }
    }
}

// (row-wise: the shift is an integer translation with zeros shifted in, so every destination row is one memcpy of the overlapping
// span; the per-pixel form of r03 took 1.1 ms per 1280 x 960 RGB-D frame, bench.py's pose_e2e leg)
template <typename T, int C>
static void translate_rows(const T* src, int w, int h, int ox, int oy, std::vector<T>& dst) {
    dst.resize((size_t)w * h * C);                                // (a reused buffer keeps its pages; every element is written below)
    ox = std::max(-w, std::min(w, ox)); oy = std::max(-h, std::min(h, oy));       // (beyond the frame: all zeros either way; no overflow in w + ox)
    const int x0 = std::max(ox, 0), x1 = std::min(w + ox, w);     // destination columns [x0, x1) have a source pixel
    const int y0 = std::max(oy, 0), y1 = std::min(h + oy, h);
    for (int y = 0; y < h; ++y) {
        T* row = &dst[(size_t)y * w * C];
        if (y < y0 || y >= y1 || x1 <= x0) { std::memset(row, 0, (size_t)w * C * sizeof(T)); continue; }
        if (x0 > 0) std::memset(row, 0, (size_t)x0 * C * sizeof(T));
        std::memcpy(row + (size_t)x0 * C, &src[((size_t)(y - oy) * w + (x0 - ox)) * C], (size_t)(x1 - x0) * C * sizeof(T));
        if (x1 < w) std::memset(row + (size_t)x1 * C, 0, (size_t)(w - x1) * C * sizeof(T));
    }
}
void translate_u8c3(const uint8_t* src, int w, int h, int ox, int oy, std::vector<uint8_t>& dst) { translate_rows<uint8_t, 3>(src, w, h, ox, oy, dst); }
void translate_u16(const uint16_t* src, int w, int h, int ox, int oy, std::vector<uint16_t>& dst) { translate_rows<uint16_t, 1>(src, w, h, ox, oy, dst); }

// ---- medianMat (:336-349) ------------------------------------------------------------------------------------------------------------
// The crop of the (translated) depth image with threshold(.., 1, 65535) inverted and added -- depths <= 1 and the zeros shifted in become
// 65535 -- as one dense vector, plus two by-products of the same pass (r05): the smallest value and how many values lie below `below`.
#if defined(__x86_64__)
// (a 16-bit lane counts one element in sixteen: no overflow for rows below 2^20 pixels)
__attribute__((target("avx2"))) static void crop_row_avx2(const uint16_t* r, int n, uint16_t* o, uint16_t below, uint16_t top, uint16_t* mn, size_t* cnt, size_t* cnt_in) {
    const __m256i lim = _mm256_set1_epi16((short)below), fe = _mm256_set1_epi16((short)0xFFFE), zero = _mm256_setzero_si256();
    const __m256i span = _mm256_set1_epi16((short)(uint16_t)(top >= below ? top - below : 0));
    __m256i vmin = _mm256_set1_epi16((short)0xFFFF), acc = zero, acc_in = zero;
    int x = 0;
    for (; x + 16 <= n; x += 16) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(r + x));
        const __m256i t = _mm256_or_si256(v, _mm256_cmpeq_epi16(_mm256_and_si256(v, fe), zero));       // v <= 1 -> 65535
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(o + x), t);
        vmin = _mm256_min_epu16(vmin, t);
        const __m256i ge = _mm256_cmpeq_epi16(_mm256_max_epu16(t, lim), t);                              // t >= below
        acc = _mm256_sub_epi16(acc, _mm256_andnot_si256(ge, _mm256_set1_epi16(-1)));                      // t < below: max(t, below) != t
        const __m256i off = _mm256_sub_epi16(t, lim);                                                     // below <= t <= top: t >= below and t - below <= top - below
        acc_in = _mm256_sub_epi16(acc_in, _mm256_and_si256(ge, _mm256_cmpeq_epi16(_mm256_min_epu16(off, span), off)));
    }
    alignas(32) uint16_t a[16];
    _mm256_store_si256(reinterpret_cast<__m256i*>(a), acc); for (uint16_t q : a) *cnt += q;
    _mm256_store_si256(reinterpret_cast<__m256i*>(a), acc_in); for (uint16_t q : a) *cnt_in += q;
    _mm256_store_si256(reinterpret_cast<__m256i*>(a), vmin); for (uint16_t q : a) *mn = std::min(*mn, q);
    for (; x < n; ++x) { const uint16_t t = r[x] > 1 ? r[x] : (uint16_t)65535; o[x] = t; *mn = std::min(*mn, t); *cnt += t < below; *cnt_in += (t >= below && t <= top); }
}
#endif
static void crop_row(const uint16_t* r, int n, uint16_t* o, uint16_t below, uint16_t top, uint16_t* mn, size_t* cnt, size_t* cnt_in) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && top >= below) { crop_row_avx2(r, n, o, below, top, mn, cnt, cnt_in); return; }
#endif
    for (int x = 0; x < n; ++x) { const uint16_t t = r[x] > 1 ? r[x] : (uint16_t)65535; o[x] = t; *mn = std::min(*mn, t); *cnt += t < below; *cnt_in += (t >= below && t <= top); }
}

// (shift_x, shift_y): `depth` is the frame before a translation by that many pixels with zeros shifted in; bb is in the translated frame.
// Returns the element count (0: empty crop).
// by-products: *mn = the smallest value, *cnt_below = values below `below`, *cnt_in = values in [below, top] (the shifted-in / invalid 65535s count when top is 65535)
static size_t crop_depth(const uint16_t* depth, int w, int h, Rect bb, int shift_x, int shift_y, std::vector<uint16_t>& v, uint16_t below, uint16_t top, uint16_t* mn, size_t* cnt_below, size_t* cnt_in) {
    shift_x = std::max(-w, std::min(w, shift_x)); shift_y = std::max(-h, std::min(h, shift_y));   // (beyond the frame: all zeros either way; no overflow in w + shift)
    int x0 = std::max(bb.x, 0), y0 = std::max(bb.y, 0);
    int x1 = (int)std::min<long long>((long long)bb.x + bb.width, w), y1 = (int)std::min<long long>((long long)bb.y + bb.height, h);   // cv::Mat ROI would assert; we clip
    *mn = 65535; *cnt_below = 0; *cnt_in = 0;
    if (x1 <= x0 || y1 <= y0) return 0;
    size_t filled = 0;                                                          // 65535s written without a source pixel
    const size_t rw = (size_t)(x1 - x0), n = rw * (size_t)(y1 - y0);
    v.resize(n);
    uint16_t* o = v.data();
    // columns of the translated frame that have a source pixel: [sx0, sx1); the others (and whole rows without a source row) are the
    // zeros shifted in, which threshold(.., 1, 65535) inverted turns into 65535 like every depth <= 1
    const int sx0 = std::min(std::max(std::max(shift_x, 0), x0), x1), sx1 = std::max(std::min(std::min(w + shift_x, w), x1), sx0);
    for (int y = y0; y < y1; ++y) {
        const int sy = y - shift_y;
        if (sy < 0 || sy >= h) { for (size_t x = 0; x < rw; ++x) o[x] = 65535; o += rw; filled += rw; continue; }
        const uint16_t* r = depth + (size_t)sy * w;                            // source row; translated column x reads r[x - shift_x] (indexed from the row base: ADVICE r4)
        for (int x = x0; x < sx0; ++x) o[x - x0] = 65535;
        if (sx1 > sx0) crop_row(r + (sx0 - shift_x), sx1 - sx0, o + (sx0 - x0), below, top, mn, cnt_below, cnt_in);
        for (int x = sx1; x < x1; ++x) o[x - x0] = 65535;
        filled += (size_t)(sx0 - x0) + (size_t)(x1 - sx1);
        o += rw;
    }
    if (top == 65535) *cnt_in += filled;
    return n;
}

uint16_t median_mat(const uint16_t* depth, int w, int h, Rect bb, uint8_t position, int shift_x, int shift_y) {
    static thread_local std::vector<uint16_t> v;      // (one buffer per thread: a call per match, thousands per frame batch)
    if (position == 0) return 65535;
    uint16_t mn; size_t cb, ci;
    const size_t n = crop_depth(depth, w, h, bb, shift_x, shift_y, v, 0, 65535, &mn, &cb, &ci);
    if (n == 0) return 65535;
    std::nth_element(v.begin(), v.begin() + (ptrdiff_t)(n / 4), v.end());
    return v[n / position];
}

// medianMat when all the caller wants to know is whether the result lies in [win_lo, win_hi], and the result itself if it does (the
// depth check, :437-457).  The reference's value is v[n / position] AFTER nth_element(begin, begin + n / 4, end): an element at or
// before the nth position, so (for position >= 4) it is at most the (n / 4)-th order statistic and at least the smallest element --
// WHICH of those elements it is depends on the library's partition order, which is why the value itself must come from the very same
// std::nth_element.  But when more than n / 4 elements lie below win_lo the (n / 4)-th order statistic does, and when NO element lies inside
// the window none of the n / 4 + 1 smallest can (those that are not below win_lo are then above win_hi -- the smallest element above win_hi
// is the special case): the verdict "outside" is then certain without the selection (r05: one vectorised pass over the crop instead of
// the partition passes; exact, never a different verdict or value).  Returns true and *median when inside.
bool median_mat_in_window(const uint16_t* depth, int w, int h, Rect bb, uint8_t position, int shift_x, int shift_y, int win_lo, int win_hi,
                          uint16_t* median, bool* decided_early) {
    static thread_local std::vector<uint16_t> v;
    if (decided_early) *decided_early = false;
    uint16_t m = 65535;
    if (position != 0) {
        uint16_t mn; size_t cb, ci;
        const uint16_t below = (uint16_t)std::max(0, std::min(win_lo, 65535));
        const uint16_t top = (uint16_t)std::max(0, std::min(win_hi, 65535));
        const size_t n = crop_depth(depth, w, h, bb, shift_x, shift_y, v, below, top, &mn, &cb, &ci);
        if (n != 0) {
            if (position >= 4 && win_lo <= 65535 && win_hi >= win_lo && win_hi >= 0 && (cb >= n / 4 + 1 || ci == 0)) { if (decided_early) *decided_early = true; return false; }
            std::nth_element(v.begin(), v.begin() + (ptrdiff_t)(n / 4), v.end());
            m = v[n / position];
        }
    }
    *median = m;
    return (int)m >= win_lo && (int)m <= win_hi;
}

std::vector<Pt> convex_hull(std::vector<Pt> p) {   // Andrew's monotone chain, counter-clockwise, collinear points dropped
    std::sort(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }), p.end());
    if (p.size() < 3) return p;
    auto crs = [](const Pt& o, const Pt& a, const Pt& b) { return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x); };
    std::vector<Pt> hull(2 * p.size());
    size_t k = 0;
    for (size_t i = 0; i < p.size(); ++i) {
        while (k >= 2 && crs(hull[k - 2], hull[k - 1], p[i]) <= 0) --k;
        hull[k++] = p[i];
    }
    for (size_t i = p.size() - 1, t = k + 1; i > 0; --i) {
        while (k >= t && crs(hull[k - 2], hull[k - 1], p[i - 1]) <= 0) --k;
        hull[k++] = p[i - 1];
    }
    hull.resize(k - 1);
    return hull;
}

// templateMask (:113-135) = fillPoly of the hull into a full-frame mask; colorCheck then counts the
// mask and mask & colour.  Here: the closed polygon (scanline interior + 8-connected boundary lines, as
// fillPoly draws both) rasterised into the hull's bounding box only.
void hull_counts(const std::vector<Pt>& hull, const uint8_t* color_mask, int w, int h, long* in_hull, long* in_both) {
    *in_hull = 0; *in_both = 0;
    if (hull.empty()) return;
    int x0 = INT_MAX, y0 = INT_MAX, x1 = INT_MIN, y1 = INT_MIN;
    for (const Pt& p : hull) { x0 = std::min(x0, p.x); y0 = std::min(y0, p.y); x1 = std::max(x1, p.x); y1 = std::max(y1, p.y); }
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    std::vector<uint8_t> m((size_t)bw * bh, 0);
    auto set = [&](int x, int y) { m[(size_t)(y - y0) * bw + (x - x0)] = 1; };
    const size_t n = hull.size();
    for (int y = y0; y <= y1; ++y) {   // interior: exact intersection of the row with the convex polygon
        double lo = 1e300, hi = -1e300;
        for (size_t i = 0; i < n; ++i) {
            const Pt& a = hull[i];
            const Pt& b = hull[(i + 1) % n];
            if (a.y == b.y) { if (a.y == y) { lo = std::min(lo, (double)std::min(a.x, b.x)); hi = std::max(hi, (double)std::max(a.x, b.x)); } continue; }
            if (y < std::min(a.y, b.y) || y > std::max(a.y, b.y)) continue;
            double x = a.x + (double)(y - a.y) * (b.x - a.x) / (double)(b.y - a.y);
            lo = std::min(lo, x); hi = std::max(hi, x);
        }
        if (n == 1) { lo = hi = hull[0].x; }
        if (lo > hi) continue;
        for (int x = (int)std::ceil(lo - 1e-9); x <= (int)std::floor(hi + 1e-9); ++x) set(x, y);
    }
    for (size_t i = 0; i < n && n > 1; ++i) {   // boundary: 8-connected Bresenham between consecutive vertices
        Pt a = hull[i], b = hull[(i + 1) % n];
        int dx = std::abs(b.x - a.x), dy = -std::abs(b.y - a.y), sx = a.x < b.x ? 1 : -1, sy = a.y < b.y ? 1 : -1, err = dx + dy;
        for (;;) {
            set(a.x, a.y);
            if (a.x == b.x && a.y == b.y) break;
            int e2 = 2 * err;
            if (e2 >= dy) { err += dy; a.x += sx; }
            if (e2 <= dx) { err += dx; a.y += sy; }
        }
    }
    for (int y = std::max(y0, 0); y <= std::min(y1, h - 1); ++y)
        for (int x = std::max(x0, 0); x <= std::min(x1, w - 1); ++x)
            if (m[(size_t)(y - y0) * bw + (x - x0)]) {
                ++*in_hull;
                if (color_mask[(size_t)y * w + x]) ++*in_both;
            }
}

// ==================================================================================================
// grouping (:206-253)
// ==================================================================================================
std::vector<MatchGroup> group_similar_matches(const std::vector<lm_match_t>& matches, float radius) {
    std::vector<MatchGroup> groups;
    for (size_t i = 0; i < matches.size(); ++i) {
        const uint16_t numCurrentGroups = (uint16_t)groups.size();   // uint16_t in the reference (:212)
        bool found = false;
        for (size_t q = 0; q < numCurrentGroups; ++q) {
            double dx = matches[i].x - groups[q].position.x, dy = matches[i].y - groups[q].position.y;
            if (std::sqrt(dx * dx + dy * dy) < radius) {   // cv::norm(Point) < radiusThresholdNewObject
                groups[q].matchIndices.push_back((uint32_t)i);
                found = true;
                break;
            }
        }
        if (!found) groups.push_back(MatchGroup{Pt{matches[i].x, matches[i].y}, {(uint32_t)i}});
    }
    return groups;
}

std::vector<MatchGroup> discard_small_groups(const std::vector<MatchGroup>& groups, float ratio) {
    size_t biggest = 0;
    for (const MatchGroup& g : groups) biggest = std::max(biggest, g.matchIndices.size());
    std::vector<MatchGroup> out;
    for (const MatchGroup& g : groups) {
        float r = (float)(g.matchIndices.size() * 100 / biggest);   // integer division first (:246)
        if (r > ratio) out.push_back(g);
    }
    return out;
}

void calculate_template_pose(Vec3 cam, int16_t inplaneRot, float t[3], float q[4]) {   // :351-379
    t[0] = 0.f; t[1] = 0.f; t[2] = length(cam);
    if (cam.x == 0 && cam.z == 0) cam.x = 0.00000000001f;   // looking straight up or down fails the cross product
    const Vec3 up{0.f, 1.f, 0.f};
    Vec3 camUp = normalize(cross(cam, cross(cam, up)));
    Vec3 rotatedUp = rotate(Vec3{-camUp.x, -camUp.y, -camUp.z}, (float)inplaneRot * 0.01745329251994329576923690768489f, normalize(cam));
    Mat4 view = lookAt(cam, Vec3{0, 0, 0}, rotatedUp);
    // openglCoordinatesystem2opencv (:371-379)
    Mat4 ct;
    std::memset(&ct, 0, sizeof(ct));
    ct.m[0][0] = 1.f; ct.m[1][1] = -1.f; ct.m[2][2] = -1.f; ct.m[3][3] = 1.f;
    Quat qq = toQuat(transpose(mul(transpose(view), ct)));
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
}

// ==================================================================================================
// PostProcessor
// ==================================================================================================
bool PostProcessor::color_counts(const lm_match_t& m, const std::vector<uint8_t>& color_mask, long* in_hull, long* in_both) const {
    std::vector<Pt> pts;
    const int M = lm_num_modalities(det);
    *in_hull = 0; *in_both = 0;
    for (int mod = 0; mod < M; ++mod) {   // templates[m].features for m < num_modalities = the level-0 templates (:120-126)
        int n = 0, tw = 0, th = 0;
        if (lm_get_template(det, m.class_idx, m.template_id, 0, mod, &tw, &th, nullptr, &n) != LM_OK) return false;
        std::vector<lm_feature> f((size_t)n);
        lm_get_template(det, m.class_idx, m.template_id, 0, mod, &tw, &th, f.data(), &n);
        for (const lm_feature& ft : f) pts.push_back(Pt{ft.x + m.x, ft.y + m.y});
    }
    hull_counts(convex_hull(pts), color_mask.data(), st.videoWidth, st.videoHeight, in_hull, in_both);
    return true;
}

static bool color_verdict(long in_hull, long in_both, uint16_t percentToPassCheck) {
    if (in_hull == 0) return false;                         // the reference would divide by zero
    float nonZer = (float)(in_both * 100 / in_hull);         // integer division first (:432)
    return nonZer > (float)percentToPassCheck;
}

bool PostProcessor::color_check(const lm_match_t& m, const std::vector<uint8_t>& color_mask) const {
    long in_hull = 0, in_both = 0;
    if (!color_counts(m, color_mask, &in_hull, &in_both)) return false;
    return color_verdict(in_hull, in_both, st.percentToPassCheck);
}

// depthDiff of a median (:441-447): monotone non-decreasing in the median (exact int -> float, a float subtraction, truncation)
static inline int32_t depth_diff(int med, const TemplatePose& tp, float depthOffset) {
    return (int32_t)((float)((int32_t)med - (int32_t)tp.medianDepth) - depthOffset);
}

void PostProcessor::depth_window(const TemplatePose& tp, int* win_lo, int* win_hi) const {
    // the medians that pass |depthDiff| < stepSize are an interval [lo, hi] of 0 .. 65535 (depth_diff is monotone): two bisections
    const int32_t step = (int32_t)st.stepSize;
    int lo = 0, hi = 65536;                              // lo: first median with depthDiff > -step
    while (lo < hi) { const int mid = (lo + hi) / 2; if (depth_diff(mid, tp, st.depthOffset) > -step) hi = mid; else lo = mid + 1; }
    *win_lo = lo;
    lo = -1; hi = 65535;                                 // hi: last median with depthDiff < step
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (depth_diff(mid, tp, st.depthOffset) < step) lo = mid; else hi = mid - 1; }
    *win_hi = lo;
}

void PostProcessor::depth_queries(const Prepared& p, const std::vector<TemplatePose>& templates, int slot, std::vector<lm_depth_query>& out) const {
    const int w = st.videoWidth, h = st.videoHeight;
    for (const lm_match_t& m : p.todo) {
        lm_depth_query q;
        std::memset(&q, 0, sizeof(q));
        q.slot = slot;
        if ((size_t)m.template_id < templates.size() && st.useDepthImprovement) {
            const TemplatePose& tp = templates[(size_t)m.template_id];
            int lo, hi;
            depth_window(tp, &lo, &hi);
            // the crop exactly as crop_depth clips it; the window exactly as median_mat_in_window clamps it
            const int x0 = std::max(m.x, 0), y0 = std::max(m.y, 0);
            const int x1 = (int)std::min<long long>((long long)m.x + tp.bb[2], w), y1 = (int)std::min<long long>((long long)m.y + tp.bb[3], h);
            if (x1 > x0 && y1 > y0 && lo <= hi && lo <= 65535 && hi >= 0) {
                q.x0 = x0; q.y0 = y0; q.x1 = x1; q.y1 = y1;
                q.lo = std::max(0, std::min(lo, 65535)); q.hi = std::max(0, std::min(hi, 65535));
            }
        }
        out.push_back(q);
    }
}

bool PostProcessor::depth_check(const lm_match_t& m, const uint16_t* depth, const std::vector<TemplatePose>& t, int32_t* tempDepth, bool* decided_early,
                                const uint32_t* counts) const {
    const TemplatePose& tp = t[(size_t)m.template_id];
    if (decided_early) *decided_early = false;
    if (st.useDepthImprovement) {
        Rect bb{m.x, m.y, tp.bb[2], tp.bb[3]};
        const int32_t step = (int32_t)st.stepSize;
        int win_lo, win_hi;
        depth_window(tp, &win_lo, &win_hi);
        // r06: median_mat_in_window's early verdict from the GPU's counts of this very crop and window (a query with an empty crop or window carries
        // size 0 and decides nothing): more than n / 4 values below the window, or none inside it
        if (counts && counts[2] != 0 && win_lo <= win_hi && win_lo <= 65535 && win_hi >= 0 &&
            ((size_t)counts[0] >= (size_t)counts[2] / 4 + 1 || counts[1] == 0)) { if (decided_early) *decided_early = true; return false; }
        uint16_t med = 0;
        if (win_lo > win_hi || !median_mat_in_window(depth, st.videoWidth, st.videoHeight, bb, 5, depth_ox, depth_oy, win_lo, win_hi, &med, decided_early)) return false;
        const int32_t depthDiff = depth_diff(med, tp, st.depthOffset);
        *tempDepth = (int32_t)(tp.translation[2] + (float)depthDiff);
        return std::abs(depthDiff) < step;                   // (true: med is inside the window)
    }
    *tempDepth = (int32_t)tp.translation[2];
    return true;
}

ObjectPose PostProcessor::make_pose(const lm_match_t& m, const std::vector<TemplatePose>& t, int32_t tempDepth) const {
    const TemplatePose& tp = t[(size_t)m.template_id];
    const int halfW = st.videoWidth / 2, halfH = st.videoHeight / 2;
    // matchToPixelCoord (:497-503)
    float pixelX = (float)(m.x + halfW - tp.bb[0]);
    float pixelY = (float)(m.y + halfH - tp.bb[1]);
    // pixelDistToCenter (:505-510), calcTrueZ (:512-515), calcPosition (:474-485)
    float cx = pixelX - (float)halfW, cy = pixelY - (float)halfH;
    float offsetFromCenter = std::sqrt(cx * cx + cy * cy);
    float direct = (float)tempDepth;
    ObjectPose pose;
    pose.translation.z = std::sqrt(direct * direct - (offsetFromCenter * offsetFromCenter));
    float mmOffsetFromCenter = pose.translation.z / st.fy;
    pose.translation.x = (pixelX - (float)halfW) * mmOffsetFromCenter;
    pose.translation.y = (pixelY - (float)halfH) * mmOffsetFromCenter;
    // calcRotation (:488-495)
    Mat4 adjust = lookAt(Vec3{-pose.translation.x, -pose.translation.y, pose.translation.z}, Vec3{0, 0, 0}, Vec3{0, 1, 0});
    Quat tq; tq.x = tp.quat_xyzw[0]; tq.y = tp.quat_xyzw[1]; tq.z = tp.quat_xyzw[2]; tq.w = tp.quat_xyzw[3];
    pose.quaternions = toQuat(mul(adjust, toMat4(tq)));
    pose.boundingBox = Rect{m.x, m.y, tp.bb[2], tp.bb[3]};
    return pose;
}

PostProcessor::Times& PostProcessor::times() { static thread_local Times t; return t; }

PostProcessor::Prepared PostProcessor::prepare_groups(const std::vector<lm_match_t>& matches, const std::vector<TemplatePose>& templates, Times* tmp) {
    using clk = std::chrono::steady_clock;
    Times& tm = tmp ? *tmp : times();
    Prepared p;
    if (matches.empty()) return p;
    const clk::time_point t_g = clk::now();
    p.groups = discard_small_groups(group_similar_matches(matches, st.radiusThresholdNewObject), st.discardGroupRatio);
    p.gpos.assign(matches.size(), (size_t)-1);
    for (const MatchGroup& g : p.groups)
        for (uint32_t idx : g.matchIndices)
            if ((size_t)matches[idx].template_id < templates.size()) { p.gpos[idx] = p.todo.size(); p.todo.push_back(matches[idx]); }
    tm.grouping += std::chrono::duration<double>(clk::now() - t_g).count(); tm.groups += (long)p.groups.size();
    return p;
}

PostProcessor::Prepared PostProcessor::prepare(const std::vector<lm_match_t>& matches, const uint8_t* bgr, size_t bgr_stride,
                                               const std::vector<TemplatePose>& templates, const ModelProperties& props, int gpu_slot, Times* tmp) {
    using clk = std::chrono::steady_clock;
    Times& tm = tmp ? *tmp : times();
    Prepared p = prepare_groups(matches, templates, tmp);
    if (matches.empty()) return p;
    const int w = st.videoWidth, h = st.videoHeight;
    const clk::time_point t_c = clk::now();
    // colour check: on the host one match at a time as the reference does (:424-434), or every match of every surviving
    // group in one GPU batch up front -- the sequential accept / break logic of finish_group then only looks the verdicts up
    if (gpu_slot >= 0) {
        std::vector<int64_t> gin(p.todo.size()), gboth(p.todo.size());
        if (lm_color_check_counts(det, gpu_slot, props.lowerColorRange, props.upperColorRange, p.todo.data(), p.todo.size(),
                                  gin.data(), gboth.data()) != LM_OK) {
            // loud, never a silent switch of implementation: frames of up to 4992 rows run on the GPU; beyond that the
            // caller selects the host check (HighLevelLineMOD::setGpuColorCheck(false))
            error = lm_last_error();
            p.failed = true;
            p.groups.clear();
        } else {
            set_counts(p, gin.data(), gboth.data());
        }
    } else {
        bgr2hsv_inrange(bgr, w, h, bgr_stride, props.lowerColorRange, props.upperColorRange, p.color_mask);   // :159-161
    }
    tm.colour += std::chrono::duration<double>(clk::now() - t_c).count(); tm.colour_checks += (long)p.gin.size();
    return p;
}

// ---- :165-174, applyPostProcessing (:382-421) for one group, in pieces (r05) -----------------------------------------------------------
// The reference walks a group's matches in order: colour check && depth check -> pose, until numberWantedPoses poses are collected.  The
// two checks of a match are pure functions of the frame, so they may be evaluated ahead of the walk (in parallel, on any thread); the
// walk itself -- which verdicts COUNT -- stays sequential: accept_range.  finish_group = one match at a time, nothing evaluated in vain.
bool PostProcessor::colour_ok(const Prepared& p, uint32_t idx, const lm_match_t& m) const {
    return p.gpu ? color_verdict((long)p.gin[p.gpos[idx]], (long)p.gboth[p.gpos[idx]], st.percentToPassCheck) : color_check(m, p.color_mask);
}

void PostProcessor::depth_part(const lm_match_t& m, const uint16_t* depth_rows, const std::vector<TemplatePose>& templates, MatchVerdict& v, Times* tm) const {
    depth_part(Prepared(), 0, m, depth_rows, templates, v, tm);
}

void PostProcessor::depth_part(const Prepared& p, uint32_t idx, const lm_match_t& m, const uint16_t* depth_rows, const std::vector<TemplatePose>& templates, MatchVerdict& v, Times* tm,
                               bool use_counts) const {
    using clk = std::chrono::steady_clock;
    v.depth_done = true; v.depth_ok = true;
    if ((size_t)m.template_id >= templates.size()) return;
    v.tempDepth = (int32_t)templates[(size_t)m.template_id].translation[2];
    if (!depth_rows) return;
    const clk::time_point t_d = clk::now();
    bool early = false;
    uint32_t cnt[3];
    const bool have = use_counts && p.depth_counts && idx < p.gpos.size() && p.gpos[idx] != (size_t)-1;
    if (have) { const size_t g = p.gpos[idx]; cnt[0] = p.dbelow[g]; cnt[1] = p.dinside[g]; cnt[2] = p.dsize[g]; }
    v.depth_ok = depth_check(m, depth_rows, templates, &v.tempDepth, &early, have ? cnt : nullptr);
    if (tm) { tm->depth += std::chrono::duration<double>(clk::now() - t_d).count(); tm->depth_checks += 1; tm->depth_decided_early += early ? 1 : 0; }
}

bool PostProcessor::accept_range(const Prepared& p, size_t group, const std::vector<lm_match_t>& matches, const std::vector<TemplatePose>& templates,
                                 size_t from, size_t to, const MatchVerdict* v, std::vector<ObjectPose>& objPoses, Times* tm) const {
    using clk = std::chrono::steady_clock;
    const std::vector<uint32_t>& idxs = p.groups[group].matchIndices;
    for (size_t k = from; k < to && k < idxs.size(); ++k) {
        const lm_match_t& m = matches[idxs[k]];
        if ((size_t)m.template_id >= templates.size()) continue;
        if (v[k - from].colour_ok && v[k - from].depth_ok) {                        // (a depth verdict evaluated ahead of a failed colour check is simply not looked at)
            const clk::time_point t_p = clk::now();
            objPoses.push_back(make_pose(m, templates, v[k - from].tempDepth));
            if (tm) { tm->pose += std::chrono::duration<double>(clk::now() - t_p).count(); tm->poses += 1; }
        }
        if (objPoses.size() == st.numberWantedPoses) return true;
    }
    return false;
}

std::vector<ObjectPose> PostProcessor::finish_group(const Prepared& p, size_t group, const std::vector<lm_match_t>& matches, const uint16_t* depth_rows,
                                                    const std::vector<TemplatePose>& templates, Times* tm) const {
    std::vector<ObjectPose> objPoses;
    const std::vector<uint32_t>& idxs = p.groups[group].matchIndices;
    for (size_t k = 0; k < idxs.size(); ++k) {
        const lm_match_t& m = matches[idxs[k]];
        if ((size_t)m.template_id >= templates.size()) continue;
        MatchVerdict v;
        v.colour_ok = colour_ok(p, idxs[k], m);
        if (v.colour_ok) depth_part(m, depth_rows, templates, v, tm);                 // && short-circuit like the reference
        if (accept_range(p, group, matches, templates, k, k + 1, &v, objPoses, tm)) break;
    }
    return objPoses;
}

std::vector<std::vector<ObjectPose>> PostProcessor::run(const std::vector<lm_match_t>& matches, const uint8_t* bgr,
                                                        size_t bgr_stride, const uint16_t* depth, size_t depth_stride,
                                                        const std::vector<TemplatePose>& templates, const ModelProperties& props,
                                                        int gpu_slot) {
    std::vector<std::vector<ObjectPose>> poses;
    if (matches.empty()) return poses;
    const int w = st.videoWidth, h = st.videoHeight;
    std::vector<uint16_t> dense_depth;
    const uint16_t* depth_rows = depth;                 // dense rows: used where they are (no copy per class and frame)
    if (depth && depth_stride != 0 && depth_stride != (size_t)w * 2) {
        dense_depth.resize((size_t)w * h);
        for (int y = 0; y < h; ++y)
            std::memcpy(&dense_depth[(size_t)y * w], reinterpret_cast<const uint8_t*>(depth) + y * depth_stride, (size_t)w * 2);
        depth_rows = dense_depth.data();
    }
    const Prepared p = prepare(matches, bgr, bgr_stride, templates, props, gpu_slot);
    for (size_t g = 0; g < p.groups.size(); ++g) {
        std::vector<ObjectPose> objPoses = finish_group(p, g, matches, depth_rows, templates, &times());
        if (!objPoses.empty()) poses.push_back(objPoses);
    }
    return poses;
}

}  // namespace lmamd
