// lm_extract.cpp -- feature selection for Detector::addTemplate (see lm_extract.h).
//
// Upstream helpers restated: ColorGradientPyramid::extractTemplate,
// DepthNormalPyramid::extractTemplate, QuantizedPyramid::selectScatteredFeatures, cropTemplates.
#include "lm_extract.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace lmh {
namespace {

struct Scored {
    lm_feature f;
    float score;
};

int label_of(unsigned q) {  // one-hot byte -> bin, -1 if not one-hot
    if (q == 0 || (q & (q - 1))) return -1;
    int b = 0;
    while (!(q & 1u)) { q >>= 1; ++b; }
    return b;
}

// 3x3 minimum filter with replicated borders, applied `times` times (cv::erode, BORDER_REPLICATE).
std::vector<u8> shrink_mask(const std::vector<u8>& mask, int w, int h, int times) {
    std::vector<u8> cur = mask, rowmin((size_t)w * h);
    for (int t = 0; t < times; ++t) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const u8* r = &cur[(size_t)y * w];
                u8 a = r[x > 0 ? x - 1 : 0], b = r[x], c = r[x + 1 < w ? x + 1 : w - 1];
                rowmin[(size_t)y * w + x] = std::min(a, std::min(b, c));
            }
        for (int y = 0; y < h; ++y) {
            const u8* up = &rowmin[(size_t)(y > 0 ? y - 1 : 0) * w];
            const u8* mid = &rowmin[(size_t)y * w];
            const u8* dn = &rowmin[(size_t)(y + 1 < h ? y + 1 : h - 1) * w];
            for (int x = 0; x < w; ++x) cur[(size_t)y * w + x] = std::min(up[x], std::min(mid[x], dn[x]));
        }
    }
    return cur;
}

// Chessboard distance to the nearest zero pixel (distanceTransform(DIST_C, 3)); two raster sweeps are
// exact for this metric.  Outside the image counts as far away.
std::vector<float> chessboard_distance(const std::vector<u8>& nz, int w, int h) {
    const int FAR = INT_MAX >> 2;
    std::vector<int> d((size_t)w * h);
    for (size_t i = 0; i < d.size(); ++i) d[i] = nz[i] ? FAR : 0;
    auto get = [&](int y, int x) { return (x < 0 || y < 0 || x >= w || y >= h) ? FAR : d[(size_t)y * w + x]; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int& v = d[(size_t)y * w + x];
            if (v == 0) continue;
            int m = std::min(std::min(get(y - 1, x - 1), get(y - 1, x)), std::min(get(y - 1, x + 1), get(y, x - 1)));
            v = std::min(v, m + 1);
        }
    for (int y = h - 1; y >= 0; --y)
        for (int x = w - 1; x >= 0; --x) {
            int& v = d[(size_t)y * w + x];
            if (v == 0) continue;
            int m = std::min(std::min(get(y + 1, x + 1), get(y + 1, x)), std::min(get(y + 1, x - 1), get(y, x + 1)));
            v = std::min(v, m + 1);
        }
    std::vector<float> out(d.size());
    for (size_t i = 0; i < d.size(); ++i) out[i] = (float)d[i];
    return out;
}

// Greedy scattered pick: walk the (score-sorted) candidates cyclically, keep one if it is at least
// `distance` from everything kept so far, relax the distance by 1 after each full pass.
void pick_scattered(const std::vector<Scored>& cands, size_t want, float distance, std::vector<lm_feature>& out) {
    out.clear();
    float d2 = distance * distance;
    size_t i = 0;
    while (out.size() < want) {
        const lm_feature& c = cands[i].f;
        bool far_enough = true;
        for (const lm_feature& k : out) {
            int dx = c.x - k.x, dy = c.y - k.y;
            if (!((float)(dx * dx + dy * dy) >= d2)) { far_enough = false; break; }
        }
        if (far_enough) out.push_back(c);
        if (++i == cands.size()) { i = 0; distance -= 1.0f; d2 = distance * distance; }
    }
}

bool by_score_desc(const Scored& a, const Scored& b) { return a.score > b.score; }

bool pick_color(const ExtractLevel& L, float strong_threshold, size_t want, Template& t) {
    const bool masked = !L.mask.empty();
    std::vector<u8> rim;
    if (masked) {  // features on the object border: mask minus its erosion
        rim = shrink_mask(L.mask, L.w, L.h, 1);
        for (size_t i = 0; i < rim.size(); ++i) rim[i] = (u8)(L.mask[i] > rim[i] ? L.mask[i] - rim[i] : 0);
    }
    const float min_mag = strong_threshold * strong_threshold;
    std::vector<Scored> cands;
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x) {
            size_t i = (size_t)y * L.w + x;
            if (masked && !rim[i]) continue;
            u8 q = L.color_q[i];
            if (q == 0 || !(L.color_mag[i] > min_mag)) continue;
            cands.push_back(Scored{{x, y, label_of(q)}, L.color_mag[i]});
        }
    if (cands.size() < want) return false;
    std::stable_sort(cands.begin(), cands.end(), by_score_desc);
    float distance = (float)(cands.size() / want + 1);
    pick_scattered(cands, want, distance, t.features);
    return true;
}

bool pick_depth(const ExtractLevel& L, int extract_threshold, size_t want, Template& t) {
    const bool masked = !L.mask.empty();
    std::vector<u8> inner;
    if (masked) inner = shrink_mask(L.mask, L.w, L.h, 2);  // features right on the border are unreliable
    std::vector<float> dist[8];
    std::vector<u8> sel((size_t)L.w * L.h);
    for (int b = 0; b < 8; ++b) {
        for (size_t i = 0; i < sel.size(); ++i) sel[i] = ((!masked || inner[i]) ? (u8)(1u << b) : (u8)0) & L.depth_q[i];
        dist[b] = chessboard_distance(sel, L.w, L.h);
    }
    int per_label[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<Scored> cands;
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x) {
            size_t i = (size_t)y * L.w + x;
            if (masked && !inner[i]) continue;
            u8 q = L.depth_q[i];
            if (q == 0 || q == 255) continue;
            int lab = label_of(q);
            if (lab < 0) continue;
            float dd = dist[lab][i];
            if (dd >= (float)extract_threshold) { cands.push_back(Scored{{x, y, lab}, dd}); ++per_label[lab]; }
        }
    if (cands.size() < want) return false;
    for (Scored& c : cands) c.score /= (float)per_label[c.f.label];  // spread the pick over all labels
    std::stable_sort(cands.begin(), cands.end(), by_score_desc);
    float area = 0.f;
    if (!masked) area = (float)sel.size();
    else for (u8 v : inner) area += v ? 1.f : 0.f;
    float distance = sqrtf(area) / sqrtf((float)want) + 1.5f;
    pick_scattered(cands, want, distance, t.features);
    return true;
}

}  // namespace

bool extract_pyramid(const std::vector<ExtractLevel>& levels, const lm_config& cfg, TemplatePyramid& tp) {
    const int M = cfg.num_modalities, L = cfg.pyramid_levels;
    tp.assign((size_t)M * L, Template());
    int nf_color = cfg.num_features, nf_depth = cfg.depth_num_features, et = cfg.extract_threshold;
    for (int l = 0; l < L; ++l) {
        if (l > 0) { nf_color /= 2; nf_depth /= 2; et /= 2; }  // pyrDown(): num_features /= 2, extract_threshold /= 2
        Template& tc = tp[(size_t)l * M];
        tc.pyramid_level = l; tc.width = tc.height = -1;
        if (!pick_color(levels[l], cfg.strong_threshold, (size_t)nf_color, tc)) return false;
        if (M == 2) {
            Template& td = tp[(size_t)l * M + 1];
            td.pyramid_level = l; td.width = td.height = -1;
            if (!pick_depth(levels[l], et, (size_t)nf_depth, td)) return false;
        }
    }
    return true;
}

lm_rect crop_templates(TemplatePyramid& tp) {
    int x0 = INT_MAX, y0 = INT_MAX, x1 = INT_MIN, y1 = INT_MIN;
    for (const Template& t : tp)
        for (const lm_feature& f : t.features) {
            int x = f.x << t.pyramid_level, y = f.y << t.pyramid_level;
            x0 = std::min(x0, x); y0 = std::min(y0, y);
            x1 = std::max(x1, x); y1 = std::max(y1, y);
        }
    if (x0 % 2 == 1) --x0;  // upstream keeps the origin even so it survives the level shift
    if (y0 % 2 == 1) --y0;
    for (Template& t : tp) {
        t.width = (x1 - x0) >> t.pyramid_level;
        t.height = (y1 - y0) >> t.pyramid_level;
        int ox = x0 >> t.pyramid_level, oy = y0 >> t.pyramid_level;
        for (lm_feature& f : t.features) { f.x -= ox; f.y -= oy; }
    }
    return lm_rect{x0, y0, x1 - x0, y1 - y0};
}

}  // namespace lmh
