// CPU unit tests of line-mod-pipeline_amd/host/PostProcess.* (SURVEY.md 8f-1 host glue).
// Exits non-zero on the first failed check; prints "OK" otherwise.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../line-mod-pipeline_amd/host/PostProcess.h"
#include "../../line-mod-pipeline_amd/host/TemplateGenerator.h"

using namespace lmamd;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
static bool near(float a, float b, float eps = 1e-5f) { return std::fabs(a - b) <= eps; }

int main() {
    // ---- BGR -> HSV (OpenCV 8-bit convention: H in [0,180)) + inRange
    {
        const uint8_t px[5 * 3] = {0, 0, 255, /*red*/ 0, 255, 0, /*green*/ 255, 0, 0, /*blue*/ 128, 128, 128, /*gray*/ 0, 255, 255 /*yellow*/};
        std::vector<uint8_t> m;
        double lo[3] = {0, 0, 0}, hi[3] = {255, 255, 255};
        bgr2hsv_inrange(px, 5, 1, 0, lo, hi, m);
        for (int i = 0; i < 5; ++i) CHECK(m[i] == 255);
        double lo2[3] = {55, 200, 200}, hi2[3] = {65, 255, 255};   // H = 60: green only
        bgr2hsv_inrange(px, 5, 1, 0, lo2, hi2, m);
        CHECK(m[0] == 0 && m[1] == 255 && m[2] == 0 && m[3] == 0 && m[4] == 0);
        double lo3[3] = {0, 0, 0}, hi3[3] = {255, 150, 255};       // the shipped model file: S <= 150
        bgr2hsv_inrange(px, 5, 1, 0, lo3, hi3, m);
        CHECK(m[0] == 0 && m[3] == 255);                            // saturated red fails, gray passes
        double lo4[3] = {115, 0, 0}, hi4[3] = {125, 255, 255};      // H = 120: blue; H = 30: yellow
        bgr2hsv_inrange(px, 5, 1, 0, lo4, hi4, m);
        CHECK(m[2] == 255 && m[4] == 0);
    }
    // ---- convex hull + fill counts
    {
        std::vector<Pt> pts = {{10, 10}, {20, 10}, {20, 20}, {10, 20}, {15, 15}, {12, 18}, {15, 10}};
        std::vector<Pt> hull = convex_hull(pts);
        CHECK(hull.size() == 4);
        std::vector<uint8_t> cm(64 * 64, 0);
        for (int y = 0; y < 64; ++y) for (int x = 0; x < 16; ++x) cm[y * 64 + x] = 255;   // left part coloured
        long a = 0, b = 0;
        hull_counts(hull, cm.data(), 64, 64, &a, &b);
        CHECK(a == 11 * 11);            // closed square 10..20
        CHECK(b == 6 * 11);             // columns 10..15
        std::vector<Pt> tri = convex_hull({{0, 0}, {8, 0}, {0, 8}});
        hull_counts(tri, cm.data(), 64, 64, &a, &b);
        CHECK(a == 45 && b == 45);      // lattice points of the closed triangle x + y <= 8
        std::vector<Pt> off = convex_hull({{60, 60}, {70, 60}, {70, 70}, {60, 70}});   // partly outside the image
        hull_counts(off, cm.data(), 64, 64, &a, &b);
        CHECK(a == 16 && b == 0);
    }
    // ---- grouping (:206-253)
    {
        std::vector<lm_match_t> ms;
        auto add = [&](int x, int y) { lm_match_t m; m.x = x; m.y = y; m.similarity = 90; m.template_id = 0; m.class_idx = 0; ms.push_back(m); };
        add(0, 0); add(10, 0); add(100, 0); add(44, 0); add(45, 0); add(130, 10); add(5, 5); add(300, 300);
        std::vector<MatchGroup> g = group_similar_matches(ms, 45.f);
        CHECK(g.size() == 4);                       // {0,1,3,6}, {2,5}, {4}, {7}: distance 45 is not < 45
        CHECK(g[0].matchIndices.size() == 4 && g[1].matchIndices.size() == 2 && g[2].matchIndices.size() == 1);
        CHECK(g[2].position.x == 45);
        std::vector<MatchGroup> k = discard_small_groups(g, 35.f);
        CHECK(k.size() == 2);                       // 4 -> 100, 2 -> 50 kept; 1 -> 25 dropped
        std::vector<MatchGroup> k2 = discard_small_groups(g, 50.f);
        CHECK(k2.size() == 1);                      // 50 is not > 50
    }
    // ---- medianMat quirk (:336-349)
    {
        std::vector<uint16_t> d(20 * 10);
        for (int i = 0; i < 200; ++i) d[i] = (uint16_t)(1000 - i);
        d[5] = 0; d[6] = 1;                          // holes count as 65535
        Rect bb{0, 0, 20, 10};
        uint16_t v = median_mat(d.data(), 20, 10, bb, 5);
        CHECK(v <= 1000 - 150);                      // some element of the lower quarter (values 801..850 are the 50 smallest)
        CHECK(v >= 801);
        Rect clip{15, 5, 20, 20};                    // partly outside: clipped instead of cv::Mat's assert
        CHECK(median_mat(d.data(), 20, 10, clip, 5) >= 801);
    }
    // ---- mini GLM
    {
        Vec3 r = rotate(Vec3{1, 0, 0}, 1.57079632679f, Vec3{0, 0, 1});
        CHECK(near(r.x, 0, 1e-6f) && near(r.y, 1, 1e-6f) && near(r.z, 0, 1e-6f));
        Mat4 v = lookAt(Vec3{0, 0, 5}, Vec3{0, 0, 0}, Vec3{0, 1, 0});
        CHECK(near(v.m[0][0], 1) && near(v.m[1][1], 1) && near(v.m[2][2], 1) && near(v.m[3][2], -5));
        Quat q; q.w = 0.5f; q.x = 0.5f; q.y = -0.5f; q.z = 0.5f;
        Quat b = toQuat(toMat4(q));
        CHECK(near(b.w, q.w) && near(b.x, q.x) && near(b.y, q.y) && near(b.z, q.z));
        Mat4 I = mul(toMat4(q), transpose(toMat4(q)));
        CHECK(near(I.m[0][0], 1) && near(I.m[1][0], 0) && near(I.m[2][1], 0) && near(I.m[2][2], 1));
        float t[3], qq[4];
        calculate_template_pose(Vec3{0, 0, 700}, 0, t, qq);
        CHECK(near(t[0], 0) && near(t[1], 0) && near(t[2], 700));
        CHECK(near(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3], 1, 1e-5f));
        calculate_template_pose(Vec3{0, 650, 0}, -45, t, qq);           // straight above: the cross-product guard (:358-362)
        CHECK(near(t[2], 650) && std::isfinite(qq[0]) && std::isfinite(qq[3]));
    }
    // ---- translateImg (PoseDetection.cpp:192-197)
    {
        std::vector<uint16_t> d(4 * 3), o;
        for (int i = 0; i < 12; ++i) d[i] = (uint16_t)(i + 1);
        translate_u16(d.data(), 4, 3, 1, -1, o);
        CHECK(o[0] == 0 && o[1] == 5 && o[2] == 6 && o[3] == 7 && o[8] == 0 && o[11] == 0);
    }
    // ---- r04: the row-wise translation against the per-pixel definition, every kind of shift (also beyond the frame), both types;
    // and the depth check's median read THROUGH a pending shift against the median on the translated copy
    {
        const int W = 37, H = 23;
        std::vector<uint8_t> c((size_t)W * H * 3), co;
        std::vector<uint16_t> d((size_t)W * H), dd;
        uint32_t r = 12345;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r >> 8; };
        for (uint8_t& v : c) v = (uint8_t)rnd();
        for (uint16_t& v : d) v = (uint16_t)(rnd() % 7 == 0 ? rnd() % 2 : 400 + rnd() % 900);     // holes (0 / 1) and depths
        const int shifts[][2] = {{0, 0}, {5, -3}, {-12, 10}, {36, 22}, {-36, -22}, {37, 0}, {0, -23}, {100, 100}, {1, 1}};
        for (const auto& sh : shifts) {
            const int ox = sh[0], oy = sh[1];
            translate_u8c3(c.data(), W, H, ox, oy, co);
            translate_u16(d.data(), W, H, ox, oy, dd);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int sx = x - ox, sy = y - oy;
                    const bool in = sx >= 0 && sx < W && sy >= 0 && sy < H;
                    CHECK(dd[(size_t)y * W + x] == (in ? d[(size_t)sy * W + sx] : 0));
                    for (int k = 0; k < 3; ++k) CHECK(co[((size_t)y * W + x) * 3 + k] == (in ? c[((size_t)sy * W + sx) * 3 + k] : 0));
                }
            const Rect boxes[] = {{0, 0, W, H}, {3, 2, 10, 8}, {-4, -4, 12, 12}, {30, 18, 20, 20}, {10, 5, 1, 1}, {40, 5, 3, 3}, {12, 0, 25, 23}};
            for (const Rect& bb : boxes)
                for (uint8_t pos : {(uint8_t)5, (uint8_t)4, (uint8_t)0})
                    CHECK(median_mat(d.data(), W, H, bb, pos, ox, oy) == median_mat(dd.data(), W, H, bb, pos));
        }
    }
    // ---- r05: the depth check's bounded median (median_mat_in_window) against the plain one on random crops, windows and shifts --
    // "inside the window" and the value agree with median_mat every time (the early verdict is a proof, never a guess), early verdicts
    // happen on both sides (count below the window / minimum above it), and widths around the 16-pixel vector step are covered
    {
        uint32_t r = 777;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r >> 8; };
        long early = 0, inside = 0, outside_late = 0, total = 0;
        for (int W : {64, 50, 33, 17, 16, 15, 129}) {
            const int H = 40;
            std::vector<uint16_t> d((size_t)W * H);
            for (int kind = 0; kind < 3; ++kind) {
                for (uint16_t& v : d) v = (uint16_t)(rnd() % 9 == 0 ? rnd() % 2 : kind == 0 ? 500 + rnd() % 600 : kind == 1 ? 800 + rnd() % 8 : rnd() % 65536);
                for (int it = 0; it < 400; ++it) {
                    const Rect bb{(int)(rnd() % (unsigned)(W + 10)) - 5, (int)(rnd() % (unsigned)(H + 10)) - 5, 1 + (int)(rnd() % (unsigned)W), 1 + (int)(rnd() % (unsigned)H)};
                    const int ox = it % 3 == 0 ? (int)(rnd() % 21) - 10 : 0, oy = it % 3 == 0 ? (int)(rnd() % 21) - 10 : 0;
                    int lo = (int)(rnd() % 1400), hi = lo + (int)(rnd() % 300) - 20;           // sometimes an empty window
                    if (it % 11 == 0) { lo = 65535 - (int)(rnd() % 3); hi = 65535 + (int)(rnd() % 2); }   // the holes' value
                    if (it % 13 == 0) { lo = -5; hi = (int)(rnd() % 900); }
                    const uint16_t want = median_mat(d.data(), W, H, bb, 5, ox, oy);
                    uint16_t got = 12345; bool e = false;
                    const bool in = median_mat_in_window(d.data(), W, H, bb, 5, ox, oy, lo, hi, &got, &e);
                    CHECK(in == ((int)want >= lo && (int)want <= hi));
                    if (in) CHECK(got == want);
                    if (!e && !in) CHECK(got == want);
                    early += e; inside += in; outside_late += (!in && !e); ++total;
                    // r06: the same early verdict from counts taken on the TRANSLATED frame -- what k_depth_counts counts on the resident frame for a query
                    // built like PostProcessor::depth_queries (crop clipped to the frame, window clamped to 0 .. 65535): it must be the host's own, crop by crop
                    {
                        std::vector<uint16_t> tr;
                        translate_u16(d.data(), W, H, ox, oy, tr);
                        const int x0 = std::max(bb.x, 0), y0 = std::max(bb.y, 0), x1 = std::min(bb.x + bb.width, W), y1 = std::min(bb.y + bb.height, H);
                        bool early2 = false;
                        if (x1 > x0 && y1 > y0 && lo <= hi && lo <= 65535 && hi >= 0) {
                            const int cl = std::max(0, std::min(lo, 65535)), ch = std::max(0, std::min(hi, 65535));
                            size_t cb = 0, ci = 0;
                            const size_t n = (size_t)(x1 - x0) * (size_t)(y1 - y0);
                            for (int y = y0; y < y1; ++y)
                                for (int x = x0; x < x1; ++x) { const int v = tr[(size_t)y * W + x], t = v <= 1 ? 65535 : v; cb += t < cl; ci += (t >= cl && t <= ch); }
                            early2 = cb >= n / 4 + 1 || ci == 0;
                        }
                        CHECK(early2 == e);
                        if (early2) CHECK(!in);
                    }
                }
            }
        }
        CHECK(early > total / 4); CHECK(inside > total / 50); CHECK(outside_late > 0);
        std::fprintf(stderr, "median_mat_in_window: %ld crops, %ld inside the window, %ld outside by the bounds, %ld outside after the selection\n", total, inside, early, outside_late);
    }
    // ---- CameraViewPoints (CameraViewPoints.cpp): shipped model = rotationally symmetric, planes (1,1,1),
    // subdivisions 3 -> 13 viewpoints on the quarter arc (13 x 15 x 10 = 1950 templates)
    {
        CameraViewPoints cv;
        SymmetryProperties sym; sym.rotationallySymmetrical = true; sym.planesOfSymmetry = Vec3{1, 1, 1};
        cv.setModelProperties(sym);
        cv.createCameraViewPoints(600.f, 3);
        CHECK(cv.getVertices().size() == 13);
        for (const Vec3& v : cv.getVertices()) { CHECK(v.x == 0 && v.y >= 0 && v.z >= 0); CHECK(near(length(v), 600.f, 1e-2f)); }
        CHECK(near(cv.getVertices()[0].z, 600.f, 1e-3f));                      // i = 0: on the +z axis
        SymmetryProperties none;                                                // no symmetry: full icosphere
        cv.setModelProperties(none);
        cv.createCameraViewPoints(500.f, 0); CHECK(cv.getVertices().size() == 12);
        cv.createCameraViewPoints(500.f, 1); CHECK(cv.getVertices().size() == 42);
        cv.createCameraViewPoints(500.f, 2); CHECK(cv.getVertices().size() == 162);   // the "~24k" bank: 162 x 15 x 10
        for (const Vec3& v : cv.getVertices()) CHECK(near(length(v), 500.f, 5e-2f));
    }
    // ---- SoftRender: a 100 mm square plate facing the camera at 800 mm
    {
        CameraParameters cam; cam.fx = 1044.87f; cam.fy = 1045.69141f; cam.cx = 320; cam.cy = 240;
        SoftRender r(cam);
        Mesh m;
        m.vertices = {{-50, -50, 0}, {50, -50, 0}, {50, 50, 0}, {-50, 50, 0}};
        m.indices = {0, 1, 2, 0, 2, 3};
        std::vector<uint8_t> bgr; std::vector<uint16_t> depth;
        r.render(m, Vec3{0, 0, 800}, bgr, depth);
        CHECK(depth[240 * 640 + 320] == 800 && bgr[(240 * 640 + 320) * 3] == 255);
        CHECK(depth[0] == 0 && bgr[0] == 0);
        long covered = 0;
        for (uint16_t d : depth) covered += d != 0;
        double side = 100.0 * 1045.69141 / 800.0;                               // projected side in pixels
        CHECK(std::fabs(covered - side * side) < 0.03 * side * side);
        // moving the camera up (+y) moves the plate down in the image (rows are flipped like the reference)
        r.render(m, Vec3{0, 200, 800}, bgr, depth);
        int ymin = 480, ymax = -1;
        for (int y = 0; y < 480; ++y) for (int x = 0; x < 640; ++x) if (depth[y * 640 + x]) { ymin = std::min(ymin, y); ymax = std::max(ymax, y); }
        CHECK(ymin < ymax && (ymin + ymax) / 2 >= 235 && (ymin + ymax) / 2 <= 245);   // lookAt keeps the origin centred
    }
    // ---- warpAffine rotation
    {
        std::vector<uint8_t> img(64 * 48, 0), out;
        for (int y = 10; y < 20; ++y) for (int x = 28; x < 36; ++x) img[y * 64 + x] = 255;
        warp_rotate_u8(img.data(), 64, 48, 1, 0.f, out);
        CHECK(out == img);
        warp_rotate_u8(img.data(), 64, 48, 1, 180.f, out);                      // about (32, 24): (x, y) -> (64 - x, 48 - y)
        CHECK(out[(48 - 15) * 64 + (64 - 30)] == 255 && out[15 * 64 + 30] == 0);
        std::vector<uint16_t> d16(64 * 48, 0), o16;
        d16[24 * 64 + 40] = 1000;
        warp_rotate_u16(d16.data(), 64, 48, 90.f, o16);                         // positive angle = counter-clockwise on screen
        CHECK(o16[(24 - 8) * 64 + 32] == 1000);
    }
    std::printf("OK\n");
    return 0;
}
