// Consumer of the f1 vectors the pinning hook writes on an OpenCV box (tests/golden/make_opencv_vectors.py, f1_vectors):
// host/PostProcess.cpp's bgr2hsv_inrange / convex_hull / hull_counts against cv::cvtColor + cv::inRange and
// cv::convexHull + cv::fillPoly + cv::countNonZero (HighLevelLinemod.cpp:113-135,159-161,424-434).
// usage: f1_vectors_check <vectors.bin>   (layout: see tests/test_opencv_vectors.py::_f1_bin)
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>

#include "../../line-mod-pipeline_amd/host/PostProcess.h"

using namespace lmamd;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    auto rd = [&](void* p, size_t n) { f.read(static_cast<char*>(p), (std::streamsize)n); return (bool)f; };
    int32_t hdr[4];   // W, H, number of ranges, number of polygons
    if (!rd(hdr, sizeof(hdr))) return 2;
    const int W = hdr[0], H = hdr[1], NR = hdr[2], NP = hdr[3];
    std::vector<uint8_t> bgr((size_t)W * H * 3);
    rd(bgr.data(), bgr.size());
    std::vector<double> ranges((size_t)NR * 6);
    rd(ranges.data(), ranges.size() * 8);
    std::vector<std::vector<uint8_t>> want((size_t)NR, std::vector<uint8_t>((size_t)W * H));
    for (auto& m : want) rd(m.data(), m.size());
    long bad_px = 0;
    std::vector<std::vector<uint8_t>> mask((size_t)NR);
    for (int r = 0; r < NR; ++r) {
        bgr2hsv_inrange(bgr.data(), W, H, 0, &ranges[(size_t)r * 6], &ranges[(size_t)r * 6 + 3], mask[(size_t)r]);
        for (size_t i = 0; i < (size_t)W * H; ++i) bad_px += (mask[(size_t)r][i] != 0) != (want[(size_t)r][i] != 0);
    }
    long bad_poly = 0;
    for (int k = 0; k < NP; ++k) {
        int32_t h2[4];   // n points, ox, oy, range index
        int64_t cnt[2];
        rd(h2, sizeof(h2)); rd(cnt, sizeof(cnt));
        std::vector<int32_t> xy((size_t)h2[0] * 2);
        rd(xy.data(), xy.size() * 4);
        std::vector<Pt> pts;
        for (int i = 0; i < h2[0]; ++i) pts.push_back(Pt{xy[2 * (size_t)i] + h2[1], xy[2 * (size_t)i + 1] + h2[2]});
        long a = 0, b = 0;
        hull_counts(convex_hull(pts), mask[(size_t)h2[3]].data(), W, H, &a, &b);
        if (a != cnt[0] || b != cnt[1]) { ++bad_poly; std::printf("polygon %d: got %ld %ld, OpenCV %lld %lld\n", k, a, b, (long long)cnt[0], (long long)cnt[1]); }
    }
    std::printf("mask pixels differing %ld, polygons differing %ld of %d\n", bad_px, bad_poly, NP);
    return (bad_px || bad_poly) ? 1 : 0;
}
