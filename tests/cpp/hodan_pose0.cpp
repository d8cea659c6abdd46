// The reference's own acceptance criterion on the reference's own fixture (test-side only): templates of models/lagergehaeuse.ply with the
// SHIPPED generation settings (linemod_settings.yml:20-27: colour-only modality, radii 500..1200 step 50, in-plane rotations -45..45 step 10; 13 viewpoints
// of the symmetry-reduced icosphere) = 1950 templates, detection of the part in benchmark/img0.png + depth0.png at threshold 80, then the error function of
// Hodan et al. exactly as /root/reference/src/Benchmark.cpp:18-38,133-169 computes it: depth renders of the ground-truth pose (benchmark/pose0.yml) and of the
// estimated pose, visibility masks against the input depth with delta = 15 mm, cost threshold tau = 20 mm, error = 1 - |visible both, |difference| < tau| /
// |visible in either|; the reference counts a pose as correct below 0.3 (Benchmark.cpp:33).
// usage: hodan_pose0 <mesh.bin> <bgr.raw> <depth.raw> <gt.txt: 9 rotation entries row-major, 3 position entries>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "../../line-mod-pipeline_amd/host/HighLevelLinemod.h"
#include "../../line-mod-pipeline_amd/host/PostProcess.h"
#include "../../line-mod-pipeline_amd/host/TemplateGenerator.h"

using namespace lmamd;

static std::vector<char> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// glm::eulerAngles(q) = (pitch, yaw, roll) and glm::qua<float>(vec3 eulerAngles)  (glm/gtc/quaternion.inl)
static void euler_angles(const Quat& q, float e[3]) {
    const float y = 2.f * (q.y * q.z + q.w * q.x), x = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    e[0] = (std::fabs(x) < 1e-12f && std::fabs(y) < 1e-12f) ? 2.f * std::atan2(q.x, q.w) : std::atan2(y, x);
    float s = -2.f * (q.x * q.z - q.w * q.y);
    s = s < -1.f ? -1.f : (s > 1.f ? 1.f : s);
    e[1] = std::asin(s);
    e[2] = std::atan2(2.f * (q.x * q.y + q.w * q.z), q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z);
}
static Quat quat_from_euler(const float e[3]) {
    const float cx = std::cos(e[0] * 0.5f), cy = std::cos(e[1] * 0.5f), cz = std::cos(e[2] * 0.5f);
    const float sx = std::sin(e[0] * 0.5f), sy = std::sin(e[1] * 0.5f), sz = std::sin(e[2] * 0.5f);
    Quat q;
    q.w = cx * cy * cz + sx * sy * sz;
    q.x = sx * cy * cz - cx * sy * sz;
    q.y = cx * sy * cz + sx * cy * sz;
    q.z = cx * cy * sz - sx * sy * cz;
    return q;
}
// Benchmark::calculateViewMat (:165-170) + OpenGLRender::renderDepthToFrontBuff(indice, rotMat, traVec) (:116-141)
static void render_pose(const SoftRender& render, const Mesh& mesh, const Quat& q, const Vec3& t, std::vector<uint16_t>& depth) {
    float e[3];
    euler_angles(q, e);
    const float f[3] = {e[0] - 3.14159265358979323846f, -e[1], -e[2]};
    Mat4 view = toMat4(quat_from_euler(f));
    view.m[3][0] = t.x; view.m[3][1] = -t.y; view.m[3][2] = -t.z; view.m[3][3] = 1.0f;
    std::vector<uint8_t> bgr;
    render.render_view(mesh, view.m, bgr, depth);
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    std::vector<char> mb = slurp(argv[1]);
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(mb.data());
    const uint32_t nv = hdr[0], nf = hdr[1];
    const float* v = reinterpret_cast<const float*>(mb.data() + 8);
    const int32_t* fi = reinterpret_cast<const int32_t*>(mb.data() + 8 + (size_t)nv * 12);
    Mesh mesh;
    mesh.vertices.resize(nv);
    for (uint32_t i = 0; i < nv; ++i) mesh.vertices[i] = Vec3{v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    mesh.indices.assign(fi, fi + (size_t)nf * 3);
    const int W = 640, H = 480;
    CameraParameters cam;   // linemod_settings.yml
    cam.fx = 1044.87f; cam.fy = 1045.69141f; cam.cx = 320; cam.cy = 240; cam.videoWidth = W; cam.videoHeight = H;
    TemplateGenerationSettings ts;   // linemod_settings.yml:20-27 as shipped
    ts.onlyUseColorModality = true;
    ts.detectorThreshold = 80.f;
    HighLevelLineMOD line(cam, ts);
    SoftRender render(cam);
    SymmetryProperties sym;   // models/lagergehaeuse.yml
    sym.rotationallySymmetrical = true; sym.planesOfSymmetry = Vec3{1, 1, 1};
    GeneratorSettings gs;     // startDistance 500, endDistance 1200, stepSize 50, subdivisions 3: 15 radii
    const int n = generate_templates(line, render, mesh, "lagergehaeuse.ply", sym, gs);
    std::printf("templates %d\n", n);
    double lo[3] = {0, 0, 0}, hi[3] = {255, 150, 255};
    line.setColorRange(0, lo, hi);
    std::vector<char> bgr = slurp(argv[2]), depth = slurp(argv[3]);
    std::vector<Image> imgs(2);
    imgs[0].data = bgr.data(); imgs[0].width = W; imgs[0].height = H;
    imgs[1].data = depth.data(); imgs[1].width = W; imgs[1].height = H; imgs[1].type = 1;
    const bool found = line.detectTemplate(imgs, 0);
    std::printf("found %d matches %zu error '%s'\n", found ? 1 : 0, line.getMatches().size(), line.lastError().c_str());
    auto poses = line.getObjectPoses();
    if (poses.empty() || poses[0].empty()) { std::printf("no pose\n"); return 0; }
    const ObjectPose est = poses[0][0];
    std::printf("pose t %.3f %.3f %.3f q %.5f %.5f %.5f %.5f\n", est.translation.x, est.translation.y, est.translation.z, est.quaternions.w,
                est.quaternions.x, est.quaternions.y, est.quaternions.z);
    // ground truth: Benchmark::readGroundTruthPose (:180-192): rotMat -> glm mat3 (same matrix) -> quaternion
    std::ifstream gt(argv[4]);
    double R[9], T[3];
    for (double& r : R) gt >> r;
    for (double& t : T) gt >> t;
    Mat4 Rm;
    std::memset(Rm.m, 0, sizeof(Rm.m));
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rm.m[c][r] = (float)R[3 * r + c];
    Rm.m[3][3] = 1.f;
    const Quat gq = toQuat(Rm);
    const Vec3 gtv{(float)T[0], (float)T[1], (float)T[2]};
    std::vector<uint16_t> dg, de;
    render_pose(render, mesh, gq, gtv, dg);
    render_pose(render, mesh, est.quaternions, est.translation, de);
    const uint16_t* in = reinterpret_cast<const uint16_t*>(depth.data());
    // calculateVisibilityMasks (:133-154) on CV_16U images: saturating subtraction, thresholds as cv::threshold on 16-bit data
    const int delta = 15, tau = 20;
    long n_gt = 0, n_est = 0, inter = 0, uni = 0, good = 0, px_gt = 0, px_est = 0;
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const int g = dg[i], e = de[i], d = in[i];
        px_gt += g > 1; px_est += e > 1;
        // groundTruthVisibility = (gtRender > 1) - ((gtRender - input) > delta), saturating
        const bool g_occl = (g > d ? g - d : 0) > delta;
        const bool vg = (g > 1) && !g_occl;
        const bool e_occl = (e > d ? e - d : 0) > delta;
        bool ve = (e > 1) && !e_occl;
        // estimateVisibility |= groundTruthVisibility & estimateDepthRender  (bitwise on 16-bit values: any common bit of the mask (65535) and the depth)
        if (vg && e != 0) ve = true;
        n_gt += vg; n_est += ve;
        inter += vg && ve; uni += vg || ve;
        const int ad = g > e ? g - e : e - g;
        good += (vg && ve) && !(ad > tau);           // THRESH_BINARY_INV at tau: kept where |difference| <= tau
    }
    const double err = uni ? 1.0 - (double)good / (double)uni : 1.0;
    std::printf("hodan error %.6f  visible gt %ld est %ld intersection %ld union %ld within_tau %ld  rendered gt %ld est %ld\n", err, n_gt, n_est, inter, uni, good, px_gt, px_est);
    return 0;
}
