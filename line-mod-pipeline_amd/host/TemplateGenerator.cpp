// TemplateGenerator.cpp -- see TemplateGenerator.h.  Offline host code, plain C++17.
#include "TemplateGenerator.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "PostProcess.h"

namespace lmamd {

// ==================================================================================================
// ModelImporter stand-in: ASCII PLY (vertex x y z [...], face lists), triangulated as fans
// ==================================================================================================
bool load_ply_ascii(const std::string& path, Mesh& mesh, std::string* err) {
    std::ifstream f(path);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::string line;
    size_t nv = 0, nf = 0;
    int vprops = 0;
    bool in_vertex = false, ascii = false;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") { std::string fmt; ss >> fmt; ascii = fmt == "ascii"; }
        else if (tok == "element") {
            std::string what; size_t n; ss >> what >> n;
            in_vertex = what == "vertex";
            if (what == "vertex") nv = n; else if (what == "face") nf = n;
        } else if (tok == "property" && in_vertex) ++vprops;
        else if (tok == "end_header") break;
    }
    if (!ascii || nv == 0 || vprops < 3) { if (err) *err = "not an ASCII PLY with x y z vertices"; return false; }
    mesh.vertices.resize(nv);
    for (size_t i = 0; i < nv; ++i) {
        if (!std::getline(f, line)) { if (err) *err = "truncated PLY"; return false; }
        std::istringstream ss(line);
        ss >> mesh.vertices[i].x >> mesh.vertices[i].y >> mesh.vertices[i].z;
    }
    mesh.indices.clear();
    for (size_t i = 0; i < nf; ++i) {
        if (!std::getline(f, line)) { if (err) *err = "truncated PLY"; return false; }
        std::istringstream ss(line);
        int n; ss >> n;
        std::vector<uint32_t> idx((size_t)std::max(n, 0));
        for (int k = 0; k < n; ++k) ss >> idx[(size_t)k];
        for (int k = 1; k + 1 < n; ++k) { mesh.indices.push_back(idx[0]); mesh.indices.push_back(idx[(size_t)k]); mesh.indices.push_back(idx[(size_t)k + 1]); }
    }
    return true;
}

// ==================================================================================================
// Viewpoint sampling.  DERIVED FROM /root/reference/src/CameraViewPoints.cpp:11-225 in WHAT it produces -- the same
// viewpoints in the same order, because the order of getVertices() decides the template ids of a bank
// (TemplateGenerator.cpp:41-62) and the ids are what linemod_tempPosFile.bin indexes -- but built its own way: the
// sphere is an edge-keyed midpoint refinement (one map lookup per edge instead of the reference's linear duplicate
// search over all earlier vertices), the arc and the symmetry filter are small free functions.  Only the two public
// entry points keep the reference's names (the API TemplateGenerator calls).  The sequences are pinned by
// tests/test_viewpoints.py (tests/golden/viewpoints.txt).
// ==================================================================================================
namespace {

// the icosahedron of the reference (CameraViewPoints.cpp:84-124): vertex i = signs/axes below with
// a = r / sqrt(phi^2 + 1), b = a * phi; faces in the reference's order (both are data: the order is the contract)
const int8_t kIcoVertex[12][3] = {{-1, 0, 2}, {1, 0, 2}, {-1, 0, -2}, {1, 0, -2}, {0, 2, 1}, {0, 2, -1},
                                  {0, -2, 1}, {0, -2, -1}, {2, 1, 0}, {-2, 1, 0}, {2, -1, 0}, {-2, -1, 0}};   // 1 = a, 2 = b
const uint8_t kIcoFace[20][3] = {{0, 4, 1}, {0, 9, 4}, {9, 5, 4}, {4, 5, 8}, {4, 8, 1}, {8, 10, 1}, {8, 3, 10},
                                 {5, 3, 8}, {5, 2, 3}, {2, 7, 3}, {7, 10, 3}, {7, 6, 10}, {7, 11, 6}, {11, 0, 6},
                                 {0, 1, 6}, {6, 1, 10}, {9, 0, 11}, {9, 11, 2}, {9, 2, 5}, {7, 2, 11}};

struct Tri { uint32_t v[3]; };

// Midpoint of an edge pushed out to the sphere, created once per undirected edge.  (p + q) / 2 is commutative in
// floating point, so keying by the edge finds exactly the vertices the reference finds by comparing coordinates.
struct Refiner {
    std::vector<Vec3>& pts;
    float radius;
    uint32_t midpoint(uint32_t p, uint32_t q) {
        const uint64_t key = p < q ? ((uint64_t)p << 32) | q : ((uint64_t)q << 32) | p;
        for (const auto& e : cache_bucket(key)) if (e.first == key) return e.second;
        Vec3 m{(pts[p].x + pts[q].x) / 2, (pts[p].y + pts[q].y) / 2, (pts[p].z + pts[q].z) / 2};
        const float k = std::sqrt(m.x * m.x + m.y * m.y + m.z * m.z) / radius;   // CameraViewPoints.cpp:216-225: divide, not multiply
        m.x /= k; m.y /= k; m.z /= k;
        pts.push_back(m);
        const uint32_t id = (uint32_t)pts.size() - 1;
        cache_bucket(key).push_back({key, id});
        return id;
    }
    std::vector<std::vector<std::pair<uint64_t, uint32_t>>> buckets = std::vector<std::vector<std::pair<uint64_t, uint32_t>>>(1024);
    std::vector<std::pair<uint64_t, uint32_t>>& cache_bucket(uint64_t key) { return buckets[(key * 0x9E3779B97F4A7C15ull) >> 54]; }
};

void sphere_points(float radius, int subdivisions, std::vector<Vec3>& pts) {
    const float phi = 1.61803398875f;
    const float a = std::sqrt((radius * radius) / (phi * phi + 1)), b = a * phi;
    const float mag[3] = {0.0f, a, b};
    pts.clear();
    for (const auto& v : kIcoVertex) {
        Vec3 p;
        p.x = (v[0] < 0 ? -1.0f : 1.0f) * mag[v[0] < 0 ? -v[0] : v[0]];
        p.y = (v[1] < 0 ? -1.0f : 1.0f) * mag[v[1] < 0 ? -v[1] : v[1]];
        p.z = (v[2] < 0 ? -1.0f : 1.0f) * mag[v[2] < 0 ? -v[2] : v[2]];
        pts.push_back(p);
    }
    std::vector<Tri> tris;
    for (const auto& f : kIcoFace) tris.push_back(Tri{{f[0], f[1], f[2]}});
    Refiner ref{pts, radius};
    for (int level = 0; level < subdivisions; ++level) {
        const size_t n = tris.size();
        for (size_t i = 0; i < n; ++i) {
            const Tri t = tris[i];
            // the reference visits the edges as (a,b), (c,b), (a,c): that order numbers the new vertices
            const uint32_t ab = ref.midpoint(t.v[0], t.v[1]);
            const uint32_t cb = ref.midpoint(t.v[2], t.v[1]);
            const uint32_t ac = ref.midpoint(t.v[0], t.v[2]);
            tris.push_back(Tri{{t.v[0], ab, ac}});
            tris.push_back(Tri{{t.v[1], ab, cb}});
            tris.push_back(Tri{{t.v[2], cb, ac}});
            tris[i] = Tri{{ab, cb, ac}};
        }
    }
}

// rotationally symmetric parts: one meridian in the y/z plane, every `60 / 2^n` degrees with the step truncated to
// whole degrees by the reference's uint16_t loop counter (CameraViewPoints.cpp:75-82: 7.5 -> 7)
void arc_points(float radius, int subdivisions, std::vector<Vec3>& pts) {
    pts.clear();
    const double step = 60 / std::pow(2, subdivisions);
    const double PI = 3.1415926535897932384626433832795;
    for (uint16_t deg = 0; deg < 360; deg = (uint16_t)(deg + step)) {
        pts.push_back(Vec3{0.0f, (float)(std::sin(deg * PI / 180.0f) * radius), (float)(std::cos(deg * PI / 180.0f) * radius)});
        if (step < 1.0) break;   // the reference's counter would never advance here
    }
}

}  // namespace

void CameraViewPoints::createCameraViewPoints(float in_radius, uint8_t in_subdivions) {
    std::vector<Vec3> all;
    if (modProps.rotationallySymmetrical) arc_points(in_radius, in_subdivions, all);
    else sphere_points(in_radius, in_subdivions, all);
    // symmetry planes (models/<name>.yml "planes of symmetry", CameraViewPoints.cpp:34-52): a viewpoint survives
    // unless one of its coordinates, multiplied by the plane flag, is negative
    vertices.clear();
    const Vec3 s = modProps.planesOfSymmetry;
    for (const Vec3& v : all)
        if (!(v.x * s.x < 0 || v.y * s.y < 0 || v.z * s.z < 0)) vertices.push_back(v);
}

// ==================================================================================================
// SoftRender
// ==================================================================================================
SoftRender::SoftRender(const CameraParameters& cam) : width(cam.videoWidth), height(cam.videoHeight) {
    // glm::perspective(radians(360/pi * atan(h / 2fy)), w/h, 100, 10000)  (OpenglRender.cpp:9-11)
    const float fovy = 2.0f * std::atan((float)height / (2 * cam.fy));
    const float aspect = (float)width / (float)height, zn = 100.0f, zf = 10000.0f;
    const float t = std::tan(fovy / 2.0f);
    std::memset(proj, 0, sizeof(proj));
    proj[0][0] = 1.0f / (aspect * t);
    proj[1][1] = 1.0f / t;
    proj[2][2] = -(zf + zn) / (zf - zn);
    proj[2][3] = -1.0f;
    proj[3][2] = -(2.0f * zf * zn) / (zf - zn);
}

void SoftRender::render(const Mesh& mesh, Vec3 cam, std::vector<uint8_t>& bgr, std::vector<uint16_t>& depth) const {
    // translateCam (:334-345)
    if (cam.x == 0 && cam.z == 0) { cam.x = 0.000001f; cam.z = 0.000001f; }
    const Mat4 view = lookAt(cam, Vec3{0, 0, 0}, Vec3{0, 1, 0});
    render_view(mesh, view.m, bgr, depth);
}

void SoftRender::render_view(const Mesh& mesh, const float view_m[4][4], std::vector<uint8_t>& bgr, std::vector<uint16_t>& depth) const {
    const int W = width, H = height;
    bgr.assign((size_t)W * H * 3, 0);
    depth.assign((size_t)W * H, 0);
    std::vector<float> zbuf((size_t)W * H, 1.0f);   // glClear depth = 1, GL_LESS
    Mat4 view;
    std::memcpy(view.m, view_m, sizeof(view.m));
    Mat4 P;
    std::memcpy(P.m, proj, sizeof(proj));
    Mat4 vp = mul(P, view);   // the reference computes modelMat but never applies it (`viewProj = projection * view`)
    struct SV { float x, y, z, w; };
    std::vector<SV> sv(mesh.vertices.size());
    for (size_t i = 0; i < mesh.vertices.size(); ++i) {
        const Vec3& v = mesh.vertices[i];
        float cx = vp.m[0][0] * v.x + vp.m[1][0] * v.y + vp.m[2][0] * v.z + vp.m[3][0];
        float cy = vp.m[0][1] * v.x + vp.m[1][1] * v.y + vp.m[2][1] * v.z + vp.m[3][1];
        float cz = vp.m[0][2] * v.x + vp.m[1][2] * v.y + vp.m[2][2] * v.z + vp.m[3][2];
        float cw = vp.m[0][3] * v.x + vp.m[1][3] * v.y + vp.m[2][3] * v.z + vp.m[3][3];
        SV s;
        s.w = cw;
        if (cw > 1e-6f) {
            s.x = (cx / cw * 0.5f + 0.5f) * W;      // window coordinates, origin bottom-left
            s.y = (cy / cw * 0.5f + 0.5f) * H;
            s.z = cz / cw * 0.5f + 0.5f;
        } else { s.x = s.y = s.z = 0; }
        sv[i] = s;
    }
    const float zn = 100.0f, zf = 10000.0f;
    for (size_t t = 0; t + 2 < mesh.indices.size(); t += 3) {
        const SV& a = sv[mesh.indices[t]];
        const SV& b = sv[mesh.indices[t + 1]];
        const SV& c = sv[mesh.indices[t + 2]];
        if (a.w <= 1e-6f || b.w <= 1e-6f || c.w <= 1e-6f) continue;   // behind the camera (objects sit at >= 500 mm)
        float area = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
        if (area == 0) continue;
        int x0 = std::max(0, (int)std::floor(std::min(a.x, std::min(b.x, c.x))));
        int x1 = std::min(W - 1, (int)std::ceil(std::max(a.x, std::max(b.x, c.x))));
        int y0 = std::max(0, (int)std::floor(std::min(a.y, std::min(b.y, c.y))));
        int y1 = std::min(H - 1, (int)std::ceil(std::max(a.y, std::max(b.y, c.y))));
        const float inv = 1.0f / area;
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) {
                float fx = px + 0.5f, fy = py + 0.5f;
                float w0 = ((b.x - fx) * (c.y - fy) - (b.y - fy) * (c.x - fx)) * inv;
                float w1 = ((c.x - fx) * (a.y - fy) - (c.y - fy) * (a.x - fx)) * inv;
                float w2 = 1.0f - w0 - w1;
                if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                float z = w0 * a.z + w1 * b.z + w2 * c.z;   // window z is linear in screen space
                if (z < 0 || z > 1) continue;               // near / far clip
                size_t o = (size_t)(H - 1 - py) * W + px;   // glReadPixels rows are bottom-up; the reference flips
                if (!(z < zbuf[o])) continue;
                zbuf[o] = z;
                bgr[o * 3] = bgr[o * 3 + 1] = bgr[o * 3 + 2] = 255;   // white unlit mesh (ModelImporter: colour (1,1,1))
                float ndc = z * 2.0f - 1.0f;                           // shader/depth.fs LinearizeDepth
                float lin = (2.0f * zn * zf) / (zf + zn - ndc * (zf - zn));
                float v = lin / zf / 6.5535f;                          // R16 unorm: v * 65535 = lin (mm)
                long q = std::lrint(v * 65535.0f);
                depth[o] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
            }
    }
}

// ==================================================================================================
// cv::getRotationMatrix2D + cv::warpAffine (INTER_LINEAR, BORDER_CONSTANT 0), fixed-point coordinates
// ==================================================================================================
namespace {
struct InvAffine { double m[6]; };
InvAffine inverse_rotation(int w, int h, float angleDegrees) {
    // getRotationMatrix2D(Point2f(w / 2, h / 2), angle, 1.0)  (HighLevelLinemod.cpp:331-332: integer halves)
    double angle = angleDegrees * 3.14159265358979323846 / 180.0;
    double alpha = std::cos(angle), beta = std::sin(angle);
    double cx = (double)(w / 2), cy = (double)(h / 2);
    double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
    // warpAffine inverts the matrix (no WARP_INVERSE_MAP)
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    InvAffine r;
    std::memcpy(r.m, M, sizeof(M));
    return r;
}
// source coordinates in 1/32 pixel for destination (x, y), as warpAffine computes them
inline void src_coord(const InvAffine& A, int x, int y, int* X, int* Y) {
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / 32 / 2;
    int adelta = (int)std::lrint(A.m[0] * x * AB_SCALE), bdelta = (int)std::lrint(A.m[3] * x * AB_SCALE);
    int X0 = (int)std::lrint((A.m[1] * y + A.m[2]) * AB_SCALE) + round_delta;
    int Y0 = (int)std::lrint((A.m[4] * y + A.m[5]) * AB_SCALE) + round_delta;
    *X = (X0 + adelta) >> (AB_BITS - 5);
    *Y = (Y0 + bdelta) >> (AB_BITS - 5);
}
}  // namespace

void warp_rotate_u8(const uint8_t* src, int w, int h, int ch, float angleDegrees, std::vector<uint8_t>& dst) {
    dst.assign((size_t)w * h * ch, 0);
    const InvAffine A = inverse_rotation(w, h, angleDegrees);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int X, Y;
            src_coord(A, x, y, &X, &Y);
            int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
            // 15-bit integer weights like OpenCV's BilinearTab_i
            int w00 = (int)std::lrint((32 - fx) * (32 - fy) * 32.0), w01 = (int)std::lrint(fx * (32 - fy) * 32.0);
            int w10 = (int)std::lrint((32 - fx) * fy * 32.0), w11 = (int)std::lrint(fx * fy * 32.0);
            for (int c = 0; c < ch; ++c) {
                auto at = [&](int yy, int xx) -> int { return (xx < 0 || yy < 0 || xx >= w || yy >= h) ? 0 : src[((size_t)yy * w + xx) * ch + c]; };
                int v = at(sy, sx) * w00 + at(sy, sx + 1) * w01 + at(sy + 1, sx) * w10 + at(sy + 1, sx + 1) * w11;
                dst[((size_t)y * w + x) * ch + c] = (uint8_t)((v + (1 << 14)) >> 15);
            }
        }
}

void warp_rotate_u16(const uint16_t* src, int w, int h, float angleDegrees, std::vector<uint16_t>& dst) {
    dst.assign((size_t)w * h, 0);
    const InvAffine A = inverse_rotation(w, h, angleDegrees);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int X, Y;
            src_coord(A, x, y, &X, &Y);
            int sx = X >> 5, sy = Y >> 5;
            float fx = (X & 31) / 32.0f, fy = (Y & 31) / 32.0f;
            auto at = [&](int yy, int xx) -> float { return (xx < 0 || yy < 0 || xx >= w || yy >= h) ? 0.f : (float)src[(size_t)yy * w + xx]; };
            float v = at(sy, sx) * (1 - fx) * (1 - fy) + at(sy, sx + 1) * fx * (1 - fy) + at(sy + 1, sx) * (1 - fx) * fy +
                      at(sy + 1, sx + 1) * fx * fy;
            long q = std::lrint(v);
            dst[(size_t)y * w + x] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
        }
}

// ==================================================================================================
// TemplateGenerator::run for one model (/root/reference/src/TemplateGenerator.cpp:41-62)
// ==================================================================================================
int generate_templates(HighLevelLineMOD& line, const SoftRender& render, const Mesh& mesh, const std::string& modelName,
                       const SymmetryProperties& sym, const GeneratorSettings& gs) {
    CameraViewPoints cams;
    cams.setModelProperties(sym);
    const uint32_t before = line.getNumTemplates();
    std::vector<uint8_t> bgr;
    std::vector<uint16_t> depth;
    for (uint32_t radius = gs.startDistance; radius <= gs.endDistance; radius += gs.stepSize) {
        cams.createCameraViewPoints((float)radius, gs.subdivisions);
        for (const Vec3& cam : cams.getVertices()) {
            render.render(mesh, cam, bgr, depth);
            std::vector<Image> imgs(2);
            imgs[0].data = bgr.data(); imgs[0].width = render.width; imgs[0].height = render.height; imgs[0].type = 0;
            imgs[1].data = depth.data(); imgs[1].width = render.width; imgs[1].height = render.height; imgs[1].type = 1;
            line.addTemplate(imgs, modelName, cam);   // failures print and are skipped, like the reference (:59)
        }
    }
    line.pushBackTemplates();
    return (int)(line.getNumTemplates() - before);
}

}  // namespace lmamd
