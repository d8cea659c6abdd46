// End-to-end run on the reference's own benchmark data: render templates of models/lagergehaeuse.ply with the
// software stand-ins (viewpoint arc x radii x in-plane rotations, like TemplateGenerator::run), then detect the
// part in benchmark/img0.png + depth0.png the way PoseDetection::detect does and print the poses, to be
// compared with benchmark/pose0.yml.
// usage: pose_e2e <mesh.bin> <bgr.raw> <depth.raw> <color_only 0|1> <startDist> <endDist> <threshold>
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char** argv) {
    if (argc < 8) return 2;
    std::vector<char> mb = slurp(argv[1]);
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(mb.data());
    uint32_t nv = hdr[0], nf = hdr[1];
    const float* v = reinterpret_cast<const float*>(mb.data() + 8);
    const int32_t* f = reinterpret_cast<const int32_t*>(mb.data() + 8 + (size_t)nv * 12);
    Mesh mesh;
    mesh.vertices.resize(nv);
    for (uint32_t i = 0; i < nv; ++i) mesh.vertices[i] = Vec3{v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    mesh.indices.assign(f, f + (size_t)nf * 3);

    CameraParameters cam;   // linemod_settings.yml
    cam.fx = 1044.87f; cam.fy = 1045.69141f; cam.cx = 320; cam.cy = 240; cam.videoWidth = 640; cam.videoHeight = 480;
    TemplateGenerationSettings ts;
    ts.onlyUseColorModality = std::atoi(argv[4]) != 0;
    ts.detectorThreshold = (float)std::atof(argv[7]);
    HighLevelLineMOD line(cam, ts);
    SoftRender render(cam);
    SymmetryProperties sym;   // models/lagergehaeuse.yml
    sym.rotationallySymmetrical = true; sym.planesOfSymmetry = Vec3{1, 1, 1};
    GeneratorSettings gs;
    gs.startDistance = (uint16_t)std::atoi(argv[5]); gs.endDistance = (uint16_t)std::atoi(argv[6]); gs.stepSize = 50; gs.subdivisions = 3;
    int n = generate_templates(line, render, mesh, "lagergehaeuse.ply", sym, gs);
    std::printf("templates %d\n", n);
    double lo[3] = {0, 0, 0}, hi[3] = {255, 150, 255};
    line.setColorRange(0, lo, hi);

    std::vector<char> bgr = slurp(argv[2]), depth = slurp(argv[3]);
    std::vector<Image> imgs(2);
    imgs[0].data = bgr.data(); imgs[0].width = 640; imgs[0].height = 480;
    imgs[1].data = depth.data(); imgs[1].width = 640; imgs[1].height = 480; imgs[1].type = 1;
    // PoseDetection::detect shifts by (w/2 - cx, h/2 - cy) = (0, 0) for the shipped camera file
    bool found = line.detectTemplate(imgs, 0);
    std::printf("found %d matches %zu error '%s'\n", found ? 1 : 0, line.getMatches().size(), line.lastError().c_str());
    for (size_t i = 0; i < line.getMatches().size() && i < 5; ++i) {
        const lm_match_t& m = line.getMatches()[i];
        std::printf("match %d %d %.5g %d\n", m.x, m.y, m.similarity, m.template_id);
    }
    auto poses = line.getObjectPoses();
    std::printf("groups %zu\n", poses.size());
    for (auto& g : poses)
        for (auto& p : g) {
            Mat4 R = toMat4(p.quaternions);
            std::printf("pose t %.3f %.3f %.3f q %.5f %.5f %.5f %.5f bb %d %d %d %d axisY %.4f %.4f %.4f\n", p.translation.x,
                        p.translation.y, p.translation.z, p.quaternions.w, p.quaternions.x, p.quaternions.y, p.quaternions.z,
                        p.boundingBox.x, p.boundingBox.y, p.boundingBox.width, p.boundingBox.height, R.m[1][0], R.m[1][1], R.m[1][2]);
        }
    return 0;
}
