// End-to-end self test of the facade on a synthetic "rendered" object (white on black, flat depth), the way
// the reference's TemplateGenerator + PoseDetection would use HighLevelLineMOD: addTemplate ->
// pushBackTemplates -> detectTemplate (GPU match + host post-processing) -> getObjectPoses.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../line-mod-pipeline_amd/host/HighLevelLinemod.h"
#include "../../line-mod-pipeline_amd/host/PostProcess.h"

using namespace lmamd;

static void draw(std::vector<uint8_t>& bgr, std::vector<uint16_t>& depth, int ox, int oy) {
    const int W = 640, H = 480;
    bgr.assign((size_t)W * H * 3, 0);
    depth.assign((size_t)W * H, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u = (x - ox) * 0.94f + (y - oy) * 0.34f, v = -(x - ox) * 0.34f + (y - oy) * 0.94f;   // rotated frame
            bool in = std::fabs(u) < 70 && std::fabs(v) < 45 && !(u > 20 && v > 5);                     // L-shaped plate
            bool hole = (u + 30) * (u + 30) + (v + 10) * (v + 10) < 15 * 15;
            if (in && !hole) {
                bgr[((size_t)y * W + x) * 3 + 0] = 235; bgr[((size_t)y * W + x) * 3 + 1] = 235; bgr[((size_t)y * W + x) * 3 + 2] = 235;
                depth[(size_t)y * W + x] = 700;
            }
        }
}

int main() {
    CameraParameters cam;
    cam.fx = 1044.87f; cam.fy = 1045.69141f; cam.cx = 320; cam.cy = 240;
    TemplateGenerationSettings ts;
    ts.detectorThreshold = 85.f;
    ts.angleStart = 0; ts.angleStop = 0;   // a single, unrotated template
    HighLevelLineMOD line(cam, ts);
    std::vector<uint8_t> bgr; std::vector<uint16_t> depth;
    draw(bgr, depth, 320, 240);
    std::vector<Image> imgs(2);
    imgs[0].data = bgr.data(); imgs[0].width = 640; imgs[0].height = 480;
    imgs[1].data = depth.data(); imgs[1].width = 640; imgs[1].height = 480; imgs[1].type = 1;
    if (!line.addTemplate(imgs, "plate.ply", Vec3{0, 0, 700})) { std::printf("addTemplate failed: %s\n", line.lastError().c_str()); return 1; }
    line.pushBackTemplates();
    double lo[3] = {0, 0, 0}, hi[3] = {255, 150, 255};
    line.setColorRange(0, lo, hi);
    std::printf("templates %u\n", (unsigned)line.getNumTemplates());
    draw(bgr, depth, 360, 270);                       // the object moved by (+40, +30) px in the scene
    bool found = line.detectTemplate(imgs, 0);
    std::printf("found %d matches %zu error '%s'\n", found ? 1 : 0, line.getMatches().size(), line.lastError().c_str());
    if (!line.getMatches().empty()) {
        const lm_match_t& m = line.getMatches()[0];
        std::printf("best %d %d %.6g\n", m.x, m.y, m.similarity);
    }
    auto poses = line.getObjectPoses();
    std::printf("groups %zu\n", poses.size());
    for (auto& g : poses)
        for (auto& p : g)
            std::printf("pose t %.4f %.4f %.4f q %.5f %.5f %.5f %.5f bb %d %d %d %d\n", p.translation.x, p.translation.y,
                        p.translation.z, p.quaternions.w, p.quaternions.x, p.quaternions.y, p.quaternions.z,
                        p.boundingBox.x, p.boundingBox.y, p.boundingBox.width, p.boundingBox.height);
    line.writeLinemod();
    HighLevelLineMOD again(cam, ts);
    again.readLinemod();
    std::printf("reloaded classes %u templates %u\n", (unsigned)again.getNumClasses(), (unsigned)again.getNumTemplates());
    again.setColorRange(0, lo, hi);
    bool found2 = again.detectTemplate(imgs, 0);
    std::printf("reloaded found %d groups %zu\n", found2 ? 1 : 0, again.getObjectPoses().size());
    return 0;
}
