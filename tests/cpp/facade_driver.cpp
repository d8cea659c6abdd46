// Test driver for the C++ facade (line-mod-pipeline_amd/host/HighLevelLinemod.h): the way the reference's
// PoseDetection uses HighLevelLineMOD (/root/reference/src/PoseDetection.cpp:17,66): readLinemod(), then
// detectTemplate(imgs, classIndex).  Usage: facade_driver <color_only 0|1> <bgr.raw> <depth.raw> <threshold>
// Run from a directory that holds linemod_templates.yml.gz; a linemod_settings.yml there is read the way the
// reference's utility.cpp does (camera + settings, the command line then overrides modality and threshold).
// Prints the class table and the match list.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../line-mod-pipeline_amd/host/HighLevelLinemod.h"

static std::vector<char> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    lmamd::CameraParameters cam;
    cam.fx = 1044.87f; cam.fy = 1045.69141f; cam.cx = 320; cam.cy = 240; cam.videoWidth = 640; cam.videoHeight = 480;
    lmamd::TemplateGenerationSettings ts;
    if (lmamd::readSettings("linemod_settings.yml", cam, ts))
        std::printf("settings %u %u %.5f %d %d %d %u %.1f %s\n", (unsigned)cam.videoWidth, (unsigned)cam.videoHeight, (double)cam.fy,
                    (int)ts.angleStart, (int)ts.angleStop, (int)ts.angleStep, (unsigned)ts.stepSize, (double)ts.depthOffset,
                    ts.modelFolder.c_str());
    ts.onlyUseColorModality = std::atoi(argv[1]) != 0;
    ts.detectorThreshold = (float)std::atof(argv[4]);
    lmamd::HighLevelLineMOD line(cam, ts);
    line.readLinemod();
    std::printf("classes %u templates %u\n", (unsigned)line.getNumClasses(), (unsigned)line.getNumTemplates());
    for (auto& id : line.getClassIds()) std::printf("class %s\n", id.c_str());
    std::vector<char> bgr = slurp(argv[2]), depth = slurp(argv[3]);
    if (bgr.size() != 640u * 480 * 3 || depth.size() != 640u * 480 * 2) return 3;
    std::vector<lmamd::Image> imgs(2);
    imgs[0].data = bgr.data(); imgs[0].width = 640; imgs[0].height = 480; imgs[0].type = 0;
    imgs[1].data = depth.data(); imgs[1].width = 640; imgs[1].height = 480; imgs[1].type = 1;
    bool found = line.detectTemplate(imgs, 0);
    std::printf("found %d error '%s'\n", found ? 1 : 0, line.lastError().c_str());
    for (const lm_match_t& m : line.getMatches())
        std::printf("match %d %d %.9g %d %d\n", m.x, m.y, m.similarity, m.template_id, m.class_idx);
    return 0;
}
