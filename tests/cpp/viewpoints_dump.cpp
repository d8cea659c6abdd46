// Prints the viewpoint sequences of lmamd::CameraViewPoints (count + every vertex as hex floats) for the
// configurations tests/test_viewpoints.py pins: the order of the viewpoints decides the template ids of a bank
// (TemplateGenerator.cpp:41-62 walks getVertices() in order), so it must never change.
#include <cstdio>
#include <cstring>

#include "../../line-mod-pipeline_amd/host/TemplateGenerator.h"

using namespace lmamd;

static void dump(const char* name, bool rotsym, Vec3 planes, float radius, int subdiv) {
    CameraViewPoints cv;
    SymmetryProperties sp;
    sp.rotationallySymmetrical = rotsym;
    sp.planesOfSymmetry = planes;
    cv.setModelProperties(sp);
    cv.createCameraViewPoints(radius, (uint8_t)subdiv);
    const std::vector<Vec3>& v = cv.getVertices();
    std::printf("%s %zu", name, v.size());
    for (const Vec3& p : v) {
        uint32_t b[3];
        std::memcpy(b, &p, 12);
        std::printf(" %08x%08x%08x", b[0], b[1], b[2]);
    }
    std::printf("\n");
}

int main() {
    dump("shipped_model_r500_s3", true, Vec3{1, 1, 1}, 500.f, 3);     // models/lagergehaeuse.yml + linemod_settings.yml
    dump("shipped_model_r1200_s2", true, Vec3{1, 1, 1}, 1200.f, 2);
    dump("rotsym_noplanes_r700_s1", true, Vec3{0, 0, 0}, 700.f, 1);
    dump("sphere_r500_s0", false, Vec3{0, 0, 0}, 500.f, 0);
    dump("sphere_r500_s1", false, Vec3{0, 0, 0}, 500.f, 1);
    dump("sphere_r850_s2", false, Vec3{0, 0, 0}, 850.f, 2);          // the 162 viewpoints of BASELINE config 4
    dump("sphere_r500_s3", false, Vec3{0, 0, 0}, 500.f, 3);
    dump("octant_r600_s2", false, Vec3{1, 1, 1}, 600.f, 2);
    dump("halfspace_r600_s2", false, Vec3{0, 0, 1}, 600.f, 2);
    return 0;
}
