// TemplateGenerator.h -- offline template generation without OpenGL / assimp / OpenCV
// (SURVEY.md section 8f-3 and 8f-4).  Stand-ins, with the reference's names and call structure, for
//   ModelImporter      (/root/reference/src/ModelImporter.cpp:13-82, assimp)   -> load_ply_ascii
//   CameraViewPoints   (/root/reference/src/CameraViewPoints.cpp)              -> CameraViewPoints
//   OpenGLRender       (/root/reference/src/OpenglRender.cpp:9-11,49-141,334-345, shader/depth.fs)
//                                                                               -> SoftRender (z-buffer rasteriser)
//   cv::getRotationMatrix2D + cv::warpAffine (HighLevelLinemod.cpp:81-91,327-334) -> warp_rotate_*
//   TemplateGenerator::run (/root/reference/src/TemplateGenerator.cpp:41-62)   -> TemplateGenerator::run
// All of it is offline host code (not on the measured path) and, like the oracle, PARITY UNPINNED against
// the real OpenGL rasteriser / OpenCV resampler: sub-pixel coverage and bilinear rounding may differ.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "HighLevelLinemod.h"

namespace lmamd {

struct Mesh {
    std::vector<Vec3> vertices;
    std::vector<uint32_t> indices;   // triangles
};
bool load_ply_ascii(const std::string& path, Mesh& mesh, std::string* err = nullptr);

// models/<name>.yml: symmetry description used to prune the viewpoint sphere (CameraViewPoints.cpp:34-67)
struct SymmetryProperties {
    bool rotationallySymmetrical = false;
    Vec3 planesOfSymmetry{0, 0, 0};
};

// Viewpoints around the model: an icosphere (12 * 4^n-ish vertices) or, for rotationally symmetric parts, one meridian
// arc; symmetry planes prune it.  Same viewpoints in the same order as the reference's CameraViewPoints (the order
// numbers the templates); the two entry points TemplateGenerator calls keep the reference's names.
class CameraViewPoints {
public:
    void setModelProperties(const SymmetryProperties& p) { modProps = p; }
    void createCameraViewPoints(float in_radius, uint8_t in_subdivions);   // CameraViewPoints.cpp:11-32
    std::vector<Vec3>& getVertices() { return vertices; }

private:
    std::vector<Vec3> vertices;
    SymmetryProperties modProps;
};

// Software stand-in for OpenGLRender: perspective(fovy = 2 atan(h / 2fy), w/h, 100, 10000) * lookAt(cam, 0, +y),
// white unlit mesh, depth image = linear eye depth in millimetres (shader/depth.fs), rows flipped like the
// reference's glReadPixels + flip.  Background colour 0 / depth 0.
class SoftRender {
public:
    explicit SoftRender(const CameraParameters& cam);
    // renderColorToFrontBuff + renderDepthToFrontBuff (camera-position overloads, :49-69, :97-117)
    void render(const Mesh& mesh, Vec3 camPosition, std::vector<uint8_t>& bgr, std::vector<uint16_t>& depth) const;
    // the same rasteriser under a given view matrix (column-major like glm): renderDepthToFrontBuff(modelIndice, rotMat, traVec) of the reference's
    // Benchmark (:116-141) builds one from a pose; the Hodan-error test renders the ground-truth and the estimated pose with it
    void render_view(const Mesh& mesh, const float view[4][4], std::vector<uint8_t>& bgr, std::vector<uint16_t>& depth) const;
    int width, height;

private:
    float proj[4][4];   // column-major
};

// cv::warpAffine(src, dst, cv::getRotationMatrix2D(center, angleDegrees, 1.0), size): bilinear, constant 0 border
void warp_rotate_u8(const uint8_t* src, int w, int h, int channels, float angleDegrees, std::vector<uint8_t>& dst);
void warp_rotate_u16(const uint16_t* src, int w, int h, float angleDegrees, std::vector<uint16_t>& dst);

// TemplateGenerator::run for one model: radii startDistance..endDistance step stepSize, every viewpoint,
// every in-plane rotation (HighLevelLineMOD::addTemplate sweeps them).  Returns the number of templates added.
struct GeneratorSettings {
    uint16_t startDistance = 500, endDistance = 1200, stepSize = 50;
    uint8_t subdivisions = 3;
};
int generate_templates(HighLevelLineMOD& line, const SoftRender& render, const Mesh& mesh, const std::string& modelName,
                       const SymmetryProperties& sym, const GeneratorSettings& gs);

}  // namespace lmamd
