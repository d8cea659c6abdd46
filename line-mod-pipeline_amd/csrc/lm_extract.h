// lm_extract.h -- host half of Detector::addTemplate (reference call site
// /root/reference/src/HighLevelLinemod.cpp:93): the quantised images come from the GPU kernels,
// feature selection (an offline, inherently serial greedy pick) runs here on the host.
#pragma once
#include <string>
#include <vector>

#include "lm_host.h"

namespace lmh {

// One pyramid level of the quantised sources of a template image, as read back from the device.
struct ExtractLevel {
    int w = 0, h = 0;
    std::vector<u8> color_q;       // ColorGradient quantised orientations (one-hot)
    std::vector<float> color_mag;  // squared gradient magnitude of the selected channel
    std::vector<u8> depth_q;       // DepthNormal quantised normals (empty for a colour-only detector)
    std::vector<u8> mask;          // object mask at this level (empty = no mask)
};

// Fills tp ([level*M + modality]) from the per-level images; returns false when some level yields
// fewer candidates than requested features (upstream addTemplate then returns -1).
bool extract_pyramid(const std::vector<ExtractLevel>& levels, const lm_config& cfg, TemplatePyramid& tp);

// cropTemplates: bounding box over all levels/modalities in level-0 units, features made relative.
lm_rect crop_templates(TemplatePyramid& tp);

}  // namespace lmh
