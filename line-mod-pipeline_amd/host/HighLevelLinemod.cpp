// HighLevelLinemod.cpp -- see HighLevelLinemod.h.  Plain C++17, links against liblinemod_hip.so only.
#include "HighLevelLinemod.h"

#include <cstdio>
#include <stdexcept>

namespace lmamd {

HighLevelLineMOD::HighLevelLineMOD(CameraParameters const& cam, TemplateGenerationSettings const& ts)
    : onlyColorModality(ts.onlyUseColorModality),
      videoWidth(cam.videoWidth),
      videoHeight(cam.videoHeight),
      detectorThreshold(ts.detectorThreshold) {
    lm_config cfg;
    lm_default_config(&cfg, onlyColorModality ? 1 : 0, videoWidth, videoHeight);  // T = {2,8} / {5,8}
    cfg.device = ts.device;
    cfg.shard_rank = ts.shardRank;
    cfg.shard_size = ts.shardSize;
    if (lm_create(&cfg, &detector) != LM_OK) throw std::runtime_error(lm_last_error());
}

HighLevelLineMOD::~HighLevelLineMOD() { lm_destroy(detector); }  // detector.release(), :50

std::vector<std::string> HighLevelLineMOD::getClassIds() {
    std::vector<std::string> ids;
    for (int i = 0; i < lm_num_classes(detector); ++i) ids.emplace_back(lm_class_id(detector, i));
    return ids;
}
uint16_t HighLevelLineMOD::getNumClasses() { return (uint16_t)lm_num_classes(detector); }
uint32_t HighLevelLineMOD::getNumTemplates() { return (uint32_t)lm_num_templates(detector); }

bool HighLevelLineMOD::detectTemplate(std::vector<Image>& in_imgs, uint16_t in_classNumber) {
    posesMultipleObj.clear();
    matches.clear();
    if (in_imgs.empty()) { error = "no images"; return false; }
    const Image& color = in_imgs[0];
    // a colour-only detector pops the depth image before match() and pushes it back afterwards (:146-156)
    const Image* depth = (!onlyColorModality && in_imgs.size() >= 2) ? &in_imgs[1] : nullptr;
    if (color.width != videoWidth || color.height != videoHeight) { error = "frame size differs from the detector's"; return false; }
    size_t cap = 4096, n = 0;
    for (;;) {
        matches.resize(cap);
        int rc = lm_match(detector, static_cast<const uint8_t*>(color.data), color.stride,
                          depth ? static_cast<const uint16_t*>(depth->data) : nullptr, depth ? depth->stride : 0,
                          detectorThreshold, in_classNumber, matches.data(), cap, &n);
        if (rc == LM_ERR_OVERFLOW && n > cap) { cap = n; continue; }  // the reference consumes ALL matches
        if (rc != LM_OK) { error = lm_last_error(); matches.clear(); return false; }
        break;
    }
    matches.resize(n);
    return !matches.empty();  // :157,187-189
}

void HighLevelLineMOD::writeLinemod() {
    if (lm_save_bank(detector, "linemod_templates.lmbk") != LM_OK) { error = lm_last_error(); std::printf("ERROR::%s\n", error.c_str()); }
}
void HighLevelLineMOD::readLinemod() {
    if (lm_load_bank(detector, "linemod_templates.lmbk") != LM_OK) { error = lm_last_error(); std::printf("ERROR::%s\n", error.c_str()); }
}

bool HighLevelLineMOD::addTemplate(std::vector<Image>& in_images, const std::string& in_modelName, Vec3) {
    if (in_images.size() < 2) { error = "addTemplate needs {colour, depth}"; return false; }
    const Image& color = in_images[0];
    const Image& depth = in_images[1];
    // mask = depth > 0 (threshold(in_images[1], mask, 1, 65535, THRESH_BINARY), :78-79; depth 1 mm counts as background)
    std::vector<uint8_t> mask((size_t)depth.width * depth.height);
    size_t dstride = depth.stride ? depth.stride : (size_t)depth.width * 2;
    for (int y = 0; y < depth.height; ++y) {
        const uint16_t* row = reinterpret_cast<const uint16_t*>(static_cast<const uint8_t*>(depth.data) + y * dstride);
        for (int x = 0; x < depth.width; ++x) mask[(size_t)y * depth.width + x] = row[x] > 1 ? 255 : 0;
    }
    int tid = -1;
    lm_rect bb;
    int rc = lm_add_template(detector, in_modelName.c_str(), static_cast<const uint8_t*>(color.data), color.stride,
                             onlyColorModality ? nullptr : static_cast<const uint16_t*>(depth.data), depth.stride,
                             mask.data(), 0, &tid, &bb);
    if (rc != LM_OK || tid < 0) {
        error = lm_last_error();
        std::printf("ERROR::Cant create Template\n");  // :99
        return false;
    }
    return true;
}

void HighLevelLineMOD::pushBackTemplates() {}  // per-template poses belong to post-processing (8f-1)

}  // namespace lmamd
