// PoseDetection.cpp -- see PoseDetection.h.
#include "PoseDetection.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "PostProcess.h"

namespace lmamd {

PoseDetection::PoseDetection(CameraParameters const& cam, TemplateGenerationSettings const& ts)
    : line(new HighLevelLineMOD(cam, ts)), camParams(cam), templateSettings(ts) {}

PoseDetection::~PoseDetection() { delete line; }

void PoseDetection::loadTemplates() {
    line->readLinemod();
    refreshClassIds();
    std::printf("Loaded with %d classes and %u templates\n", (int)line->getNumClasses(), (unsigned)line->getNumTemplates());
}

void PoseDetection::refreshClassIds() { ids = line->getClassIds(); }

uint16_t PoseDetection::findIndexInVector(std::string const& s, std::vector<std::string>& v) {
    return (uint16_t)std::distance(v.begin(), std::find(v.begin(), v.end(), s));
}

// translateImg(img, -cx + videoWidth / 2, -cy + videoHeight / 2) on clones of both images (:54-59): a pure integer
// shift (the offsets are ints, :192-197), zeros shifted in.  in_imgs[0] must be dense BGR, in_imgs[1] dense depth.
void PoseDetection::shiftFrame(const std::vector<Image>& in, Shifted& buf, std::vector<Image>& out) {
    const int ox = (int)(-camParams.cx + camParams.videoWidth / 2), oy = (int)(-camParams.cy + camParams.videoHeight / 2);
    out = in;
    const int w = in[0].width, h = in[0].height;
    std::vector<uint8_t> dense;
    const uint8_t* src = static_cast<const uint8_t*>(in[0].data);
    if (in[0].stride && in[0].stride != (size_t)w * 3) {
        dense.resize((size_t)w * h * 3);
        for (int y = 0; y < h; ++y) std::memcpy(&dense[(size_t)y * w * 3], src + (size_t)y * in[0].stride, (size_t)w * 3);
        src = dense.data();
    }
    translate_u8c3(src, w, h, ox, oy, buf.color);
    out[0].data = buf.color.data(); out[0].stride = 0;
    if (in.size() >= 2) {
        std::vector<uint16_t> dd;
        const uint16_t* ds = static_cast<const uint16_t*>(in[1].data);
        if (in[1].stride && in[1].stride != (size_t)w * 2) {
            dd.resize((size_t)w * h);
            for (int y = 0; y < h; ++y) std::memcpy(&dd[(size_t)y * w], reinterpret_cast<const uint8_t*>(ds) + (size_t)y * in[1].stride, (size_t)w * 2);
            ds = dd.data();
        }
        translate_u16(ds, w, h, ox, oy, buf.depth);
        out[1].data = buf.depth.data(); out[1].stride = 0;
    }
}

// :70-95 without the ICP branch: the first pose of every group, until in_numberOfObjects poses are collected
void PoseDetection::pickFinal(const std::vector<std::vector<ObjectPose>>& groups, uint16_t nObjects, std::vector<ObjectPose>& out) {
    out.clear();
    for (const auto& g : groups) {
        if (g.empty()) continue;
        out.push_back(g[0]);
        if (out.size() == nObjects) break;
    }
}

void PoseDetection::detect(std::vector<Image>& in_imgs, std::string const& in_className, uint16_t const& in_numberOfObjects,
                           std::vector<ObjectPose>& in_objPose, bool in_displayResults) {
    const uint16_t numClassIndex = findIndexInVector(in_className, ids);
    Shifted buf;
    std::vector<Image> inputImg;
    shiftFrame(in_imgs, buf, inputImg);
    finalObjectPoses.clear();
    line->detectTemplate(inputImg, numClassIndex);
    detectedPoses = line->getObjectPoses();
    pickFinal(detectedPoses, in_numberOfObjects, finalObjectPoses);
    if (in_displayResults)
        for (const ObjectPose& p : finalObjectPoses) in_objPose.push_back(p);
}

bool PoseDetection::detectBatch(std::vector<std::vector<Image>>& in_frames, std::string const& in_className,
                                uint16_t const& in_numberOfObjects, std::vector<std::vector<ObjectPose>>& out) {
    error.clear();
    out.assign(in_frames.size(), {});
    const uint16_t numClassIndex = findIndexInVector(in_className, ids);
    if (numClassIndex >= ids.size()) { error = "unknown class name: " + in_className; return false; }   // (find's not-found index = ids.size())
    if (in_frames.size() > (size_t)HighLevelLineMOD::kBatchSlots) { error = "batch of " + std::to_string(in_frames.size()) + " frames exceeds the detector's frame slots"; return false; }
    if (batchBufs.size() < in_frames.size()) batchBufs.resize(in_frames.size());
    std::vector<std::vector<Image>> shifted(in_frames.size());
    for (size_t i = 0; i < in_frames.size(); ++i) shiftFrame(in_frames[i], batchBufs[i], shifted[i]);
    std::vector<std::vector<lm_match_t>> m;
    std::vector<std::vector<std::vector<ObjectPose>>> groups;
    line->detectTemplateBatch(shifted, numClassIndex, m, groups);
    if (!line->lastError().empty()) { error = line->lastError(); return false; }     // (the batch entry points clear it on entry)
    for (size_t i = 0; i < in_frames.size() && i < groups.size(); ++i) pickFinal(groups[i], in_numberOfObjects, out[i]);
    finalObjectPoses = out.empty() ? std::vector<ObjectPose>() : out.back();
    return true;
}

bool PoseDetection::detectBatch(std::vector<std::vector<Image>>& in_frames, std::vector<std::string> const& in_classNames,
                                uint16_t const& in_numberOfObjects, std::vector<std::vector<std::vector<ObjectPose>>>& out) {
    out.assign(in_classNames.size(), std::vector<std::vector<ObjectPose>>(in_frames.size()));
    if (line->batchesInFlight() != 0) { error = "batches are in flight: collect them with detectBatchEnd first"; return false; }
    if (!detectBatchBegin(in_frames, in_classNames)) return false;
    return detectBatchEnd(in_numberOfObjects, out);
}

bool PoseDetection::detectBatchBegin(std::vector<std::vector<Image>>& in_frames, std::vector<std::string> const& in_classNames) {
    error.clear();
    std::vector<uint16_t> idx;
    for (const std::string& nme : in_classNames) {
        const uint16_t k = findIndexInVector(nme, ids);
        if (k >= ids.size()) { error = "unknown class name: " + nme; return false; }
        idx.push_back(k);
    }
    if (in_frames.size() > (size_t)HighLevelLineMOD::kBatchSlots) { error = "batch of " + std::to_string(in_frames.size()) + " frames exceeds the detector's frame slots"; return false; }
    // (before anything is translated: a refused Begin must not touch the buffers of a batch in flight)
    if (line->batchesInFlight() >= HighLevelLineMOD::kBatchSets) { error = "all slot sets are in flight: call detectBatchEnd first"; return false; }
    std::vector<std::vector<Image>> shifted(in_frames.size());
    if (line->usesGpuColorCheck()) {
        // r04: nothing is translated on the host -- the frames go up with the shift applied while the staging buffer is filled (or by the
        // DMA engine's row-offset copy for pinned frames) and the host depth check reads the untranslated depth image through the same shift
        const int ox = (int)(-camParams.cx + camParams.videoWidth / 2), oy = (int)(-camParams.cy + camParams.videoHeight / 2);
        for (size_t i = 0; i < in_frames.size(); ++i) {
            shifted[i] = in_frames[i];
            for (Image& im : shifted[i]) { im.shift_x = ox; im.shift_y = oy; }
        }
    } else {
        // host colour check: translated copies, one buffer set per batch in flight (they must outlive the batch's End)
        const size_t base = streamBufNext * (size_t)HighLevelLineMOD::kBatchSlots;
        streamBufNext = (streamBufNext + 1) % (size_t)HighLevelLineMOD::kBatchSets;
        if (batchBufs.size() < (size_t)HighLevelLineMOD::kBatchSlots * HighLevelLineMOD::kBatchSets) batchBufs.resize((size_t)HighLevelLineMOD::kBatchSlots * HighLevelLineMOD::kBatchSets);
        for (size_t i = 0; i < in_frames.size(); ++i) shiftFrame(in_frames[i], batchBufs[base + i], shifted[i]);
    }
    if (!line->detectTemplatesBatchBegin(shifted, idx)) { error = line->lastError(); return false; }
    streamShape.push_back({in_classNames.size(), in_frames.size()});
    return true;
}

bool PoseDetection::detectBatchEnd(uint16_t const& in_numberOfObjects, std::vector<std::vector<std::vector<ObjectPose>>>& out) {
    error.clear();
    if (streamShape.empty()) { error = "no batch in flight"; out.clear(); return false; }
    const std::pair<size_t, size_t> shape = streamShape.front();
    streamShape.pop_front();
    out.assign(shape.first, std::vector<std::vector<ObjectPose>>(shape.second));
    std::vector<std::vector<std::vector<lm_match_t>>> m;
    std::vector<std::vector<std::vector<std::vector<ObjectPose>>>> groups;
    line->detectTemplatesBatchEnd(m, groups);
    if (!line->lastError().empty()) { error = line->lastError(); return false; }     // (the batch entry points clear it on entry)
    for (size_t c = 0; c < shape.first && c < groups.size(); ++c)
        for (size_t i = 0; i < shape.second && i < groups[c].size(); ++i) pickFinal(groups[c][i], in_numberOfObjects, out[c][i]);
    finalObjectPoses = (out.empty() || out.back().empty()) ? std::vector<ObjectPose>() : out.back().back();
    return true;
}

}  // namespace lmamd
