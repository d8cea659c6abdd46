// HighLevelLinemod.cpp -- see HighLevelLinemod.h.  Plain C++17, links against liblinemod_hip.so only.
#include "HighLevelLinemod.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <fstream>
#include <stdexcept>
#include <thread>

#include "PostProcess.h"
#include "TemplateGenerator.h"

namespace lmamd {

HighLevelLineMOD::HighLevelLineMOD(CameraParameters const& cam, TemplateGenerationSettings const& ts)
    : onlyColorModality(ts.onlyUseColorModality),
      videoWidth(cam.videoWidth),
      videoHeight(cam.videoHeight),
      fy(cam.fy),
      settings(ts),
      detectorThreshold(ts.detectorThreshold),
      templates(new std::vector<TemplatePose>()),
      modelTemplates(new std::vector<std::vector<TemplatePose>>()),
      modProps(new std::vector<ModelProperties>()) {
    lm_config cfg;
    lm_default_config(&cfg, onlyColorModality ? 1 : 0, videoWidth, videoHeight);  // T = {2,8} / {5,8}
    cfg.device = ts.device;
    cfg.shard_rank = ts.shardRank;
    cfg.shard_size = ts.shardSize;
    if (lm_create(&cfg, &detector) != LM_OK) {
        delete templates; delete modelTemplates; delete modProps;
        throw std::runtime_error(lm_last_error());
    }
}

HighLevelLineMOD::~HighLevelLineMOD() {
    lm_destroy(detector);  // detector.release(), :50
    delete templates; delete modelTemplates; delete modProps;
}

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
    const Image* depth_img = in_imgs.size() >= 2 ? &in_imgs[1] : nullptr;
    // a colour-only detector pops the depth image before match() and pushes it back afterwards (:146-156)
    const Image* match_depth = onlyColorModality ? nullptr : depth_img;
    if (color.width != videoWidth || color.height != videoHeight) { error = "frame size differs from the detector's"; return false; }
    size_t cap = 4096, n = 0;
    for (;;) {
        matches.resize(cap);
        int rc = lm_match(detector, static_cast<const uint8_t*>(color.data), color.stride,
                          match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr,
                          match_depth ? match_depth->stride : 0, detectorThreshold, in_classNumber, matches.data(), cap, &n);
        if (rc == LM_ERR_OVERFLOW && n > cap) { cap = n; continue; }  // the reference consumes ALL matches
        if (rc != LM_OK) { error = lm_last_error(); matches.clear(); return false; }
        break;
    }
    matches.resize(n);
    if (matches.empty()) return false;  // :157,187-189
    // the frame lm_match uploaded is still resident in slot 0: the colour checks can run there
    posesMultipleObj = postProcess(matches, color, depth_img, in_classNumber, gpuColorCheck ? 0 : -1);
    return true;
}

// :157-175 post-processing, when this class has template poses (built by addTemplate or read back)
std::vector<std::vector<ObjectPose>> HighLevelLineMOD::postProcess(const std::vector<lm_match_t>& in_matches, const Image& color,
                                                                   const Image* depth_img, uint16_t in_classNumber, int gpu_slot) {
    std::vector<std::vector<ObjectPose>> out;
    if (!(in_classNumber < modelTemplates->size()) || (*modelTemplates)[in_classNumber].empty()) return out;
    PostProcessSettings ps;
    ps.onlyColorModality = onlyColorModality;
    ps.videoWidth = videoWidth; ps.videoHeight = videoHeight; ps.fy = fy;
    ps.stepSize = settings.stepSize; ps.percentToPassCheck = settings.percentToPassCheck;
    ps.numberWantedPoses = settings.numberWantedPoses;
    ps.radiusThresholdNewObject = settings.radiusThresholdNewObject;
    ps.discardGroupRatio = settings.discardGroupRatio;
    ps.useDepthImprovement = settings.useDepthImprovement; ps.depthOffset = settings.depthOffset;
    ModelProperties props;
    if (in_classNumber < modProps->size()) props = (*modProps)[in_classNumber];
    PostProcessor pp(detector, ps);
    out = pp.run(in_matches, static_cast<const uint8_t*>(color.data), color.stride,
                 depth_img ? static_cast<const uint16_t*>(depth_img->data) : nullptr, depth_img ? depth_img->stride : 0,
                 (*modelTemplates)[in_classNumber], props, gpu_slot);
    if (!pp.lastError().empty()) error = pp.lastError();
    return out;
}

// A batch of frames against one class (BASELINE config 5): the frames go to the detector's slots [0, n) (asynchronous
// uploads), ONE lm_match_batch matches them all, then every frame is post-processed like detectTemplate does, with its
// colour checks on the GPU slot the frame is resident in.  in_frames[i] = {colour} or {colour, depth}.
bool HighLevelLineMOD::detectTemplateBatch(std::vector<std::vector<Image>>& in_frames, uint16_t in_classNumber,
                                           std::vector<std::vector<lm_match_t>>& out_matches,
                                           std::vector<std::vector<std::vector<ObjectPose>>>& out_poses) {
    const int n = (int)in_frames.size();
    error.clear();
    out_matches.assign((size_t)n, {});
    out_poses.assign((size_t)n, {});
    if (n == 0) return false;
    for (int i = 0; i < n; ++i) {
        if (in_frames[(size_t)i].empty()) { error = "no images"; return false; }
        const Image& color = in_frames[(size_t)i][0];
        const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
        const Image* match_depth = onlyColorModality ? nullptr : depth_img;
        if (color.width != videoWidth || color.height != videoHeight) { error = "frame size differs from the detector's"; return false; }
        if (lm_upload_frame(detector, i, static_cast<const uint8_t*>(color.data), color.stride,
                            match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr,
                            match_depth ? match_depth->stride : 0) != LM_OK) { error = lm_last_error(); return false; }
    }
    size_t cap = 4096;
    std::vector<lm_match_t> buf;
    std::vector<int32_t> counts((size_t)n);
    for (;;) {
        buf.resize(cap * (size_t)n);
        int rc = lm_match_batch(detector, n, detectorThreshold, in_classNumber, buf.data(), cap, counts.data());
        size_t need = 0;
        for (int32_t c : counts) need = std::max(need, (size_t)c);
        if (rc == LM_ERR_OVERFLOW && need > cap) { cap = need; continue; }   // the reference consumes ALL matches
        if (rc != LM_OK) { error = lm_last_error(); return false; }
        break;
    }
    bool any = false;
    for (int i = 0; i < n; ++i) {
        out_matches[(size_t)i].assign(buf.begin() + (ptrdiff_t)(cap * (size_t)i), buf.begin() + (ptrdiff_t)(cap * (size_t)i + (size_t)counts[(size_t)i]));
        if (out_matches[(size_t)i].empty()) continue;
        any = true;
        const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
        out_poses[(size_t)i] = postProcess(out_matches[(size_t)i], in_frames[(size_t)i][0], depth_img, in_classNumber, gpuColorCheck ? i : -1);
    }
    return any;
}

bool HighLevelLineMOD::detectTemplatesBatch(std::vector<std::vector<Image>>& in_frames, const std::vector<uint16_t>& in_classNumbers,
                                            std::vector<std::vector<std::vector<lm_match_t>>>& out_matches,
                                            std::vector<std::vector<std::vector<std::vector<ObjectPose>>>>& out_poses) {
    const int n = (int)in_frames.size();
    const size_t nc = in_classNumbers.size();
    error.clear();
    out_matches.assign(nc, std::vector<std::vector<lm_match_t>>((size_t)n));
    out_poses.assign(nc, std::vector<std::vector<std::vector<ObjectPose>>>((size_t)n));
    if (n == 0 || nc == 0) return false;
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t_up = clk::now();
    for (int i = 0; i < n; ++i) {
        if (in_frames[(size_t)i].empty()) { error = "no images"; return false; }
        const Image& color = in_frames[(size_t)i][0];
        const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
        const Image* match_depth = onlyColorModality ? nullptr : depth_img;
        if (color.width != videoWidth || color.height != videoHeight) { error = "frame size differs from the detector's"; return false; }
        // a pending shift (Image::shift_*): applied while the staging buffer is filled; both images of the frame carry the same one
        int urc;
        if (color.shift_x || color.shift_y) {
            if (depth_img && (depth_img->shift_x != color.shift_x || depth_img->shift_y != color.shift_y)) { error = "colour and depth image carry different pending shifts"; return false; }
            if (!gpuColorCheck) { error = "the host colour check reads the colour image: shift it before the call (pending shifts need the GPU colour check)"; return false; }
            urc = lm_upload_frame_shifted(detector, i, static_cast<const uint8_t*>(color.data), color.stride,
                                          match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr, match_depth ? match_depth->stride : 0,
                                          color.shift_x, color.shift_y);
        } else {
            urc = lm_upload_frame(detector, i, static_cast<const uint8_t*>(color.data), color.stride,
                                  match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr, match_depth ? match_depth->stride : 0);
        }
        if (urc != LM_OK) { error = lm_last_error(); return false; }
    }
    const clk::time_point t_match = clk::now();
    std::vector<int32_t> cls(in_classNumbers.begin(), in_classNumbers.end());
    size_t cap = 4096;
    std::vector<lm_match_t> buf;
    std::vector<int32_t> counts((size_t)n);
    for (;;) {
        buf.resize(cap * (size_t)n);
        int rc = lm_match_batch_classes(detector, 0, n, detectorThreshold, cls.data(), (int)cls.size(), buf.data(), cap, counts.data());
        size_t need = 0;
        for (int32_t c : counts) need = std::max(need, (size_t)c);
        if (rc == LM_ERR_OVERFLOW && need > cap) {
            // the frames are still resident and prepared: only a11-a15 run again (the reference consumes ALL matches)
            cap = need;
            buf.resize(cap * (size_t)n);
            rc = lm_match_prepared(detector, 0, n, detectorThreshold, cls.data(), (int)cls.size(), buf.data(), cap, counts.data());
        }
        if (rc != LM_OK) { error = lm_last_error(); return false; }
        break;
    }
    const clk::time_point t_post = clk::now();
    stageTimes.upload += secs(t_up, t_match); stageTimes.match += secs(t_match, t_post); stageTimes.frames += n;
    bool any = false;
    // ---- step 1, this thread (it owns the detector): split the mixed lists by class; per (class, frame) grouping + the GPU colour counts
    struct Unit { size_t c; int i; PostProcessor pp; PostProcessor::Prepared prep; const std::vector<TemplatePose>* tpl; const uint16_t* depth; std::vector<uint16_t> dense; ModelProperties props; };
    std::vector<Unit> units;
    PostProcessSettings ps;
    ps.onlyColorModality = onlyColorModality;
    ps.videoWidth = videoWidth; ps.videoHeight = videoHeight; ps.fy = fy;
    ps.stepSize = settings.stepSize; ps.percentToPassCheck = settings.percentToPassCheck;
    ps.numberWantedPoses = settings.numberWantedPoses;
    ps.radiusThresholdNewObject = settings.radiusThresholdNewObject;
    ps.discardGroupRatio = settings.discardGroupRatio;
    ps.useDepthImprovement = settings.useDepthImprovement; ps.depthOffset = settings.depthOffset;
    units.reserve(nc * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const lm_match_t* m = buf.data() + cap * (size_t)i;
        stageTimes.matches += counts[(size_t)i];
        // the mixed list is in the total order; a class's sub-list keeps it
        for (size_t c = 0; c < nc; ++c) {
            std::vector<lm_match_t>& dst = out_matches[c][(size_t)i];
            // Match::operator== compares x, y, similarity and class; std::unique removed the ADJACENT duplicates of the mixed
            // list, where a match of another class may sit between two equal ones of this class: filtering and removing
            // adjacent duplicates once more gives exactly the list a one-class match() returns
            for (int32_t k = 0; k < counts[(size_t)i]; ++k) {
                if (m[k].class_idx != (int32_t)in_classNumbers[c]) continue;
                if (!dst.empty() && dst.back().x == m[k].x && dst.back().y == m[k].y && dst.back().similarity == m[k].similarity) continue;
                dst.push_back(m[k]);
            }
            if (dst.empty()) continue;
            any = true;
            const uint16_t cls_no = in_classNumbers[c];
            if (!(cls_no < modelTemplates->size()) || (*modelTemplates)[cls_no].empty()) continue;       // no template poses: no post-processing
            ModelProperties props;
            if (cls_no < modProps->size()) props = (*modProps)[cls_no];
            const Image& color = in_frames[(size_t)i][0];
            const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
            units.push_back(Unit{c, i, PostProcessor(detector, ps), {}, &(*modelTemplates)[cls_no], nullptr, {}, {}});
            Unit& u = units.back();
            u.pp.setDepthShift(color.shift_x, color.shift_y);       // the host depth check reads the UNSHIFTED image through the pending shift
            if (depth_img) {
                u.depth = static_cast<const uint16_t*>(depth_img->data);
                if (depth_img->stride && depth_img->stride != (size_t)videoWidth * 2) {
                    u.dense.resize((size_t)videoWidth * videoHeight);
                    for (int y = 0; y < videoHeight; ++y)
                        std::memcpy(&u.dense[(size_t)y * videoWidth], reinterpret_cast<const uint8_t*>(depth_img->data) + (size_t)y * depth_img->stride, (size_t)videoWidth * 2);
                    u.depth = u.dense.data();
                }
            }
            if (gpuColorCheck) {
                u.prep = u.pp.prepare_groups(dst, *u.tpl);          // the colour counts follow, one GPU call per frame and HSV range
                u.props = props;
            } else {
                u.prep = u.pp.prepare(dst, static_cast<const uint8_t*>(color.data), color.stride, *u.tpl, props, -1);
                if (!u.pp.lastError().empty()) error = u.pp.lastError();
            }
        }
        if (gpuColorCheck) {
            // the units of this frame (they are the tail of `units`): classes with the same HSV range share ONE lm_color_check_counts call
            // (the colour mask of the frame is computed once per call: three classes = one mask instead of three)
            size_t first = units.size();
            while (first > 0 && units[first - 1].i == i) --first;
            std::vector<char> done(units.size() - first, 0);
            for (size_t a = first; a < units.size(); ++a) {
                if (done[a - first]) continue;
                std::vector<size_t> same;
                std::vector<lm_match_t> todo;
                for (size_t b = a; b < units.size(); ++b) {
                    if (done[b - first]) continue;
                    bool eq = true;
                    for (int k = 0; k < 3; ++k) eq = eq && units[a].props.lowerColorRange[k] == units[b].props.lowerColorRange[k] && units[a].props.upperColorRange[k] == units[b].props.upperColorRange[k];
                    if (!eq) continue;
                    done[b - first] = 1; same.push_back(b);
                    todo.insert(todo.end(), units[b].prep.todo.begin(), units[b].prep.todo.end());
                }
                std::vector<int64_t> gin(todo.size()), gboth(todo.size());
                const clk::time_point t_c = clk::now();
                if (!todo.empty() && lm_color_check_counts(detector, i, units[a].props.lowerColorRange, units[a].props.upperColorRange, todo.data(), todo.size(),
                                                           gin.data(), gboth.data()) != LM_OK) {
                    error = lm_last_error();
                    for (size_t b : same) { units[b].prep.failed = true; units[b].prep.groups.clear(); }
                    continue;
                }
                PostProcessor::times().colour += secs(t_c, clk::now()); PostProcessor::times().colour_checks += (long)todo.size();
                size_t at = 0;
                for (size_t b : same) {
                    PostProcessor::set_counts(units[b].prep, gin.data() + at, gboth.data() + at);
                    at += units[b].prep.todo.size();
                }
            }
        }
    }
    // ---- step 2, any thread: one task per group of every unit (the sequential accept / break loop of a group: colour verdict,
    // depth check, pose); results land in per-task slots and are put together in group order afterwards
    struct Task { size_t unit, group; };
    std::vector<Task> tasks;
    for (size_t k = 0; k < units.size(); ++k)
        for (size_t g = 0; g < units[k].prep.groups.size(); ++g) tasks.push_back(Task{k, g});
    std::vector<std::vector<ObjectPose>> results(tasks.size());
    int nthreads = postThreads > 0 ? postThreads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nthreads = (int)std::min<size_t>((size_t)nthreads, std::max<size_t>(tasks.size(), 1));
    std::vector<PostProcessor::Times> wt((size_t)nthreads);
    std::atomic<size_t> next{0};
    auto worker = [&](int w) {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= tasks.size()) break;
            const Unit& u = units[tasks[t].unit];
            results[t] = u.pp.finish_group(u.prep, tasks[t].group, out_matches[u.c][(size_t)u.i], u.depth, *u.tpl, &wt[(size_t)w]);
        }
    };
    if (nthreads <= 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int w = 1; w < nthreads; ++w) th.emplace_back(worker, w);
        worker(0);
        for (std::thread& t : th) t.join();
    }
    for (const PostProcessor::Times& t : wt) PostProcessor::times().add(t);
    for (size_t t = 0; t < tasks.size(); ++t) {
        if (results[t].empty()) continue;
        const Unit& u = units[tasks[t].unit];
        stageTimes.poses += (long)results[t].size();
        out_poses[u.c][(size_t)u.i].push_back(std::move(results[t]));
    }
    stageTimes.post += secs(t_post, clk::now());
    return any;
}

void HighLevelLineMOD::writeLinemod() {
    // cv::FileStorage fs("linemod_templates.yml.gz", WRITE); detector->write(fs); classes [ { writeClass } ]  (:256-270)
    if (lm_save_yaml(detector, "linemod_templates.yml.gz") != LM_OK) { error = lm_last_error(); std::printf("ERROR::%s\n", error.c_str()); }
    // linemod_tempPosFile.bin: u32 nClasses; per class {u64 n; n raw Template records}  (:272-284)
    std::ofstream f("linemod_tempPosFile.bin", std::ios::binary | std::ios::out);
    uint32_t nvec = (uint32_t)modelTemplates->size();
    f.write(reinterpret_cast<const char*>(&nvec), sizeof(nvec));
    for (const auto& v : *modelTemplates) {
        uint64_t n = v.size();
        f.write(reinterpret_cast<const char*>(&n), sizeof(n));
        f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(n * sizeof(TemplatePose)));
    }
}

void HighLevelLineMOD::readLinemod() { readLinemodFrom("linemod_templates.yml.gz", "linemod_tempPosFile.bin"); }

void HighLevelLineMOD::readLinemodFrom(const std::string& templateFile, const std::string& poseFile) {
    templates->clear();
    modelTemplates->clear();
    // detector->read(fs.root()); readClass per entry of "classes"  (:292-303)
    auto ends_with = [&](const char* suf) { const std::string t(suf); return templateFile.size() >= t.size() && templateFile.compare(templateFile.size() - t.size(), t.size(), t) == 0; };
    const bool yaml = ends_with(".yml") || ends_with(".yml.gz") || ends_with(".yaml");
    if ((yaml ? lm_load_yaml(detector, templateFile.c_str()) : lm_load_bank(detector, templateFile.c_str())) != LM_OK) { error = lm_last_error(); std::printf("ERROR::%s\n", error.c_str()); }
    std::ifstream f(poseFile, std::ios::in | std::ios::binary);
    uint32_t nvec = 0;
    if (f && f.read(reinterpret_cast<char*>(&nvec), sizeof(nvec))) {
        for (uint32_t c = 0; c < nvec; ++c) {
            uint64_t n = 0;
            if (!f.read(reinterpret_cast<char*>(&n), sizeof(n)) || n > (1ull << 28)) break;
            std::vector<TemplatePose> v((size_t)n);
            if (n && !f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(TemplatePose)))) break;
            modelTemplates->push_back(std::move(v));
        }
    }
    readColorRanges();
}

// models/<class id minus ".ply">.yml : "lower color range: [ h, s, v, 0 ]", "upper color range: [...]" (:523-543)
void HighLevelLineMOD::readColorRanges() {
    modProps->clear();
    for (const std::string& id : getClassIds()) {
        ModelProperties p;
        std::string stem = id.size() > 4 ? id.substr(0, id.size() - 4) : id;
        const std::string file = settings.modelFolder + stem + ".yml";
        double v[4];
        size_t n = 0;
        if (lm_yaml_numbers(file.c_str(), "lower color range", v, 4, &n) == LM_OK && n >= 3)
            for (int k = 0; k < 3; ++k) p.lowerColorRange[k] = v[k];
        if (lm_yaml_numbers(file.c_str(), "upper color range", v, 4, &n) == LM_OK && n >= 3)
            for (int k = 0; k < 3; ++k) p.upperColorRange[k] = v[k];
        modProps->push_back(p);
    }
}

// utility.cpp readSettings: linemod_settings.yml -> CameraParameters + TemplateGenerationSettings
bool readSettings(const std::string& file, CameraParameters& cam, TemplateGenerationSettings& ts) {
    auto num = [&](const char* key, double* out) {
        size_t n = 0;
        return lm_yaml_numbers(file.c_str(), key, out, 1, &n) == LM_OK && n == 1;
    };
    double v;
    if (!num("video width", &v)) return false;
    cam.videoWidth = (uint16_t)v;
    if (!num("video height", &v)) return false;
    cam.videoHeight = (uint16_t)v;
    if (num("camera fx", &v)) cam.fx = (float)v;
    if (num("camera fy", &v)) cam.fy = (float)v;
    if (num("camera cx", &v)) cam.cx = (float)v;
    if (num("camera cy", &v)) cam.cy = (float)v;
    char buf[512];
    if (lm_yaml_string(file.c_str(), "model folder", buf, sizeof buf) == LM_OK) ts.modelFolder = buf;
    if (num("only use color modality", &v)) ts.onlyUseColorModality = v != 0;
    if (num("in plane rotation starting angle", &v)) ts.angleStart = (int16_t)v;
    if (num("in plane rotation stopping angle", &v)) ts.angleStop = (int16_t)v;
    if (num("in plane rotation angle step", &v)) ts.angleStep = (int16_t)v;
    if (num("distance step", &v)) ts.stepSize = (uint16_t)v;
    if (num("detector threshold", &v)) ts.detectorThreshold = (float)v;
    if (num("percent to pass check", &v)) ts.percentToPassCheck = (uint16_t)v;
    if (num("number of poses to compare", &v)) ts.numberWantedPoses = (uint16_t)v;
    if (num("distance to match to be considered same object", &v)) ts.radiusThresholdNewObject = (float)v;
    if (num("ratio to determine if group is too small", &v)) ts.discardGroupRatio = (float)v;
    if (num("use depth improvement", &v)) ts.useDepthImprovement = v != 0;
    if (num("depth offset", &v)) ts.depthOffset = (float)v;
    return true;
}

void HighLevelLineMOD::setColorRange(uint16_t classNumber, const double lo[3], const double hi[3]) {
    if (modProps->size() <= classNumber) modProps->resize((size_t)classNumber + 1);
    for (int k = 0; k < 3; ++k) { (*modProps)[classNumber].lowerColorRange[k] = lo[k]; (*modProps)[classNumber].upperColorRange[k] = hi[k]; }
}

bool HighLevelLineMOD::addTemplate(std::vector<Image>& in_images, const std::string& in_modelName, Vec3 in_cameraPosition) {
    if (in_images.size() < 2) { error = "addTemplate needs {colour, depth}"; return false; }
    const Image& color = in_images[0];
    const Image& depth = in_images[1];
    const int w = depth.width, h = depth.height;
    // colorToBinary = threshold(colour, 1, 255); mask = threshold(depth, 1, 65535) as 8-bit  (:77-79)
    std::vector<uint16_t> dense((size_t)w * h);
    size_t dstride = depth.stride ? depth.stride : (size_t)w * 2;
    for (int y = 0; y < h; ++y) std::memcpy(&dense[(size_t)y * w], static_cast<const uint8_t*>(depth.data) + y * dstride, (size_t)w * 2);
    std::vector<uint8_t> mask((size_t)w * h), bin((size_t)w * h * 3);
    for (size_t i = 0; i < mask.size(); ++i) mask[i] = dense[i] > 1 ? 255 : 0;
    size_t cstride = color.stride ? color.stride : (size_t)w * 3;
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = static_cast<const uint8_t*>(color.data) + y * cstride;
        for (int x = 0; x < w * 3; ++x) bin[(size_t)y * w * 3 + x] = row[x] > 1 ? 255 : 0;
    }
    std::vector<uint8_t> maskRotated, colorRotated, er((size_t)w * h);
    std::vector<uint16_t> depthRotated;
    // one template per in-plane rotation (generateRotMatForInplaneRotation :327-334, loop :81-108)
    int q = 0;
    for (int angle = settings.angleStart; angle <= settings.angleStop; angle += std::max<int>(settings.angleStep, 1), ++q) {
        warp_rotate_u8(mask.data(), w, h, 1, (float)angle, maskRotated);
        warp_rotate_u8(bin.data(), w, h, 3, (float)angle, colorRotated);
        warp_rotate_u16(dense.data(), w, h, (float)angle, depthRotated);
        for (int y = 0; y < h; ++y)   // erode(maskRotated, 3x3, 1 iteration), border pixels do not erode (:91)
            for (int x = 0; x < w; ++x) {
                uint8_t v = 255;
                for (int j = -1; j <= 1; ++j)
                    for (int i = -1; i <= 1; ++i) {
                        int yy = y + j, xx = x + i;
                        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                        v = std::min(v, maskRotated[(size_t)yy * w + xx]);
                    }
                er[(size_t)y * w + x] = v;
            }
        int tid = -1;
        lm_rect bb;
        int rc = lm_add_template(detector, in_modelName.c_str(), colorRotated.data(), 0,
                                 onlyColorModality ? nullptr : depthRotated.data(), 0, er.data(), 0, &tid, &bb);
        if (rc != LM_OK || tid < 0) {
            error = lm_last_error();
            std::printf("ERROR::Cant create Template\n");  // :99
            return false;
        }
        TemplatePose tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.bb[0] = bb.x; tp.bb[1] = bb.y; tp.bb[2] = bb.width; tp.bb[3] = bb.height;
        tp.medianDepth = median_mat(depthRotated.data(), w, h, Rect{bb.x, bb.y, bb.width, bb.height}, 5);   // :104
        const int16_t currentInplaneAngle = (int16_t)(-(settings.angleStart + q * settings.angleStep));    // :105
        calculate_template_pose(in_cameraPosition, currentInplaneAngle, tp.translation, tp.quat_xyzw);       // :106
        templates->push_back(tp);                                                                             // :107
    }
    return true;
}

void HighLevelLineMOD::pushBackTemplates() {   // :517-521
    modelTemplates->push_back(*templates);
    templates->clear();
}

}  // namespace lmamd
