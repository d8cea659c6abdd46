// HighLevelLinemod.cpp -- see HighLevelLinemod.h.  Plain C++17, links against liblinemod_hip.so only.
#include "HighLevelLinemod.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <memory>
#include <fstream>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "PostProcess.h"
#include "GroupWaves.h"
#include "WorkerPool.h"
#include "TemplateGenerator.h"

namespace lmamd {

// ---- the batch entry point for several classes, as a stream (r05) ---------------------------------------------------------------
// One batch in flight: its slot set (= lane), the frames' views, the class list, the staging copies' task group.
struct HighLevelLineMOD::Batch {
    // the staging copies of a batch's pageable frames, running on the pool after Begin has returned
    // (one task group per frame: a frame is handed to the DMA engine as soon as ITS rows are in, while the pool still copies the others)
    struct Staging { WorkerPool::Group frame[kBatchSlots]; std::atomic<long long> ns{0}; std::atomic<int> rc{LM_OK}; };
    int set = 0, n = 0;
    std::vector<std::vector<Image>> frames;
    std::vector<uint16_t> classes;
    std::unique_ptr<Staging> staging;     // pending staging copies (nullptr: none were needed)
    bool begun = false;                   // the transfers and the match are enqueued (finishBegin has run)
    std::string begin_error;              // why finishBegin failed (reported by the batch's End)
};
struct HighLevelLineMOD::Stream {
    std::deque<Batch> inflight;            // oldest first
    bool set_busy[kBatchSets] = {};
    int next_set = 0;
    std::unique_ptr<WorkerPool> pool;
    int pool_threads = 0;
};

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
    cfg.frame_slots = kBatchSlots * kBatchSets;      // slot sets: a batch is transferred and matched while the ones before it are matched / post-processed
    if (lm_create(&cfg, &detector) != LM_OK) {
        delete templates; delete modelTemplates; delete modProps;
        throw std::runtime_error(lm_last_error());
    }
}

HighLevelLineMOD::~HighLevelLineMOD() {
    if (stream_) {
        // batches still in flight: their lanes must finish before the detector goes away (results are dropped)
        while (!stream_->inflight.empty()) {
            Batch& fb = stream_->inflight.front();
            if (fb.staging && stream_->pool) for (WorkerPool::Group& g : fb.staging->frame) stream_->pool->wait(g);       // (the tasks hold pointers into the batch)
            if (fb.begun && fb.begin_error.empty()) (void)lm_match_end(detector, fb.set, nullptr, 0, nullptr);
            stream_->inflight.pop_front();
        }
        delete stream_;
    }
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
    if (batchesInFlight() != 0) { error = "batches are in flight: collect them with detectTemplatesBatchEnd first"; return false; }
    for (const Image& im : in_imgs)       // (only detectTemplatesBatch applies Image::shift_*; silently matching the unshifted frame would return wrong poses)
        if (im.shift_x || im.shift_y) { error = "pending image shifts (Image::shift_*) are applied by detectTemplatesBatch only: translate the images before this call"; return false; }
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
    if (batchesInFlight() != 0) { error = "batches are in flight: collect them with detectTemplatesBatchEnd first"; return false; }
    if (n > kBatchSlots) { error = "batch of " + std::to_string(n) + " frames exceeds a slot set (" + std::to_string(kBatchSlots) + ")"; return false; }
    for (int i = 0; i < n; ++i) {
        if (in_frames[(size_t)i].empty()) { error = "no images"; return false; }
        for (const Image& im : in_frames[(size_t)i])
            if (im.shift_x || im.shift_y) { error = "pending image shifts (Image::shift_*) are applied by detectTemplatesBatch only: translate the images before this call"; return false; }
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

HighLevelLineMOD::Stream& HighLevelLineMOD::stream() {
    if (!stream_) stream_ = new Stream();
    Stream& st = *stream_;
    const int want = postThreads > 0 ? postThreads : std::min(usable_cpus(), 32);
    if ((!st.pool || st.pool_threads != want) && st.inflight.empty()) {
        st.pool.reset();                       // joins the old workers first
        st.pool.reset(new WorkerPool(want));
        st.pool_threads = want;
    }
    return st;
}

void HighLevelLineMOD::setPostThreads(int n) { postThreads = n < 0 ? 0 : n; }
int HighLevelLineMOD::postThreadsInUse() const { return stream_ && stream_->pool ? stream_->pool_threads : 0; }
int HighLevelLineMOD::batchesInFlight() const { return stream_ ? (int)stream_->inflight.size() : 0; }

bool HighLevelLineMOD::detectTemplatesBatch(std::vector<std::vector<Image>>& in_frames, const std::vector<uint16_t>& in_classNumbers,
                                            std::vector<std::vector<std::vector<lm_match_t>>>& out_matches,
                                            std::vector<std::vector<std::vector<std::vector<ObjectPose>>>>& out_poses) {
    out_matches.assign(in_classNumbers.size(), std::vector<std::vector<lm_match_t>>(in_frames.size()));
    out_poses.assign(in_classNumbers.size(), std::vector<std::vector<std::vector<ObjectPose>>>(in_frames.size()));
    if (batchesInFlight() != 0) { error = "batches are in flight: collect them with detectTemplatesBatchEnd first"; return false; }
    if (!detectTemplatesBatchBegin(in_frames, in_classNumbers)) return false;
    return detectTemplatesBatchEnd(out_matches, out_poses);
}

bool HighLevelLineMOD::detectTemplatesBatchBegin(std::vector<std::vector<Image>>& in_frames, const std::vector<uint16_t>& in_classNumbers) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const int n = (int)in_frames.size();
    error.clear();
    if (n == 0 || in_classNumbers.empty()) { error = "no frames or no classes"; return false; }
    if (n > kBatchSlots) { error = "batch of " + std::to_string(n) + " frames exceeds a slot set (" + std::to_string(kBatchSlots) + ")"; return false; }
    Stream& st = stream();
    if ((int)st.inflight.size() >= kBatchSets) { error = "all slot sets are in flight: call detectTemplatesBatchEnd first"; return false; }
    const clk::time_point t_up = clk::now();
    for (int i = 0; i < n; ++i) {
        if (in_frames[(size_t)i].empty()) { error = "no images"; return false; }
        const Image& color = in_frames[(size_t)i][0];
        const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
        if (color.width != videoWidth || color.height != videoHeight) { error = "frame size differs from the detector's"; return false; }
        if (depth_img && (depth_img->width != videoWidth || depth_img->height != videoHeight)) { error = "depth image size differs from the detector's"; return false; }
        // a pending shift (Image::shift_*): both images of the frame carry the same one (checked whether or not it is zero: ADVICE r4)
        if (depth_img && (depth_img->shift_x != color.shift_x || depth_img->shift_y != color.shift_y)) { error = "colour and depth image carry different pending shifts"; return false; }
        if ((color.shift_x || color.shift_y) && !gpuColorCheck) { error = "the host colour check reads the colour image: shift it before the call (pending shifts need the GPU colour check)"; return false; }
        if (!onlyColorModality && !depth_img) { error = "sources.size() != modalities.size(): depth image missing"; return false; }
    }
    int set = -1;
    for (int k = 0; k < kBatchSets; ++k) { const int c = (st.next_set + k) % kBatchSets; if (!st.set_busy[c]) { set = c; break; } }
    if (set < 0) { error = "no free slot set"; return false; }
    const int first = set * kBatchSlots;
    // pageable frames: staging copies on the pool, ~1 MB of rows per task, all frames side by side; pinned frames: straight to the DMA engine
    bool any_pageable = false;
    for (int i = 0; i < n; ++i) any_pageable |= !in_frames[(size_t)i][0].pinned;
    if (any_pageable && lm_stage_reserve(detector, first, n) != LM_OK) { error = lm_last_error(); return false; }
    Batch b;
    b.set = set; b.n = n; b.frames = in_frames; b.classes = in_classNumbers;
    if (any_pageable) b.staging.reset(new Batch::Staging());
    const int rows_per_task = std::max(16, (int)((1 << 20) / ((size_t)videoWidth * (onlyColorModality ? 3 : 5))));
    for (int i = 0; i < n; ++i) {
        const Image& color = in_frames[(size_t)i][0];
        const Image* depth_img = in_frames[(size_t)i].size() >= 2 ? &in_frames[(size_t)i][1] : nullptr;
        const Image* match_depth = onlyColorModality ? nullptr : depth_img;
        if (color.pinned) {
            const int urc = lm_upload_frame_pinned_shifted(detector, first + i, static_cast<const uint8_t*>(color.data), color.stride,
                                                           match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr, match_depth ? match_depth->stride : 0,
                                                           color.shift_x, color.shift_y);
            if (urc != LM_OK) { error = lm_last_error(); if (b.staging) for (WorkerPool::Group& g : b.staging->frame) st.pool->wait(g); return false; }
            continue;
        }
        std::vector<std::function<void()>> copies;          // a frame's staging tasks in one submission (WorkerPool::submit_many)
        for (int r0 = 0; r0 < videoHeight; r0 += rows_per_task) {
            const int r1 = std::min<int>(r0 + rows_per_task, videoHeight);
            lm_detector* det = detector;
            const uint8_t* cp = static_cast<const uint8_t*>(color.data);
            const uint16_t* dp = match_depth ? static_cast<const uint16_t*>(match_depth->data) : nullptr;
            const size_t cs = color.stride, ds = match_depth ? match_depth->stride : 0;
            const int sx = color.shift_x, sy = color.shift_y, slot = first + i;
            Batch::Staging* sg = b.staging.get();
            copies.push_back([det, slot, cp, cs, dp, ds, sx, sy, r0, r1, sg] {
                const clk::time_point t0 = clk::now();
                const int rc = lm_stage_rows(det, slot, cp, cs, dp, ds, sx, sy, r0, r1);
                if (rc != LM_OK) sg->rc.store(rc);
                sg->ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count());
            });
        }
        st.pool->submit_many(b.staging->frame[i], std::move(copies), true);
    }
    st.inflight.push_back(std::move(b));
    st.set_busy[set] = true;
    st.next_set = (set + 1) % kBatchSets;
    // Nothing older in flight: nobody else will drive this batch on, so the transfers and the match are enqueued here and now (the serial
    // use, Begin; End).  Otherwise Begin returns at once: the staging copies run on the pool beside the older batch's End, which completes
    // this batch (finishBegin) while its own colour check is on the GPU -- the host's share of an upload disappears behind work that is
    // waiting anyway (r05: "in Begin" 68-160 us per frame with pageable frames before this).
    bool ok = true;
    if (st.inflight.size() == 1) {
        ok = finishBegin(st.inflight.back());
        if (!ok) {
            // a Begin that reports failure leaves nothing in flight -- the serial callers (detectTemplatesBatch, PoseDetection::detectBatch)
            // return without calling End, and every later call would otherwise fail with "batches are in flight".  finishBegin has waited for the
            // batch's staging tasks, and a failed lm_match_begin_classes leaves its lane idle.
            error = st.inflight.back().begin_error;
            st.set_busy[set] = false;
            st.next_set = set;
            st.inflight.pop_back();
        }
    }
    stageTimes.upload += secs(t_up, clk::now());
    return ok;
}

// The second half of Begin: the staging copies are in, the transfers and the class-list match go to the batch's lane.
bool HighLevelLineMOD::finishBegin(Batch& b) {
    using clk = std::chrono::steady_clock;
    if (b.begun) return b.begin_error.empty();
    b.begun = true;
    Stream& st = *stream_;
    const clk::time_point t0 = clk::now();
    const int n = b.n, first = b.set * kBatchSlots, set = b.set;
    // (the failure is kept in the batch: the caller that RETURNS it -- Begin, or the End of this batch -- sets lastError(); an End that only drives
    // the next batch's Begin on must not report that batch's failure as its own)
    auto failed = [&](const std::string& why) { b.begin_error = why; stageTimes.upload += std::chrono::duration<double>(clk::now() - t0).count(); return false; };
    // the tasks were queued at the FRONT one after the other, so the pool takes them last frame first: frames are waited for, and sent,
    // in that order -- the transfer of a frame runs while the rows of the next ones are still being copied
    for (int i = n - 1; i >= 0; --i) {
        if (b.frames[(size_t)i][0].pinned) continue;
        st.pool->wait(b.staging->frame[i]);
        if (!b.staging->frame[i].error.empty() || b.staging->rc.load() != LM_OK) {
            for (int k = i - 1; k >= 0; --k) st.pool->wait(b.staging->frame[k]);
            return failed(b.staging->frame[i].error.empty() ? "staging copy failed (lm_stage_rows)" : b.staging->frame[i].error);
        }
        if (lm_upload_staged(detector, first + i) != LM_OK) {
            const std::string why = lm_last_error();
            for (int k = i - 1; k >= 0; --k) st.pool->wait(b.staging->frame[k]);
            return failed(why);
        }
    }
    if (b.staging) stageTimes.staging_cpu += (double)b.staging->ns.load() * 1e-9;
    std::vector<int32_t> cls(b.classes.begin(), b.classes.end());
    if (gpuColorCheck) {
        // when the classes that will be post-processed share ONE HSV range (the usual case: one kind of part), the frames' colour masks are
        // computed on the lane ahead of the match, so that End's colour check is the hull launch alone
        const ModelProperties* range = nullptr;
        bool one = true;
        for (uint16_t c : b.classes) {
            if (!(c < modelTemplates->size()) || (*modelTemplates)[c].empty()) continue;
            static const ModelProperties kDefault;
            const ModelProperties* p = c < modProps->size() ? &(*modProps)[c] : &kDefault;
            if (!range) { range = p; continue; }
            for (int k = 0; k < 3; ++k) one = one && range->lowerColorRange[k] == p->lowerColorRange[k] && range->upperColorRange[k] == p->upperColorRange[k];
        }
        if (range && one && lm_color_mask_prepare(detector, /*lane*/ set, first, n, range->lowerColorRange, range->upperColorRange) != LM_OK) return failed(lm_last_error());
    }
    if (lm_match_begin_classes(detector, /*lane*/ set, first, n, detectorThreshold, cls.data(), (int)cls.size()) != LM_OK) return failed(lm_last_error());
    stageTimes.upload += std::chrono::duration<double>(clk::now() - t0).count();
    return true;
}

bool HighLevelLineMOD::detectTemplatesBatchEnd(std::vector<std::vector<std::vector<lm_match_t>>>& out_matches,
                                               std::vector<std::vector<std::vector<std::vector<ObjectPose>>>>& out_poses) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    error.clear();
    if (!stream_ || stream_->inflight.empty()) { error = "no batch in flight"; out_matches.clear(); out_poses.clear(); return false; }
    Stream& st = *stream_;
    if (!finishBegin(st.inflight.front())) {      // (a no-op for a batch that has begun; a failed second half of Begin is reported here)
        const std::string why = st.inflight.front().begin_error;
        st.set_busy[st.inflight.front().set] = false;
        st.inflight.pop_front();
        out_matches.clear(); out_poses.clear();
        error = why;
        return false;
    }
    const Batch b = std::move(st.inflight.front());
    st.inflight.pop_front();
    st.set_busy[b.set] = false;
    // the NEXT batch, if its Begin was deferred (completed below, right after this batch's lists are in)
    double driven = 0;                    // (seconds of this End spent on the next batch's Begin: accounted as upload, not as post-processing)
    auto drive_next = [&] {
        for (Batch& nb : st.inflight) {       // (oldest first: every batch whose Begin was deferred)
            if (nb.begun) continue;
            const clk::time_point t = clk::now();
            (void)finishBegin(nb);
            driven += secs(t, clk::now());
        }
    };
    const int n = b.n, first = b.set * kBatchSlots;
    const size_t nc = b.classes.size();
    out_matches.assign(nc, std::vector<std::vector<lm_match_t>>((size_t)n));
    out_poses.assign(nc, std::vector<std::vector<std::vector<ObjectPose>>>((size_t)n));
    const clk::time_point t_match = clk::now();
    size_t cap = 4096;
    std::vector<lm_match_t> buf(cap * (size_t)n);
    std::vector<int32_t> counts((size_t)n);
    {
        int rc = lm_match_end(detector, b.set, buf.data(), cap, counts.data());
        size_t need = 0;
        for (int32_t c : counts) need = std::max(need, (size_t)c);
        if (rc == LM_ERR_OVERFLOW && need > cap) {
            // the lists are still in the slots' result blocks: fetch them again with room for all (the reference consumes ALL matches)
            cap = need;
            buf.resize(cap * (size_t)n);
            rc = lm_match_collect(detector, first, n, buf.data(), cap, counts.data());
        }
        if (rc != LM_OK) { error = lm_last_error(); return false; }
    }
    // the next batch, if its Begin was deferred: its staging copies have been running on the pool since its Begin (this thread helps with
    // what is left), now its transfers and its match go to the other lane -- before this batch's post-processing, which they run beside
    drive_next();
    const clk::time_point t_post = clk::now();
    stageTimes.match += secs(t_match, t_post) - driven; stageTimes.frames += n;
    driven = 0;
    PostProcessSettings ps;
    ps.onlyColorModality = onlyColorModality;
    ps.videoWidth = videoWidth; ps.videoHeight = videoHeight; ps.fy = fy;
    ps.stepSize = settings.stepSize; ps.percentToPassCheck = settings.percentToPassCheck;
    ps.numberWantedPoses = settings.numberWantedPoses;
    ps.radiusThresholdNewObject = settings.radiusThresholdNewObject;
    ps.discardGroupRatio = settings.discardGroupRatio;
    ps.useDepthImprovement = settings.useDepthImprovement; ps.depthOffset = settings.depthOffset;
    // ---- step 1, pool (one task per frame): split the mixed list by class, per (class, frame) the grouping (host only)
    struct Unit { size_t c = 0; int i = 0; bool live = false; PostProcessor pp; PostProcessor::Prepared prep; const std::vector<TemplatePose>* tpl = nullptr;
                  const uint16_t* depth = nullptr; std::vector<uint16_t> dense; ModelProperties props; const std::vector<lm_match_t>* matches = nullptr;
                  std::vector<lm_depth_query> dq;      // r06: the depth checks' GPU queries of the unit's matches (built with the grouping, on the pool)
                  Unit(lm_detector* d, const PostProcessSettings& s) : pp(d, s) {} };
    std::vector<Unit> units;
    units.reserve(nc * (size_t)n);
    for (int i = 0; i < n; ++i) for (size_t c = 0; c < nc; ++c) { units.emplace_back(detector, ps); units.back().c = c; units.back().i = i; }
    std::vector<PostProcessor::Times> frame_times((size_t)n * nc);
    // r06: the depth checks' early verdicts from GPU counts -- by default only when this is the one batch in flight (setGpuDepthCounts)
    const bool depth_counts_on = !onlyColorModality && settings.useDepthImprovement && (gpuDepthCounts == 2 || (gpuDepthCounts == 1 && st.inflight.empty()));
    WorkerPool::Group grouping;
    std::vector<std::function<void()>> grouping_tasks;
    for (int i = 0; i < n; ++i) {
        stageTimes.matches += counts[(size_t)i];
        for (size_t c = 0; c < nc; ++c)              // one task per (frame, class): 24 tasks for config 5's batch of 8 x 3 (a frame's three at once left half the pool idle)
        grouping_tasks.push_back([&, i, c] {
            const lm_match_t* m = buf.data() + cap * (size_t)i;
            {
                // the mixed list is in the total order; a class's sub-list keeps it.  Match::operator== compares x, y, similarity and
                // class; std::unique removed the ADJACENT duplicates of the mixed list, where a match of another class may sit between
                // two equal ones of this class: filtering and removing adjacent duplicates once more gives exactly the list a one-class
                // match() returns
                std::vector<lm_match_t>& dst = out_matches[c][(size_t)i];
                for (int32_t k = 0; k < counts[(size_t)i]; ++k) {
                    if (m[k].class_idx != (int32_t)b.classes[c]) continue;
                    if (!dst.empty() && dst.back().x == m[k].x && dst.back().y == m[k].y && dst.back().similarity == m[k].similarity) continue;
                    dst.push_back(m[k]);
                }
                if (dst.empty()) return;
                const uint16_t cls_no = b.classes[c];
                if (!(cls_no < modelTemplates->size()) || (*modelTemplates)[cls_no].empty()) return;       // no template poses: no post-processing
                Unit& u = units[(size_t)i * nc + c];
                u.live = true;
                u.tpl = &(*modelTemplates)[cls_no];
                if (cls_no < modProps->size()) u.props = (*modProps)[cls_no];
                const Image& color = b.frames[(size_t)i][0];
                const Image* depth_img = b.frames[(size_t)i].size() >= 2 ? &b.frames[(size_t)i][1] : nullptr;
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
                    u.prep = u.pp.prepare_groups(dst, *u.tpl, &frame_times[(size_t)i * nc + c]);   // the colour counts follow, one GPU call per HSV range
                    if (depth_counts_on && u.depth) u.pp.depth_queries(u.prep, *u.tpl, first + i, u.dq);
                }
                else u.prep = u.pp.prepare(dst, static_cast<const uint8_t*>(color.data), color.stride, *u.tpl, u.props, -1, &frame_times[(size_t)i * nc + c]);
            }
        });
    }
    st.pool->submit_many(grouping, std::move(grouping_tasks));
    st.pool->wait(grouping);
    static const bool post_trace = std::getenv("LM_POST_TRACE") != nullptr;      // where a batch's post-processing spends its wall time (stderr)
    const clk::time_point t_grouped = clk::now();
    if (!grouping.error.empty()) { error = grouping.error; return false; }
    bool any = false;
    for (size_t c = 0; c < nc; ++c) for (int i = 0; i < n; ++i) any = any || !out_matches[c][(size_t)i].empty();
    for (const Unit& u : units) if (u.live && !u.pp.lastError().empty()) error = u.pp.lastError();
    // ---- step 2 + 3: colour counts of the WHOLE batch on the GPU, depth checks + poses on the pool.
    // Colour: classes with the same HSV range share ONE check (one mask launch for the batch's frames, one hull launch for all their
    // matches) on the detector's colour-check stream, beside whatever the other lane is matching; the first range's check is only BEGUN
    // here and collected after the first wave of depth checks has been handed to the pool.
    // Depth: the reference walks a group's matches in order until numberWantedPoses poses are found -- a chain of up to N dependent
    // nth_element calls per group, and the longest chain bounded the batch (measured: 1.8 ms of a 2.3-ms post-processing with 0.8 ms of
    // work per thread).  A match's depth check is a pure function of the frame, so a group's checks are evaluated in WAVES of 1, 2, 4,
    // .. 16 matches on the pool and the walk (PostProcessor::accept_range) consumes a wave when it is complete: same verdicts, same
    // poses; at most the checks of the wave in which the walk stops are evaluated in vain (none when the first match passes, the
    // common case on real frames).  The first wave starts before the colour counts are back: a token per group holds its walk.
    struct GroupState {     // what the walk of one group reads and writes (host/GroupWaves.h schedules the evaluations and the walk)
        const Unit* u = nullptr; size_t g = 0;
        std::vector<PostProcessor::MatchVerdict> v;      // one per match of the group
        std::vector<ObjectPose> poses;
        std::mutex mu; PostProcessor::Times tm;
        void add(const PostProcessor::Times& t) { std::lock_guard<std::mutex> g_(mu); tm.add(t); }
    };
    size_t n_groups = 0;
    for (const Unit& u : units) if (u.live) n_groups += u.prep.groups.size();
    std::vector<GroupState> runs(n_groups);
    std::vector<size_t> lengths(n_groups);
    {
        size_t at = 0;
        for (const Unit& u : units)
            if (u.live) for (size_t g = 0; g < u.prep.groups.size(); ++g) {
                runs[at].u = &u; runs[at].g = g; lengths[at] = u.prep.groups[g].matchIndices.size();
                runs[at].v.assign(lengths[at], PostProcessor::MatchVerdict());
                ++at;
            }
    }
    std::atomic<bool> depth_ready{false};          // r06: the depth checks' GPU counts are in (they arrive after the walks have started)
    WorkerPool::Group finishing;
    GroupWaves waves(*st.pool, finishing, lengths,
        // a match's checks, on any pool thread.  early (the first wave, before the GPU's colour counts are back): the depth check runs ahead
        // of the colour verdict; otherwise the depth check only runs behind a passed colour check, as in the reference (the host colour
        // check computes its verdict here)
        [&runs, &depth_ready](size_t gi, size_t k, bool early) {
            GroupState& r = runs[gi];
            const Unit& u = *r.u;
            const uint32_t idx = u.prep.groups[r.g].matchIndices[k];
            const lm_match_t& m = (*u.matches)[idx];
            PostProcessor::MatchVerdict& v = r.v[k];
            PostProcessor::Times t;
            if ((size_t)m.template_id < u.tpl->size()) {
                if (!early) { v.colour_ok = u.pp.colour_ok(u.prep, idx, m); if (v.colour_ok) u.pp.depth_part(u.prep, idx, m, u.depth, *u.tpl, v, &t, depth_ready.load(std::memory_order_acquire)); }
                else u.pp.depth_part(m, u.depth, *u.tpl, v, &t);
            }
            r.add(t);
        },
        // the reference's walk over a complete wave (PostProcessor::accept_range); true = the group has its poses
        [&runs](size_t gi, size_t from, size_t to) {
            GroupState& r = runs[gi];
            const Unit& u = *r.u;
            const std::vector<lm_match_t>& ms = *u.matches;
            if (u.prep.failed) return true;
            for (size_t k = from; k < to; ++k) {          // (first wave of the GPU colour check: the verdicts came in after the depth checks)
                const uint32_t idx = u.prep.groups[r.g].matchIndices[k];
                if (!r.v[k].colour_ok && u.prep.gpu && (size_t)ms[idx].template_id < u.tpl->size()) r.v[k].colour_ok = u.pp.colour_ok(u.prep, idx, ms[idx]);
            }
            PostProcessor::Times t;
            const bool done = u.pp.accept_range(u.prep, r.g, ms, *u.tpl, from, to, r.v.data() + from, r.poses, &t);
            r.add(t);
            return done;
        });
    for (Unit& u : units) if (u.live) u.matches = &out_matches[u.c][(size_t)u.i];
    // r06: the depth checks' early verdicts for ALL matches of the surviving groups, counted on the GPU in one launch ahead of the colour work on the same
    // stream (the frames are resident, translated as the depth check wants them): about four checks in five then need no pass over the frame at all
    std::vector<lm_depth_query> dq;
    std::vector<size_t> dq_at(units.size(), 0);
    bool depth_begun = false;
    if (gpuColorCheck && depth_counts_on) {
        for (size_t a = 0; a < units.size(); ++a) {
            dq_at[a] = dq.size();
            if (units[a].live) dq.insert(dq.end(), units[a].dq.begin(), units[a].dq.end());
        }
    }
    // (the counts are enqueued BEHIND the first colour check on the stream and collected after the walks have been released: nothing waits for them --
    // a check evaluated before they are in takes the host's own crop pass, same verdict; measured r06: waiting for them ahead of the release cost 55 us
    // per batch alone and 40-50 us per frame in the streamed steady state, where the next batch's kernels hold the GPU)
    auto begin_depth_counts = [&] {
        if (depth_begun || dq.empty()) return;
        if (lm_depth_counts_begin(detector, dq.data(), dq.size()) == LM_OK) depth_begun = true;
        else error = lm_last_error();          // (the host's own crop pass decides then: same verdicts)
    };
    if (gpuColorCheck) {
        // the units by HSV range; the first range's check is asynchronous
        std::vector<char> done(units.size(), 0);
        bool first_range = true;
        for (size_t a = 0; a < units.size(); ++a) {
            if (done[a] || !units[a].live) continue;
            std::vector<size_t> same;
            std::vector<lm_match_t> todo;
            std::vector<int32_t> slot_of;
            for (size_t q = a; q < units.size(); ++q) {
                if (done[q] || !units[q].live) continue;
                bool eq = true;
                for (int k = 0; k < 3; ++k) eq = eq && units[a].props.lowerColorRange[k] == units[q].props.lowerColorRange[k] && units[a].props.upperColorRange[k] == units[q].props.upperColorRange[k];
                if (!eq) continue;
                done[q] = 1; same.push_back(q);
                todo.insert(todo.end(), units[q].prep.todo.begin(), units[q].prep.todo.end());
                slot_of.insert(slot_of.end(), units[q].prep.todo.size(), (int32_t)(first + units[q].i));
            }
            std::vector<int64_t> gin(todo.size()), gboth(todo.size());
            const clk::time_point t_c = clk::now();
            int crc = lm_color_check_begin_slots(detector, slot_of.data(), units[a].props.lowerColorRange, units[a].props.upperColorRange, todo.data(), todo.size());
            const clk::time_point t_cb = clk::now();
            if (first_range) begin_depth_counts();
            if (crc == LM_OK && first_range) waves.start(true);
            const clk::time_point t_ws = clk::now();
            if (post_trace) std::fprintf(stderr, "post-trace: colour check begun in %.0f us (%zu matches), first wave submitted in %.0f us\n", secs(t_c, t_cb) * 1e6, todo.size(), secs(t_cb, t_ws) * 1e6);      // while the GPU counts: the first wave of every group (all ranges' groups: their walks wait for the tokens)
            if (crc == LM_OK) crc = lm_color_check_end(detector, gin.data(), gboth.data());
            if (crc != LM_OK) {
                // loud, never a silent switch of implementation (frames of more than 4992 rows: setGpuColorCheck(false))
                error = lm_last_error();
                for (size_t q : same) { units[q].prep.failed = true; }
                if (first_range) waves.start(true);      // (every group holds a token below; groups already started are left alone)
                first_range = false;
                continue;
            }
            first_range = false;
            PostProcessor::times().colour += secs(t_c, clk::now()); PostProcessor::times().colour_checks += (long)todo.size();
            size_t at = 0;
            for (size_t q : same) {
                PostProcessor::set_counts(units[q].prep, gin.data() + at, gboth.data() + at);
                at += units[q].prep.todo.size();
            }
        }
        waves.start(true);                // (no live unit at all: nothing to start; otherwise a no-op)
        const clk::time_point t_rel = clk::now();
        waves.release_tokens();           // every colour count is in
        begin_depth_counts();             // (no colour range at all: still collected below)
        if (depth_begun) {
            std::vector<uint32_t> below(dq.size()), inside(dq.size());
            if (lm_depth_counts_end(detector, below.data(), inside.data()) == LM_OK) {
                for (size_t a = 0; a < units.size(); ++a)
                    if (units[a].live && !units[a].dq.empty())
                        PostProcessor::set_depth_counts(units[a].prep, dq.data() + dq_at[a], below.data() + dq_at[a], inside.data() + dq_at[a]);
                depth_ready.store(true, std::memory_order_release);
            } else error = lm_last_error();
        }
        if (post_trace) std::fprintf(stderr, "post-trace: grouping %.0f us, colour counts in after %.0f us more, tokens released in %.0f us\n", secs(t_post, t_grouped) * 1e6, secs(t_grouped, t_rel) * 1e6, secs(t_rel, clk::now()) * 1e6);
    } else {
        waves.start(false);
    }
    const clk::time_point t_w = clk::now();
    st.pool->wait(finishing);
    if (post_trace) std::fprintf(stderr, "post-trace: waited %.0f us for the walks (%zu groups)\n", secs(t_w, clk::now()) * 1e6, n_groups);
    if (!finishing.error.empty()) { error = finishing.error; return false; }
    for (const PostProcessor::Times& t : frame_times) PostProcessor::times().add(t);
    for (GroupState& r : runs) {
        PostProcessor::times().add(r.tm);
        if (r.poses.empty() || r.u->prep.failed) continue;
        stageTimes.poses += (long)r.poses.size();
        out_poses[r.u->c][(size_t)r.u->i].push_back(std::move(r.poses));
    }
    stageTimes.post += secs(t_post, clk::now()) - driven;
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
