// bench.py --config 5, leg `pose_e2e`: BASELINE configs[4] as it is worded -- "end-to-end PoseDetection incl.
// depth/color check" -- timed around lmamd::PoseDetection on the bench's own synthetic workload: principal-point shift of both images
// (PoseDetection.cpp:54-59), upload, ONE class-list match on the GPU (the hot path bench.py's `value` measures), then the reference's
// post-processing of every (class, frame): grouping, colour check (GPU counts), depth check, poses (HighLevelLinemod.cpp:157-175,
// 206-253,424-515).  Three passes over the same batches, poses compared bit for bit between them:
//   serial     detectBatch, one batch at a time (r04's figure: a sum of phases)
//   pipelined  detectBatchBegin(k + 1) before detectBatchEnd(k): upload + GPU of batch k + 1 behind the host work of batch k (pageable frames)
//   pinned     the same with the frames in pinned host memory (no staging copy)
// Prints one JSON object.
// usage: pose_e2e_bench <bank file> <pose file> <frames.raw> <W> <H> <n frames> <threshold> <iterations> <host colour check 0|1> [threads]
//   frames.raw: per frame W*H*3 bytes BGR then W*H u16 depth.  Build: g++ -O2 tools/pose_e2e_bench.cpp host/*.cpp -llinemod_hip
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../line-mod-pipeline_amd/host/HighLevelLinemod.h"
#include "../line-mod-pipeline_amd/host/PoseDetection.h"
#include "../line-mod-pipeline_amd/host/PostProcess.h"

using namespace lmamd;
using clk = std::chrono::steady_clock;
typedef std::vector<std::vector<std::vector<ObjectPose>>> Poses;       // [class][frame][pose]

static bool same_poses(const Poses& a, const Poses& b) {
    if (a.size() != b.size()) return false;
    for (size_t c = 0; c < a.size(); ++c) {
        if (a[c].size() != b[c].size()) return false;
        for (size_t i = 0; i < a[c].size(); ++i) {
            if (a[c][i].size() != b[c][i].size()) return false;
            for (size_t k = 0; k < a[c][i].size(); ++k)
                if (std::memcmp(&a[c][i][k], &b[c][i][k], sizeof(ObjectPose)) != 0) return false;
        }
    }
    return true;
}

struct Pass {
    double total = 0;
    HighLevelLineMOD::StageTimes st;
    PostProcessor::Times pt;
    double gpu_us[4] = {0, 0, 0, 0};
    long long gpu_frames = 0;
    std::vector<Poses> poses;      // per iteration
};

static void print_pass(const char* name, const Pass& p, double nf, double link_us_per_frame, bool last) {
    const double wall = p.total / nf * 1e6;
    const double gpu = p.gpu_frames ? (p.gpu_us[0] + p.gpu_us[1] + p.gpu_us[2] + p.gpu_us[3]) / (double)p.gpu_frames : 0.0;
    std::printf("\"%s\": {\"us_per_frame\": %.2f, \"in_begin_us_per_frame\": %.2f, \"waiting_for_the_gpu_us_per_frame\": %.2f, \"post_us_per_frame\": %.2f, "
                "\"staging_copy_cpu_us_per_frame\": %.2f, \"gpu_hot_path_us_per_frame\": %.2f, \"gpu_share_of_wall\": %.4f, \"link_us_per_frame\": %.2f, \"link_share_of_wall\": %.4f, "
                "\"post_cpu_us_per_frame_by_part\": {\"grouping\": %.2f, \"colour_check_wall\": %.2f, \"depth_check\": %.2f, \"poses\": %.2f}, "
                "\"per_frame_counts\": {\"groups\": %.1f, \"colour_checks\": %.1f, \"depth_checks\": %.1f, \"depth_checks_decided_without_nth_element\": %.1f, \"poses\": %.1f}}%s",
                name, wall, p.st.upload / nf * 1e6, p.st.match / nf * 1e6, p.st.post / nf * 1e6, p.st.staging_cpu / nf * 1e6, gpu, wall > 0 ? gpu / wall : 0.0,
                link_us_per_frame, wall > 0 ? link_us_per_frame / wall : 0.0,
                p.pt.grouping / nf * 1e6, p.pt.colour / nf * 1e6, p.pt.depth / nf * 1e6, p.pt.pose / nf * 1e6,
                (double)p.pt.groups / nf, (double)p.pt.colour_checks / nf, (double)p.pt.depth_checks / nf, (double)p.pt.depth_decided_early / nf, (double)p.pt.poses / nf,
                last ? "" : ", ");
}

int main(int argc, char** argv) {
    if (argc < 10) { std::fprintf(stderr, "usage: see the head of tools/pose_e2e_bench.cpp\n"); return 2; }
    const std::string bank = argv[1], pose = argv[2], raw = argv[3];
    const int W = std::atoi(argv[4]), H = std::atoi(argv[5]), NF = std::atoi(argv[6]);
    const float thr = (float)std::atof(argv[7]);
    const int iters = std::atoi(argv[8]);
    const bool host_colour = std::atoi(argv[9]) != 0;
    const int threads = argc > 10 ? std::atoi(argv[10]) : 0;
    const int sleep_us = std::getenv("LM_E2E_SLEEP_US") ? std::atoi(std::getenv("LM_E2E_SLEEP_US")) : 0;
    CameraParameters cam;      // the shipped camera scaled to the frame, principal point off centre: the shift is (-12, +10) pixels
    cam.fx = 2089.74f * (float)W / 1280.f; cam.fy = 2091.38282f * (float)H / 960.f;
    cam.cx = (float)(W / 2 + 12); cam.cy = (float)(H / 2 - 10); cam.videoWidth = (uint16_t)W; cam.videoHeight = (uint16_t)H;
    TemplateGenerationSettings ts;
    ts.onlyUseColorModality = false;
    ts.detectorThreshold = thr;
    PoseDetection pd(cam, ts);
    HighLevelLineMOD& line = *pd.lineMod();
    line.readLinemodFrom(bank, pose);
    if (!line.lastError().empty()) { std::fprintf(stderr, "load failed: %s\n", line.lastError().c_str()); return 1; }
    pd.refreshClassIds();
    const std::vector<std::string> names = line.getClassIds();
    for (size_t c = 0; c < names.size(); ++c) {
        const double lo[3] = {0, 0, 50}, hi[3] = {255, 255, 255};      // V >= 50 (models/<name>.yml in the reference): dark pixels fail the colour test
        line.setColorRange((uint16_t)c, lo, hi);
    }
    line.setGpuColorCheck(!host_colour);
    if (std::getenv("LM_E2E_HOST_DEPTH_COUNTS")) line.setGpuDepthCounts(0);
    if (std::getenv("LM_E2E_GPU_DEPTH_COUNTS")) line.setGpuDepthCounts(2);      // A/B (r06): the depth checks' early verdicts from the host's own crop pass
    line.setPostThreads(threads);
    std::ifstream f(raw, std::ios::binary);
    const size_t cb = (size_t)W * H * 3, db = (size_t)W * H * 2;
    std::vector<std::vector<uint8_t>> fb((size_t)NF, std::vector<uint8_t>(cb));
    std::vector<std::vector<uint16_t>> fd((size_t)NF, std::vector<uint16_t>((size_t)W * H));
    for (int i = 0; i < NF; ++i) {
        f.read(reinterpret_cast<char*>(fb[(size_t)i].data()), (std::streamsize)cb);
        f.read(reinterpret_cast<char*>(fd[(size_t)i].data()), (std::streamsize)db);
        if (!f) { std::fprintf(stderr, "frames file too short\n"); return 1; }
    }
    // two batches of the same frames in opposite order: consecutive batches of the stream differ, as a camera's would
    auto views = [&](bool pinned, uint8_t* pin_base, bool reversed) {
        std::vector<std::vector<Image>> frames((size_t)NF, std::vector<Image>(2));
        for (int i = 0; i < NF; ++i) {
            const int src = reversed ? NF - 1 - i : i;
            Image& c = frames[(size_t)i][0]; Image& d = frames[(size_t)i][1];
            c.width = d.width = W; c.height = d.height = H; c.type = 0; d.type = 1; c.pinned = d.pinned = pinned;
            if (pinned) { c.data = pin_base + (size_t)src * (cb + db); d.data = pin_base + (size_t)src * (cb + db) + cb; }
            else { c.data = fb[(size_t)src].data(); d.data = fd[(size_t)src].data(); }
        }
        return frames;
    };
    uint8_t* pin = nullptr;
    if (lm_host_alloc((cb + db) * (size_t)NF, reinterpret_cast<void**>(&pin)) != LM_OK) { std::fprintf(stderr, "lm_host_alloc: %s\n", lm_last_error()); return 1; }
    for (int i = 0; i < NF; ++i) { std::memcpy(pin + (size_t)i * (cb + db), fb[(size_t)i].data(), cb); std::memcpy(pin + (size_t)i * (cb + db) + cb, fd[(size_t)i].data(), db); }
    std::vector<std::vector<Image>> batch[2] = {views(false, nullptr, false), views(false, nullptr, true)};
    std::vector<std::vector<Image>> pbatch[2] = {views(true, pin, false), views(true, pin, true)};

    // the link alone: the batch's pinned frames through lm_upload_frame_pinned, host waits for all of them
    double link_us_per_frame = 0;
    {
        lm_detector* det = line.handle();
        for (int rep = 0; rep < 3; ++rep) {
            const clk::time_point t0 = clk::now();
            for (int i = 0; i < NF; ++i)
                if (lm_upload_frame_pinned(det, i, pin + (size_t)i * (cb + db), 0, reinterpret_cast<const uint16_t*>(pin + (size_t)i * (cb + db) + cb), 0) != LM_OK) { std::fprintf(stderr, "upload: %s\n", lm_last_error()); return 1; }
            lm_upload_wait(det, -1);
            link_us_per_frame = std::chrono::duration<double>(clk::now() - t0).count() / NF * 1e6;
        }
    }

    const char* ahead_env = std::getenv("LM_E2E_AHEAD");
    const int ahead_cfg = std::max(1, std::min(HighLevelLineMOD::kBatchSets - 1, ahead_env ? std::atoi(ahead_env) : HighLevelLineMOD::kBatchSets - 1));
    auto run = [&](Pass& p, std::vector<std::vector<Image>>* bt, bool pipelined) -> bool {
        Poses out;
        p.poses.clear();
        // (two untimed batches first; warm-up and timed region share one loop so that the stream is in steady state when the clock starts)
        line.resetTimes(); PostProcessor::resetTimes();
        const int warm = 2, total_it = warm + iters;
        clk::time_point t0 = clk::now();
        if (!pipelined) {
            for (int it = 0; it < total_it; ++it) {
                if (it == warm) { line.resetTimes(); PostProcessor::resetTimes(); lm_set_profiling(line.handle(), 1); t0 = clk::now(); }
                if (!pd.detectBatch(bt[it & 1], names, 1, out)) { std::fprintf(stderr, "detectBatch failed: %s\n", pd.lastError().c_str()); return false; }
                if (it >= warm) p.poses.push_back(out);
            }
        } else {
            // `ahead` batches are begun before the oldest is collected: all the slot sets but one stay in flight behind the batch being post-processed
            // (LM_E2E_AHEAD: 1 = r05's first streamed form, two batches in flight).  The two host batches alternate; they are only read.
            const int ahead = ahead_cfg;
            for (int k = 0; k < ahead; ++k)
                if (!pd.detectBatchBegin(bt[k & 1], names)) { std::fprintf(stderr, "detectBatchBegin failed: %s\n", pd.lastError().c_str()); return false; }
            for (int it = 0; it < total_it; ++it) {
                // the clock starts with the stream full, as it is when the region ends (the batches begun last are never collected inside it)
                if (it == warm) { line.resetTimes(); PostProcessor::resetTimes(); lm_set_profiling(line.handle(), 1); t0 = clk::now(); }
                if (sleep_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));     // experiment: is the GPU done when nobody asks?
                if (!pd.detectBatchBegin(bt[(it + ahead) & 1], names)) { std::fprintf(stderr, "detectBatchBegin failed: %s\n", pd.lastError().c_str()); return false; }
                if (!pd.detectBatchEnd(1, out)) { std::fprintf(stderr, "detectBatchEnd failed: %s\n", pd.lastError().c_str()); return false; }
                if (it >= warm) p.poses.push_back(out);
            }
        }
        p.total = std::chrono::duration<double>(clk::now() - t0).count();
        p.st = line.times(); p.pt = PostProcessor::times();
        if (pipelined) for (int k = 0; k < ahead_cfg; ++k) { Poses drop; if (!pd.detectBatchEnd(1, drop)) { std::fprintf(stderr, "detectBatchEnd failed: %s\n", pd.lastError().c_str()); return false; } }
        int64_t launches = 0, frames = 0;
        lm_get_profile(line.handle(), p.gpu_us, nullptr, &launches, &frames);
        p.gpu_frames = frames;
        lm_set_profiling(line.handle(), 0);
        return true;
    };
    Pass serial, piped, pinned;
    if (!run(serial, batch, false)) return 1;
    if (!run(piped, batch, true)) return 1;
    if (!run(pinned, pbatch, true)) return 1;
    bool identical = serial.poses.size() == piped.poses.size() && serial.poses.size() == pinned.poses.size();
    for (size_t k = 0; identical && k < serial.poses.size(); ++k)
        identical = same_poses(serial.poses[k], piped.poses[k]) && same_poses(serial.poses[k], pinned.poses[k]);
    long final_poses = 0;
    if (!serial.poses.empty()) for (const auto& c : serial.poses.back()) for (const auto& fr : c) final_poses += (long)fr.size();
    const double nf = (double)iters * NF;
    std::printf("{\"frames\": %d, \"iterations\": %d, \"classes\": %zu, \"templates\": %u, \"host_threads\": %d, \"colour_check\": \"%s\", "
                "\"matches_per_frame\": %.1f, \"grouped_poses_per_frame\": %.1f, \"final_poses_last_batch\": %ld, \"poses_identical_across_passes\": %s, ",
                NF, iters, names.size(), (unsigned)line.getNumTemplates(), line.postThreadsInUse(), host_colour ? "host" : "gpu",
                (double)serial.st.matches / nf, (double)serial.st.poses / nf, final_poses, identical ? "true" : "false");
    print_pass("serial", serial, nf, link_us_per_frame, false);
    print_pass("pipelined", piped, nf, link_us_per_frame, false);
    print_pass("pipelined_pinned", pinned, nf, link_us_per_frame, true);
    std::printf("}\n");
    lm_host_free(pin);
    return identical ? 0 : 3;
}
