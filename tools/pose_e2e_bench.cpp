// bench.py --config 5, leg `pose_e2e`: BASELINE configs[4] as it is worded -- "end-to-end PoseDetection incl.
// depth/color check" -- timed around lmamd::PoseDetection::detectBatch on the bench's own synthetic workload: principal-point
// shift of both images on the host (PoseDetection.cpp:54-59), upload, ONE class-list match on the GPU (the hot path bench.py's
// `value` measures), then the reference's post-processing of every (class, frame): grouping, colour check (GPU counts),
// depth check, poses (HighLevelLinemod.cpp:157-175,206-253,424-515).  Prints one JSON object.
// usage: pose_e2e_bench <bank file> <pose file> <frames.raw> <W> <H> <n frames> <threshold> <iterations> <host colour check 0|1>
//   frames.raw: per frame W*H*3 bytes BGR then W*H u16 depth.  Build: g++ -O2 tools/pose_e2e_bench.cpp host/*.cpp -llinemod_hip
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../line-mod-pipeline_amd/host/HighLevelLinemod.h"
#include "../line-mod-pipeline_amd/host/PoseDetection.h"
#include "../line-mod-pipeline_amd/host/PostProcess.h"

using namespace lmamd;

int main(int argc, char** argv) {
    if (argc < 10) { std::fprintf(stderr, "usage: see the head of tools/pose_e2e_bench.cpp\n"); return 2; }
    const std::string bank = argv[1], pose = argv[2], raw = argv[3];
    const int W = std::atoi(argv[4]), H = std::atoi(argv[5]), NF = std::atoi(argv[6]);
    const float thr = (float)std::atof(argv[7]);
    const int iters = std::atoi(argv[8]);
    const bool host_colour = std::atoi(argv[9]) != 0;
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
    std::ifstream f(raw, std::ios::binary);
    std::vector<std::vector<uint8_t>> fb((size_t)NF, std::vector<uint8_t>((size_t)W * H * 3));
    std::vector<std::vector<uint16_t>> fd((size_t)NF, std::vector<uint16_t>((size_t)W * H));
    for (int i = 0; i < NF; ++i) {
        f.read(reinterpret_cast<char*>(fb[(size_t)i].data()), (std::streamsize)fb[(size_t)i].size());
        f.read(reinterpret_cast<char*>(fd[(size_t)i].data()), (std::streamsize)(fd[(size_t)i].size() * 2));
        if (!f) { std::fprintf(stderr, "frames file too short\n"); return 1; }
    }
    std::vector<std::vector<Image>> frames((size_t)NF, std::vector<Image>(2));
    for (int i = 0; i < NF; ++i) {
        frames[(size_t)i][0].data = fb[(size_t)i].data(); frames[(size_t)i][0].width = W; frames[(size_t)i][0].height = H; frames[(size_t)i][0].type = 0;
        frames[(size_t)i][1].data = fd[(size_t)i].data(); frames[(size_t)i][1].width = W; frames[(size_t)i][1].height = H; frames[(size_t)i][1].type = 1;
    }
    std::vector<std::vector<std::vector<ObjectPose>>> poses;
    using clk = std::chrono::steady_clock;
    double total = 0;
    for (int it = -2; it < iters; ++it) {              // two untimed passes first
        if (it == 0) { line.resetTimes(); PostProcessor::resetTimes(); }
        const clk::time_point t0 = clk::now();
        if (!pd.detectBatch(frames, names, 1, poses)) { std::fprintf(stderr, "detectBatch failed: %s\n", pd.lastError().c_str()); return 1; }
        if (it >= 0) total += std::chrono::duration<double>(clk::now() - t0).count();
    }
    const HighLevelLineMOD::StageTimes& st = line.times();
    const PostProcessor::Times pt = PostProcessor::times();
    long final_poses = 0;
    for (const auto& c : poses) for (const auto& fr : c) final_poses += (long)fr.size();
    const double nf = (double)iters * NF;
    std::printf("{\"frames\": %d, \"iterations\": %d, \"classes\": %zu, \"templates\": %u, \"us_per_frame\": %.2f, "
                "\"shift_us_per_frame\": %.2f, \"upload_us_per_frame\": %.2f, \"hot_path_us_per_frame\": %.2f, \"post_us_per_frame\": %.2f, "
                "\"matches_per_frame\": %.1f, \"grouped_poses_per_frame\": %.1f, \"final_poses_last_batch\": %ld, \"colour_check\": \"%s\", "
                "\"post_us_per_frame_by_part\": {\"grouping\": %.2f, \"colour_check\": %.2f, \"depth_check\": %.2f, \"poses\": %.2f}, "
                "\"per_frame_counts\": {\"groups\": %.1f, \"colour_checks\": %.1f, \"depth_checks\": %.1f, \"poses\": %.1f}}\n",
                NF, iters, names.size(), (unsigned)line.getNumTemplates(), total / nf * 1e6,
                (total - st.upload - st.match - st.post) / nf * 1e6, st.upload / nf * 1e6, st.match / nf * 1e6, st.post / nf * 1e6,
                (double)st.matches / nf, (double)st.poses / nf, final_poses, host_colour ? "host" : "gpu",
                pt.grouping / nf * 1e6, pt.colour / nf * 1e6, pt.depth / nf * 1e6, pt.pose / nf * 1e6,
                (double)pt.groups / nf, (double)pt.colour_checks / nf, (double)pt.depth_checks / nf, (double)pt.poses / nf);
    return 0;
}
