// BASELINE config 5 end to end on one GPU, headless: a batch of 8 frames of 1280x960 RGB-D with three objects each,
// three classes (.ply models), lmamd::PoseDetection::detectBatch = principal-point shift -> upload -> ONE
// lm_match_batch per class -> the reference's post-processing (grouping, colour check, depth check, poses), with the
// colour checks batched on the GPU (lm_color_check_counts).  Prints, for pytest to compare:
//   - "counts <n> <mismatches>": GPU in_hull / in_both of EVERY match of frame 0 against the host's hull_counts
//   - the final poses with the colour check on the GPU and on the host (must be identical), and where each object was put
// usage: config5_e2e <mesh.bin> [full]
//   full: the STATED bank size of BASELINE config 5 -- 162 viewpoints (subdivisions 2, no symmetry reduction) x 5 radii x
//   10 in-plane rotations = 8 100 templates per class, 24 300 in the bank (CameraViewPoints.cpp:84-124 x
//   linemod_settings.yml:21-27); takes minutes (24 300 renders / rotations / GPU quantisations), so pytest runs it only
//   with LM_CONFIG5_FULL=1 (log: profiles/r03_config5_full.log)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../line-mod-pipeline_amd/host/HighLevelLinemod.h"
#include "../../line-mod-pipeline_amd/host/PoseDetection.h"
#include "../../line-mod-pipeline_amd/host/PostProcess.h"
#include "../../line-mod-pipeline_amd/host/TemplateGenerator.h"

using namespace lmamd;

static const int W = 1280, H = 960, NF = 8;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const bool full = argc >= 3 && std::string(argv[2]) == "full";
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> mb((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(mb.data());
    const uint32_t nv = hdr[0], nf = hdr[1];
    const float* v = reinterpret_cast<const float*>(mb.data() + 8);
    const int32_t* fi = reinterpret_cast<const int32_t*>(mb.data() + 8 + (size_t)nv * 12);
    // three "models": the reference's part and two anisotropically scaled variants of it
    const float scale[3][3] = {{1.f, 1.f, 1.f}, {1.35f, 1.f, 0.8f}, {0.75f, 1.25f, 1.1f}};
    const char* names[3] = {"lagergehaeuse.ply", "wide.ply", "tall.ply"};
    Mesh mesh[3];
    for (int c = 0; c < 3; ++c) {
        mesh[c].vertices.resize(nv);
        for (uint32_t i = 0; i < nv; ++i) mesh[c].vertices[i] = Vec3{v[3 * i] * scale[c][0], v[3 * i + 1] * scale[c][1], v[3 * i + 2] * scale[c][2]};
        mesh[c].indices.assign(fi, fi + (size_t)nf * 3);
    }
    CameraParameters cam;   // the shipped camera scaled to 1280x960, principal point off centre: the shift is (-12, +10)
    cam.fx = 2089.74f; cam.fy = 2091.38282f; cam.cx = 652; cam.cy = 470; cam.videoWidth = W; cam.videoHeight = H;
    TemplateGenerationSettings ts;
    ts.onlyUseColorModality = false;
    ts.detectorThreshold = 85.f;
    ts.angleStart = -30; ts.angleStop = 30; ts.angleStep = 30;
    if (full) { ts.angleStart = -45; ts.angleStop = 45; ts.angleStep = 10; }     // linemod_settings.yml:25-27
    PoseDetection pd(cam, ts);
    HighLevelLineMOD& line = *pd.lineMod();
    // templates are rendered with the principal point at the image centre (the reference's renderer, OpenglRender.cpp:9-11)
    CameraParameters rcam = cam; rcam.cx = W / 2; rcam.cy = H / 2;
    SoftRender render(rcam);
    SymmetryProperties sym; sym.rotationallySymmetrical = true; sym.planesOfSymmetry = Vec3{1, 1, 1};
    GeneratorSettings gs; gs.startDistance = 600; gs.endDistance = 700; gs.stepSize = 50; gs.subdivisions = 3;
    SymmetryProperties gen_sym = sym;
    if (full) {   // every viewpoint of the subdivision-2 sphere, five radii
        gen_sym.rotationallySymmetrical = false; gen_sym.planesOfSymmetry = Vec3{0, 0, 0};
        gs.startDistance = 600; gs.endDistance = 800; gs.stepSize = 50; gs.subdivisions = 2;
    }
    for (int c = 0; c < 3; ++c) {
        int n = generate_templates(line, render, mesh[c], names[c], gen_sym, gs);
        std::printf("class %d templates %d\n", c, n);
        double lo[3] = {0, 0, 50}, hi[3] = {255, 150, 255};   // V >= 50: the black background fails the colour test
        line.setColorRange((uint16_t)c, lo, hi);
    }
    pd.refreshClassIds();

    // ---- scenes: per frame one rendered view of every class, pasted at its own place (in the SHIFTED frame the
    // detector sees; the camera frame is that moved back by (+12, -10))
    CameraViewPoints cams; cams.setModelProperties(sym);
    std::vector<std::vector<uint8_t>> fb(NF, std::vector<uint8_t>((size_t)W * H * 3, 0));
    std::vector<std::vector<uint16_t>> fd(NF, std::vector<uint16_t>((size_t)W * H, 0));
    int placed[NF][3][2];
    for (int i = 0; i < NF; ++i)
        for (int c = 0; c < 3; ++c) {
            const float radius = 600.f + 50.f * (float)((i + c) % 3);
            cams.createCameraViewPoints(radius, 3);
            const Vec3 vp = cams.getVertices()[(size_t)((3 * i + 5 * c) % (int)cams.getVertices().size())];
            std::vector<uint8_t> bgr, rot; std::vector<uint16_t> depth, drot;
            render.render(mesh[c], vp, bgr, depth);
            const float ang = (float)(-30 + 30 * ((i + 2 * c) % 3));
            warp_rotate_u8(bgr.data(), W, H, 3, ang, rot);
            warp_rotate_u16(depth.data(), W, H, ang, drot);
            const int ox = -380 + 380 * c + 12 * (i % 3), oy = -200 + 55 * i - 30 * c;     // object centre away from the image centre
            placed[i][c][0] = ox; placed[i][c][1] = oy;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int sx = x - ox - 12, sy = y - oy + 10;                             // camera frame = shifted frame moved by (+12, -10)
                    if (sx < 0 || sy < 0 || sx >= W || sy >= H) continue;
                    const uint16_t d = drot[(size_t)sy * W + sx];
                    if (d > 1) {
                        fd[i][(size_t)y * W + x] = d;
                        std::memcpy(&fb[i][((size_t)y * W + x) * 3], &rot[((size_t)sy * W + sx) * 3], 3);
                    }
                }
        }
    std::vector<std::vector<Image>> frames(NF, std::vector<Image>(2));
    for (int i = 0; i < NF; ++i) {
        frames[i][0].data = fb[i].data(); frames[i][0].width = W; frames[i][0].height = H; frames[i][0].type = 0;
        frames[i][1].data = fd[i].data(); frames[i][1].width = W; frames[i][1].height = H; frames[i][1].type = 1;
    }

    // ---- the two counts of colorCheck, GPU batch vs host, for every raw match of every frame and class (the frames
    // as they are, unshifted: after detectTemplateBatch frame i is resident in slot i)
    {
        PostProcessSettings ps; ps.videoWidth = W; ps.videoHeight = H;
        PostProcessor pp(line.handle(), ps);
        const double lo[3] = {0, 0, 50}, hi[3] = {255, 150, 255};
        size_t total = 0, bad = 0;
        long sum_hull = 0, sum_both = 0;
        std::vector<std::vector<uint8_t>> cmask(NF);
        for (int i = 0; i < NF; ++i) bgr2hsv_inrange(fb[i].data(), W, H, 0, lo, hi, cmask[i]);
        for (int c = 0; c < 3; ++c) {
            std::vector<std::vector<lm_match_t>> m;
            std::vector<std::vector<std::vector<ObjectPose>>> groups;
            line.detectTemplateBatch(frames, (uint16_t)c, m, groups);
            for (int i = 0; i < NF; ++i) {
                std::vector<int64_t> gi(m[i].size()), gb(m[i].size());
                if (lm_color_check_counts(line.handle(), i, lo, hi, m[i].data(), m[i].size(), gi.data(), gb.data()) != LM_OK) {
                    std::printf("lm_color_check_counts failed: %s\n", lm_last_error());
                    return 1;
                }
                for (size_t k = 0; k < m[i].size(); ++k) {
                    if (full && k >= 100) break;           // the host check fills a full-frame mask per match: a sample is enough here
                    long a = 0, b = 0;
                    pp.color_counts(m[i][k], cmask[i], &a, &b);
                    if (a != (long)gi[k] || b != (long)gb[k]) ++bad;
                    sum_hull += a; sum_both += b; ++total;
                }
            }
        }
        std::printf("counts %zu %zu hull %ld both %ld\n", total, bad, sum_hull, sum_both);
    }

    // ---- the whole path, colour checks on the GPU and on the host: ALL THREE classes in one detectBatch = one upload and
    // one pre-processing (a3-a10) per frame for the three classes (lm_match_batch_classes); "percls" = the older
    // one-call-per-class path, whose poses must be the same
    const std::vector<std::string> all_names = {names[0], names[1], names[2]};
    for (int mode = 0; mode < 2; ++mode) {
        line.setGpuColorCheck(mode == 0);
        lm_set_profiling(line.handle(), 0);        // resets the stage counters
        std::vector<std::vector<std::vector<ObjectPose>>> poses;
        pd.detectBatch(frames, all_names, 1, poses);
        int64_t sc[4];
        lm_get_stage_counts(line.handle(), sc);
        std::printf("stagecounts %s preprocess_frames %lld scan_launches %lld refine_launches %lld sort_launches %lld\n", mode == 0 ? "gpu" : "host",
                    (long long)sc[0], (long long)sc[1], (long long)sc[2], (long long)sc[3]);
        for (int c = 0; c < 3; ++c) {
            for (int i = 0; i < NF; ++i) {
                std::printf("%s frame %d class %d placed %d %d poses %zu", mode == 0 ? "gpu" : "host", i, c, placed[i][c][0], placed[i][c][1], poses[c][i].size());
                for (const ObjectPose& p : poses[c][i])
                    std::printf(" t %.9g %.9g %.9g q %.9g %.9g %.9g %.9g bb %d %d %d %d", p.translation.x, p.translation.y, p.translation.z,
                                p.quaternions.w, p.quaternions.x, p.quaternions.y, p.quaternions.z, p.boundingBox.x, p.boundingBox.y,
                                p.boundingBox.width, p.boundingBox.height);
                std::printf("\n");
            }
        }
    }
    line.setGpuColorCheck(true);
    for (int c = 0; c < 3; ++c) {
        std::vector<std::vector<ObjectPose>> poses;
        pd.detectBatch(frames, std::string(names[c]), 1, poses);
        for (int i = 0; i < NF; ++i) {
            std::printf("percls frame %d class %d placed %d %d poses %zu", i, c, placed[i][c][0], placed[i][c][1], poses[i].size());
            for (const ObjectPose& p : poses[i])
                std::printf(" t %.9g %.9g %.9g q %.9g %.9g %.9g %.9g bb %d %d %d %d", p.translation.x, p.translation.y, p.translation.z,
                            p.quaternions.w, p.quaternions.x, p.quaternions.y, p.quaternions.z, p.boundingBox.x, p.boundingBox.y,
                            p.boundingBox.width, p.boundingBox.height);
            std::printf("\n");
        }
    }
    // ---- r05: the same batches as a STREAM (detectBatchBegin / detectBatchEnd, two batches in flight): three different batches (the
    // frames in order, reversed, rotated by three), serial = one detectBatch each; streamed with pageable frames, with pinned frames
    // (DMA row-offset copy instead of a staging copy) and with the host colour check -- every pose of every pass printed for pytest
    {
        uint8_t* pin = nullptr;
        const size_t cb = (size_t)W * H * 3, db = (size_t)W * H * 2;
        if (lm_host_alloc((cb + db) * NF, reinterpret_cast<void**>(&pin)) != LM_OK) { std::printf("lm_host_alloc failed: %s\n", lm_last_error()); return 1; }
        for (int i = 0; i < NF; ++i) { std::memcpy(pin + (size_t)i * (cb + db), fb[i].data(), cb); std::memcpy(pin + (size_t)i * (cb + db) + cb, fd[i].data(), db); }
        auto make = [&](int which, bool pinned) {
            std::vector<std::vector<Image>> b(NF, std::vector<Image>(2));
            for (int i = 0; i < NF; ++i) {
                const int src = which == 0 ? i : which == 1 ? NF - 1 - i : (i + 3) % NF;
                b[i][0] = frames[src][0]; b[i][1] = frames[src][1];
                if (pinned) { b[i][0].data = pin + (size_t)src * (cb + db); b[i][1].data = pin + (size_t)src * (cb + db) + cb; b[i][0].pinned = b[i][1].pinned = true; }
            }
            return b;
        };
        auto dump = [&](const char* tag, int bi, const std::vector<std::vector<std::vector<ObjectPose>>>& poses) {
            for (int c = 0; c < 3; ++c)
                for (int i = 0; i < NF; ++i) {
                    std::printf("stream %s batch %d frame %d class %d poses %zu", tag, bi, i, c, poses[c][i].size());
                    for (const ObjectPose& p : poses[c][i])
                        std::printf(" t %.9g %.9g %.9g q %.9g %.9g %.9g %.9g bb %d %d %d %d", p.translation.x, p.translation.y, p.translation.z,
                                    p.quaternions.w, p.quaternions.x, p.quaternions.y, p.quaternions.z, p.boundingBox.x, p.boundingBox.y,
                                    p.boundingBox.width, p.boundingBox.height);
                    std::printf("\n");
                }
        };
        std::vector<std::vector<std::vector<ObjectPose>>> poses;
        for (int pass = 0; pass < 6; ++pass) {       // 0 serial, 1 streamed pageable, 2 streamed pinned, 3 streamed with the host colour check, r06: 4 the depth checks' early verdicts never / 5 always from GPU counts
            const char* tag = pass == 0 ? "serial" : pass == 1 ? "piped" : pass == 2 ? "pinned" : pass == 3 ? "hostcc" : pass == 4 ? "hostdc" : "gpudc";
            line.setGpuColorCheck(pass != 3);
            line.setGpuDepthCounts(pass == 4 ? 0 : pass == 5 ? 2 : 1);
            std::vector<std::vector<Image>> bt[3] = {make(0, pass == 2), make(1, pass == 2), make(2, pass == 2)};
            if (pass == 0) {
                for (int bi = 0; bi < 3; ++bi) {
                    if (!pd.detectBatch(bt[bi], all_names, 1, poses)) { std::printf("detectBatch failed: %s\n", pd.lastError().c_str()); return 1; }
                    dump(tag, bi, poses);
                }
                continue;
            }
            // all slot sets in flight (HighLevelLineMOD::kBatchSets = 3 batches), then one End per batch; passes 2 and 3 keep only two in flight
            const int ahead = pass == 1 ? HighLevelLineMOD::kBatchSets - 1 : 1;
            for (int bi = 0; bi < ahead; ++bi)
                if (!pd.detectBatchBegin(bt[bi], all_names)) { std::printf("detectBatchBegin failed: %s\n", pd.lastError().c_str()); return 1; }
            for (int bi = 0; bi < 3; ++bi) {
                if (bi + ahead < 3 && !pd.detectBatchBegin(bt[bi + ahead], all_names)) { std::printf("detectBatchBegin failed: %s\n", pd.lastError().c_str()); return 1; }
                if (bi == 0 && pass == 1) {     // one batch more than there are slot sets is refused, loudly, and the stream goes on
                    const bool extra = pd.detectBatchBegin(bt[2], all_names);
                    std::printf("stream %s third_begin_refused %d\n", tag, extra ? 0 : 1);
                    if (extra) return 1;
                } else if (bi == 0) std::printf("stream %s third_begin_refused 1\n", tag);
                if (!pd.detectBatchEnd(1, poses)) { std::printf("detectBatchEnd failed: %s\n", pd.lastError().c_str()); return 1; }
                dump(tag, bi, poses);
            }
            std::vector<std::vector<std::vector<ObjectPose>>> none;
            std::printf("stream %s end_without_batch_refused %d\n", tag, pd.detectBatchEnd(1, none) ? 0 : 1);
        }
        line.setGpuColorCheck(true);
        line.setGpuDepthCounts(1);
        lm_host_free(pin);
    }
    // ---- a Begin whose second half fails (here: a class the bank does not hold -> lm_match_begin_classes refuses) leaves nothing in
    // flight, so the serial API goes on working: the next detectBatch gives the poses of the first one
    {
        std::vector<std::vector<std::vector<lm_match_t>>> mm;
        std::vector<std::vector<std::vector<std::vector<ObjectPose>>>> gg;
        const bool bad = line.detectTemplatesBatch(frames, std::vector<uint16_t>{99}, mm, gg);
        const int left = line.batchesInFlight();
        const bool said = !line.lastError().empty();
        std::vector<std::vector<ObjectPose>> poses;
        const bool again = pd.detectBatch(frames, std::string(names[0]), 1, poses);
        size_t found = 0;
        if (again) for (int i = 0; i < NF; ++i) found += poses[(size_t)i].size();
        std::printf("recovery failed_begin_refused %d in_flight %d error_text %d next_call_ok %d poses %zu\n", bad ? 0 : 1, left, said ? 1 : 0, again ? 1 : 0, found);
        if (!again) std::printf("detectBatch after a failed Begin: %s\n", pd.lastError().c_str());
    }
    if (!line.lastError().empty() && line.lastError() != "no batch in flight") std::printf("last error: %s\n", line.lastError().c_str());
    return 0;
}
