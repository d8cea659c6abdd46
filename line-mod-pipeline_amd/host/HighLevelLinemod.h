// HighLevelLinemod.h -- C++ host facade over the C ABI (include/linemod_hip.h).
//
// Mirrors the public interface of the reference's wrapper class
//     class HighLevelLineMOD            /root/reference/include/HighLevelLinemod.h:21-99
// (same method names, argument meaning and bool/void error behaviour) without OpenCV or GLM types,
// neither of which exists on the GPU box.  Where the reference holds
//     cv::Ptr<cv::linemod::Detector> detector;   (HighLevelLinemod.h:102)
// this class holds an lm_detector* and every call that crossed that seam goes through liblinemod_hip.so.
//
// Scope: the hot path behind detectTemplate (Detector::match, GPU), the detector
// queries, the host glue of 8f-1 (match post-processing into ObjectPose, PostProcess.{h,cpp}) and of
// 8f-3 (addTemplate with its in-plane rotation sweep; the resampler lives in TemplateGenerator.cpp).
// and of 8f-2: writeLinemod / readLinemod use the reference's own file, linemod_templates.yml.gz, in
// cv::FileStorage's YAML layout (csrc/lm_yaml.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/linemod_hip.h"

namespace lmamd {

// cv::Mat stand-in: a borrowed view.  type 0 = CV_8UC3 (BGR), 1 = CV_16UC1 (depth mm), 2 = CV_8UC1 (mask).
struct Image {
    const void* data = nullptr;
    int width = 0, height = 0;
    size_t stride = 0;  // bytes per row, 0 = dense
    int type = 0;
    // r04: a translation still to be applied (zeros shifted in): detectTemplatesBatch applies it while the staging buffer of the upload
    // is filled (lm_upload_frame_shifted) and the host depth check reads the depth image through it -- PoseDetection's principal-point
    // shift without a translated host copy.  Both images of a frame carry the same value; needs the GPU colour check (the host colour
    // check reads the colour image as it is).
    int shift_x = 0, shift_y = 0;
    // r05: `data` lies in PINNED host memory (lm_host_alloc, hipHostMalloc, hipHostRegister): the batch entry points hand it to the DMA
    // engine as it is (lm_upload_frame_pinned[_shifted]) -- no staging copy; it must stay untouched until the batch has been collected.
    bool pinned = false;
};

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Quat { float w = 1, x = 0, y = 0, z = 0; };
struct Rect { int x = 0, y = 0, width = 0, height = 0; };

// defines.h:37-45
struct ObjectPose {
    Vec3 translation;
    Quat quaternions;
    Rect boundingBox;
};

// defines.h:47-57 (the fields the hot path reads)
struct CameraParameters {
    float fx = 0, fy = 0, cx = 0, cy = 0;
    uint16_t videoWidth = 640, videoHeight = 480;
};

// defines.h:59-83 (the fields HighLevelLineMOD's constructor copies, HighLevelLinemod.cpp:5-24)
struct TemplateGenerationSettings {
    std::string modelFolder = "models/";
    bool onlyUseColorModality = false;
    uint16_t stepSize = 50;
    int16_t angleStart = -45, angleStop = 45, angleStep = 10;
    float detectorThreshold = 80.f;
    uint16_t percentToPassCheck = 50;
    uint16_t numberWantedPoses = 1;
    float radiusThresholdNewObject = 45.f;
    float discardGroupRatio = 35.f;
    bool useDepthImprovement = true;
    float depthOffset = 30.f;
    int device = 0;          // not in the reference: HIP device ordinal
    int shardRank = 0;       // not in the reference: template-bank shard of this process
    int shardSize = 1;
};

struct TemplatePose;       // PostProcess.h
struct ModelProperties;    // PostProcess.h

class HighLevelLineMOD {
public:
    // HighLevelLinemod.cpp:3-46: {ColorGradient, DepthNormal} with T={5,8}, or {ColorGradient} with T={2,8}
    HighLevelLineMOD(CameraParameters const& in_camParams, TemplateGenerationSettings const& in_templateSettings);
    ~HighLevelLineMOD();
    HighLevelLineMOD(const HighLevelLineMOD&) = delete;
    HighLevelLineMOD& operator=(const HighLevelLineMOD&) = delete;

    std::vector<std::string> getClassIds();   // :53-56
    uint16_t getNumClasses();                 // :58-61
    uint32_t getNumTemplates();               // :63-66

    // :138-190.  in_imgs = {colour} or {colour, depth}; a colour-only detector ignores the depth image for
    // matching exactly like the reference (:146-151) but still uses it for the depth check (:394-399).
    // Runs the match on the GPU, then the reference's post-processing (grouping, colour / depth checks,
    // poses) when template poses are known for the class.  Returns true iff the raw match list is non-empty.
    bool detectTemplate(std::vector<Image>& in_imgs, uint16_t in_classNumber);

    // The same for a batch of frames in one lm_match_batch (not in the reference, which sees one camera frame at a
    // time; BASELINE config 5): per frame the raw match list and the pose groups detectTemplate would produce.
    bool detectTemplateBatch(std::vector<std::vector<Image>>& in_frames, uint16_t in_classNumber,
                             std::vector<std::vector<lm_match_t>>& out_matches,
                             std::vector<std::vector<std::vector<ObjectPose>>>& out_poses);
    // Several classes against the same batch of frames: upstream's match(sources, threshold, matches, class_ids) takes a
    // class LIST (:145,152) and builds the linear memories once.  ONE upload and ONE pre-processing (a3-a10) per frame for
    // all the named classes (lm_match_batch_classes); the mixed lists are split by class and every (class, frame) is
    // post-processed like detectTemplate does.  out_*[c][i]: class in_classNumbers[c], frame i.
    bool detectTemplatesBatch(std::vector<std::vector<Image>>& in_frames, const std::vector<uint16_t>& in_classNumbers,
                              std::vector<std::vector<std::vector<lm_match_t>>>& out_matches,
                              std::vector<std::vector<std::vector<std::vector<ObjectPose>>>>& out_poses);
    // The same as a STREAM (r05): up to kBatchSets batches in flight.  Begin hands the frames to the detector's next free
    // slot set (staging copies of pageable frames on the host pool, transfers on the copy streams), enqueues the class-list match on
    // that set's lane and returns; End collects the OLDEST batch begun (match lists, then the reference's post-processing with the colour
    // counts on the GPU's colour-check stream and the depth checks on the host pool).  With
    //     Begin(0);  loop { Begin(k + 1); End(k); }  End(last)
    // the upload and the GPU hot path of batch k + 1 run while the host post-processes batch k.  The frames of a batch must stay valid
    // and unchanged until its End has returned (the depth check reads the depth image, the DMA engine reads pinned frames).  Results are
    // the same lists and poses, bit for bit, as detectTemplatesBatch's, which is Begin + End.  Returns of End as detectTemplatesBatch.
    bool detectTemplatesBatchBegin(std::vector<std::vector<Image>>& in_frames, const std::vector<uint16_t>& in_classNumbers);
    bool detectTemplatesBatchEnd(std::vector<std::vector<std::vector<lm_match_t>>>& out_matches,
                                 std::vector<std::vector<std::vector<std::vector<ObjectPose>>>>& out_poses);
    int batchesInFlight() const;
    static constexpr int kBatchSets = 3;     // (r05: two until the host's share of a batch fell below the link's + the GPU's; with three, a batch's transfer, the match of the one before
                                             //  it and the post-processing of the one before that run at the same time)
    // colour checks of the post-processing on the GPU (default) or on the host (the reference's one-match-at-a-time way)
    void setGpuColorCheck(bool on) { gpuColorCheck = on; }
    // r06: the depth checks' early verdicts from counts taken on the GPU (lm_depth_counts_begin) in the streamed / batched path; off: every check's crop pass
    // runs on the host (same verdicts, same poses)
    // 0: never; 1 (default): when no other batch is in flight (the serial detectBatch: -8 % per frame, measured r06 -- in the streamed steady state the
    // next batch's kernels hold the GPU and the counts buy the post-processing nothing); 2: always.
    void setGpuDepthCounts(int mode) { gpuDepthCounts = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
    bool usesGpuColorCheck() const { return gpuColorCheck; }
    // host threads of detectTemplatesBatch's post-processing (r04): the groups of all (class, frame) pairs of a batch are independent
    // once their colour counts are known, and the reference's depth check (an nth_element over the template's bounding box per
    // tested match) is 80 % of the batch's wall time on ONE thread.  0 (default) = one per hardware thread, at most 32; 1 = serial.
    // r05: the threads are a persistent pool owned by this object (WorkerPool.h; built on first use, rebuilt when the number changes
    // while no batch is in flight); 0 = one per CPU this process may use (hardware threads cut by the cgroup CPU quota), at most 32.
    void setPostThreads(int n);
    int postThreadsInUse() const;
    // frames per batch: a slot set of the detector (the detector is created with kBatchSlots * kBatchSets frame slots)
    static constexpr int kBatchSlots = 8;

    // The raw, sorted, unique match list of the last detectTemplate (the reference's private `matches`).
    const std::vector<lm_match_t>& getMatches() const { return matches; }
    std::vector<std::vector<ObjectPose>> getObjectPoses() { return posesMultipleObj; }   // :322-325

    // :256-320.  "linemod_templates.yml.gz" (cv::FileStorage YAML of Detector::write + writeClass) plus the
    // reference's own raw pose file "linemod_tempPosFile.bin" (:272-284, :302-318).
    void writeLinemod();
    void readLinemod();
    // readLinemod from named files: a template file in cv::FileStorage's YAML layout (".yml" / ".yml.gz") or this library's
    // compact bank file (anything else: lm_load_bank), plus the raw pose file.  readLinemod() = the reference's two fixed names.
    void readLinemodFrom(const std::string& templateFile, const std::string& poseFile);

    // Where the wall time of the batch entry points went (not in the reference; bench.py's pose_e2e leg): seconds accumulated since
    // resetTimes() -- upload = lm_upload_frame of every frame (staging copy + H2D enqueue), match = the hot path (lm_match_batch*,
    // a3-a15 on the GPU, synchronous), post = the reference's post-processing of every (class, frame): grouping, colour check
    // (GPU counts or host masks), depth check, poses.
    // r05, streamed use: upload = wall time inside Begin (staging copies on the pool + enqueue), match = time End spent WAITING for the
    // lane (0 when the GPU finished behind the previous batch's post-processing), post = the rest of End; staging_cpu = the staging
    // copies' time summed over the pool's threads.
    struct StageTimes { double upload = 0, match = 0, post = 0, staging_cpu = 0; long frames = 0, matches = 0, poses = 0; };
    const StageTimes& times() const { return stageTimes; }
    void resetTimes() { stageTimes = StageTimes(); }

    // :68-110.  From one rendered colour+depth pair: one template per in-plane rotation angleStart..angleStop
    // (warpAffine of mask / binarised colour / depth, erode, Detector::addTemplate), with the template
    // pose and median depth the post-processing needs (:102-107).  false as soon as one extraction fails.
    bool addTemplate(std::vector<Image>& in_images, const std::string& in_modelName, Vec3 in_cameraPosition);
    void pushBackTemplates();                 // :517-521

    // readColorRanges (:523-543) reads models/<name>.yml; this sets the same data directly.
    void setColorRange(uint16_t classNumber, const double lowerHSV[3], const double upperHSV[3]);

    lm_detector* handle() { return detector; }
    const std::string& lastError() const { return error; }

private:
    lm_detector* detector = nullptr;
    bool onlyColorModality;
    uint16_t videoWidth, videoHeight;
    float fy;
    TemplateGenerationSettings settings;
    float detectorThreshold;
    std::vector<lm_match_t> matches;
    std::vector<std::vector<ObjectPose>> posesMultipleObj;
    std::vector<TemplatePose>* templates;                     // current class being generated (:164 `templates`)
    std::vector<std::vector<TemplatePose>>* modelTemplates;   // :165
    std::vector<ModelProperties>* modProps;                   // :169
    std::string error;
    bool gpuColorCheck = true;
    int gpuDepthCounts = 1;
    int postThreads = 0;
    StageTimes stageTimes;
    struct Stream;                       // the batches in flight + the pool (HighLevelLinemod.cpp)
    struct Batch;                        // one batch in flight
    Stream* stream_ = nullptr;
    Stream& stream();
    bool finishBegin(Batch& b);          // second half of Begin: staging copies in -> transfers + match enqueued on the batch's lane
    void readColorRanges();
    std::vector<std::vector<ObjectPose>> postProcess(const std::vector<lm_match_t>& in_matches, const Image& color,
                                                     const Image* depth_img, uint16_t in_classNumber, int gpu_slot);
};

// utility.cpp: linemod_settings.yml -> the two settings structs (keys as in the reference's file).
bool readSettings(const std::string& file, CameraParameters& cam, TemplateGenerationSettings& ts);

}  // namespace lmamd
