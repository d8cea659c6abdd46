// PoseDetection.h -- headless equivalent of the reference's online pipeline class
//     class PoseDetection            /root/reference/include/PoseDetection.h:18-106, src/PoseDetection.cpp
// over the HighLevelLineMOD facade: class name -> index (:49), principal-point shift of both images (:54-59,192-197),
// detectTemplate (:66), first pose of every group until in_numberOfObjects (:86-92).  What the reference does besides
// (OpenGL renderer, ICP refinement :72-84 -- off in the shipped settings --, Hodan error :96-104, drawing and imshow
// :105-123) needs a display / OpenGL / OpenCV and is out of scope.
#pragma once
#include <deque>
#include <string>
#include <utility>
#include <vector>

#include "HighLevelLinemod.h"

namespace lmamd {

class PoseDetection {
public:
    // The reference's constructor reads linemod_settings.yml and the template files itself (:3-18); here the settings
    // are handed in (readSettings() fills them from the same file) and loadTemplates() is readLinemodFromFile (:142-160).
    PoseDetection(CameraParameters const& in_camParams, TemplateGenerationSettings const& in_templateSettings);
    ~PoseDetection();
    PoseDetection(const PoseDetection&) = delete;
    PoseDetection& operator=(const PoseDetection&) = delete;

    void loadTemplates();                                   // readLinemodFromFile: line->readLinemod(), class ids
    HighLevelLineMOD* lineMod() { return line; }            // to add templates in-process instead of reading files
    void refreshClassIds();                                 // after templates were added through lineMod()

    // :45-126.  in_imgs = {colour, depth}.  Like the reference, the poses are appended to in_objPose only when
    // in_displayResults is set (:105-112: the push_back sits inside `if (in_displayResults)`); nothing is drawn here.
    // getFinalObjectPoses() returns them either way.
    void detect(std::vector<Image>& in_imgs, std::string const& in_className, uint16_t const& in_numberOfObjects,
                std::vector<ObjectPose>& in_objPose, bool in_displayResults);
    // The same for a batch of frames in one pass over the GPU (BASELINE config 5): out[i] = final poses of frame i.
    // Returns false -- with the reason in lastError() -- when the class name is unknown, the batch exceeds the detector's frame
    // slots, or the match fails (capacity overflow, ...): "nothing detected" is true with empty pose lists (r04, ADVICE r3: the
    // reference's void detect cannot tell the two apart; the C ABI underneath can).
    bool detectBatch(std::vector<std::vector<Image>>& in_frames, std::string const& in_className,
                     uint16_t const& in_numberOfObjects, std::vector<std::vector<ObjectPose>>& out_objPoses);
    // Several objects in the same batch of frames (BASELINE config 5: >= 3 models): one upload and one pre-processing per
    // frame for ALL the classes (the reference calls detect once per class name on the same camera frame and pays
    // Detector::match's pyramid each time, PoseDetection.cpp:45-66).  out[c][i] = final poses of class in_classNames[c] in
    // frame i.
    bool detectBatch(std::vector<std::vector<Image>>& in_frames, std::vector<std::string> const& in_classNames,
                     uint16_t const& in_numberOfObjects, std::vector<std::vector<std::vector<ObjectPose>>>& out_objPoses);
    // The several-objects form as a STREAM (r05): Begin hands a batch to the GPU (upload + class-list match on the next free slot set)
    // and returns, End collects the oldest batch begun and runs its post-processing --
    //     detectBatchBegin(b0);  loop { detectBatchBegin(b[k + 1]); detectBatchEnd(n, out[k]); }  detectBatchEnd(n, out[last])
    // overlaps the transfer and the GPU hot path of batch k + 1 with the host's colour / depth checks of batch k.  Up to
    // HighLevelLineMOD::kBatchSets batches in flight; a batch's frames must stay valid until its End has returned.  Same poses, bit
    // for bit, as detectBatch (which is Begin + End).
    bool detectBatchBegin(std::vector<std::vector<Image>>& in_frames, std::vector<std::string> const& in_classNames);
    bool detectBatchEnd(uint16_t const& in_numberOfObjects, std::vector<std::vector<std::vector<ObjectPose>>>& out_objPoses);
    const std::vector<ObjectPose>& getFinalObjectPoses() const { return finalObjectPoses; }
    const std::string& lastError() const { return error; }

private:
    uint16_t findIndexInVector(std::string const& in_stringToFind, std::vector<std::string>& in_vectorToLookIn);   // :134-140
    HighLevelLineMOD* line;
    CameraParameters camParams;
    TemplateGenerationSettings templateSettings;
    std::vector<std::string> ids;
    std::vector<std::vector<ObjectPose>> detectedPoses;
    std::vector<ObjectPose> finalObjectPoses;
    std::string error;
    // shifted copies of one frame (translateImg works in place on clones, :54-59)
    struct Shifted { std::vector<uint8_t> color; std::vector<uint16_t> depth; };
    std::vector<Shifted> batchBufs;      // detectBatch's shifted frames, kept between calls (6 MB of fresh pages per 1280 x 960 frame otherwise)
    size_t streamBufNext = 0;            // which buffer set the next detectBatchBegin translates into (host colour check only)
    std::deque<std::pair<size_t, size_t>> streamShape;   // (classes, frames) of the batches in flight, oldest first
    void shiftFrame(const std::vector<Image>& in_imgs, Shifted& buf, std::vector<Image>& out);
    void pickFinal(const std::vector<std::vector<ObjectPose>>& groups, uint16_t nObjects, std::vector<ObjectPose>& out);
};

}  // namespace lmamd
