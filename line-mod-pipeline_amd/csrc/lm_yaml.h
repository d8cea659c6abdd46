// lm_yaml.h -- the subset of YAML 1.0 that cv::FileStorage writes, enough for the files the reference
// reads and writes:
//   linemod_templates.yml.gz   cv::linemod::Detector::write + writeClass   (HighLevelLinemod.cpp:256-270, 292-303)
//   linemod_settings.yml, models/<name>.yml, benchmark/pose0.yml            (utility.cpp, HighLevelLinemod.cpp:523-543)
// Block mappings (keys may contain spaces), block sequences, flow sequences that wrap over lines,
// "!!opencv-matrix" tags, comments, quoted strings.  Files may be gzip-compressed (zlib).
// Host-only; OpenCV itself is not available in this image, so the writer restates FileStorage's layout.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "lm_host.h"

namespace lmy {

struct Node {
    enum Kind { Null, Scalar, Seq, Map, Nums } kind = Null;
    std::string scalar;                                  // Scalar (quotes removed)
    std::vector<Node> seq;                               // Seq
    std::vector<std::pair<std::string, Node>> map;       // Map, file order
    std::vector<double> nums;                            // Nums: a flow sequence of numbers only
    const Node* get(const char* key) const;
    bool number(double* out) const;                      // Scalar that parses as a number
    size_t size() const { return kind == Seq ? seq.size() : kind == Nums ? nums.size() : kind == Map ? map.size() : 0; }
};

bool read_text_file(const char* path, std::string& out, std::string& err);   // plain or gzip
bool parse(const std::string& text, Node& root, std::string& err);

// cv::linemod::Detector::write(fs) + "classes" [ { writeClass } ... ]
bool save_templates_yaml(const lmh::Bank& bank, const lm_config& cfg, const char* path, std::string& err);
// Detector::read(fs.root()) + readClass per entry of "classes".  The file's pyramid_levels, T and modality
// types must equal the detector's (its buffers are sized for them); the modality parameters are copied into
// cfg.  Classes already present are left alone (std::map::insert semantics upstream).
bool load_templates_yaml(lmh::Bank& bank, lm_config& cfg, const char* path, std::string& err);

}  // namespace lmy
