// lm_host.h -- host-only parts of liblinemod_hip.so: the template bank as the caller sees it
// (classes -> template pyramids -> features, cv::linemod::Detector's class_templates map), its
// translation into the device bank of one shard, the bank file format, the default tables and the
// host-side sort/merge of match records.
#pragma once
#include <string>
#include <vector>

#include "../../include/linemod_hip.h"
#include "lm_common.h"

namespace lmh {

struct Template {  // cv::linemod::Template
    int width = 0, height = 0, pyramid_level = 0;
    std::vector<lm_feature> features;
};
typedef std::vector<Template> TemplatePyramid;  // [level * M + modality]

struct ClassEntry {
    std::string id;
    std::vector<TemplatePyramid> pyramids;  // index = template_id
};

struct Bank {
    std::vector<ClassEntry> classes;
    int find(const std::string& id) const;
    // returns class index or -1 (err set)
    int add_class(const std::string& id, int n_templates, const lm_template_desc* descs, const lm_feature* features,
                  int levels, int modalities, std::string& err);
    int add_pyramid(const std::string& id, TemplatePyramid&& tp);  // returns template id
};

// What every template pyramid of the bank must satisfy, whoever built it (lm_add_class, lm_load_bank, lm_load_yaml --
// the YAML reader parses files this library did not write): levels x modalities templates ordered [level * M + modality],
// at most 63 features each (upstream CV_Assert), at least one feature per level (similarity divides by their number),
// labels 0..7, coordinates and sizes in 0..32767 (16-bit fields of the device records).
bool check_template_pyramid(const TemplatePyramid& tp, int levels, int modalities, std::string& err);
// The modality parameters a detector can run with (lm_create, lm_load_yaml): finite non-negative thresholds, feature
// counts in 1..63.
bool check_modality_params(const lm_config& cfg, std::string& err);

// Host image of the device bank of one shard (uploaded verbatim by lm_detector.hip).
struct DeviceBankHost {
    int fpad = LM_SCAN_FPAD;
    std::vector<int> t_global, t_class;            // [nt] global template id / class index
    std::vector<u32> scan_off;                     // [nt][M][fpad]
    std::vector<int> scan_P, scan_n;               // [nt]
    std::vector<u32> item_t, item_chunk;           // scan work items, contiguous per class
    std::vector<int> class_item_lo, class_item_hi; // [n_classes]
    std::vector<int> class_t_lo, class_t_hi;       // bank-local template range per class
    std::vector<double> class_alg_bytes;           // sum_t sum_m F_m(t) * P(t) per class
    std::vector<double> class_load_bytes;          // bytes the scan's vector loads request per frame, per class
    std::vector<LmRefMeta> ref_meta[LM_MAX_LEVELS];
    std::vector<LmRefFeat> ref_feat[LM_MAX_LEVELS];
    // bit-plane scan (k_scan1, r05; only when the scanned level has planes): the in-bounds features of all modalities as ONE list per
    // template, in the lists' order -- bit offsets of the miss planes and the same features' nibble offsets
    int fpad1 = 0;
    std::vector<u32> off1, offn, offs3;            // [nt][fpad1]; offs3: orientation << 29 | byte offset of the feature in the modality's spread memory (the level written without response memories)
    long long items1_by_L[65] = {};                // work items when a frame takes L lanes: sum over templates of ceil(P / (128 L - 31))
    // r06, the same scan with a frame's planes in LDS (k_scanl; lds_ok: the level's planes of all modalities fit LM_SCANL_IMAGE_MAX bytes, a plane is a
    // whole number of 16-byte pieces, positions and templates fit the survivor entry): the lists once more in the LDS image's layout
    // [modality][orientation][T*T*wh / 8 bytes], without the arena's pads
    bool lds_ok = false;
    std::vector<u32> offl;                         // [nt][fpad1] (LDS byte address of the dword holding the feature's first bit) << 8 | bit shift; pad: the zero block
    std::vector<u32> offsl;                        // [nt][fpad1] orientation << 29 | byte offset in the LDS image of the spread bytes [modality][T*T*wh]
    std::vector<u32> litem;                        // lane items, template-major: template << 8 | unit of 128 positions (0xFFFFFFFF: a filler)
    std::vector<u32> lrec;                         // [n lane items][4] the same with the template's records: item, scan_n, scan_P, 0 -- ONE load per lane item
    std::vector<int> lbegin;                       // [nt + 1] first lane item of a bank-local template
};
// k_scan1's work items for L lanes per frame (chunks of 128 L - 31 positions), template-major like item_t / item_chunk; begin[t] =
// first item of bank-local template t (begin[nt] = number of items)
void build_items1(const DeviceBankHost& hb, int L, std::vector<u32>& item_t, std::vector<u32>& item_chunk, std::vector<int>& begin);

// Contiguous template_id range of shard `rank` of `size` for a class of n templates.
inline void shard_range(int n, int rank, int size, int* lo, int* hi) {
    *lo = (int)((long long)n * rank / size);
    *hi = (int)((long long)n * (rank + 1) / size);
}

// scan_list_order: how a (template, modality) list of the scanned level is ordered -- 0 ascending offsets (r01-r03), 1 dealt round-robin over the
// orientation labels, 2 descending offsets, 3 (default) greedy farthest-point order in (x, y, orientation); the sums do not depend on it
bool build_device_bank(const Bank& bank, const lm_config& cfg, const LmLevelGeom* geom, DeviceBankHost& out,
                       int scan_list_order, std::string& err);

// Convex hulls of the level-0 features (all modalities) of EVERY template of the bank, not only this shard's: the
// colour check runs on merged match lists (host/PostProcess.cpp color_check; HighLevelLinemod.cpp:113-135).
// Monotone chain: points sorted by (x, y), duplicates and collinear points dropped, counter-clockwise from the
// leftmost-lowest point -- the vertex ORDER matters, the outline is drawn edge by edge in that direction.
struct HullTable {
    std::vector<u32> class_base;   // [n_classes] index of the class's first template in hull_off
    std::vector<u32> hull_off;     // [n_templates + 1]
    std::vector<int16_t> hull_xy;  // x, y per vertex
};
void build_hull_table(const Bank& bank, int modalities, HullTable& out);

void default_similarity_lut(u8 lut[256]);
void default_normal_lut(u8 lut[8000]);

// Total order of SURVEY.md A.9 and upstream Match::operator==.
bool match_less(const lm_match_t& a, const lm_match_t& b);
bool match_eq(const lm_match_t& a, const lm_match_t& b);
void sort_unique(std::vector<lm_match_t>& v);

bool save_bank(const Bank& bank, const lm_config& cfg, const char* path, std::string& err);
bool load_bank(Bank& bank, const lm_config& cfg, const char* path, std::string& err);


// memcpy for the staging copies of uploads (r05): the destination -- a pinned staging buffer -- is read next by the DMA engine, never by
// this CPU, so the bytes go out with non-temporal stores (no read-for-ownership of the destination lines, no cache pollution: a
// third less DRAM traffic for a copy that is bound by it).  Falls back to memcpy without AVX2 or for short runs.  copy_stream_fence()
// once after a batch of calls, before the buffer is handed to the DMA engine.
void copy_stream(void* dst, const void* src, size_t n);
void copy_stream_fence();
}  // namespace lmh
