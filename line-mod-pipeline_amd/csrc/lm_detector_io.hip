// lm_detector_io.hip -- f2: persistence of the template bank (the library's binary format; OpenCV's linemod_templates.yml(.gz) through csrc/lm_yaml) and
// the generic FileStorage readers the facade uses for the reference's settings files.  C ABI: lm_save_bank, lm_load_bank, lm_save_yaml, lm_load_yaml, lm_yaml_*.
#include "lm_detector_impl.h"

extern "C" {

int lm_save_bank(const lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::save_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}
int lm_load_bank(lm_detector* d, const char* path) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmh::load_bank(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    d->bank_dirty = true; d->hulls_dirty = true;
    return LM_OK;
}

int lm_save_yaml(const lm_detector* d, const char* path) {
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmy::save_templates_yaml(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    return LM_OK;
}
int lm_load_yaml(lm_detector* d, const char* path) {
    if (d && any_lane_busy(d)) return fail(LM_ERR_INVALID, "a lane has a match in flight: call lm_match_end first");
    if (!d || !path) return fail(LM_ERR_INVALID, "null argument");
    std::string err;
    if (!lmy::load_templates_yaml(d->bank, d->cfg, path, err)) return fail(LM_ERR_IO, err);
    d->bank_dirty = true; d->hulls_dirty = true;
    for (Slot& sl : d->slots) sl.prepared = false;   // the file's modality parameters (thresholds) replaced the detector's: a3-a10 results are stale
    if (d->cfg.num_modalities == 2 && d->normal_lut_substitute) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            std::fprintf(stderr, "liblinemod_hip: warning: %s holds DepthNormal templates, but the built-in NORMAL_LUT is a "
                                 "substitute for OpenCV's normal_lut.i (SURVEY.md A.4): a bank WRITTEN BY OpenCV will be scored "
                                 "against differently quantised normals.  Install the real table with lm_set_normal_lut, or "
                                 "regenerate the bank with this library.\n", path);
        }
    }
    return LM_OK;
}

// Top-level scalars / number lists of a cv::FileStorage YAML file (linemod_settings.yml, models/<name>.yml,
// benchmark/pose0.yml): the host glue reads its settings through these.
// Returns the status code (LM_ERR_IO: unreadable / unparsable file; LM_ERR_INVALID: no such key) and the node.
static int yaml_top(const char* path, const char* key, lmy::Node& root, const lmy::Node** out) {
    std::string text, err;
    *out = nullptr;
    if (!lmy::read_text_file(path, text, err)) return fail(LM_ERR_IO, err);
    if (!lmy::parse(text, root, err)) return fail(LM_ERR_IO, std::string(path) + ": " + err);
    const lmy::Node* n = root.get(key);
    if (!n) return fail(LM_ERR_INVALID, std::string("no key '") + key + "' in " + path);
    *out = n;
    return LM_OK;
}
int lm_yaml_numbers(const char* path, const char* key, double* out, size_t cap, size_t* n_out) {
    if (!path || !key) return fail(LM_ERR_INVALID, "null argument");
    lmy::Node root;
    const lmy::Node* n = nullptr;
    int rc;
    if ((rc = yaml_top(path, key, root, &n))) return rc;
    if (n->kind == lmy::Node::Map && n->get("data")) n = n->get("data");   // !!opencv-matrix
    std::vector<double> v;
    double d;
    if (n->kind == lmy::Node::Nums) v = n->nums;
    else if (n->number(&d)) v.push_back(d);
    else return fail(LM_ERR_INVALID, std::string("'") + key + "' is not numeric");
    if (n_out) *n_out = v.size();
    if (out) {
        if (cap < v.size()) return fail(LM_ERR_INVALID, "buffer too small");
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    }
    return LM_OK;
}
int lm_yaml_string(const char* path, const char* key, char* out, size_t cap) {
    if (!path || !key || !out || !cap) return fail(LM_ERR_INVALID, "null argument");
    lmy::Node root;
    const lmy::Node* n = nullptr;
    int rc;
    if ((rc = yaml_top(path, key, root, &n))) return rc;
    if (n->kind != lmy::Node::Scalar) return fail(LM_ERR_INVALID, std::string("'") + key + "' is not a scalar");
    if (n->scalar.size() + 1 > cap) return fail(LM_ERR_INVALID, "buffer too small");
    std::memcpy(out, n->scalar.c_str(), n->scalar.size() + 1);
    return LM_OK;
}


}  // extern "C"
