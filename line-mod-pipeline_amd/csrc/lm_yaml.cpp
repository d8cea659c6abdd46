// lm_yaml.cpp -- see lm_yaml.h.
#include "lm_yaml.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lmy {

const Node* Node::get(const char* key) const {
    if (kind != Map) return nullptr;
    for (const auto& kv : map)
        if (kv.first == key) return &kv.second;
    return nullptr;
}

bool Node::number(double* out) const {
    if (kind == Nums && nums.size() == 1) { *out = nums[0]; return true; }
    if (kind != Scalar || scalar.empty()) return false;
    char* end = nullptr;
    double v = std::strtod(scalar.c_str(), &end);
    if (end == scalar.c_str() || *end) return false;
    *out = v;
    return true;
}

bool read_text_file(const char* path, std::string& out, std::string& err) {
    gzFile f = gzopen(path, "rb");   // reads plain files transparently
    if (!f) { err = std::string("cannot open ") + path; return false; }
    out.clear();
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
    const bool ok = n == 0;
    gzclose(f);
    if (!ok) err = std::string("read error in ") + path;
    return ok;
}

namespace {

struct Line { int indent; std::string text; };

// comment = '#' at the start of the content or after whitespace, outside quotes
void strip_comment(std::string& s) {
    bool q = false;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '"') q = !q;
        else if (s[i] == '#' && !q && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) { s.erase(i); break; }
    }
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
}

int bracket_balance(const std::string& s) {
    int b = 0;
    bool q = false;
    for (char c : s) {
        if (c == '"') q = !q;
        else if (!q && (c == '[' || c == '{')) ++b;
        else if (!q && (c == ']' || c == '}')) --b;
    }
    return b;
}

std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) {
        std::string o;
        for (size_t i = 1; i + 1 < s.size(); ++i) {
            if (s[i] == '\\' && i + 2 < s.size()) { ++i; o.push_back(s[i] == 'n' ? '\n' : s[i] == 't' ? '\t' : s[i]); }
            else o.push_back(s[i]);
        }
        return o;
    }
    return s;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
    return s.substr(a, b - a);
}

// cv::FileStorage files nest a handful of levels (the template file: 9).  Collections nested deeper than this are
// refused: the parser and the Node destructor recurse once per level, and a hostile or damaged file must not be able to
// run them off the stack.
#define LMY_MAX_DEPTH 64

struct Parser {
    std::vector<Line> lines;
    std::string err;
    int depth = 0;
    struct Deeper {
        int& d;
        explicit Deeper(int& dd) : d(dd) { ++d; }
        ~Deeper() { --d; }
    };

    bool fail(size_t i, const char* what) {
        err = std::string(what) + " near: " + (i < lines.size() ? lines[i].text : std::string("<end of file>"));
        return false;
    }

    // flow collection starting at s[pos] == '['; pos is left behind the closing bracket
    bool flow(const std::string& s, size_t& pos, Node& out) {
        Deeper guard(depth);
        if (depth > LMY_MAX_DEPTH) { err = "collections nested deeper than " + std::to_string(LMY_MAX_DEPTH) + " levels"; return false; }
        ++pos;
        std::vector<Node> items;
        bool all_num = true;
        for (;;) {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == ',' || s[pos] == '\t')) ++pos;
            if (pos >= s.size()) { err = "unterminated flow sequence: " + s.substr(0, 60); return false; }
            if (s[pos] == ']') { ++pos; break; }
            Node it;
            if (s[pos] == '[') {
                if (!flow(s, pos, it)) return false;
                all_num = false;
            } else {
                size_t b = pos;
                if (s[pos] == '"') { ++pos; while (pos < s.size() && s[pos] != '"') ++pos; if (pos < s.size()) ++pos; }
                else while (pos < s.size() && s[pos] != ',' && s[pos] != ']') ++pos;
                std::string tok = trim(s.substr(b, pos - b));
                it.kind = Node::Scalar;
                const bool quoted = !tok.empty() && tok.front() == '"';
                it.scalar = unquote(tok);
                double v;
                if (quoted || !it.number(&v)) all_num = false;
            }
            items.push_back(std::move(it));
        }
        if (all_num && !items.empty()) {
            out.kind = Node::Nums;
            out.nums.reserve(items.size());
            for (const Node& it : items) { double v = 0; it.number(&v); out.nums.push_back(v); }
        } else {
            out.kind = Node::Seq;
            out.seq = std::move(items);
        }
        return true;
    }

    bool value(const std::string& rest_in, size_t& i, int indent, Node& out) {
        // `rest` is what follows "key:" or "- " on line i (already consumed); i points at the NEXT line
        std::string rest = trim(rest_in);
        if (rest.rfind("!!", 0) == 0) {   // tag, e.g. !!opencv-matrix
            size_t sp = rest.find(' ');
            rest = sp == std::string::npos ? std::string() : trim(rest.substr(sp));
        }
        if (rest.empty()) {
            if (i < lines.size() && lines[i].indent > indent) return block(i, lines[i].indent, out);
            if (i < lines.size() && lines[i].indent == indent && lines[i].text[0] == '-' ) return block(i, indent, out);
            out.kind = Node::Null;
            return true;
        }
        if (rest[0] == '[') { size_t p = 0; return flow(rest, p, out); }
        out.kind = Node::Scalar;
        out.scalar = unquote(rest);
        return true;
    }

    static size_t key_end(const std::string& t) {   // position of the ':' ending a mapping key, npos if none
        bool q = false;
        for (size_t k = 0; k < t.size(); ++k) {
            if (t[k] == '"') q = !q;
            else if (!q && t[k] == '[') return std::string::npos;
            else if (!q && t[k] == ':' && (k + 1 == t.size() || t[k + 1] == ' ')) return k;
        }
        return std::string::npos;
    }

    bool block(size_t& i, int indent, Node& out) {
        Deeper guard(depth);
        if (depth > LMY_MAX_DEPTH) return fail(i, "collections nested too deeply");
        if (i >= lines.size()) { out.kind = Node::Null; return true; }
        if (lines[i].text[0] == '-' && (lines[i].text.size() == 1 || lines[i].text[1] == ' ')) {
            out.kind = Node::Seq;
            while (i < lines.size() && lines[i].indent == indent && lines[i].text[0] == '-' &&
                   (lines[i].text.size() == 1 || lines[i].text[1] == ' ')) {
                std::string rest = lines[i].text.size() > 1 ? trim(lines[i].text.substr(1)) : std::string();
                Node it;
                size_t ke = rest.empty() || rest[0] == '[' ? std::string::npos : key_end(rest);
                if (ke != std::string::npos) {
                    // "- key: value": a mapping whose first entry sits on the dash line
                    const int sub = indent + 1 + (int)(lines[i].text.size() - 1 - trim(lines[i].text.substr(1)).size());
                    lines[i].indent = sub;
                    lines[i].text = rest;
                    if (!block(i, sub, it)) return false;
                } else {
                    ++i;
                    if (!value(rest, i, indent, it)) return false;
                }
                out.seq.push_back(std::move(it));
            }
            return true;
        }
        out.kind = Node::Map;
        while (i < lines.size() && lines[i].indent == indent) {
            const std::string& t = lines[i].text;
            if (t[0] == '-' && (t.size() == 1 || t[1] == ' ')) break;
            size_t ke = key_end(t);
            if (ke == std::string::npos) return fail(i, "expected 'key: value'");
            std::string key = unquote(trim(t.substr(0, ke)));
            std::string rest = ke + 1 < t.size() ? t.substr(ke + 1) : std::string();
            ++i;
            Node v;
            if (!value(rest, i, indent, v)) return false;
            out.map.emplace_back(std::move(key), std::move(v));
        }
        if (i < lines.size() && lines[i].indent > indent) return fail(i, "unexpected indentation");
        return true;
    }
};

}  // namespace

bool parse(const std::string& text, Node& root, std::string& err) {
    Parser p;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        std::string raw = text.substr(pos, e - pos);
        pos = e + 1;
        strip_comment(raw);
        size_t ind = 0;
        while (ind < raw.size() && raw[ind] == ' ') ++ind;
        std::string body = raw.substr(ind);
        if (body.empty() || body[0] == '%' || body == "---" || body == "...") continue;
        // a flow collection may wrap: join continuation lines until the brackets balance
        // (the balance is kept incrementally: re-scanning the joined text per line is quadratic in the length of a
        // collection that never closes; a quote left open across lines makes the count approximate, which only decides
        // where joining stops -- flow() then reports the malformed collection)
        int bal = bracket_balance(body);
        while (bal > 0 && pos < text.size()) {
            size_t e2 = text.find('\n', pos);
            if (e2 == std::string::npos) e2 = text.size();
            std::string more = text.substr(pos, e2 - pos);
            pos = e2 + 1;
            strip_comment(more);
            more = trim(more);
            bal += bracket_balance(more);
            body += " ";
            body += more;
        }
        p.lines.push_back(Line{(int)ind, std::move(body)});
    }
    root = Node();
    if (p.lines.empty()) { root.kind = Node::Map; return true; }
    size_t i = 0;
    if (!p.block(i, p.lines[0].indent, root)) { err = p.err; return false; }
    if (i != p.lines.size()) { err = "trailing content near: " + p.lines[i].text; return false; }
    return true;
}

// ---- cv::linemod::Detector files -----------------------------------------------------------------
namespace {

// cv::FileStorage writes floats that are integers as "10." and others in %.8e form
std::string fs_float(float v) {
    char b[64];
    if (v == std::floor(v) && std::fabs(v) < 1e9f) std::snprintf(b, sizeof b, "%d.", (int)v);
    else std::snprintf(b, sizeof b, "%.8e", (double)v);
    return b;
}

const char* modality_name(int m) { return m == 0 ? "ColorGradient" : "DepthNormal"; }

// a double that is an int32 (the cast of anything else is undefined behaviour)
bool to_int(double d, int* out) {
    if (!(d >= -2147483648.0 && d <= 2147483647.0)) return false;   // also rejects NaN
    *out = (int)d;
    return true;
}
bool get_int(const Node& n, const char* key, int* out) {
    const Node* v = n.get(key);
    double d;
    if (!v || !v->number(&d)) return false;
    return to_int(d, out);
}
bool get_float(const Node& n, const char* key, float* out) {
    const Node* v = n.get(key);
    double d;
    if (!v || !v->number(&d)) return false;
    if (!(d >= -3.0e38 && d <= 3.0e38)) return false;
    *out = (float)d;
    return true;
}

}  // namespace

bool save_templates_yaml(const lmh::Bank& bank, const lm_config& cfg, const char* path, std::string& err) {
    const size_t plen = std::strlen(path);
    const bool gz = plen > 3 && std::strcmp(path + plen - 3, ".gz") == 0;
    gzFile f = gzopen(path, gz ? "wb" : "wbT");   // 'T': transparent (no compression)
    if (!f) { err = std::string("cannot create ") + path; return false; }
    const int M = cfg.num_modalities, L = cfg.pyramid_levels;
    std::string o;
    o.reserve(1 << 20);
    auto flush = [&](bool force) {
        if (o.size() > (1 << 19) || force) { if (!o.empty()) gzwrite(f, o.data(), (unsigned)o.size()); o.clear(); }
    };
    char b[160];
    o += "%YAML:1.0\n---\n";
    std::snprintf(b, sizeof b, "pyramid_levels: %d\nT: [ ", L); o += b;
    for (int l = 0; l < L; ++l) { std::snprintf(b, sizeof b, l ? ", %d" : "%d", cfg.T[l]); o += b; }
    o += " ]\nmodalities:\n";
    o += "   -\n      type: ColorGradient\n      weak_threshold: " + fs_float(cfg.weak_threshold) + "\n";
    std::snprintf(b, sizeof b, "      num_features: %d\n", cfg.num_features); o += b;
    o += "      strong_threshold: " + fs_float(cfg.strong_threshold) + "\n";
    if (M == 2) {
        std::snprintf(b, sizeof b,
                      "   -\n      type: DepthNormal\n      distance_threshold: %d\n      difference_threshold: %d\n"
                      "      num_features: %d\n      extract_threshold: %d\n",
                      cfg.distance_threshold, cfg.difference_threshold, cfg.depth_num_features, cfg.extract_threshold);
        o += b;
    }
    o += "classes:\n";
    for (const lmh::ClassEntry& c : bank.classes) {
        o += "   -\n      class_id: \"" + c.id + "\"\n      modalities: [ ColorGradient";
        if (M == 2) o += ", DepthNormal";
        std::snprintf(b, sizeof b, " ]\n      pyramid_levels: %d\n      template_pyramids:\n", L); o += b;
        for (size_t t = 0; t < c.pyramids.size(); ++t) {
            std::snprintf(b, sizeof b, "         -\n            template_id: %d\n            templates:\n", (int)t); o += b;
            for (const lmh::Template& tp : c.pyramids[t]) {
                std::snprintf(b, sizeof b,
                              "               -\n                  width: %d\n                  height: %d\n"
                              "                  pyramid_level: %d\n                  features:\n",
                              tp.width, tp.height, tp.pyramid_level);
                o += b;
                for (const lm_feature& ft : tp.features) {
                    std::snprintf(b, sizeof b, "                     - [ %d, %d, %d ]\n", ft.x, ft.y, ft.label);
                    o += b;
                }
                flush(false);
            }
        }
    }
    flush(true);
    const int rc = gzclose(f);
    if (rc != Z_OK) { err = std::string("write error in ") + path; return false; }
    return true;
}

bool load_templates_yaml(lmh::Bank& bank, lm_config& cfg, const char* path, std::string& err) {
    std::string text;
    if (!read_text_file(path, text, err)) return false;
    Node root;
    if (!parse(text, root, err)) { err = std::string(path) + ": " + err; return false; }
    text.clear();
    text.shrink_to_fit();
    const int M = cfg.num_modalities, L = cfg.pyramid_levels;
    // Detector::read: pyramid_levels, T, modalities
    int levels = 0;
    if (!get_int(root, "pyramid_levels", &levels)) { err = "pyramid_levels missing"; return false; }
    const Node* T = root.get("T");
    if (!T || T->kind != Node::Nums || (int)T->nums.size() != levels) { err = "T missing or not pyramid_levels long"; return false; }
    if (levels != L) { err = "file has " + std::to_string(levels) + " pyramid levels, the detector " + std::to_string(L); return false; }
    for (int l = 0; l < L; ++l)
        if (T->nums[l] != (double)cfg.T[l]) { err = "T of the file differs from the detector's at level " + std::to_string(l); return false; }
    const Node* mods = root.get("modalities");
    if (!mods || mods->kind != Node::Seq || (int)mods->seq.size() != M) { err = "the file's modalities differ from the detector's"; return false; }
    lm_config nc = cfg;
    for (int m = 0; m < M; ++m) {
        const Node& mn = mods->seq[m];
        const Node* ty = mn.get("type");
        if (!ty || ty->scalar != modality_name(m)) { err = std::string("modality ") + std::to_string(m) + " is not " + modality_name(m); return false; }
        bool ok;
        if (m == 0) ok = get_float(mn, "weak_threshold", &nc.weak_threshold) && get_int(mn, "num_features", &nc.num_features) &&
                         get_float(mn, "strong_threshold", &nc.strong_threshold);
        else ok = get_int(mn, "distance_threshold", &nc.distance_threshold) && get_int(mn, "difference_threshold", &nc.difference_threshold) &&
                  get_int(mn, "num_features", &nc.depth_num_features) && get_int(mn, "extract_threshold", &nc.extract_threshold);
        if (!ok) { err = std::string("incomplete parameters of modality ") + modality_name(m); return false; }
    }
    if (!lmh::check_modality_params(nc, err)) { err = std::string(path) + ": " + err; return false; }
    // readClass per entry
    const Node* classes = root.get("classes");
    std::vector<lmh::ClassEntry> incoming;
    if (classes && classes->kind == Node::Seq) {
        for (const Node& cn : classes->seq) {
            const Node* id = cn.get("class_id");
            const Node* cm = cn.get("modalities");
            int cl = 0;
            if (!id || id->kind != Node::Scalar) { err = "class without class_id"; return false; }
            if (!cm || cm->kind != Node::Seq || (int)cm->seq.size() != M) { err = "class " + id->scalar + ": modalities differ from the detector's"; return false; }
            for (int m = 0; m < M; ++m)
                if (cm->seq[m].scalar != modality_name(m)) { err = "class " + id->scalar + ": modalities differ from the detector's"; return false; }
            if (!get_int(cn, "pyramid_levels", &cl) || cl != L) { err = "class " + id->scalar + ": pyramid_levels differs from the detector's"; return false; }
            lmh::ClassEntry ce;
            ce.id = id->scalar;
            const Node* tps = cn.get("template_pyramids");
            if (tps && tps->kind == Node::Seq) {
                ce.pyramids.reserve(tps->seq.size());
                int expected = 0;
                for (const Node& pn : tps->seq) {
                    int tid = -1;
                    if (!get_int(pn, "template_id", &tid) || tid != expected) { err = "class " + ce.id + ": template_id out of sequence"; return false; }
                    ++expected;
                    const Node* tl = pn.get("templates");
                    if (!tl || tl->kind != Node::Seq || (int)tl->seq.size() != L * M) { err = "class " + ce.id + ": a pyramid needs levels x modalities templates"; return false; }
                    lmh::TemplatePyramid tp(tl->seq.size());
                    for (size_t j = 0; j < tl->seq.size(); ++j) {
                        const Node& tn = tl->seq[j];
                        lmh::Template& t = tp[j];
                        if (!get_int(tn, "width", &t.width) || !get_int(tn, "height", &t.height) || !get_int(tn, "pyramid_level", &t.pyramid_level)) {
                            err = "class " + ce.id + ": template without width/height/pyramid_level"; return false;
                        }
                        const Node* fl = tn.get("features");
                        if (fl && fl->kind == Node::Seq) {
                            t.features.reserve(fl->seq.size());
                            for (const Node& fn : fl->seq) {
                                if (fn.kind != Node::Nums || fn.nums.size() != 3) { err = "class " + ce.id + ": a feature is [x, y, label]"; return false; }
                                lm_feature ft;
                                if (!to_int(fn.nums[0], &ft.x) || !to_int(fn.nums[1], &ft.y) || !to_int(fn.nums[2], &ft.label)) {
                                    err = "class " + ce.id + ": feature value is not an integer"; return false;
                                }
                                if (t.features.size() >= LM_MAX_FEATURES) { err = "class " + ce.id + ": template with more than 63 features"; return false; }
                                t.features.push_back(ft);
                            }
                        }
                    }
                    std::string why;
                    if (!lmh::check_template_pyramid(tp, L, M, why)) { err = "class " + ce.id + ", template " + std::to_string(tid) + ": " + why; return false; }
                    ce.pyramids.push_back(std::move(tp));
                }
            }
            incoming.push_back(std::move(ce));
        }
    }
    cfg = nc;
    for (lmh::ClassEntry& ce : incoming)
        if (bank.find(ce.id) < 0) bank.classes.push_back(std::move(ce));
    return true;
}

}  // namespace lmy
