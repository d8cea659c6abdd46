// lm_host.cpp -- host-only parts of liblinemod_hip.so (see lm_host.h).
#include "lm_host.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lmh {

int Bank::find(const std::string& id) const {
    for (size_t i = 0; i < classes.size(); ++i)
        if (classes[i].id == id) return (int)i;
    return -1;
}

int Bank::add_pyramid(const std::string& id, TemplatePyramid&& tp) {
    int ci = find(id);
    if (ci < 0) { classes.push_back(ClassEntry{id, {}}); ci = (int)classes.size() - 1; }
    classes[ci].pyramids.push_back(std::move(tp));
    return (int)classes[ci].pyramids.size() - 1;
}

bool check_template_pyramid(const TemplatePyramid& tp, int levels, int modalities, std::string& err) {
    if (levels < 1 || modalities < 1 || tp.size() != (size_t)levels * (size_t)modalities) { err = "a pyramid needs levels x modalities templates"; return false; }
    for (int l = 0; l < levels; ++l) {
        size_t nf = 0;
        for (int m = 0; m < modalities; ++m) {
            const Template& t = tp[(size_t)l * modalities + m];
            if (t.pyramid_level != l) { err = "templates must be ordered [level*M + modality]"; return false; }
            if (t.width < 0 || t.height < 0 || t.width > 32767 || t.height > 32767) { err = "template size outside 0..32767"; return false; }
            if (t.features.size() > LM_MAX_FEATURES) { err = "template with more than 63 features (upstream CV_Assert(features.size() <= 63))"; return false; }
            for (const lm_feature& ft : t.features) {
                if (ft.label < 0 || ft.label > 7) { err = "feature label outside 0..7"; return false; }
                if (ft.x < 0 || ft.y < 0 || ft.x > 32767 || ft.y > 32767) { err = "feature coordinate outside 0..32767"; return false; }
            }
            nf += t.features.size();
        }
        if (nf == 0) { err = "template without features at a pyramid level (similarity would divide by zero)"; return false; }
    }
    return true;
}

bool check_modality_params(const lm_config& c, std::string& err) {
    auto fin = [](float v) { return v == v && v >= 0.0f && v < 1e18f; };
    if (!fin(c.weak_threshold) || !fin(c.strong_threshold)) { err = "ColorGradient thresholds must be finite and >= 0"; return false; }
    if (c.num_features < 1 || c.num_features > LM_MAX_FEATURES) { err = "ColorGradient num_features must be in 1..63"; return false; }
    if (c.num_modalities >= 2) {
        if (c.distance_threshold < 0 || c.difference_threshold < 0) { err = "DepthNormal thresholds must be >= 0"; return false; }
        if (c.depth_num_features < 1 || c.depth_num_features > LM_MAX_FEATURES) { err = "DepthNormal num_features must be in 1..63"; return false; }
        if (c.extract_threshold < 0) { err = "DepthNormal extract_threshold must be >= 0"; return false; }
    }
    return true;
}

int Bank::add_class(const std::string& id, int n_templates, const lm_template_desc* descs, const lm_feature* features,
                    int levels, int modalities, std::string& err) {
    const int per = levels * modalities;
    // validate first so a bad call leaves the bank untouched
    size_t fo = 0;
    for (int t = 0; t < n_templates; ++t)
        for (int k = 0; k < per; ++k) {
            const lm_template_desc& ds = descs[(size_t)t * per + k];
            if (ds.num_features < 0 || ds.num_features > LM_MAX_FEATURES) {
                err = "template with more than 63 features (upstream CV_Assert(features.size() <= 63))";
                return -1;
            }
            if (ds.pyramid_level != k / modalities) { err = "template descs must be ordered [level*M + modality]"; return -1; }
            if (ds.width < 0 || ds.height < 0 || ds.width > 32767 || ds.height > 32767) { err = "template size outside 0..32767"; return -1; }
            for (int f = 0; f < ds.num_features; ++f) {
                const lm_feature& ft = features[fo + f];
                if (ft.label < 0 || ft.label > 7) { err = "feature label outside 0..7"; return -1; }
                if (ft.x < 0 || ft.y < 0 || ft.x > 32767 || ft.y > 32767) { err = "feature coordinate outside 0..32767"; return -1; }
            }
            fo += ds.num_features;
        }
    for (int t = 0; t < n_templates; ++t)
        for (int l = 0; l < levels; ++l) {
            int nf = 0;
            for (int m = 0; m < modalities; ++m) nf += descs[(size_t)t * per + l * modalities + m].num_features;
            if (nf == 0) { err = "template without features at a pyramid level (similarity would divide by zero)"; return -1; }
        }
    int ci = find(id);
    if (ci < 0) { classes.push_back(ClassEntry{id, {}}); ci = (int)classes.size() - 1; }
    fo = 0;
    for (int t = 0; t < n_templates; ++t) {
        TemplatePyramid tp(per);
        for (int k = 0; k < per; ++k) {
            const lm_template_desc& ds = descs[(size_t)t * per + k];
            tp[k].width = ds.width; tp[k].height = ds.height; tp[k].pyramid_level = ds.pyramid_level;
            tp[k].features.assign(features + fo, features + fo + ds.num_features);
            fo += ds.num_features;
        }
        classes[ci].pyramids.push_back(std::move(tp));
    }
    return ci;
}

// ------------------------------------------------------------------------------------------------
// Host bank -> device bank of this shard.
//   scan (lowest level, upstream similarity()):  per (template, modality) a list of byte offsets
//     off = m*mod_stride + label*ori_stride + ((y%T)*T + x%T)*W*H + (y/T)*W + x/T
//   into the level arena, padded to `fpad` with offsets of the arena's zero block;
//     P = span_y*W + span_x + 1 = template_positions.
//   refine (levels above the lowest, upstream similarityLocal()): the same offset for the
//     unshifted feature plus (x, y) for the bounds test after the patch offset is applied.
// ------------------------------------------------------------------------------------------------
void schedule_lds_lists(DeviceBankHost& out);
bool build_device_bank(const Bank& bank, const lm_config& cfg, const LmLevelGeom* geom, DeviceBankHost& out,
                       int scan_list_order, std::string& err) {
    const int M = cfg.num_modalities, L = cfg.pyramid_levels;
    out = DeviceBankHost();
    const LmLevelGeom& gl = geom[L - 1];
    int maxf = 1;
    for (const ClassEntry& c : bank.classes)
        for (const TemplatePyramid& tp : c.pyramids)
            for (int m = 0; m < M; ++m) maxf = std::max(maxf, (int)tp[(size_t)(L - 1) * M + m].features.size());
    // byte scan: lists padded to LM_SCAN_FPAD; nibble scan (k_scan4): loops to the pair's own feature count, offsets in nibbles
    const int fq = gl.nibble ? 1 : LM_SCAN_FPAD;
    const u32 osc = gl.nibble ? 2u : 1u;
    out.fpad = (maxf + fq - 1) / fq * fq;
    const int nc = (int)bank.classes.size();
    out.class_item_lo.assign(nc, 0); out.class_item_hi.assign(nc, 0);
    out.class_t_lo.assign(nc, 0); out.class_t_hi.assign(nc, 0);
    out.class_alg_bytes.assign(nc, 0.0);
    out.class_load_bytes.assign(nc, 0.0);
    const bool planes = gl.nibble && gl.plane_ori != 0;
    if (planes) out.fpad1 = (std::min(M * maxf, 2 * LM_MAX_FEATURES) + 7) / 8 * 8;       // (k_scan1_exact reads whole batches of eight: the padding is the zero block)
    // r06: the LDS image of a frame's planes (k_scanl) -- [modality][orientation][pb bytes], pb = T*T*wh / 8, nothing between the planes
    const u32 ttwh = (u32)gl.T * (u32)gl.T * gl.wh, pb = ttwh / 8u;
    out.lds_ok = planes && (ttwh % 128u) == 0 && (size_t)M * ttwh <= LM_SCANL_IMAGE_MAX && gl.wh <= (1u << LM_SCANL_POS_BITS);
    // a list entry's bit offset from its nibble offset: the planes of a modality follow its 8 response memories
    auto plane_bit_off = [&](u32 noff) {
        const u32 base = noff / 2u, m = base / gl.mod_stride, label = (base - m * gl.mod_stride) / gl.ori_stride;
        const u32 rest = noff - 2u * (m * gl.mod_stride + label * gl.ori_stride);
        return 8u * (m * gl.mod_stride + 8u * gl.ori_stride + label * gl.plane_ori) + rest;
    };
    for (int ci = 0; ci < nc; ++ci) {
        const ClassEntry& c = bank.classes[ci];
        int lo, hi;
        shard_range((int)c.pyramids.size(), cfg.shard_rank, cfg.shard_size, &lo, &hi);
        out.class_t_lo[ci] = (int)out.t_global.size();
        out.class_item_lo[ci] = (int)out.item_t.size();
        // (k_scanl: a class starts a fresh wave item, so that the lanes that meet in one LDS access are the same whichever classes a launch scans;
        // the fillers are "no item")
        if (out.lds_ok) while (out.litem.size() % 64u) out.litem.push_back(0xFFFFFFFFu);
        for (int tid = lo; tid < hi; ++tid) {
            const TemplatePyramid& tp = c.pyramids[tid];
            if ((int)tp.size() != L * M) { err = "template pyramid size mismatch"; return false; }
            const u32 ti = (u32)out.t_global.size();
            out.t_global.push_back(tid);
            out.t_class.push_back(ci);
            // ---- scan level
            const Template& t0 = tp[(size_t)(L - 1) * M];
            int n_total = 0, cnt_packed = 0;
            int P = 0;
            double fcount = 0;
            for (int m = 0; m < M; ++m) {
                const Template& t = tp[(size_t)(L - 1) * M + m];
                n_total += (int)t.features.size();
                // upstream computes the span per modality from that modality's own template size; all
                // templates of one pyramid level share width/height after cropTemplates, so use each
                // modality's own value and require them equal.
                if (t.width != t0.width || t.height != t0.height) { err = "modalities of one pyramid level must share width/height"; return false; }
                int wf = (t.width - 1) / gl.T + 1, hf = (t.height - 1) / gl.T + 1;
                const int span_x = gl.W - wf, span_y = gl.H - hf;
                const long long P64 = (long long)span_y * gl.W + span_x + 1;
                P = (int)std::min<long long>(std::max<long long>(P64, 0), (long long)gl.wh);
                int k = 0;
                const size_t list_begin = out.scan_off.size();
                for (const lm_feature& f : t.features) {
                    if (f.x < 0 || f.x >= gl.w || f.y < 0 || f.y >= gl.h) continue;  // similarity(): "discard feature if out of bounds"
                    u32 off = osc * ((u32)m * gl.mod_stride + (u32)f.label * gl.ori_stride) +
                              (u32)((f.y % gl.T) * gl.T + (f.x % gl.T)) * gl.wh + (u32)(f.y / gl.T) * gl.W + (u32)(f.x / gl.T);
                    out.scan_off.push_back(off);
                    ++k;
                }
                fcount += k;
                cnt_packed |= k << (8 + 8 * m);
                // the sum is order-independent: ascending offsets make the waves resident on one CU (they start
                // together and step through their lists in step) read from the same few linear memories at a
                // time, so part of the traffic is served by the CU's L1 instead of L2
                std::sort(out.scan_off.begin() + (ptrdiff_t)list_begin, out.scan_off.end());
                {
                    // r04: the ORDER of a list decides how soon the scan's exact pruning gives up on a work item (the sums do not depend
                    // on it).  Sorted by offset (0) a list starts with ALL its features of orientation 0, whose responses rise and fall
                    // together over the frame.  3 (default): greedy farthest-point order in (x, y, orientation) -- every next feature is the
                    // one farthest from all chosen so far, an orientation step counting like 4 pixels -- so the first features sample the
                    // template's whole extent and all its orientations.  Measured, share of the feature loads the pruned scan makes /
                    // scan launch (profiles/r04_ab_experiments.log): config 2 49.7 % / 167 us (0), 48.2 % / 164 (1: round-robin over the
                    // orientations), 50.5 % (2: descending offsets), 46.3 % / 158.5 (3); config 3 42.7 % / 409 -> 39.8 % / 392.
                    const int feat_order = scan_list_order;
                    if (feat_order == 2) std::reverse(out.scan_off.begin() + (ptrdiff_t)list_begin, out.scan_off.end());
                    if (feat_order == 1) {
                        std::vector<std::vector<u32>> by_label(8);
                        for (size_t q = list_begin; q < out.scan_off.size(); ++q) {
                            const u32 rel = out.scan_off[q] / osc - (u32)m * gl.mod_stride;
                            by_label[std::min<u32>(rel / gl.ori_stride, 7u)].push_back(out.scan_off[q]);
                        }
                        size_t q = list_begin;
                        for (size_t r = 0; q < out.scan_off.size(); ++r)
                            for (int lb = 0; lb < 8; ++lb) if (r < by_label[lb].size()) out.scan_off[q++] = by_label[lb][r];
                    }
                    if (feat_order == 3) {
                        struct FP { u32 off; int x, y, lab; };
                        std::vector<FP> fp;
                        for (const lm_feature& f : t.features) {
                            if (f.x < 0 || f.x >= gl.w || f.y < 0 || f.y >= gl.h) continue;
                            const u32 off = osc * ((u32)m * gl.mod_stride + (u32)f.label * gl.ori_stride) +
                                            (u32)((f.y % gl.T) * gl.T + (f.x % gl.T)) * gl.wh + (u32)(f.y / gl.T) * gl.W + (u32)(f.x / gl.T);
                            fp.push_back(FP{off, f.x, f.y, f.label});
                        }
                        std::vector<long> best(fp.size(), (long)1 << 40);
                        std::vector<char> used(fp.size(), 0);
                        size_t cur = 0;
                        for (size_t q = 1; q < fp.size(); ++q) if (fp[q].off < fp[cur].off) cur = q;     // starts at the smallest offset
                        for (size_t r = 0; r < fp.size(); ++r) {
                            out.scan_off[list_begin + r] = fp[cur].off;
                            used[cur] = 1;
                            size_t nxt = cur; long far = -1;
                            for (size_t q = 0; q < fp.size(); ++q) {
                                if (used[q]) continue;
                                const int dl = std::min((fp[q].lab - fp[cur].lab) & 7, (fp[cur].lab - fp[q].lab) & 7);
                                const long dd = (long)(fp[q].x - fp[cur].x) * (fp[q].x - fp[cur].x) + (long)(fp[q].y - fp[cur].y) * (fp[q].y - fp[cur].y) + 16L * dl * dl;
                                best[q] = std::min(best[q], dd);
                                if (best[q] > far || (best[q] == far && fp[q].off < fp[nxt].off)) { far = best[q]; nxt = q; }
                            }
                            cur = nxt;
                        }
                    }
                }
                for (; k < out.fpad; ++k) out.scan_off.push_back(osc * gl.zero_off);
            }
            if (planes) {
                const size_t b1 = out.off1.size();
                for (int m = 0; m < M; ++m) {
                    const int k = (cnt_packed >> (8 + 8 * m)) & 0xFF;
                    const size_t lb = ((size_t)ti * M + m) * out.fpad;
                    for (int q = 0; q < k; ++q) {
                        const u32 noff = out.scan_off[lb + q];
                        out.offn.push_back(noff); out.off1.push_back(plane_bit_off(noff));
                        const u32 base = noff / 2u, mm = base / gl.mod_stride, label = (base - mm * gl.mod_stride) / gl.ori_stride;
                        out.offs3.push_back((label << 29) | (mm * gl.mod_stride + (noff - 2u * (mm * gl.mod_stride + label * gl.ori_stride))));
                    }
                }
                if (out.lds_ok) {
                    for (size_t q = b1; q < out.off1.size(); ++q) {
                        const u32 e3 = out.offs3[q], label = e3 >> 29, so = e3 & 0x1FFFFFFFu, mm = so / gl.mod_stride, rest = so - mm * gl.mod_stride;     // rest = memory * wh + cell
                        const u32 bit = 8u * (mm * 8u + label) * pb + rest;
                        out.offl.push_back((((bit >> 5) << 2) << 8) | (bit & 31u));
                        out.offsl.push_back((label << 29) | (mm * ttwh + rest));
                    }
                    // (padding: the image's zero block, right behind the planes; the second stage stops at a template's own feature count)
                    while (out.offl.size() < b1 + (size_t)out.fpad1) { out.offl.push_back(((u32)M * ttwh) << 8); out.offsl.push_back(0u); }
                    out.lbegin.push_back((int)out.litem.size());
                    for (int un = 0; un * 128 < P; ++un) out.litem.push_back((ti << 8) | (u32)un);
                }
                // (padding: the zero block through orientation 0 -- response 0 whatever the table, it maps an empty spread byte to 0)
                while (out.off1.size() < b1 + (size_t)out.fpad1) { out.offn.push_back(2u * gl.zero_off); out.off1.push_back(8u * gl.zero_off); out.offs3.push_back(gl.zero_off); }
                for (int L1 = 1; L1 <= 64; ++L1) out.items1_by_L[L1] += (P + (128 * L1 - 31) - 1) / (128 * L1 - 31);
            }
            const int chunk = gl.nibble ? LM_SCAN4_CHUNK : LM_SCAN_CHUNK;
            out.scan_P.push_back(P);
            out.scan_n.push_back(n_total | cnt_packed);   // n (bits 0-7) | in-bounds features of modality 0 / 1 (bits 8-15 / 16-23)
            out.class_alg_bytes[ci] += fcount * (double)P;
            // what the scan's vector loads request: per feature and work item one half-wave x 16 B (nibble
            // layout) or one wave x 16 B (byte layout)
            out.class_load_bytes[ci] += fcount * (double)((P + chunk - 1) / chunk) * (gl.nibble ? 512.0 : 1024.0);
            for (int ch = 0; ch * chunk < P; ++ch) { out.item_t.push_back(ti); out.item_chunk.push_back((u32)ch); }
            // ---- refinement levels
            for (int l = 0; l + 1 < L; ++l) {
                const LmLevelGeom& g = geom[l];
                LmRefMeta mt;
                std::memset(&mt, 0, sizeof(mt));
                mt.width = tp[(size_t)l * M].width;
                mt.height = tp[(size_t)l * M].height;
                for (int m = 0; m < M; ++m) {
                    const Template& t = tp[(size_t)l * M + m];
                    mt.nfeat_total += (int)t.features.size();
                    mt.start[m] = (u32)out.ref_feat[l].size();
                    mt.count[m] = (u32)t.features.size();
                    for (const lm_feature& f : t.features) {
                        LmRefFeat rf;
                        // spread arena: [modality][memory (y%T)*T + x%T][(y/T)*W + x/T]; label rides in the top bits
                        rf.off = ((u32)m * g.mod_stride + (u32)((f.y % g.T) * g.T + (f.x % g.T)) * g.wh +
                                  (u32)(f.y / g.T) * g.W + (u32)(f.x / g.T)) |
                                 ((u32)f.label << 29);
                        if ((u64)m * g.mod_stride + (u64)((f.y % g.T) * g.T + (f.x % g.T)) * g.wh + (u64)(f.y / g.T) * g.W +
                                (u64)(f.x / g.T) >= (1ull << 29)) { err = "feature too far outside the frame"; return false; }
                        rf.x = (int16_t)f.x; rf.y = (int16_t)f.y;
                        out.ref_feat[l].push_back(rf);
                    }
                }
                out.ref_meta[l].push_back(mt);
            }
        }
        out.class_t_hi[ci] = (int)out.t_global.size();
        out.class_item_hi[ci] = (int)out.item_t.size();
    }
    if (out.lds_ok) {
        out.lbegin.push_back((int)out.litem.size());
        if (out.t_global.size() >= (1u << (32 - LM_SCANL_POS_BITS)) || out.litem.empty()) out.lds_ok = false;     // (the survivor entry's template field)
    }
    if (out.lds_ok) {
        schedule_lds_lists(out);
        out.lrec.assign(out.litem.size() * 4u, 0u);
        for (size_t i = 0; i < out.litem.size(); ++i) {
            const u32 it = out.litem[i];
            out.lrec[4 * i] = it;
            if (it != 0xFFFFFFFFu) { out.lrec[4 * i + 1] = (u32)out.scan_n[it >> 8]; out.lrec[4 * i + 2] = (u32)out.scan_P[it >> 8]; }
        }
    }
    return true;
}

// r06: the ORDER of k_scanl's lists against LDS bank conflicts.  A lane reads the dwords D + 4 u + q of a feature's plane (D: the feature's first dword,
// u: the lane's unit, q = 0..4), one ds_read_b32-sized access per q; the hardware serves 32 lanes per cycle from 32 banks.  The lanes of ONE template hit
// the eight banks = (D + q) mod 4; two templates of a 32-lane group whose features have the same D mod 4 at the same step collide on all of them
// (measured: 25 of 109 us per 96-frame launch of config 2 are conflict cycles; four templates at random residues serve an access in 2.1 cycles instead of
// 1).  The sums do not depend on the order of a list, so every template's list is permuted -- greedily, first fit in its own (farthest-point) order -- such
// that at each step its residue differs from those of the templates it shares a 32-lane group with.  Only the order of `offl` changes.
void schedule_lds_lists(DeviceBankHost& out) {
    const size_t nt = out.t_global.size();
    const int fp = out.fpad1;
    auto residue = [](u32 e) { return (int)(((e >> 8) >> 2) & 3u); };
    std::vector<u32> tmp((size_t)fp);
    std::vector<char> used((size_t)fp);
    for (size_t t = 0; t < nt; ++t) {
        const int lb = out.lbegin[t], le = out.lbegin[t + 1];
        // lane items of template t: [lb, le) minus trailing fillers; none: nothing to schedule
        int last = le;
        while (last > lb && out.litem[(size_t)last - 1] == 0xFFFFFFFFu) --last;
        if (last <= lb) continue;
        const int g_first = lb / 32;
        // neighbours: earlier templates with a lane in one of this template's groups (they are scheduled already)
        std::vector<size_t> nb;
        for (size_t p = t; p-- > 0 && nb.size() < 8;) {
            int pl = out.lbegin[p + 1];
            while (pl > out.lbegin[p] && out.litem[(size_t)pl - 1] == 0xFFFFFFFFu) --pl;
            if (pl <= out.lbegin[p]) continue;
            if ((pl - 1) / 32 < g_first) break;
            nb.push_back(p);
        }
        if (nb.empty()) continue;
        u32* list = out.offl.data() + t * (size_t)fp;
        // the list's real entries = the template's in-bounds features (the padding, entries of the zero block, stays at the end)
        const int cnt = out.scan_n[t];
        const int nreal = std::min(fp, ((cnt >> 8) & 0xFF) + ((cnt >> 16) & 0xFF));
        std::fill(used.begin(), used.end(), 0);
        for (int k = 0; k < nreal; ++k) {
            bool taken[4] = {false, false, false, false};
            for (size_t p : nb) taken[residue(out.offl[p * (size_t)fp + (size_t)k])] = true;
            int pick = -1, first = -1;
            for (int q = 0; q < nreal; ++q) {
                if (used[(size_t)q]) continue;
                if (first < 0) first = q;
                if (!taken[residue(list[q])]) { pick = q; break; }
            }
            if (pick < 0) pick = first;
            used[(size_t)pick] = 1;
            tmp[(size_t)k] = list[pick];
        }
        for (int k = 0; k < nreal; ++k) list[k] = tmp[(size_t)k];
    }
}

void build_items1(const DeviceBankHost& hb, int L, std::vector<u32>& item_t, std::vector<u32>& item_chunk, std::vector<int>& begin) {
    const int chunk = 128 * L - 31;
    item_t.clear(); item_chunk.clear(); begin.clear();
    for (size_t t = 0; t < hb.scan_P.size(); ++t) {
        begin.push_back((int)item_t.size());
        for (int ch = 0; ch * chunk < hb.scan_P[t]; ++ch) { item_t.push_back((u32)t); item_chunk.push_back((u32)ch); }
    }
    begin.push_back((int)item_t.size());
}

void build_hull_table(const Bank& bank, int M, HullTable& out) {
    out = HullTable();
    struct P { int x, y; };
    std::vector<P> p, hull;
    for (const ClassEntry& c : bank.classes) {
        out.class_base.push_back((u32)out.hull_off.size());
        for (const TemplatePyramid& tp : c.pyramids) {
            out.hull_off.push_back((u32)(out.hull_xy.size() / 2));
            p.clear();
            for (int m = 0; m < M && m < (int)tp.size(); ++m)          // tp[m], m < M: the level-0 templates
                for (const lm_feature& f : tp[(size_t)m].features) p.push_back(P{f.x, f.y});
            std::sort(p.begin(), p.end(), [](const P& a, const P& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
            p.erase(std::unique(p.begin(), p.end(), [](const P& a, const P& b) { return a.x == b.x && a.y == b.y; }), p.end());
            if (p.size() >= 3) {
                auto crs = [](const P& o, const P& a, const P& b) { return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x); };
                hull.assign(2 * p.size(), P{0, 0});
                size_t k = 0;
                for (size_t i = 0; i < p.size(); ++i) {
                    while (k >= 2 && crs(hull[k - 2], hull[k - 1], p[i]) <= 0) --k;
                    hull[k++] = p[i];
                }
                for (size_t i = p.size() - 1, t = k + 1; i > 0; --i) {
                    while (k >= t && crs(hull[k - 2], hull[k - 1], p[i - 1]) <= 0) --k;
                    hull[k++] = p[i - 1];
                }
                hull.resize(k - 1);
            } else {
                hull = p;
            }
            for (const P& q : hull) { out.hull_xy.push_back((int16_t)q.x); out.hull_xy.push_back((int16_t)q.y); }
        }
    }
    out.hull_off.push_back((u32)(out.hull_xy.size() / 2));
    if (out.class_base.empty()) out.class_base.push_back(0);
}

// SIMILARITY_LUT default = the table cv::linemod ships (layout [ori][lo nibble 16 | hi nibble 16],
// entry = max over the nibble's set bits of the single-bit score).  Rows 3-7 are circular in the 8 orientation
// bins, rows 0-2 are not (orientation 0 scores 0 against bits 5-7): an upstream quirk that is part of the
// reference's behaviour, so it is the default here; lm_set_similarity_lut replaces it.
void default_similarity_lut(u8 lut[256]) {
    static const u8 kUpstream[256] = {
        0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,  0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1,
        0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,  0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2,
        0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,  0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,
        0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,  0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,
        0, 1, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,  0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,
        0, 2, 1, 2, 0, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2,  0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,
        0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,  0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};
    std::memcpy(lut, kUpstream, 256);
}

// NORMAL_LUT default (our documented rule): azimuth of the cell centre, 8 bins.
void default_normal_lut(u8 lut[8000]) {
    const double PI = 3.14159265358979323846;
    for (int v3 = 0; v3 < 20; ++v3)
        for (int v2 = 0; v2 < 20; ++v2)
            for (int v1 = 0; v1 < 20; ++v1) {
                double nx = (v1 + 0.5 - 10.0) / 10.0, ny = (v2 + 0.5 - 10.0) / 10.0;
                double a = std::atan2(ny, nx);
                if (a < 0) a += 2 * PI;
                int bin = (int)std::floor(a / (PI / 4.0));
                if (bin > 7) bin = 7;
                lut[v3 * 400 + v2 * 20 + v1] = (u8)(1u << bin);
            }
}

bool match_less(const lm_match_t& a, const lm_match_t& b) {
    if (a.similarity != b.similarity) return a.similarity > b.similarity;
    if (a.template_id != b.template_id) return a.template_id < b.template_id;
    if (a.class_idx != b.class_idx) return a.class_idx < b.class_idx;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
}
bool match_eq(const lm_match_t& a, const lm_match_t& b) {
    return a.x == b.x && a.y == b.y && a.similarity == b.similarity && a.class_idx == b.class_idx;
}
void sort_unique(std::vector<lm_match_t>& v) {
    std::sort(v.begin(), v.end(), match_less);
    v.erase(std::unique(v.begin(), v.end(), match_eq), v.end());
}

// ------------------------------------------------------------------------------------------------
// Bank file: "LMBK0001" | u32 levels | u32 modalities | u32 T[levels] | u32 n_classes |
//   per class { u32 id_len | id bytes | u32 n_templates | descs[n*levels*M] | u32 n_features | features[] }
// little endian, descs/features in the lm_template_desc / lm_feature layouts.
// ------------------------------------------------------------------------------------------------
namespace {
struct File {
    FILE* f;
    explicit File(FILE* fp) : f(fp) {}
    ~File() { if (f) fclose(f); }
};
bool wr(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
}  // namespace

bool save_bank(const Bank& bank, const lm_config& cfg, const char* path, std::string& err) {
    File fh(fopen(path, "wb"));
    if (!fh.f) { err = std::string("cannot open for writing: ") + path; return false; }
    const u32 L = (u32)cfg.pyramid_levels, M = (u32)cfg.num_modalities;
    bool ok = wr(fh.f, "LMBK0001", 8) && wr(fh.f, &L, 4) && wr(fh.f, &M, 4);
    for (u32 l = 0; l < L; ++l) { u32 t = (u32)cfg.T[l]; ok = ok && wr(fh.f, &t, 4); }
    u32 nc = (u32)bank.classes.size();
    ok = ok && wr(fh.f, &nc, 4);
    for (const ClassEntry& c : bank.classes) {
        u32 len = (u32)c.id.size(), nt = (u32)c.pyramids.size();
        ok = ok && wr(fh.f, &len, 4) && wr(fh.f, c.id.data(), len) && wr(fh.f, &nt, 4);
        std::vector<lm_template_desc> descs;
        std::vector<lm_feature> feats;
        for (const TemplatePyramid& tp : c.pyramids)
            for (const Template& t : tp) {
                descs.push_back(lm_template_desc{t.width, t.height, t.pyramid_level, (int32_t)t.features.size()});
                feats.insert(feats.end(), t.features.begin(), t.features.end());
            }
        u32 nf = (u32)feats.size();
        ok = ok && wr(fh.f, descs.data(), descs.size() * sizeof(lm_template_desc)) && wr(fh.f, &nf, 4) &&
             wr(fh.f, feats.data(), feats.size() * sizeof(lm_feature));
    }
    if (!ok) { err = std::string("short write: ") + path; return false; }
    return true;
}

bool load_bank(Bank& bank, const lm_config& cfg, const char* path, std::string& err) {
    File fh(fopen(path, "rb"));
    if (!fh.f) { err = std::string("cannot open: ") + path; return false; }
    char magic[8];
    u32 L = 0, M = 0;
    if (!rd(fh.f, magic, 8) || std::memcmp(magic, "LMBK0001", 8) != 0) { err = "not a linemod bank file"; return false; }
    if (!rd(fh.f, &L, 4) || !rd(fh.f, &M, 4)) { err = "truncated bank file"; return false; }
    if ((int)L != cfg.pyramid_levels || (int)M != cfg.num_modalities) { err = "bank was written for a different detector (levels/modalities)"; return false; }
    for (u32 l = 0; l < L; ++l) {
        u32 t = 0;
        if (!rd(fh.f, &t, 4)) { err = "truncated bank file"; return false; }
        if ((int)t != cfg.T[l]) { err = "bank was written for a different T pyramid"; return false; }
    }
    u32 nc = 0;
    if (!rd(fh.f, &nc, 4)) { err = "truncated bank file"; return false; }
    // counts read from the file size buffers below: none may promise more bytes than the file holds
    long here = std::ftell(fh.f);
    if (here < 0 || std::fseek(fh.f, 0, SEEK_END) != 0) { err = "cannot seek in bank file"; return false; }
    const unsigned long long file_bytes = (unsigned long long)std::ftell(fh.f);
    if (std::fseek(fh.f, here, SEEK_SET) != 0) { err = "cannot seek in bank file"; return false; }
    Bank nb;
    for (u32 c = 0; c < nc; ++c) {
        u32 len = 0, nt = 0, nf = 0;
        if (!rd(fh.f, &len, 4) || len > 4096) { err = "corrupt bank file"; return false; }
        std::string id(len, '\0');
        if (!rd(fh.f, &id[0], len) || !rd(fh.f, &nt, 4)) { err = "truncated bank file"; return false; }
        if ((unsigned long long)nt * L * M * sizeof(lm_template_desc) > file_bytes) { err = "corrupt bank file (template count)"; return false; }
        std::vector<lm_template_desc> descs((size_t)nt * L * M);
        if (!rd(fh.f, descs.data(), descs.size() * sizeof(lm_template_desc)) || !rd(fh.f, &nf, 4)) { err = "truncated bank file"; return false; }
        unsigned long long want = 0;
        for (const auto& ds : descs) want += (unsigned)std::max(ds.num_features, 0);
        if (want != nf || (unsigned long long)nf * sizeof(lm_feature) > file_bytes) { err = "corrupt bank file (feature count)"; return false; }
        std::vector<lm_feature> feats(nf);
        if (!rd(fh.f, feats.data(), feats.size() * sizeof(lm_feature))) { err = "truncated bank file"; return false; }
        if (nb.add_class(id, (int)nt, descs.data(), feats.data(), (int)L, (int)M, err) < 0) return false;
        if (nt == 0 && nb.find(id) < 0) nb.classes.push_back(ClassEntry{id, {}});
    }
    bank = std::move(nb);
    return true;
}


#if defined(__x86_64__)
__attribute__((target("avx2"))) static void copy_stream_avx2(unsigned char* d, const unsigned char* s, size_t n) {
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;
    if (head) { std::memcpy(d, s, head); d += head; s += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 64)), e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), a); _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 64), c); _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 96), e);
    }
    for (; i + 32 <= n; i += 32) _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i)));
    if (i < n) std::memcpy(d + i, s + i, n - i);
}
#endif
void copy_stream(void* dst, const void* src, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 256) { copy_stream_avx2(static_cast<unsigned char*>(dst), static_cast<const unsigned char*>(src), n); return; }
#endif
    std::memcpy(dst, src, n);
}
void copy_stream_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#endif
}
}  // namespace lmh
