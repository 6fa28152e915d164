// Host-side tables of the extractor (see orbx_host.h).  Compiled with -ffp-contract=off.
#include "orbx_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace orbx {

namespace {
// cvRound: round half to even (x86 cvtss2si/cvtsd2si under the default rounding mode).
inline int round_even(float v) { return (int)std::nearbyintf(v); }
inline int round_even(double v) { return (int)std::nearbyint(v); }
}  // namespace

// Reference src/ORBextractor.cc:426-463 (scale tables, per-level quotas) and :473-493 (umax).
// The reference keeps scaleFactor as a double initialised from the float argument
// (include/ORBextractor.h:98), so the products below are float*double rounded to float.
bool build_tables(int nfeatures, float scale_factor, int nlevels, Tables* t) {
    if (nlevels < 1 || nlevels > ORBX_MAX_LEVELS || nfeatures < 0 || !(scale_factor > 1.0f)) return false;
    std::memset(t, 0, sizeof(*t));
    t->nlevels = nlevels;
    const double sf = (double)scale_factor;
    t->scale[0] = 1.0f;
    t->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        t->scale[i] = (float)((double)t->scale[i - 1] * sf);
        t->sigma2[i] = t->scale[i] * t->scale[i];
    }
    for (int i = 0; i < nlevels; ++i) {
        t->inv_scale[i] = 1.0f / t->scale[i];
        t->inv_sigma2[i] = 1.0f / t->sigma2[i];
    }
    const float factor = (float)(1.0 / sf);
    const float denom = 1.0f - (float)std::pow((double)factor, (double)nlevels);
    float want = (float)nfeatures * (1.0f - factor) / denom;
    int given = 0;
    for (int level = 0; level + 1 < nlevels; ++level) {
        t->quota[level] = round_even(want);
        given += t->quota[level];
        want *= factor;
    }
    t->quota[nlevels - 1] = std::max(nfeatures - given, 0);

    // Row half-widths of the radius-15 disc: exact for v <= 11, mirrored above so that the
    // disc is symmetric about the diagonal.
    const float half_diag = kHalfPatch * std::sqrt(2.f) / 2;
    const int vmax = (int)std::floor(half_diag + 1);
    const int vmin = (int)std::ceil(half_diag);
    for (int v = 0; v <= vmax; ++v) t->umax[v] = round_even(std::sqrt((double)(kHalfPatch * kHalfPatch - v * v)));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t->umax[v0] == t->umax[v0 + 1]) ++v0;
        t->umax[v] = v0;
        ++v0;
    }
    return true;
}

// Level sizes (:1220-1221), the FAST cell grid with its skip rules (:862-897), quadtree roots
// (:576-578).  Cells are listed in visiting order: cell row major, cell column minor.
int build_level_geometry(const Tables& t, int width, int height, std::vector<LevelGeom>* levels,
                         std::vector<CellDesc>* cells) {
    levels->clear();
    cells->clear();
    int slot = 0;
    for (int l = 0; l < t.nlevels; ++l) {
        LevelGeom g;
        std::memset(&g, 0, sizeof(g));
        g.w = round_even((float)width * t.inv_scale[l]);
        g.h = round_even((float)height * t.inv_scale[l]);
        const int max_bx = g.w - kBorder, max_by = g.h - kBorder;
        g.rect_w = max_bx - kBorder;
        g.rect_h = max_by - kBorder;
        if (g.rect_w < kCellTarget || g.rect_h < kCellTarget) return ORBX_E_GEOMETRY;  // nCols or nRows would be 0
        if (g.rect_w > 4095 || g.rect_h > 4095) return ORBX_E_GEOMETRY;                // 12-bit packed coordinates
        const float fw = (float)g.rect_w, fh = (float)g.rect_h;
        g.n_cols = (int)(fw / (float)kCellTarget);
        g.n_rows = (int)(fh / (float)kCellTarget);
        g.w_cell = (int)std::ceil(fw / (float)g.n_cols);
        g.h_cell = (int)std::ceil(fh / (float)g.n_rows);
        g.n_ini = (int)std::round(fw / (float)g.rect_h);
        if (g.n_ini < 1) return ORBX_E_GEOMETRY;  // tall images: the reference divides by zero
        g.h_x = fw / (float)g.n_ini;
        g.quota = t.quota[l];
        g.max_kp = std::max(g.quota + 3, 4 * g.n_ini);
        g.max_slots = g.max_kp + g.n_ini + 4;
        g.cell_cap = ((g.w_cell + 1) / 2) * ((g.h_cell + 1) / 2);
        g.cell_base = (int)cells->size();
        for (int i = 0; i < g.n_rows; ++i) {
            const int ini_y = kBorder + i * g.h_cell;
            if (ini_y >= max_by - 3) continue;
            const int end_y = std::min(ini_y + g.h_cell + 6, max_by);
            for (int j = 0; j < g.n_cols; ++j) {
                const int ini_x = kBorder + j * g.w_cell;
                if (ini_x >= max_bx - 6) continue;
                const int end_x = std::min(ini_x + g.w_cell + 6, max_bx);
                CellDesc c;
                std::memset(&c, 0, sizeof(c));
                c.x0 = (int16_t)ini_x;
                c.y0 = (int16_t)ini_y;
                c.cw = (int16_t)(end_x - ini_x);
                c.ch = (int16_t)(end_y - ini_y);
                c.off_x = (int16_t)(j * g.w_cell);
                c.off_y = (int16_t)(i * g.h_cell);
                c.level = (int16_t)l;
                c.slot = slot++;
                cells->push_back(c);
            }
        }
        g.n_cells = (int)cells->size() - g.cell_base;
        g.max_cand = g.n_cells * g.cell_cap;
        levels->push_back(g);
    }
    return ORBX_OK;
}

// Cells of one cell row are consecutive in the cell list and their detection areas
// [x0+3, x0+cw-3) x [y0+3, y0+ch-3) tile the row without gaps, so a run of cells is one rectangle.
void build_fast_tiles(const std::vector<LevelGeom>& levels, const std::vector<CellDesc>& cells,
                      std::vector<TileDesc>* tiles) {
    tiles->clear();
    size_t k = 0;
    while (k < cells.size()) {
        const CellDesc& first = cells[k];
        size_t e = k + 1;
        while (e < cells.size() && e - k < (size_t)kFastTileCells && cells[e].level == first.level &&
               cells[e].y0 == first.y0)
            ++e;
        const CellDesc& last = cells[e - 1];
        TileDesc t;
        std::memset(&t, 0, sizeof(t));
        t.level = first.level;
        t.nc = (int16_t)(e - k);
        t.x_first = (int16_t)(first.x0 + 3);
        t.y_first = (int16_t)(first.y0 + 3);
        t.wd = (int16_t)((last.x0 + last.cw - 3) - (first.x0 + 3));
        t.hd = (int16_t)(first.ch - 6);
        t.w_cell = (int16_t)levels[first.level].w_cell;
        t.slot0 = first.slot;
        t.off_x = (int16_t)(first.off_x + 3);
        t.off_y = (int16_t)(first.off_y + 3);
        tiles->push_back(t);
        k = e;
    }
}

// cv::resize(..., INTER_LINEAR) 8UC1 tap setup, OpenCV 3.4.5 (reference call site :1237):
// scale = 1/(dn/sn) in double, f = float((d+0.5)*scale-0.5), index = floor(f), weight = f-index
// in float, 11-bit weights rounded half to even.
void build_resize_taps(int sn, int dn, bool clamp_index, std::vector<ResizeTap>* taps) {
    taps->resize(dn);
    const double inv_scale = (double)dn / sn;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (clamp_index) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= sn - 1) { f = 0.f; s = sn - 1; }
        }
        ResizeTap tp;
        tp.c0 = (int16_t)round_even((1.f - f) * 2048.f);
        tp.c1 = (int16_t)round_even(f * 2048.f);
        // rows: both source rows are clamped into the image, weights untouched
        tp.ofs = (int16_t)std::min(std::max(s, 0), sn - 1);
        tp.ofs1 = (int16_t)std::min(std::max(s + 1, 0), sn - 1);
        (*taps)[d] = tp;
    }
}

int max_keypoints(const std::vector<LevelGeom>& levels) {
    int n = 0;
    for (const LevelGeom& g : levels) n += g.max_kp;
    return n;
}

int build_vocabulary(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                     const uint8_t* desc, const double* weight, VocHost* out) {
    if (n_nodes < 1 || L < 1 || L > 10 || k < 0 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
        return ORBX_E_INVALID;
    VocHost v;
    v.k = k;
    v.L = L;
    v.scoring = scoring;
    v.weighting = weighting;
    std::vector<uint32_t> count((size_t)n_nodes + 1, 0);
    for (int i = 1; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] >= i) return ORBX_E_INVALID;
        ++count[(size_t)parent[i]];
    }
    v.child_start.assign((size_t)n_nodes + 1, 0);
    for (int i = 0; i < n_nodes; ++i) v.child_start[(size_t)i + 1] = v.child_start[(size_t)i] + count[(size_t)i];
    v.child.assign((size_t)std::max(1, n_nodes - 1), 0);
    std::vector<uint32_t> cursor(v.child_start.begin(), v.child_start.end() - 1);
    v.word.assign((size_t)n_nodes, 0);
    for (int i = 1; i < n_nodes; ++i) {
        v.child[cursor[(size_t)parent[i]]++] = (uint32_t)i;  // children in node order (:1390-1391)
        if (is_leaf[i]) v.word[(size_t)i] = (uint32_t)v.n_words++;  // :1408-1415
    }
    v.desc.assign(desc, desc + (size_t)n_nodes * 32);
    v.weight.assign(weight, weight + n_nodes);
    *out = std::move(v);
    return ORBX_OK;
}

int load_vocabulary_text(const char* path, VocHost* out) {
    std::FILE* f = std::fopen(path, "r");
    if (!f) return ORBX_E_INVALID;
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (std::fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 ||
        n2 < 0 || n2 > 3) {  // :1359-1363
        std::fclose(f);
        return ORBX_E_INVALID;
    }
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    for (;;) {
        int pid, is_leaf;
        if (std::fscanf(f, "%d %d", &pid, &is_leaf) != 2) break;  // end of file (a trailing empty line adds no node here)
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32 && ok; ++i) {
            int b;
            ok = std::fscanf(f, "%d", &b) == 1;
            d[i] = (uint8_t)b;
        }
        double w = 0.0;
        ok = ok && std::fscanf(f, "%lf", &w) == 1;
        if (!ok) {
            std::fclose(f);
            return ORBX_E_INVALID;
        }
        parent.push_back(pid);
        leaf.push_back(is_leaf > 0);
        desc.insert(desc.end(), d, d + 32);
        weight.push_back(w);
    }
    std::fclose(f);
    return build_vocabulary(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

}  // namespace orbx

// ---- C ABI, host-only helpers -------------------------------------------------------------
using namespace orbx;

extern "C" int orbx_host_tables(int nfeatures, float scale_factor, int nlevels, float* scale,
                                float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* quota,
                                int32_t* umax16) {
    Tables t;
    if (!build_tables(nfeatures, scale_factor, nlevels, &t)) return ORBX_E_INVALID;
    for (int i = 0; i < nlevels; ++i) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (quota) quota[i] = t.quota[i];
    }
    if (umax16)
        for (int i = 0; i < 16; ++i) umax16[i] = t.umax[i];
    return ORBX_OK;
}

extern "C" int orbx_host_level_geometry(int nfeatures, float scale_factor, int nlevels, int width,
                                        int height, int level, orbx_level_geom* out) {
    Tables t;
    if (!out || !build_tables(nfeatures, scale_factor, nlevels, &t)) return ORBX_E_INVALID;
    if (level < 0 || level >= nlevels || width < 1 || height < 1) return ORBX_E_INVALID;
    std::vector<LevelGeom> levels;
    std::vector<CellDesc> cells;
    int rc = build_level_geometry(t, width, height, &levels, &cells);
    if (rc != ORBX_OK) return rc;
    const LevelGeom& g = levels[level];
    out->width = g.w;
    out->height = g.h;
    out->n_cols = g.n_cols;
    out->n_rows = g.n_rows;
    out->w_cell = g.w_cell;
    out->h_cell = g.h_cell;
    out->n_cells = g.n_cells;
    out->n_ini = g.n_ini;
    out->quota = g.quota;
    return ORBX_OK;
}

extern "C" int orbx_host_resize_taps(int sn, int dn, int clamp_index, int32_t* ofs, int16_t* c0,
                                     int16_t* c1) {
    if (sn < 1 || dn < 1 || sn > 32767 || !ofs || !c0 || !c1) return ORBX_E_INVALID;
    // Reported exactly as OpenCV holds them: the unclamped row index for the vertical axis.
    const double scale = 1.0 / ((double)dn / sn);
    std::vector<ResizeTap> taps;
    build_resize_taps(sn, dn, clamp_index != 0, &taps);
    for (int d = 0; d < dn; ++d) {
        if (clamp_index) {
            ofs[d] = taps[d].ofs;
        } else {
            float f = (float)((d + 0.5) * scale - 0.5);
            ofs[d] = (int)std::floor(f);
        }
        c0[d] = taps[d].c0;
        c1[d] = taps[d].c1;
    }
    return ORBX_OK;
}

extern "C" int orbx_host_max_keypoints(int nfeatures, float scale_factor, int nlevels, int width,
                                       int height) {
    Tables t;
    if (!build_tables(nfeatures, scale_factor, nlevels, &t)) return ORBX_E_INVALID;
    std::vector<LevelGeom> levels;
    std::vector<CellDesc> cells;
    int rc = build_level_geometry(t, width, height, &levels, &cells);
    if (rc != ORBX_OK) return rc;
    return max_keypoints(levels);
}
