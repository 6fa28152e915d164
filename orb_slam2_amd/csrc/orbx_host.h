// Host-side tables of the extractor: everything the reference computes once per instance or
// once per image size, evaluated with the same float32/float64 operation order so that the
// integer results (level sizes, quotas, cell grids, resize taps) are identical.
#ifndef ORBX_HOST_H
#define ORBX_HOST_H

#include <cstdint>
#include <vector>

#include "../../include/orbx.h"

namespace orbx {

constexpr int kEdgeThreshold = 19;                  // reference src/ORBextractor.cc:75
constexpr int kBorder = kEdgeThreshold - 3;         // 16: FAST rect origin (:862)
constexpr int kHalfPatch = 15;                      // :74
constexpr int kCellTarget = 30;                     // :858

struct Tables {
    int nlevels = 0;
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS];
    float sigma2[ORBX_MAX_LEVELS], inv_sigma2[ORBX_MAX_LEVELS];
    int quota[ORBX_MAX_LEVELS];
    int umax[16];
};

// One visited FAST cell (sub-image of a level), in visiting order.
struct CellDesc {
    int16_t x0, y0;      // sub-image origin in level coordinates
    int16_t cw, ch;      // sub-image size (detection area is the interior minus a 3-px rim)
    int16_t off_x, off_y;  // j*wCell, i*hCell: added to sub-image coords -> rect-relative coords
    int16_t level;
    int16_t pad;
    int32_t slot;        // index of this cell's candidate slot (global over levels)
    int32_t pad2;
};
static_assert(sizeof(CellDesc) == 24, "CellDesc layout");

// One FAST work tile: up to kFastTileCells horizontally adjacent cells of one cell row.
// Detection coordinates (u, v) run over the union of the cells' detection areas.
struct TileDesc {
    int16_t level;
    int16_t nc;          // cells in the tile
    int16_t x_first;     // level x of detection column u = 0 (= 19 + j0*wCell)
    int16_t y_first;     // level y of detection row v = 0 (= 19 + i*hCell)
    int16_t wd, hd;      // detection width (all cells) and height
    int16_t w_cell;      // width of every cell but possibly the last
    int16_t pad;
    int32_t slot0;       // candidate slot of the first cell; the others follow
    int16_t off_x, off_y;  // rect-relative coordinates of (u, v) = (0, 0): j0*wCell + 3, i*hCell + 3
};
static_assert(sizeof(TileDesc) == 24, "TileDesc layout");
constexpr int kFastTileCells = 4;

struct ResizeTap {  // one destination index of one axis
    int16_t ofs;    // source index (clamped for columns; unclamped first row for rows)
    int16_t ofs1;   // second source index, clamped into the image
    int16_t c0, c1; // 11-bit fixed-point weights
};
static_assert(sizeof(ResizeTap) == 8, "ResizeTap layout");

struct LevelGeom {
    int w, h;
    int n_cols, n_rows, w_cell, h_cell;
    int n_cells;      // visited cells
    int cell_base;    // first entry in the cell table
    int n_ini;
    float h_x;        // root width (float division, :578)
    int rect_w, rect_h;
    int quota;
    int max_kp;       // upper bound on retained nodes
    int max_slots;    // quadtree node slots needed
    int cell_cap;     // candidate capacity of one cell of this level
    int max_cand;     // upper bound on candidates of the level
};

bool build_tables(int nfeatures, float scale_factor, int nlevels, Tables* t);
// Returns ORBX_OK or ORBX_E_GEOMETRY.
int build_level_geometry(const Tables& t, int width, int height, std::vector<LevelGeom>* levels,
                         std::vector<CellDesc>* cells);
// Groups the visited cells (as listed by build_level_geometry) into FAST work tiles.
void build_fast_tiles(const std::vector<LevelGeom>& levels, const std::vector<CellDesc>& cells,
                      std::vector<TileDesc>* tiles);
void build_resize_taps(int sn, int dn, bool clamp_index, std::vector<ResizeTap>* taps);
int max_keypoints(const std::vector<LevelGeom>& levels);

// A DBoW2 vocabulary tree flattened for the device (reference
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:296-320, :1338-1425).  Node 0 is the root.
struct VocHost {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<uint32_t> child_start, child;  // CSR of the children lists, in node order
    std::vector<uint8_t> desc;                 // n_nodes x 32
    std::vector<uint32_t> word;                // word id of the nodes the file flags as leaves, else 0
    std::vector<double> weight;
    int n_words = 0;
    int n_nodes() const { return (int)word.size(); }
};
// Builds the tree from per-node arrays (entry 0 = root, its parent / is_leaf are ignored).
// Returns ORBX_OK or ORBX_E_INVALID (a parent that is not an earlier node, enum values out of range).
int build_vocabulary(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                     const uint8_t* desc, const double* weight, VocHost* out);
// Reads the text format of loadFromTextFile (:1338-1425): header "k L scoring weighting", then one
// line per node: parent, is-leaf flag, 32 descriptor bytes, weight.
int load_vocabulary_text(const char* path, VocHost* out);

}  // namespace orbx
#endif
