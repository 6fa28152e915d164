// Device-side data layout shared by the kernels and the launcher (gfx950, wave64).
#ifndef ORBX_DEVICE_H
#define ORBX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_host.h"

namespace orbx {

constexpr int kWave = 64;
constexpr int kBlock = 256;

// A candidate / retained point: response << 24 | y << 12 | x, with (x, y) relative to the
// level's (16,16) FAST origin -- the coordinates DistributeOctTree works in
// (reference src/ORBextractor.cc:909-917).
__host__ __device__ inline uint32_t pack_pt(int x, int y, int score) {
    return ((uint32_t)score << 24) | ((uint32_t)y << 12) | (uint32_t)x;
}
__host__ __device__ inline int pt_x(uint32_t p) { return (int)(p & 0xFFFu); }
__host__ __device__ inline int pt_y(uint32_t p) { return (int)((p >> 12) & 0xFFFu); }
__host__ __device__ inline int pt_score(uint32_t p) { return (int)(p >> 24); }

// Loads at byte-granular addresses: gfx950 global loads of 2, 4 and 16 bytes and
// LDS loads of 4 bytes need no alignment.
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) uint4_unaligned {
    uint32_t x, y, z, w;
    __device__ operator uint4() const { return make_uint4(x, y, z, w); }
};

// Words of BatchDev::status.  A kernel that reads a count or a coordinate out of device memory checks it against the
// capacity it indexes, works on with the value cut to that capacity and sets the word; the host reports the batch as
// failed (ORBX_E_DEVICE_DATA) at its next synchronisation instead of the GPU faulting on the access.
enum { kStatusCellCount = 0, kStatusPoint = 1, kStatusLevelCount = 2, kStatusWords = 4 };

// Per-level view used by every kernel.  Images of one level for all frames of the batch are
// contiguous: frame f of level l starts at img + f*plane, rows `pitch` bytes apart.
struct LevelDev {
    const uint8_t* img;   // un-blurred level (level 0 may alias the caller's buffer)
    uint8_t* blur;        // blurred level
    uint64_t plane;       // bytes between frames in img
    uint64_t bplane;      // bytes between frames in blur
    int32_t pitch;        // row pitch of img
    int32_t bpitch;       // row pitch of blur
    int32_t w, h;
    // FAST / quadtree geometry
    int32_t n_cells, cell_base;   // cell table range of this level
    int32_t n_ini;
    float h_x;
    int32_t rect_w, rect_h;
    int32_t quota, max_kp, max_slots;
    int32_t dense_off;    // offset of this level inside a frame's dense candidate arrays
    int32_t kp_off;       // offset of this level inside a frame's retained-point array
    float scale;          // mvScaleFactor[level]
    float size;           // float(int(31*scale))
    int32_t pad;
};

struct BatchDev {
    LevelDev lv[ORBX_MAX_LEVELS];
    int32_t nlevels;
    int32_t nframes;
    int32_t total_cells;   // cells per frame over all levels
    int32_t cell_cap;      // candidate capacity per cell
    int32_t dense_total;   // dense candidates per frame over all levels
    int32_t kp_total;      // retained points per frame over all levels (sum of max_kp)
    int32_t ini_th, min_th;
    unsigned long long* stamps;  // cycle sums per phase, diagnostic builds only (null otherwise)
    uint32_t* status;            // kStatus* words (host memory the device can write): non-zero = a kernel met a value outside its bounds
    int32_t pad0;
    int32_t blur_first[ORBX_MAX_LEVELS + 1];  // first blur block of each level; [nlevels] = total
};

}  // namespace orbx
#endif
