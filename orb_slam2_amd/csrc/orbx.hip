// C ABI of the gfx950 ORB extractor: context, device memory, launch sequence.
// Declared in include/orbx.h.  Compiled with: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kernels_describe.hpp"
#include "kernels_fast.hpp"
#include "kernels_octree.hpp"
#include "kernels_post.hpp"
#include "kernels_bow.hpp"
#include "kernels_bowmatch.hpp"
#include "kernels_initmatch.hpp"
#include "kernels_project.hpp"
#include "kernels_pyramid.hpp"
#include "kernels_staging.hpp"
#include "kernels_stereo.hpp"
#include "orbx_device.h"
#include "orbx_host.h"

using namespace orbx;

namespace {

const int8_t kPattern[1024] = {
#include "../../include/orb_pattern_31.inc"
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Dynamic LDS a launch of the LDS-heavy kernels may ask for: the CU's 160 KB less room for their
// small static arrays.  The limit is a property of the kernel on a device, shared by every context,
// so it is set to this one value and never to a context's own need.
constexpr int kMaxDynamicLds = 160 * 1024 - 256;
constexpr uint64_t kResizeChainMaxPixels = 1u << 19;  // launches of up to this many level-0 pixels build the pyramid three levels per launch
constexpr int kPipeSlots = 3;  // pinned upload slots and pinned result slots of the pipelined host batch path

// Scratch of the entry point that allocates nothing (orbx_stereo_match_batch_device_noalloc): a
// fixed array in the code object, one copy per GPU.  Host code reaches it only through the address
// hipGetSymbolAddress returns for the current device (kept per device below), never by name.
constexpr size_t kScratchBytes = ORBX_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) uint8_t scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
uint8_t* g_scratch_ptr[kMaxDevices];  // per device ordinal, null until looked up
// The array has one user at a time per GPU: every call that uses it waits, on its own stream, for the event the
// previous user recorded behind its kernels, and records the event again behind its own (under the mutex, so that
// wait, launches and record of two contexts do not interleave).  Two contexts on one GPU -- the stereo pattern of two
// extractor instances on two threads -- so take turns instead of racing.  Created with the address, never destroyed.
hipEvent_t g_scratch_event[kMaxDevices];
bool g_scratch_used[kMaxDevices];     // the event has been recorded at least once

// Address of the current device's copy of `scratch` (device `dev` must be current).
int scratch_address(int dev, uint8_t** out, int* hip_err) {
    if (dev < 0 || dev >= kMaxDevices) return ORBX_E_INVALID;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            *hip_err = (int)e;
            return ORBX_E_HIP;
        }
        hipEvent_t ev;
        const hipError_t e2 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e2 != hipSuccess) {
            *hip_err = (int)e2;
            return ORBX_E_HIP;
        }
        g_scratch_event[dev] = ev;
        g_scratch_ptr[dev] = (uint8_t*)p;
    }
    *out = g_scratch_ptr[dev];
    return ORBX_OK;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

constexpr int kArenaSets = 2;  // pyramid + blur arenas a context alternates between

enum {  // timing events of one batch
    kEvPyr0, kEvPyr1, kEvFastA0, kEvFastA1, kEvFastB0, kEvFastB1, kEvOct0, kEvOct1, kEvBlur0, kEvBlur1, kEvDesc0, kEvDesc1, kEvCount
};

struct orbx_ctx {
    orbx_params prm;
    Tables tab;
    hipStream_t stream = nullptr;
    int last_hip = 0;

    // geometry of the current image size
    int cur_w = -1, cur_h = -1;
    std::vector<LevelGeom> levels;
    std::vector<CellDesc> cells;
    std::vector<TileDesc> tiles;
    int rs_pitch[ORBX_MAX_LEVELS], rs_rows[ORBX_MAX_LEVELS];  // resize source tile in LDS
    bool rs_stream[ORBX_MAX_LEVELS];                          // level built by k_resize_rows
    bool fast_fixed_pitch = false;
    int fast_tile_bytes = 0, fast_plane_bytes = 0, fast_queue_bytes = 0, fast_kept_bytes = 0, fast_verdict_bytes = 0;
    size_t fast_lds = 0;
    int pitch[ORBX_MAX_LEVELS];
    size_t plane[ORBX_MAX_LEVELS];
    int cell_cap = 0, total_cells = 0, dense_total = 0, kp_total = 0, max_kp_level = 0;
    size_t oct_lds = 0;

    // device tables
    DevBuf d_cells, d_tiles, d_pat;
    DevBuf d_xt[ORBX_MAX_LEVELS], d_yt[ORBX_MAX_LEVELS];
    // device arenas (grown on demand)
    // Two sets of everything the descriptor stage reads (pyramid, blurred pyramid, retained points):
    // the descriptors of one batch run while the next batch is detected into the other set.
    DevBuf d_pyr[kArenaSets], d_blur[kArenaSets], d_lvl_kps[2], d_lvl_count[2];
    DevBuf d_cand[2], d_ccount[2];  // FAST output, alternating like the retained points (index kp_set)
    DevBuf d_dense_pts, d_dense_node;
    DevBuf d_oct_ws;          // quadtree tables of quotas that do not fit the LDS (oct_global), per (frame, level)
    bool oct_global = false;
    size_t oct_ws_stride = 0;
    int arena_set = 0, kp_set = 0;            // sets of the last batch
    bool arena_pending[kArenaSets] = {};      // ev_arena[i] recorded (behind the descriptors that last read arena set i) and not yet drained
    bool desc_pending[2] = {false, false};    // ev_desc[i] recorded and not yet joined
    bool oct_pending[2] = {false, false};     // ev_oct_set[i] recorded and not yet joined
    DevBuf d_out_kps, d_out_desc, d_out_n;  // results of the host-memory entry points
    DevBuf d_stamps;                        // diagnostic builds (-DORBX_STAMPS)

    // last batch (for read-backs and for the stereo stage)
    BatchDev B;
    bool have_last = false;
    bool last_chunked = false;           // the last batch ran chunk by chunk: FAST and quadtree lists exist for its last chunks only
    const orbx_kp* last_kps = nullptr;   // where the last extraction left its results (device)
    const uint8_t* last_desc = nullptr;
    const int32_t* last_n = nullptr;
    int last_kp_cap = 0;
    // pinned host staging for the host-memory entry points (pageable 2-D copies are done row by
    // row by the runtime and cost milliseconds; one contiguous copy from pinned memory does not)
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // pipelined host batches: two upload slots and two result slots in pinned memory, a stream per
    // copy direction, events chaining upload -> compute -> download per slot
    void* h_up[kPipeSlots] = {};
    void* h_res[kPipeSlots] = {};
    size_t h_up_bytes = 0, h_res_bytes = 0;
    hipStream_t st_up = nullptr, st_down = nullptr;
    hipEvent_t ev_up[kPipeSlots], ev_done[kPipeSlots], ev_down[kPipeSlots];
    bool pipe_made = false;
    // auxiliary compute stream (pyramid + blur beside FAST + quadtree) and its ordering events
    hipStream_t st_aux = nullptr;
    hipStream_t st_desc = nullptr;            // descriptor stage
    hipStream_t st_oct = nullptr;             // quadtree stage
    hipEvent_t ev_start, ev_pyr_done, ev_fast_done, ev_blur_done, ev_oct_done, ev_desc[2], ev_oct_set[2], ev_arena[kArenaSets];
    bool aux_made = false;
    int tiles_level0 = 0;  // the first tiles_level0 entries of `tiles` belong to level 0
    // stereo stage scratch (grown on demand)
    DevBuf d_st_rows, d_st_meta, d_st_wmax, d_st_sad, d_st_u, d_st_depth;
    DevBuf d_local_pick;  // scratch of orbx_search_local_map_batch_device (it shares minq, state and list with the call above)
    DevBuf d_proj_minq, d_proj_state, d_proj_list, d_proj_cache, d_proj_ncache;  // scratch of orbx_search_by_projection_batch_device

    // diagnostic switches: read from the environment once, in orbx_create; orbx_set_debug_flag changes them
    bool dbg_no_overlap = false, dbg_resize_tiles = false, dbg_stereo_border = false, dbg_trace = false, dbg_no_graph = false;
    int dbg_pipe_chunk = 0;
    int dbg_init_list = 0;
    bool dbg_aux_low_priority = false;
    // launch graph of the single-frame host path (captured once per image size, replayed by orbx_extract)
    hipGraphExec_t graph_exec = nullptr;
    int graph_w = -1, graph_h = -1, graph_cap = 0;
    int graph_n = 0;             // frames per launch: 1, or 2 (a stereo pair)
    bool graph_stereo = false;   // the pair's stereo association is part of the graph
    float graph_mb = 0.f, graph_mbf = 0.f;
    int graph_pad = 0;           // border of the padded pyramid the graph leaves in h_pad (0: none)
    bool capturing = false;
    void* h_gin = nullptr;   // pinned: the frames at device pitch (source of the graph's upload node)
    void* h_gout = nullptr;  // pinned: [n * cap keypoints][n * cap descriptors][4 counts][cap mvuRight][cap mvDepth] (targets of its download nodes)
    // The pyramid with its reflected border (mvImagePyramid's layout), for single-image calls: built on the device after
    // the extraction and copied to pinned host memory the caller reads in place (orbx_set_keep_pyramid)
    int pad_border = 0;
    DevBuf d_pad;
    void* h_pad = nullptr;
    size_t h_pad_bytes = 0;
    uint32_t pad_off[ORBX_MAX_LEVELS + 1] = {};
    bool pad_valid = false;

    uint8_t* scratch = nullptr;  // this device's copy of the library's static scratch array
    // kStatusWords words in pinned host memory that the kernels can write (see orbx_device.h): a kernel that meets a
    // count or a coordinate outside its bounds sets one; sync_ctx reads them after the streams have drained
    uint32_t* h_status = nullptr;
    uint32_t* d_status = nullptr;  // the same words as the device addresses them
    bool main_touched = false;     // something other than a batch was put on the main stream since the last batch (or drain)
    bool dbg_ordered = false;      // diagnostic: no batch runs ahead of the main stream (see run_batch, free_run)
    int dbg_poison = 0;            // test hook: 1 a cell count, 2 a level count, 3 a candidate point is overwritten on the device

    bool profiling = false;
    hipEvent_t ev[kEvCount];  // timing events, recorded when profiling is on
    bool ev_made = false;
    bool ev_valid = false;
};

namespace {

#define HIP_TRY(ctx, expr)                        \
    do {                                          \
        hipError_t e_ = (expr);                   \
        if (e_ != hipSuccess) {                   \
            (ctx)->last_hip = (int)e_;            \
            return ORBX_E_HIP;                    \
        }                                         \
    } while (0)

// The descriptors of a batch run on their own stream (see run_batch).  join() makes everything
// enqueued on the main stream from here on wait for them; sync_ctx() also drains the main stream.
int join(orbx_ctx* c) {
    c->main_touched = true;  // every caller is about to put work on the main stream (see run_batch, free_run)
    for (int i = 0; i < 2; ++i)
        if (c->desc_pending[i]) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_desc[i], 0));
            c->desc_pending[i] = false;
            c->oct_pending[i] = false;  // a batch's descriptors run after its quadtree
        }
    return ORBX_OK;
}
// `stream` waits until the descriptor stage that last read arena set `set` has finished.
int wait_arena_free(orbx_ctx* c, int set, hipStream_t stream) {
    if (c->arena_pending[set]) HIP_TRY(c, hipStreamWaitEvent(stream, c->ev_arena[set], 0));
    return ORBX_OK;
}
// Non-zero status words = some kernel since the last check read a value outside its bounds from device memory (it
// went on with the value cut to the bound, so nothing was accessed out of range).  Reported once, then cleared.
int check_status(orbx_ctx* c) {
    if (!c->h_status) return ORBX_OK;
    uint32_t any = 0;
    for (int i = 0; i < kStatusWords; ++i) {
        any |= reinterpret_cast<volatile uint32_t*>(c->h_status)[i];
        c->h_status[i] = 0;
    }
    return any ? ORBX_E_DEVICE_DATA : ORBX_OK;
}
int sync_ctx(orbx_ctx* c) {
    int rc = join(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->main_touched = false;  // drained
    for (bool& p : c->arena_pending) p = false;
    return check_status(c);
}

// Forgets the captured single-frame launch graph (see build_graph) and its pinned buffers.
void drop_graph(orbx_ctx* c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    c->graph_w = c->graph_h = -1;
    if (c->h_gin) (void)hipHostFree(c->h_gin);
    if (c->h_gout) (void)hipHostFree(c->h_gout);
    c->h_gin = c->h_gout = nullptr;
}

int ensure(orbx_ctx* c, DevBuf* b, size_t bytes) {
    if (bytes <= b->bytes) return ORBX_OK;
    if (b->p) {
        { int rc_ = sync_ctx(c); if (rc_) return rc_; }
        drop_graph(c);  // the graph's nodes may hold this buffer's address
        HIP_TRY(c, hipFree(b->p));
        b->p = nullptr;
        b->bytes = 0;
    }
    bytes = align_up(bytes + 256, 4096);
    HIP_TRY(c, hipMalloc(&b->p, bytes));
    b->bytes = bytes;
    return ORBX_OK;
}

void release(DevBuf* b) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->bytes = 0;
}

// Rebuilds every size-dependent table for a w x h input and uploads it.
int set_geometry(orbx_ctx* c, int w, int h) {
    if (w == c->cur_w && h == c->cur_h) return ORBX_OK;
    // Until the last table has been uploaded the context describes no size: a failure on the way
    // leaves device tables that match neither the old nor the new geometry, so the next call must
    // rebuild them all, and nothing may read the last batch through the old level table.
    c->cur_w = c->cur_h = -1;
    c->have_last = false;
    drop_graph(c);  // its kernel arguments describe the old size
    std::vector<LevelGeom> levels;
    std::vector<CellDesc> cells;
    int rc = build_level_geometry(c->tab, w, h, &levels, &cells);
    if (rc != ORBX_OK) return rc;
    int cell_cap = 0, dense_total = 0, kp_total = 0, max_kp_level = 0;
    size_t oct_lds = 0;
    for (LevelGeom& g : levels) {
        if (g.max_kp > 8192 || g.max_slots > 60000) return ORBX_E_GEOMETRY;
        if (g.w_cell + 6 > kFastMaxCell - 1 || g.h_cell + 6 > kFastMaxCell - 1) return ORBX_E_GEOMETRY;
        cell_cap = std::max(cell_cap, g.cell_cap);
        max_kp_level = std::max(max_kp_level, g.max_kp);
        oct_lds = std::max(oct_lds, oct_lds_bytes(g.max_slots, g.max_kp, g.n_cells));
    }
    const bool oct_global = oct_lds > 160 * 1024 - 256;  // the quadtree then keeps its tables in device memory
    for (LevelGeom& g : levels) {
        g.max_cand = g.n_cells * cell_cap;
        if (g.max_cand > 0xFFFFFF) return ORBX_E_GEOMETRY;  // the quadtree packs a candidate's index into 24 bits
        dense_total += g.max_cand;
        kp_total += g.max_kp;
    }
    std::vector<TileDesc> tiles;
    build_fast_tiles(levels, cells, &tiles);
    int tile_b = 0, plane_b = 0, queue_b = 0, kept_b = 0, verdict_b = 0;
    // Tiles up to kFastFixedPitch bytes wide (cells up to 36 pixels) all use that pitch, so that the kernel's
    // row offsets are constants; wider ones take the kernel with per-tile pitches.
    bool fixed_pitch = true;
    for (const TileDesc& t : tiles) fixed_pitch = fixed_pitch && fast_tile_pitch(t.wd) <= kFastFixedPitch;
    for (const TileDesc& t : tiles) {
        FastLds need = fast_lds_need(t.wd, t.hd, t.nc);
        if (fixed_pitch) need.tile = kFastFixedPitch * (t.hd + 6) + 16;
        tile_b = std::max(tile_b, need.tile);
        plane_b = std::max(plane_b, need.plane);
        queue_b = std::max(queue_b, need.queue);
        kept_b = std::max(kept_b, need.kept);
        verdict_b = std::max(verdict_b, need.verdict);
    }
    kept_b = (int)align_up((size_t)kept_b, 16);
    verdict_b = (int)align_up((size_t)verdict_b, 16);
    queue_b = (int)align_up((size_t)queue_b, 16);
    tile_b = (int)align_up((size_t)tile_b, 16);
    plane_b = (int)align_up((size_t)plane_b, 16);
    const size_t fast_lds = (size_t)tile_b + (size_t)plane_b + (size_t)queue_b + (size_t)kept_b + (size_t)verdict_b + 16;
    if (fast_lds > 160 * 1024 - 256) return ORBX_E_GEOMETRY;
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    rc = ensure(c, &c->d_cells, cells.size() * sizeof(CellDesc));
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(c->d_cells.p, cells.data(), cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    rc = ensure(c, &c->d_tiles, tiles.size() * sizeof(TileDesc));
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(c->d_tiles.p, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    for (int l = 0; l < c->tab.nlevels; ++l) {
        c->pitch[l] = (int)align_up((size_t)levels[l].w, 64);
        c->plane[l] = (size_t)c->pitch[l] * levels[l].h;
        if (l == 0) continue;
        std::vector<ResizeTap> xt, yt;
        build_resize_taps(levels[l - 1].w, levels[l].w, true, &xt);
        build_resize_taps(levels[l - 1].h, levels[l].h, false, &yt);
        {
            // largest source window of a 256 x 16 destination tile
            int span_c = 0, span_r = 0;
            for (size_t d0 = 0; d0 < xt.size(); d0 += kRsTW) {
                const size_t d1 = std::min(xt.size(), d0 + kRsTW) - 1;
                span_c = std::max(span_c, xt[d1].ofs1 - (xt[d0].ofs & ~15) + 1);
            }
            for (size_t d0 = 0; d0 < yt.size(); d0 += kRsTH) {
                const size_t d1 = std::min(yt.size(), d0 + kRsTH) - 1;
                span_r = std::max(span_r, yt[d1].ofs1 - yt[d0].ofs + 1);
            }
            c->rs_pitch[l] = resize_src_pitch(span_c);
            c->rs_rows[l] = span_r;
            const size_t need = (size_t)c->rs_pitch[l] * span_r + (kRsTW + kRsTH) * sizeof(ResizeTap);
            if (need > 60 * 1024 || (size_t)c->rs_pitch[l] * span_r > 32767) return ORBX_E_GEOMETRY;  // scale factor too large
        }
        {
            // the streaming kernel needs the taps of four adjacent columns within eight source bytes
            // and row taps that step forward by whole rows
            bool ok = true;
            for (size_t d0 = 0; d0 < xt.size() && ok; d0 += 4) {
                const size_t d1 = std::min(xt.size(), d0 + 4) - 1;
                for (size_t d = d0; d <= d1; ++d) ok = ok && xt[d].ofs >= xt[d0].ofs && xt[d].ofs1 >= xt[d0].ofs && xt[d].ofs1 - xt[d0].ofs <= 7;
            }
            for (size_t d = 0; d < yt.size() && ok; ++d) {
                ok = ok && (yt[d].ofs1 == yt[d].ofs || yt[d].ofs1 == yt[d].ofs + 1);
                if (d) ok = ok && yt[d].ofs >= yt[d - 1].ofs && yt[d].ofs1 >= yt[d - 1].ofs1;
                // a source row completes at most two destination rows (two only where the taps are clamped to the last row)
                if (d >= 2) ok = ok && yt[d].ofs1 != yt[d - 2].ofs1;
            }
            c->rs_stream[l] = ok;
        }
        rc = ensure(c, &c->d_xt[l], xt.size() * sizeof(ResizeTap));
        if (rc) return rc;
        rc = ensure(c, &c->d_yt[l], yt.size() * sizeof(ResizeTap));
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(c->d_xt[l].p, xt.data(), xt.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_yt[l].p, yt.data(), yt.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    }
    c->levels.swap(levels);
    c->cells.swap(cells);
    c->tiles.swap(tiles);
    c->tiles_level0 = 0;
    for (const TileDesc& t : c->tiles) c->tiles_level0 += t.level == 0 ? 1 : 0;
    c->fast_tile_bytes = tile_b;
    c->fast_plane_bytes = plane_b;
    c->fast_queue_bytes = queue_b;
    c->fast_fixed_pitch = fixed_pitch;
    c->fast_kept_bytes = kept_b;
    c->fast_verdict_bytes = verdict_b;
    c->fast_lds = fast_lds;
    c->cell_cap = cell_cap;
    c->total_cells = (int)c->cells.size();
    c->dense_total = dense_total;
    c->kp_total = kp_total;
    c->max_kp_level = max_kp_level;
    c->oct_lds = oct_global ? 0 : oct_lds;
    c->oct_global = oct_global;
    c->oct_ws_stride = oct_global ? align_up(oct_lds, 256) : 0;
    c->cur_w = w;
    c->cur_h = h;
    c->have_last = false;
    return ORBX_OK;
}

constexpr size_t kStageMax = (size_t)64 << 20;  // staging chunk for uploads

int ensure_stage(orbx_ctx* c, size_t bytes) {
    if (bytes <= c->h_stage_bytes) return ORBX_OK;
    if (c->h_stage) {
        { int rc_ = sync_ctx(c); if (rc_) return rc_; }
        HIP_TRY(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
    }
    HIP_TRY(c, hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault));
    c->h_stage_bytes = bytes;
    return ORBX_OK;
}

// Device and pinned host room for the padded pyramid of one frame of the current geometry; fills pad_off.
int ensure_pad(orbx_ctx* c) {
    const int b = c->pad_border;
    size_t total = 0;
    for (int l = 0; l < c->tab.nlevels; ++l) {
        c->pad_off[l] = (uint32_t)total;
        total += align_up((size_t)(c->levels[l].w + 2 * b) * (c->levels[l].h + 2 * b), 64);
    }
    c->pad_off[c->tab.nlevels] = (uint32_t)total;
    int rc = ensure(c, &c->d_pad, total);
    if (rc) return rc;
    if (total > c->h_pad_bytes) {
        { int rc_ = sync_ctx(c); if (rc_) return rc_; }
        drop_graph(c);
        if (c->h_pad) HIP_TRY(c, hipHostFree(c->h_pad));
        c->h_pad = nullptr;
        c->h_pad_bytes = 0;
        c->pad_valid = false;
        HIP_TRY(c, hipHostMalloc(&c->h_pad, total, hipHostMallocDefault));
        c->h_pad_bytes = total;
    }
    return ORBX_OK;
}

// Enqueues, on the main stream, the padded pyramid of frame 0 of c->B and its copy to c->h_pad (ensure_pad has run).
int enqueue_pad(orbx_ctx* c) {
    PadDev P;
    std::memset(&P, 0, sizeof(P));
    const BatchDev& B = c->B;
    for (int l = 0; l < B.nlevels; ++l) {
        P.src[l] = B.lv[l].img;
        P.spitch[l] = B.lv[l].pitch;
        P.w[l] = B.lv[l].w;
        P.h[l] = B.lv[l].h;
    }
    for (int l = 0; l <= B.nlevels; ++l) P.off[l] = c->pad_off[l];
    P.dst = (uint8_t*)c->d_pad.p;
    P.nlevels = B.nlevels;
    P.border = c->pad_border;
    const uint32_t total = c->pad_off[B.nlevels];
    hipLaunchKernelGGL(k_pad_levels, dim3((total + 255) / 256), dim3(256), 0, c->stream, P);
    HIP_TRY(c, hipMemcpyAsync(c->h_pad, c->d_pad.p, total, hipMemcpyDeviceToHost, c->stream));
    return ORBX_OK;
}

// Host images -> level 0 of the pyramid arena (device pitch), through the pinned staging buffer.
int upload_level0(orbx_ctx* c, int set, int n, const uint8_t* const* imgs, const int* strides, int width, int height) {
    const size_t plane = c->plane[0];
    const int pitch = c->pitch[0];
    const int per_chunk = (int)std::max<size_t>(1, kStageMax / plane);
    int rc = ensure_stage(c, plane * (size_t)std::min(n, per_chunk));
    if (rc) return rc;
    if ((rc = wait_arena_free(c, set, c->stream))) return rc;  // level 0 of this set may still be read by descriptors
    for (int i0 = 0; i0 < n; i0 += per_chunk) {
        const int m = std::min(per_chunk, n - i0);
        if (i0 > 0) { int rc_ = sync_ctx(c); if (rc_) return rc_; }  // the previous chunk has left the staging buffer
        for (int i = 0; i < m; ++i) {
            uint8_t* d = (uint8_t*)c->h_stage + (size_t)i * plane;
            const uint8_t* s = imgs[i0 + i];
            for (int y = 0; y < height; ++y) std::memcpy(d + (size_t)y * pitch, s + (size_t)y * strides[i0 + i], (size_t)width);
        }
        HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->d_pyr[set].p + (size_t)i0 * plane, c->h_stage, (size_t)m * plane,
                                  hipMemcpyHostToDevice, c->stream));
    }
    return ORBX_OK;
}

int ensure_arenas(orbx_ctx* c, int n, bool need_level0) {
    size_t pyr = 0, blur = 0;
    for (int l = 0; l < c->tab.nlevels; ++l) {
        if (l > 0 || need_level0) pyr += c->plane[l] * n;
        blur += c->plane[l] * n;
    }
    int rc;
    for (int k = 0; k < kArenaSets; ++k) {
        if ((rc = ensure(c, &c->d_pyr[k], pyr))) return rc;
        if ((rc = ensure(c, &c->d_blur[k], blur))) return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure(c, &c->d_lvl_kps[k], (size_t)n * c->kp_total * 4))) return rc;
        if ((rc = ensure(c, &c->d_lvl_count[k], (size_t)n * c->tab.nlevels * 4))) return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure(c, &c->d_cand[k], (size_t)n * c->total_cells * c->cell_cap * 4))) return rc;
        if ((rc = ensure(c, &c->d_ccount[k], (size_t)n * c->total_cells * 4))) return rc;
    }
    if ((rc = ensure(c, &c->d_dense_pts, (size_t)n * c->dense_total * 4))) return rc;
    if (c->oct_global && (rc = ensure(c, &c->d_oct_ws, (size_t)n * c->tab.nlevels * c->oct_ws_stride))) return rc;
    if ((rc = ensure(c, &c->d_dense_node, (size_t)n * c->dense_total * 2))) return rc;
    return ORBX_OK;
}

// Describes n frames in c->B: frames frame0 .. frame0+n-1 of arenas laid out for n_total frames
// (level-major), level 0 at (l0, l0_pitch, l0_plane) when given, else in the pyramid arena too.
void fill_batch(orbx_ctx* c, int n, const uint8_t* l0, int l0_pitch, size_t l0_plane, int n_total, int frame0, int set) {
    const int L = c->tab.nlevels;
    BatchDev& B = c->B;
    std::memset(&B, 0, sizeof(B));
    B.nlevels = L;
    B.nframes = n;
    B.total_cells = c->total_cells;
    B.cell_cap = c->cell_cap;
    B.dense_total = c->dense_total;
    B.kp_total = c->kp_total;
    B.ini_th = c->prm.ini_th_fast;
    B.min_th = c->prm.min_th_fast;
    B.status = c->d_status;
#if defined(ORBX_STAMPS) || defined(ORBX_STAMPS_DESC)
    if (!c->d_stamps.p && ensure(c, &c->d_stamps, 1024 * 16 * 8) == ORBX_OK) (void)hipMemset(c->d_stamps.p, 0, 1024 * 16 * 8);
    B.stamps = (unsigned long long*)c->d_stamps.p;
#endif
    size_t poff = 0, boff = 0;
    int dense_off = 0, kp_off = 0;
    const bool l0_in_arena = (l0 == nullptr);
    for (int l = 0; l < L; ++l) {
        const LevelGeom& g = c->levels[l];
        LevelDev& D = B.lv[l];
        if (l == 0 && !l0_in_arena) {
            D.img = l0;
            D.pitch = l0_pitch;
            D.plane = l0_plane;
        } else {
            D.img = (const uint8_t*)c->d_pyr[set].p + poff + c->plane[l] * (size_t)frame0;
            D.pitch = c->pitch[l];
            D.plane = c->plane[l];
            poff += c->plane[l] * n_total;
        }
        D.blur = (uint8_t*)c->d_blur[set].p + boff + c->plane[l] * (size_t)frame0;
        D.bpitch = c->pitch[l];
        D.bplane = c->plane[l];
        boff += c->plane[l] * n_total;
        D.w = g.w;
        D.h = g.h;
        D.n_cells = g.n_cells;
        D.cell_base = g.cell_base;
        D.n_ini = g.n_ini;
        D.h_x = g.h_x;
        D.rect_w = g.rect_w;
        D.rect_h = g.rect_h;
        D.quota = g.quota;
        D.max_kp = g.max_kp;
        D.max_slots = g.max_slots;
        D.dense_off = dense_off;
        D.kp_off = kp_off;
        D.scale = c->tab.scale[l];
        D.size = (float)(int)(31 * c->tab.scale[l]);  // reference :930
        dense_off += g.max_cand;
        kp_off += g.max_kp;
        B.blur_first[l + 1] = B.blur_first[l] + blur_main_blocks(g.w, g.h);
    }
}

// The auxiliary and the descriptor stream with the events that order them (see run_batch).
int ensure_aux_streams(orbx_ctx* c) {
    if (c->aux_made) return ORBX_OK;
    // All streams at the default priority: with low-priority streams a wait on one of them from a
    // default-priority stream stalled for tens of seconds (measured with the chunked host path).
#ifdef ORBX_DIAG_LOW_PRIORITY  // diagnostic build only: the configuration in which the chunked host path stalled
    if (c->dbg_aux_low_priority) {
        int least = 0, greatest = 0;
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->st_aux, hipStreamNonBlocking, least));
    } else
#endif
    {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->st_aux, hipStreamNonBlocking));
    }
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pyr_done, hipEventDisableTiming));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->st_desc, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_blur_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fast_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_oct_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_desc[0], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_desc[1], hipEventDisableTiming));
    for (int i = 0; i < kArenaSets; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_arena[i], hipEventDisableTiming));
#ifdef ORBX_DIAG_FOUR_STREAMS  // diagnostic build: the quadtree on a fourth stream of its own (the layout the comment below describes the cost of)
    HIP_TRY(c, hipStreamCreateWithFlags(&c->st_oct, hipStreamNonBlocking));
#else
    // The quadtree shares the descriptor stream.  A process has four hardware queues, one of them the null stream's;
    // with a fourth compute stream two of the context's streams share a queue, the runtime picks which, and kernels
    // on one queue run strictly one after the other.  The first context of a process got the quadtree and the
    // descriptors together (harmless: a batch's descriptors wait for its quadtree anyway), a context created after
    // another one had been destroyed got FAST and the quadtree together and ran every batch 5 % slower (10 % on 4K
    // frames).  With three streams every one has a queue to itself whatever the process did before, and the
    // quadtree and the descriptors are ordered the way the lucky mapping ordered them.
    c->st_oct = c->st_desc;
#endif
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_oct_set[0], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_oct_set[1], hipEventDisableTiming));
    c->aux_made = true;
    return ORBX_OK;
}

// Enqueues the whole path for n frames (see fill_batch) on the context's stream.
int run_batch(orbx_ctx* c, int n, const uint8_t* l0, int l0_pitch, size_t l0_plane, orbx_kp* d_kps, int kp_cap,
              uint8_t* d_desc, int32_t* d_n_out, int arena_set, int kp_set, int n_total = -1, int frame0 = 0,
              bool arena_known_free = false, bool inputs_at_call = false) {
    fill_batch(c, n, l0, l0_pitch, l0_plane, n_total < 0 ? n : n_total, frame0, arena_set);
    c->arena_set = arena_set;
    c->kp_set = kp_set;
    const int L = c->tab.nlevels;
    BatchDev& B = c->B;
    // Three streams.  The main stream runs FAST: on level 0 (which needs no pyramid) at once and on
    // the other levels when the pyramid is there, or in one launch (below).  The auxiliary
    // stream builds the pyramid and blurs it.  The descriptor stream runs the quadtree once FAST has
    // finished and the last stage once the blur has finished too.  The memory-bound pyramid and blur so fill what the
    // VALU-bound FAST and the latency-bound quadtree leave idle, and the next batch, which works in
    // the other set of arenas, is detected while this one is described.
    // The no-overlap switch (ORBX_NO_OVERLAP=1) puts everything on the main stream in stage order.
    const bool overlap = !c->dbg_no_overlap;
    // A batch whose level-0 images were complete in memory when the call was made (inputs_at_call), with nothing put on
    // the main stream since the batch before it, need not take its place in the main stream's order before its
    // pyramid is built: the auxiliary stream starts on it as soon as the arena set is free (the descriptors of the
    // batch before the previous one), builds the pyramid and blurs it while the main stream still detects the batch
    // before, and the main stream finds the pyramid ready.  Any other batch (its images come from a copy or a staging
    // kernel on the main stream, or the main stream holds reads of the arenas) is ordered the other way: pyramid
    // behind the main stream's position, blur behind FAST.  Measured on 512-frame batches enqueued back to back:
    // 1.94 -> 1.87 ms per batch; with the blur moved up but the pyramid still behind the main stream, 1.96 ms.
#ifdef ORBX_DIAG_BREAK_ORDER  // diagnostic build: the ordering tests must fail on it
    const bool free_run = overlap && inputs_at_call && !c->dbg_ordered;
#else
    const bool free_run = overlap && inputs_at_call && !c->main_touched && !c->dbg_ordered;
#endif
    const bool pipeline_busy = c->desc_pending[0] || c->desc_pending[1];
    // A quadtree of 1024-thread workgroups with the node tables of a level of more than a thousand features in LDS (4K
    // frames: most of a CU's 160 KB) finds no room on a CU while FAST's four-wave, 18 KB workgroups keep arriving:
    // beside FAST it crawls until FAST's grid is exhausted, and everything behind it (the descriptors, the arena set,
    // the next pyramid, the next FAST) waits.  With such a quadtree the previous batch's quadtree goes before this
    // batch's FAST, and this batch's pyramid runs beside it.  128 4K frames per step: 9.5 -> 9.1 ms.
    const bool wide_octree = c->oct_global || c->max_kp_level > kOctWideFrom;
    hipStream_t st = c->stream;
    if (overlap) { int rc_ = ensure_aux_streams(c); if (rc_) return rc_; }
    hipStream_t aux = overlap ? c->st_aux : st;
    const bool prof = c->profiling && !c->capturing;  // timing events are not captured into a graph
    auto mark = [&](int i, hipStream_t s) -> int {  // timing event i (see orbx_get_stage_ms)
        if (prof) HIP_TRY(c, hipEventRecord(c->ev[i], s));
        return ORBX_OK;
    };
    int rc;
    const int n_tiles = (int)c->tiles.size(), n_tiles0 = c->tiles_level0;
    const bool blur_fm = n >= 8;  // as for the FAST tiles: a frame's blocks on one XCD
    const dim3 blur_grid = blur_fm ? dim3(8, (unsigned)B.blur_first[L], (unsigned)((n + 7) / 8)) : dim3((unsigned)B.blur_first[L], n);
    auto launch_fast = [&](int first, int count) {
        if (count > 0) {
            // batches of 8 frames or more: the frames of a group of 8 in grid x, so that each XCD gets
            // whole frames (neighbouring tiles share halo rows: 1.67 GB -> 0.72 GB fetched per 512 frames)
            const bool fm = n >= 8;
            const dim3 grid = fm ? dim3(8, (unsigned)count, (unsigned)((n + 7) / 8)) : dim3((unsigned)count, n);
            if (c->fast_fixed_pitch)
                hipLaunchKernelGGL(k_fast_tiles<kFastFixedPitch>, grid, dim3(256), c->fast_lds, st, B,
                                   (const TileDesc*)c->d_tiles.p + first, (uint32_t*)c->d_cand[kp_set].p, (uint32_t*)c->d_ccount[kp_set].p,
                                   c->fast_tile_bytes, c->fast_plane_bytes, c->fast_queue_bytes, c->fast_kept_bytes,
                                   c->fast_verdict_bytes, fm ? 1 : 0);
            else
                hipLaunchKernelGGL(k_fast_tiles<0>, grid, dim3(256), c->fast_lds, st, B,
                                   (const TileDesc*)c->d_tiles.p + first, (uint32_t*)c->d_cand[kp_set].p, (uint32_t*)c->d_ccount[kp_set].p,
                                   c->fast_tile_bytes, c->fast_plane_bytes, c->fast_queue_bytes, c->fast_kept_bytes,
                                   c->fast_verdict_bytes, fm ? 1 : 0);
        }
    };
    if (overlap) {
        // level 0 and the arenas are ready when the main stream gets here
        // (a batch with a wide quadtree, below: its pyramid starts when the previous batch's FAST has finished, beside that
        // batch's quadtree, which leaves the memory system idle)
        if (!free_run || wide_octree) {
            HIP_TRY(c, hipEventRecord(c->ev_start, st));
            HIP_TRY(c, hipStreamWaitEvent(aux, c->ev_start, 0));
        }
        c->main_touched = false;  // whatever was put on the main stream so far is in front of this batch, and of every later one
        // the pyramid and the blur overwrite this arena set: its last reader must have finished
        if (!arena_known_free && (rc = wait_arena_free(c, arena_set, aux))) return rc;
    }
    // stage 0 (aux): pyramid
    if ((rc = mark(kEvPyr0, aux))) return rc;
    // a launch of a few frames: three levels per launch (k_resize_chain) instead of one
    const bool chain = !c->dbg_resize_tiles && (uint64_t)n * B.lv[0].w * B.lv[0].h <= kResizeChainMaxPixels;
    for (int l = 1; chain && l < L; l += 3) {
        const LevelDev& S = B.lv[l - 1];
        ResizeChain C;
        C.src = S.img;
        C.spitch = S.pitch;
        C.splane = S.plane;
        C.nlev = std::min(3, L - l);
        int px = 0;
        for (int k = 0; k < 3; ++k) {
            const LevelDev& D = B.lv[std::min(l + k, L - 1)];
            C.dst[k] = const_cast<uint8_t*>(D.img);
            C.dpitch[k] = D.pitch;
            C.dplane[k] = D.plane;
            C.dw[k] = D.w;
            C.dh[k] = D.h;
            C.xt[k] = (const ResizeTap*)c->d_xt[std::min(l + k, L - 1)].p;
            C.yt[k] = (const ResizeTap*)c->d_yt[std::min(l + k, L - 1)].p;
            if (k < C.nlev) px = std::max(px, D.w * D.h);
        }
        hipLaunchKernelGGL(k_resize_chain, dim3((px + 255) / 256, C.nlev, n), dim3(256), 0, aux, C);
    }
    for (int l = 1; !chain && l < L; ++l) {
        const LevelDev& S = B.lv[l - 1];
        const LevelDev& D = B.lv[l];
        // the resize-tiles switch (diagnostic) keeps the LDS-tiled kernel for every level
        if (c->rs_stream[l] && !c->dbg_resize_tiles && S.pitch >= 8 &&
            (uint64_t)n * S.plane < 0x7FFFFFF0ull) {
            const int nd = (D.w + 3) / 4;
            // destination rows per lane: long bands filter fewer source rows twice, short ones give
            // the small levels enough waves to hide their load latency
            const unsigned blocks_x = ((unsigned)n * nd + 255) / 256;
#ifndef ORBX_RS_ROWS
#define ORBX_RS_ROWS 32
#endif
            int rows = ORBX_RS_ROWS;
#ifndef ORBX_RS_WAVES
#define ORBX_RS_WAVES 4096
#endif
            while (rows > 8 && (uint64_t)blocks_x * 4 * ((D.h + rows - 1) / rows) < ORBX_RS_WAVES) rows >>= 1;
            dim3 grid_rows(blocks_x, (D.h + rows - 1) / rows);
            hipLaunchKernelGGL(k_resize_rows, grid_rows, dim3(256), 0, aux, S.img, S.pitch, S.plane, const_cast<uint8_t*>(D.img),
                               D.pitch, D.plane, D.w, D.h, (const ResizeTap*)c->d_xt[l].p, (const ResizeTap*)c->d_yt[l].p, n,
                               rows);
            continue;
        }
        dim3 grid((D.w + kRsTW - 1) / kRsTW, (D.h + kRsTH - 1) / kRsTH, n);
        const size_t lds = (size_t)c->rs_pitch[l] * c->rs_rows[l] + (kRsTW + kRsTH) * sizeof(ResizeTap);
        hipLaunchKernelGGL(k_resize, grid, dim3(256), lds, aux, S.img, S.pitch, S.plane, S.w, S.h,
                           const_cast<uint8_t*>(D.img), D.pitch, D.plane, D.w, D.h, (const ResizeTap*)c->d_xt[l].p,
                           (const ResizeTap*)c->d_yt[l].p, c->rs_pitch[l], c->rs_rows[l]);
    }
    if ((rc = mark(kEvPyr1, aux))) return rc;
    if (overlap) {
        HIP_TRY(c, hipEventRecord(c->ev_pyr_done, aux));
    }
    // stage 1 (main): FAST cells, level 0 first.  It overwrites the candidate lists of set kp_set, which the
    // quadtree of the batch before the previous one may still be reading.
    if (overlap && c->oct_pending[kp_set]) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_oct_set[kp_set], 0));
    if (overlap && wide_octree && c->oct_pending[kp_set ^ 1]) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_oct_set[kp_set ^ 1], 0));
    if ((rc = mark(kEvFastA0, st))) return rc;
    // Level 0 needs no pyramid and goes first, the other levels when the pyramid is there.  A free-running batch behind
    // another one finds its pyramid built long before the main stream gets to it: one launch for all levels then
    // (0.7 % of the step: no drained GPU between the two launches).
    const bool split = overlap && !(free_run && pipeline_busy);
    launch_fast(0, split ? n_tiles0 : 0);
    if ((rc = mark(kEvFastA1, st))) return rc;
    if (overlap) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_pyr_done, 0));
    if ((rc = mark(kEvFastB0, st))) return rc;
    launch_fast(split ? n_tiles0 : 0, split ? n_tiles - n_tiles0 : n_tiles);
    if ((rc = mark(kEvFastB1, st))) return rc;
    // test hook (orbx_set_debug_flag, ORBX_FLAG_POISON): what the detector left for frame 0, cell 0 is overwritten,
    // so that a test can see the consumers of device-side counts cut them to their bounds and report them
    if (c->dbg_poison == 1) HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_ccount[kp_set].p, 0x7FFFFFFF, 1, st));
    if (c->dbg_poison == 3) {
        HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_ccount[kp_set].p, 1, 1, st));
        HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_cand[kp_set].p, 0x00FFFFFF, 1, st));
    }
    if (overlap) {
        // stage 3 (aux): blur of all levels.  It is held back until FAST has finished: started earlier
        // it only slows FAST down (both are bound by vector instruction issue), started here it runs
        // beside the quadtree and the descriptor stage, which leave the vector units idle.
        HIP_TRY(c, hipEventRecord(c->ev_fast_done, st));
        // (a batch with nothing in front of it has nothing else to run beside: its blur then waits for FAST as in the
        // ordered schedule, 2.25 -> 2.22 ms for one 512-frame batch at a time and 0.39 -> 0.37 ms for 64 frames; a batch
        // of a few frames leaves most of the GPU idle anyway and is 7 % faster with the blur beside FAST)
        const bool blur_behind_fast = !free_run || (!pipeline_busy && n >= 32);
        if (blur_behind_fast) HIP_TRY(c, hipStreamWaitEvent(aux, c->ev_fast_done, 0));
        if ((rc = mark(kEvBlur0, aux))) return rc;
        hipLaunchKernelGGL(k_blur7_all, blur_grid, dim3(256), 0, aux, B, blur_fm ? 1 : 0);
        if ((rc = mark(kEvBlur1, aux))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_blur_done, aux));
    }
    // stage 2: quadtree retention, off the main stream so that the next batch's FAST (which writes the other
    // set of candidate lists) starts beside it: the quadtree is bound by latency and leaves the vector units to
    // FAST.  It overwrites the retained points of set kp_set, which the descriptor stage of the batch before the
    // previous one may still be reading.
    hipStream_t so = overlap ? c->st_oct : st;
    if (overlap) {
        HIP_TRY(c, hipStreamWaitEvent(so, c->ev_fast_done, 0));
        if (c->desc_pending[kp_set]) HIP_TRY(c, hipStreamWaitEvent(so, c->ev_desc[kp_set], 0));
    }
    if ((rc = mark(kEvOct0, so))) return rc;
    // 1024 threads per (frame, level) when a level keeps more than a thousand features (the loops over the
    // points and the rank-by-counting of the nodes are then long) or when the batch is so small that the launch
    // leaves most of the GPU idle anyway (a single frame: the level's sweeps are the critical path), else 256
    if (c->oct_global)
        hipLaunchKernelGGL((k_octree<1024, true>), dim3(n, L), dim3(1024), 0, so, B, (const uint32_t*)c->d_cand[kp_set].p,
                           (const uint32_t*)c->d_ccount[kp_set].p, (uint32_t*)c->d_dense_pts.p, (uint16_t*)c->d_dense_node.p,
                           (uint32_t*)c->d_lvl_kps[kp_set].p, (int32_t*)c->d_lvl_count[kp_set].p, (uint8_t*)c->d_oct_ws.p,
                           c->oct_ws_stride);
    else if (c->max_kp_level > kOctWideFrom || n * L <= kOctWideBelowBlocks)
        hipLaunchKernelGGL((k_octree<1024, false>), dim3(n, L), dim3(1024), c->oct_lds, so, B, (const uint32_t*)c->d_cand[kp_set].p,
                           (const uint32_t*)c->d_ccount[kp_set].p, (uint32_t*)c->d_dense_pts.p, (uint16_t*)c->d_dense_node.p,
                           (uint32_t*)c->d_lvl_kps[kp_set].p, (int32_t*)c->d_lvl_count[kp_set].p, (uint8_t*)nullptr, (size_t)0);
    else
        hipLaunchKernelGGL((k_octree<256, false>), dim3(n, L), dim3(256), c->oct_lds, so, B, (const uint32_t*)c->d_cand[kp_set].p,
                           (const uint32_t*)c->d_ccount[kp_set].p, (uint32_t*)c->d_dense_pts.p, (uint16_t*)c->d_dense_node.p,
                           (uint32_t*)c->d_lvl_kps[kp_set].p, (int32_t*)c->d_lvl_count[kp_set].p, (uint8_t*)nullptr, (size_t)0);
    if ((rc = mark(kEvOct1, so))) return rc;
    if (c->dbg_poison == 2) HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_lvl_count[kp_set].p, 0x7FFFFFFF, 1, so));
    hipStream_t sd = st;  // stream of the descriptor stage
    if (overlap) {
        // stage 4 on its own stream, after the quadtree and the blur; the other streams are then free
        // for the next batch (join() orders anything that needs this batch's results behind it)
        sd = c->st_desc;
        HIP_TRY(c, hipEventRecord(c->ev_oct_done, so));
        HIP_TRY(c, hipEventRecord(c->ev_oct_set[kp_set], so));
        c->oct_pending[kp_set] = true;
        if (sd != so) HIP_TRY(c, hipStreamWaitEvent(sd, c->ev_oct_done, 0));  // (one stream: already in order)
        HIP_TRY(c, hipStreamWaitEvent(sd, c->ev_blur_done, 0));
    } else {
        if ((rc = mark(kEvBlur0, st))) return rc;
        hipLaunchKernelGGL(k_blur7_all, blur_grid, dim3(256), 0, st, B, blur_fm ? 1 : 0);
        if ((rc = mark(kEvBlur1, st))) return rc;
    }
    // stage 4: orientation + descriptors + output assembly
    if ((rc = mark(kEvDesc0, sd))) return rc;
    {
        // every level gets the block count of the largest one; the blocks beyond a level's own count leave at once.
        // (A grid with each level's own block count, 40 % fewer workgroups, left this kernel's time alone unchanged
        // and made the overlapped step 4 % slower.)
        const int kp_blocks = (c->max_kp_level + kDescKp - 1) / kDescKp;
        const unsigned grid = 8u * (unsigned)((n + 7) / 8) * (unsigned)L * (unsigned)kp_blocks;
        hipLaunchKernelGGL(k_orient_describe, dim3(grid), dim3(256), 0, sd, B, (const uint32_t*)c->d_lvl_kps[kp_set].p,
                           (const int32_t*)c->d_lvl_count[kp_set].p, (const uint4*)c->d_pat.p, d_kps, d_desc, d_n_out, kp_cap,
                           kp_blocks);
    }
    if ((rc = mark(kEvDesc1, sd))) return rc;
    if (overlap) {
        HIP_TRY(c, hipEventRecord(c->ev_desc[kp_set], sd));
        c->desc_pending[kp_set] = true;
        HIP_TRY(c, hipEventRecord(c->ev_arena[arena_set], sd));
        c->arena_pending[arena_set] = true;
    }
    if (prof) c->ev_valid = true;
    HIP_TRY(c, hipGetLastError());
    c->have_last = true;
    c->last_chunked = false;
    c->last_kps = d_kps;
    c->last_desc = d_desc;
    c->last_n = d_n_out;
    c->last_kp_cap = kp_cap;
    return ORBX_OK;
}

// Per-pair scratch of the stereo stage (see StereoDev); every word the kernels read they wrote first.
struct StereoScratch {
    uint32_t* row_start;
    StereoMeta* meta;
    int32_t* wmax;
    int32_t* sad;
};
// The context's own (grown on demand) scratch for n_pairs pairs of kp_cap keypoints and `rows` image rows.
int stereo_scratch_ctx(orbx_ctx* l, int n_pairs, int kp_cap, int rows, StereoScratch* out) {
    int rc;
    if ((rc = ensure(l, &l->d_st_rows, (size_t)n_pairs * (rows + 1) * 4))) return rc;
    if ((rc = ensure(l, &l->d_st_meta, (size_t)n_pairs * kp_cap * sizeof(StereoMeta)))) return rc;
    if ((rc = ensure(l, &l->d_st_wmax, (size_t)n_pairs * 4))) return rc;
    if ((rc = ensure(l, &l->d_st_sad, (size_t)n_pairs * kp_cap * 4))) return rc;
    out->row_start = (uint32_t*)l->d_st_rows.p;
    out->meta = (StereoMeta*)l->d_st_meta.p;
    out->wmax = (int32_t*)l->d_st_wmax.p;
    out->sad = (int32_t*)l->d_st_sad.p;
    return ORBX_OK;
}

// Enqueues the stereo stage for n_pairs pairs on `l`'s stream.  Left images are first_l + p*step of
// `l`'s last batch, right images first_r + p*step of `r`'s; keypoints/descriptors/counts of pair p
// are at kps_x + p*kp_step (descriptors 32 bytes each, counts at n_x[p*n_step]).
int run_stereo(orbx_ctx* l, orbx_ctx* r, int n_pairs, int first_l, int first_r, int step, const orbx_kp* kps_l,
               const orbx_kp* kps_r, const uint8_t* desc_l, const uint8_t* desc_r, const int32_t* n_l,
               const int32_t* n_r, int64_t kp_step, int n_step, int kp_cap, float mb, float mbf, float* d_u_right,
               float* d_depth, const StereoScratch& scr) {
    const BatchDev& BL = l->B;
    const BatchDev& BR = r->B;
    const int L = BL.nlevels;
    StereoDev S;
    std::memset(&S, 0, sizeof(S));
    for (int i = 0; i < L; ++i) {
        S.img_l[i] = BL.lv[i].img;
        S.img_r[i] = BR.lv[i].img;
        S.plane_l[i] = BL.lv[i].plane;
        S.plane_r[i] = BR.lv[i].plane;
        S.pitch_l[i] = BL.lv[i].pitch;
        S.pitch_r[i] = BR.lv[i].pitch;
        S.w[i] = BL.lv[i].w;
        S.h[i] = BL.lv[i].h;
        S.scale[i] = l->tab.scale[i];
        S.inv_scale[i] = l->tab.inv_scale[i];
    }
    S.nlevels = L;
    S.first_l = first_l;
    S.first_r = first_r;
    S.step = step;
    S.kps_l = kps_l;
    S.kps_r = kps_r;
    S.desc_l = desc_l;
    S.desc_r = desc_r;
    S.n_l = n_l;
    S.n_r = n_r;
    S.kp_step = kp_step;
    S.n_step = n_step;
    S.kp_cap = kp_cap;
    S.rows = BL.lv[0].h;
    S.max_d = mbf / mb;  // :517, float division
    S.mbf = mbf;
    // diagnostic switch: every window is loaded through the reflecting byte loader
    S.border_only = l->dbg_stereo_border ? 1 : 0;
    S.row_start = scr.row_start;
    S.meta = scr.meta;
    S.wmax = scr.wmax;
    S.sad = scr.sad;
    S.u_right = d_u_right;
    S.depth = d_depth;
    hipStream_t st = l->stream;
    hipLaunchKernelGGL(k_stereo_rows, dim3(n_pairs), dim3(256), (size_t)(S.rows + 1) * 4, st, S);
    hipLaunchKernelGGL(k_stereo_search, dim3((kp_cap + kStereoKpPerBlock - 1) / kStereoKpPerBlock, n_pairs), dim3(256), 0,
                       st, S);
    hipLaunchKernelGGL(k_stereo_cut, dim3(n_pairs), dim3(1024), (size_t)((kp_cap + 3) & ~3) * 4, st, S);
    HIP_TRY(l, hipGetLastError());
    return ORBX_OK;
}

// Runs fn(i) for i in [0, n) on up to `threads` host threads (the calling thread is one of them).
template <class F>
void parallel_for(int n, int threads, F fn) {
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
        pool.emplace_back([=] {
            for (int i = t; i < n; i += threads) fn(i);
        });
    for (int i = 0; i < n; i += threads) fn(i);
    for (std::thread& th : pool) th.join();
}

constexpr int kPipeMinFrames = 16;            // smaller batches take the single-shot path
constexpr size_t kPipeChunkBytes = (size_t)32 << 20;  // level-0 bytes per chunk

int ensure_pipeline(orbx_ctx* c, size_t up_bytes, size_t res_bytes) {
    if (!c->pipe_made) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->st_up, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->st_down, hipStreamNonBlocking));
        for (int i = 0; i < kPipeSlots; ++i) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_down[i], hipEventDisableTiming));
        }
        c->pipe_made = true;
    }
    if (up_bytes > c->h_up_bytes) {
        for (int i = 0; i < kPipeSlots; ++i) {
            if (c->h_up[i]) HIP_TRY(c, hipHostFree(c->h_up[i]));
            c->h_up[i] = nullptr;
            HIP_TRY(c, hipHostMalloc(&c->h_up[i], up_bytes, hipHostMallocDefault));
        }
        c->h_up_bytes = up_bytes;
    }
    if (res_bytes > c->h_res_bytes) {
        for (int i = 0; i < kPipeSlots; ++i) {
            if (c->h_res[i]) HIP_TRY(c, hipHostFree(c->h_res[i]));
            c->h_res[i] = nullptr;
            HIP_TRY(c, hipHostMalloc(&c->h_res[i], res_bytes, hipHostMallocDefault));
        }
        c->h_res_bytes = res_bytes;
    }
    return ORBX_OK;
}

// The runtime sets up its copy engines lazily: the first asynchronous copies in each direction that find an engine
// busy each block the calling thread for several milliseconds inside hipMemcpyAsync while another engine's queue
// is created (seen as one 5-6 ms stall in each of the first three or four batches of a process).  Copies of the
// pipeline's own shape -- both directions at once, several in flight per stream -- issued once when the pipeline is
// built move that set-up out of the first batches.
int warm_pipeline(orbx_ctx* c, size_t up_bytes, size_t res_bytes) {
    up_bytes = std::min(up_bytes, c->d_pyr[0].bytes);
    res_bytes = std::min(res_bytes, c->d_out_kps.bytes);
    for (int round = 0; round < 4; ++round)
        for (int i = 0; i < kPipeSlots; ++i) {
            HIP_TRY(c, hipMemcpyAsync(c->d_pyr[0].p, c->h_up[i], up_bytes, hipMemcpyHostToDevice, c->st_up));
            HIP_TRY(c, hipMemcpyAsync(c->h_res[i], c->d_out_kps.p, res_bytes / 2, hipMemcpyDeviceToHost, c->st_down));
            HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->h_res[i] + res_bytes / 2, c->d_out_kps.p, res_bytes / 2, hipMemcpyDeviceToHost, c->st_down));
            HIP_TRY(c, hipMemcpyAsync(c->h_res[i], c->d_out_n.p, 16, hipMemcpyDeviceToHost, c->st_down));
        }
    HIP_TRY(c, hipStreamSynchronize(c->st_up));
    HIP_TRY(c, hipStreamSynchronize(c->st_down));
    return ORBX_OK;
}

// A counter one thread advances and others wait on; fail() releases every waiter.
struct Gate {
    std::mutex m;
    std::condition_variable cv;
    int value = -1;
    bool failed = false;
    void set(int v) {
        { std::lock_guard<std::mutex> l(m); value = v; }
        cv.notify_all();
    }
    void fail() {
        { std::lock_guard<std::mutex> l(m); failed = true; }
        cv.notify_all();
    }
    bool wait_for(int v) {  // false: some stage of the pipeline failed
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return value >= v || failed; });
        return !failed;
    }
};

// Host batch in chunks, three host roles side by side: a staging thread re-pitches chunk k+1 into a
// pinned upload slot, the calling thread enqueues upload, extraction and download of chunk k, and a
// delivery thread hands the results of chunk k-1 to the caller -- so neither the link nor the GPU
// waits for a host copy.  kPipeSlots upload slots and result slots; the device arenas hold the whole
// batch (pyramids of every frame stay readable afterwards).
int extract_batch_pipelined(orbx_ctx* c, int n, const uint8_t* const* imgs, int width, int height, const int* strides,
                            orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out, int cap) {
    const size_t plane = c->plane[0];
    const int pitch = c->pitch[0];
    int m = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, kPipeChunkBytes / plane));
    if (c->dbg_pipe_chunk > 0) m = std::max(1, std::min(n, c->dbg_pipe_chunk));  // frames per chunk (tests)
    const size_t res_kps = (size_t)m * cap * sizeof(orbx_kp), res_desc = (size_t)m * cap * 32, res_cnt = (size_t)m * 4;
    const bool fresh = !c->pipe_made || (size_t)m * plane > c->h_up_bytes || res_kps + res_desc + res_cnt > c->h_res_bytes;
    int rc = ensure_pipeline(c, (size_t)m * plane, res_kps + res_desc + res_cnt);
    if (rc) return rc;
    if ((rc = sync_ctx(c))) return rc;  // nothing of an earlier call is still in flight
#ifndef ORBX_DIAG_NO_PIPE_WARM  // diagnostic build: the first batches then show the runtime's stalls (kept trace under profiles/)
    if (fresh && (rc = warm_pipeline(c, (size_t)m * plane, res_kps + res_desc + res_cnt))) return rc;
#else
    (void)fresh;
#endif
    const int aset = c->arena_set ^ 1;
    const int hw = (int)std::max(2u, std::min(12u, std::thread::hardware_concurrency()));
    const int stage_threads = std::max(1, hw * 2 / 3), deliver_threads = std::max(1, hw - stage_threads);
    // chunk boundaries.  Short chunks at both ends (less time to fill and drain the pipeline) were measured and
    // lost: the fixed cost of enqueueing a chunk outweighs them.
    std::vector<int> first;
    for (int f = 0; f < n; f += m) first.push_back(f);
    first.push_back(n);
    const int nchunks = (int)first.size() - 1;
    const int device = c->prm.device_id;
    orbx_kp* d_kps = (orbx_kp*)c->d_out_kps.p;
    uint8_t* d_desc = (uint8_t*)c->d_out_desc.p;
    int32_t* d_cnt = (int32_t*)c->d_out_n.p;

    Gate staged, uploaded, enqueued, delivered;  // staged: chunk k lies in its slot; uploaded: its copy is enqueued; ...
    std::atomic<int> worker_rc{ORBX_OK}, worker_hip{0};
    auto fail_all = [&](int code, int hip) {
        int expect = ORBX_OK;
        if (worker_rc.compare_exchange_strong(expect, code)) worker_hip = hip;
        staged.fail(); uploaded.fail(); enqueued.fail(); delivered.fail();
    };
    const bool ptrace = c->dbg_trace;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = tnow();

    std::thread stager([&] {
        if (hipSetDevice(device) != hipSuccess) return fail_all(ORBX_E_HIP, (int)hipGetLastError());
        for (int k = 0; k < nchunks; ++k) {
            const int slot = k % kPipeSlots, f0 = first[k], mk = first[k + 1] - f0;
            if (k >= kPipeSlots) {  // the slot's previous upload has left the pinned buffer
                if (!uploaded.wait_for(k - kPipeSlots)) return;
                const hipError_t e = hipEventSynchronize(c->ev_up[slot]);
                if (e != hipSuccess) return fail_all(ORBX_E_HIP, (int)e);
            }
            uint8_t* up = (uint8_t*)c->h_up[slot];
            parallel_for(mk, stage_threads, [&](int i) {
                uint8_t* d = up + (size_t)i * plane;
                const uint8_t* s = imgs[f0 + i];
                const int st = strides[f0 + i];
                for (int y = 0; y < height; ++y) std::memcpy(d + (size_t)y * pitch, s + (size_t)y * st, (size_t)width);
            });
            if (ptrace) std::fprintf(stderr, "chunk %d staged %.3f\n", k, tnow() - tp0);
            staged.set(k);
        }
    });
    std::thread deliverer([&] {
        if (hipSetDevice(device) != hipSuccess) return fail_all(ORBX_E_HIP, (int)hipGetLastError());
        for (int k = 0; k < nchunks; ++k) {  // results of chunk k: pinned slot -> the caller's arrays
            const int slot = k % kPipeSlots, f0 = first[k], mk = first[k + 1] - f0;
            if (!enqueued.wait_for(k)) return;
            const hipError_t e = hipEventSynchronize(c->ev_down[slot]);
            if (e != hipSuccess) return fail_all(ORBX_E_HIP, (int)e);
            const uint8_t* base = (const uint8_t*)c->h_res[slot];
            const int32_t* cnt = (const int32_t*)(base + res_kps + res_desc);
            parallel_for(mk, deliver_threads, [&](int i) {
                const int kk = std::min(cnt[i], kp_cap);
                n_out[f0 + i] = kk;
                if (kk > 0) {
                    std::memcpy(kps + (size_t)(f0 + i) * kp_cap, base + (size_t)i * cap * sizeof(orbx_kp), (size_t)kk * sizeof(orbx_kp));
                    std::memcpy(desc + (size_t)(f0 + i) * kp_cap * 32, base + res_kps + (size_t)i * cap * 32, (size_t)kk * 32);
                }
            });
            if (ptrace) std::fprintf(stderr, "chunk %d delivered %.3f\n", k, tnow() - tp0);
            delivered.set(k);
        }
    });

    auto enqueue_all = [&]() -> int {
        for (int k = 0; k < nchunks; ++k) {
            const int slot = k % kPipeSlots, f0 = first[k], mk = first[k + 1] - f0;
            if (!staged.wait_for(k)) return ORBX_OK;  // the failing thread left its code in worker_rc
            const double te0 = ptrace ? tnow() : 0.0;
            HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->d_pyr[aset].p + (size_t)f0 * plane, c->h_up[slot], (size_t)mk * plane,
                                      hipMemcpyHostToDevice, c->st_up));
            HIP_TRY(c, hipEventRecord(c->ev_up[slot], c->st_up));
            uploaded.set(k);
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up[slot], 0));
            const double te1 = ptrace ? tnow() : 0.0;
            // every chunk works in the same arena set (different frames of it) and alternates the retained-point set
            int rc_ = run_batch(c, mk, nullptr, 0, 0, d_kps + (size_t)f0 * cap, cap, d_desc + (size_t)f0 * cap * 32, d_cnt + f0, aset,
                                k & 1, n, f0, true);
            if (rc_) return rc_;
            HIP_TRY(c, hipEventRecord(c->ev_done[slot], c->desc_pending[k & 1] ? c->st_desc : c->stream));  // after the descriptors
            const double te2 = ptrace ? tnow() : 0.0;
            if (k >= kPipeSlots && !delivered.wait_for(k - kPipeSlots)) return ORBX_OK;  // the result slot has been handed over
            const double te3 = ptrace ? tnow() : 0.0;
            HIP_TRY(c, hipStreamWaitEvent(c->st_down, c->ev_done[slot], 0));
            uint8_t* res = (uint8_t*)c->h_res[slot];
            HIP_TRY(c, hipMemcpyAsync(res, d_kps + (size_t)f0 * cap, (size_t)mk * cap * sizeof(orbx_kp), hipMemcpyDeviceToHost,
                                      c->st_down));
            HIP_TRY(c, hipMemcpyAsync(res + res_kps, d_desc + (size_t)f0 * cap * 32, (size_t)mk * cap * 32, hipMemcpyDeviceToHost,
                                      c->st_down));
            HIP_TRY(c, hipMemcpyAsync(res + res_kps + res_desc, d_cnt + f0, (size_t)mk * 4, hipMemcpyDeviceToHost, c->st_down));
            HIP_TRY(c, hipEventRecord(c->ev_down[slot], c->st_down));
            if (ptrace)
                std::fprintf(stderr, "chunk %d enqueued %.3f (upload calls %.3f, launches %.3f, wait for the result slot %.3f, download calls %.3f)\n", k,
                             tnow() - tp0, te1 - te0, te2 - te1, te3 - te2, tnow() - te3);
            enqueued.set(k);
        }
        return ORBX_OK;
    };
    rc = enqueue_all();
    if (rc) fail_all(rc, c->last_hip);
    stager.join();
    deliverer.join();
    if (worker_rc.load() != ORBX_OK) {
        c->last_hip = worker_hip.load();
        sync_ctx(c);  // leave nothing in flight that still names the pinned slots
        return worker_rc.load();
    }
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    fill_batch(c, n, nullptr, 0, 0, n, 0, aset);  // read-backs and the stereo stage see the whole batch
    c->last_kps = d_kps;
    c->last_desc = d_desc;
    c->last_n = d_cnt;
    c->last_chunked = nchunks > 1;
    return ORBX_OK;
}

// One frame -- or one stereo pair -- from host memory as ONE graph launch: upload, the whole launch sequence of
// run_batch on its streams, for a pair the stereo association behind it, and the download of the results are captured
// once per image size and replayed per call, so that a small image costs one submission instead of thirteen launches
// and a dozen event operations.  Everything the graph touches is fixed for the life of the graph: pinned input and
// output buffers of its own, arena set 0, retained-point set 0, the context's result buffers and stereo scratch.
struct GraphWant {
    int w, h, cap, n;      // n = 1, or 2: frames 0 and 1 are the left and the right image of a pair
    bool stereo;           // n == 2: Frame::ComputeStereoMatches on the pair, inside the graph
    float mb, mbf;
    int pad;               // n == 1: border of the padded pyramid to leave in h_pad (0: none)
};
size_t graph_counts_off(const GraphWant& g) { return (size_t)g.n * g.cap * (sizeof(orbx_kp) + 32); }
size_t graph_stereo_off(const GraphWant& g) { return graph_counts_off(g) + 16; }

int build_graph(orbx_ctx* c, const GraphWant& g) {
    drop_graph(c);
    int rc = sync_ctx(c);  // no event of an earlier batch may be waited for inside the capture
    if (rc) return rc;
    if ((rc = ensure_aux_streams(c))) return rc;
    const size_t plane = c->plane[0];
    const size_t out_bytes = graph_stereo_off(g) + (g.stereo ? (size_t)g.cap * 8 : 0);
    // everything that may allocate happens before the capture starts
    StereoScratch scr = {};
    if (g.stereo) {
        if ((rc = ensure(c, &c->d_st_u, (size_t)g.cap * 4))) return rc;
        if ((rc = ensure(c, &c->d_st_depth, (size_t)g.cap * 4))) return rc;
        if ((rc = stereo_scratch_ctx(c, 1, g.cap, c->levels[0].h, &scr))) return rc;
    }
    if (g.pad && (rc = ensure_pad(c))) return rc;
    HIP_TRY(c, hipHostMalloc(&c->h_gin, plane * g.n, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&c->h_gout, out_bytes, hipHostMallocDefault));
    std::memset(c->h_gin, 0, plane * g.n);
    for (bool& p : c->arena_pending) p = false;
    hipGraph_t graph = nullptr;
    HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    orbx_kp* d_kps = (orbx_kp*)c->d_out_kps.p;
    uint8_t* d_desc = (uint8_t*)c->d_out_desc.p;
    int32_t* d_cnt = (int32_t*)c->d_out_n.p;
    auto body = [&]() -> int {
        HIP_TRY(c, hipMemcpyAsync(c->d_pyr[0].p, c->h_gin, plane * g.n, hipMemcpyHostToDevice, c->stream));
        int rc_ = run_batch(c, g.n, nullptr, 0, 0, d_kps, g.cap, d_desc, d_cnt, 0, 0, -1, 0, true);
        if (rc_) return rc_;
        if ((rc_ = join(c))) return rc_;  // the descriptor stream rejoins the capturing stream
        if (g.stereo &&
            (rc_ = run_stereo(c, c, 1, 0, 1, 0, d_kps, d_kps + g.cap, d_desc, d_desc + (size_t)g.cap * 32, d_cnt, d_cnt + 1, 0, 0,
                              g.cap, g.mb, g.mbf, (float*)c->d_st_u.p, (float*)c->d_st_depth.p, scr)))
            return rc_;
        if (g.pad && (rc_ = enqueue_pad(c))) return rc_;
        uint8_t* o = (uint8_t*)c->h_gout;
        const size_t nk = (size_t)g.n * g.cap;
        HIP_TRY(c, hipMemcpyAsync(o, d_kps, nk * sizeof(orbx_kp), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(o + nk * sizeof(orbx_kp), d_desc, nk * 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(o + graph_counts_off(g), d_cnt, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream));
        if (g.stereo) {
            HIP_TRY(c, hipMemcpyAsync(o + graph_stereo_off(g), c->d_st_u.p, (size_t)g.cap * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(o + graph_stereo_off(g) + (size_t)g.cap * 4, c->d_st_depth.p, (size_t)g.cap * 4,
                                      hipMemcpyDeviceToHost, c->stream));
        }
        return ORBX_OK;
    };
    rc = body();
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    c->capturing = false;
    c->desc_pending[0] = c->desc_pending[1] = false;
    c->oct_pending[0] = c->oct_pending[1] = false;
    for (bool& p : c->arena_pending) p = false;
    c->have_last = false;  // nothing has run yet
    if (rc == ORBX_OK && e_end != hipSuccess) {
        c->last_hip = (int)e_end;
        rc = ORBX_E_HIP;
    }
    if (rc == ORBX_OK) {
        const hipError_t e_inst = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
        if (e_inst != hipSuccess) {
            c->last_hip = (int)e_inst;
            c->graph_exec = nullptr;
            rc = ORBX_E_HIP;
        }
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (rc) {
        drop_graph(c);
        return rc;
    }
    c->graph_w = g.w;
    c->graph_h = g.h;
    c->graph_cap = g.cap;
    c->graph_n = g.n;
    c->graph_stereo = g.stereo;
    c->graph_mb = g.mb;
    c->graph_mbf = g.mbf;
    c->graph_pad = g.pad;
    return ORBX_OK;
}

// imgs / strides: g.n host images.  kps[g.n * kp_cap], desc[g.n * kp_cap * 32], n_out[g.n]; u_right / depth [kp_cap] when g.stereo.
int extract_graph(orbx_ctx* c, const GraphWant& g, const uint8_t* const* imgs, const int* strides, orbx_kp* kps, int kp_cap,
                  uint8_t* desc, int* n_out, float* u_right, float* depth) {
    int rc;
    if (!c->graph_exec || c->graph_w != g.w || c->graph_h != g.h || c->graph_cap != g.cap || c->graph_n != g.n ||
        c->graph_stereo != g.stereo || c->graph_pad != g.pad || (g.stereo && (c->graph_mb != g.mb || c->graph_mbf != g.mbf))) {
        if ((rc = build_graph(c, g))) return rc;
    } else if ((rc = sync_ctx(c))) {  // an earlier asynchronous batch may still use arena set 0
        return rc;
    }
    const int pitch = c->pitch[0];
    const size_t plane = c->plane[0];
    for (int i = 0; i < g.n; ++i)
        for (int y = 0; y < g.h; ++y)
            std::memcpy((uint8_t*)c->h_gin + i * plane + (size_t)y * pitch, imgs[i] + (size_t)y * strides[i], (size_t)g.w);
    c->pad_valid = false;
    HIP_TRY(c, hipGraphLaunch(c->graph_exec, c->stream));
    // the host-side view of the batch the graph computes (read-backs and the stereo stage use it)
    fill_batch(c, g.n, nullptr, 0, 0, g.n, 0, 0);
    c->arena_set = 0;
    c->kp_set = 0;
    c->last_kps = (orbx_kp*)c->d_out_kps.p;
    c->last_desc = (uint8_t*)c->d_out_desc.p;
    c->last_n = (int32_t*)c->d_out_n.p;
    c->last_kp_cap = g.cap;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_last = true;
    c->last_chunked = false;
    { int rc_ = check_status(c); if (rc_) return rc_; }
    c->pad_valid = g.pad != 0;
    const uint8_t* o = (const uint8_t*)c->h_gout;
    for (int i = 0; i < g.n; ++i) {
        int32_t k = 0;
        std::memcpy(&k, o + graph_counts_off(g) + 4 * i, 4);
        k = std::max(0, std::min(k, std::min(g.cap, kp_cap)));
        n_out[i] = k;
        if (k > 0) {
            std::memcpy(kps + (size_t)i * kp_cap, o + (size_t)i * g.cap * sizeof(orbx_kp), (size_t)k * sizeof(orbx_kp));
            std::memcpy(desc + (size_t)i * kp_cap * 32, o + (size_t)g.n * g.cap * sizeof(orbx_kp) + (size_t)i * g.cap * 32, (size_t)k * 32);
        }
    }
    if (g.stereo && n_out[0] > 0) {
        std::memcpy(u_right, o + graph_stereo_off(g), (size_t)n_out[0] * 4);
        std::memcpy(depth, o + graph_stereo_off(g) + (size_t)g.cap * 4, (size_t)n_out[0] * 4);
    }
    return ORBX_OK;
}

int check_call(orbx_ctx* c, int n, int w, int h, int kp_cap) {
    if (!c || n < 1 || w < 1 || h < 1) return ORBX_E_INVALID;
    if (n > c->prm.max_batch || n > 65535) return ORBX_E_CAPACITY;
    if (w > c->prm.max_width || h > c->prm.max_height) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    int rc = set_geometry(c, w, h);
    if (rc) return rc;
    if (kp_cap < max_keypoints(c->levels)) return ORBX_E_CAPACITY;
    return ORBX_OK;
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------

extern "C" int orbx_create(const orbx_params* params, orbx_ctx** out) {
    if (!params || !out) return ORBX_E_INVALID;
    *out = nullptr;
    if (params->max_width < 1 || params->max_height < 1 || params->max_batch < 1) return ORBX_E_INVALID;
    if (params->ini_th_fast < 0 || params->ini_th_fast > 255 || params->min_th_fast < 0 ||
        params->min_th_fast > params->ini_th_fast)
        return ORBX_E_INVALID;
    Tables tab;
    if (!build_tables(params->nfeatures, params->scale_factor, params->nlevels, &tab)) return ORBX_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return ORBX_E_NO_DEVICE;
    if (params->device_id < 0 || params->device_id >= ndev) return ORBX_E_INVALID;
    orbx_ctx* c = new (std::nothrow) orbx_ctx();
    if (!c) return ORBX_E_INVALID;
    c->prm = *params;
    c->tab = tab;
    // diagnostic switches: the environment is read here, once, and never on the per-batch path
    c->dbg_no_overlap = std::getenv("ORBX_NO_OVERLAP") != nullptr;
    c->dbg_resize_tiles = std::getenv("ORBX_RESIZE_TILES") != nullptr;
    c->dbg_stereo_border = std::getenv("ORBX_STEREO_BORDER_PATH") != nullptr;
    c->dbg_trace = std::getenv("ORBX_TRACE") != nullptr;
    c->dbg_no_graph = std::getenv("ORBX_NO_GRAPH") != nullptr;
    c->dbg_ordered = std::getenv("ORBX_ORDERED") != nullptr;
#ifdef ORBX_DIAG_LOW_PRIORITY
    c->dbg_aux_low_priority = std::getenv("ORBX_AUX_LOW_PRIORITY") != nullptr;
#endif
    if (const char* e = std::getenv("ORBX_PIPE_CHUNK")) c->dbg_pipe_chunk = std::max(0, std::atoi(e));
    int rc = ORBX_OK;
    auto fail = [&](int code) {
        orbx_destroy(c);
        return code;
    };
    if (hipSetDevice(params->device_id) != hipSuccess) return fail(ORBX_E_HIP);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(ORBX_E_HIP);
    for (int i = 0; i < kEvCount; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return fail(ORBX_E_HIP);
    c->ev_made = true;
    // every context sets the same limit, so a later context cannot lower what an earlier one relies on
    for (const void* fn : {reinterpret_cast<const void*>(k_fast_tiles<0>), reinterpret_cast<const void*>(k_fast_tiles<kFastFixedPitch>),
                           reinterpret_cast<const void*>(k_octree<256, false>), reinterpret_cast<const void*>(k_octree<1024, false>),
                           reinterpret_cast<const void*>(k_bow_vectors)})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds) != hipSuccess)
            return fail(ORBX_E_HIP);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_init), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)init_match_lds_bytes(kInitMaxCap, kInitMaxCap, kInitListCap)) != hipSuccess)
        return fail(ORBX_E_HIP);
    if ((rc = scratch_address(params->device_id, &c->scratch, &c->last_hip))) return fail(rc);
    if (hipHostMalloc((void**)&c->h_status, kStatusWords * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return fail(ORBX_E_HIP);
    std::memset(c->h_status, 0, kStatusWords * sizeof(uint32_t));
    if (hipHostGetDevicePointer((void**)&c->d_status, c->h_status, 0) != hipSuccess) return fail(ORBX_E_HIP);
    {
        std::vector<uint32_t> tab((size_t)kDescTabVec * 64 * 4);
        build_describe_lane_table(kPattern, tab.data());
        if ((rc = ensure(c, &c->d_pat, tab.size() * 4))) return fail(rc);
        if (hipMemcpy(c->d_pat.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(ORBX_E_HIP);
    }
    // size everything for the largest image and batch now, so steady-state calls do not allocate
    if ((rc = set_geometry(c, params->max_width, params->max_height))) return fail(rc);
    if ((rc = ensure_arenas(c, params->max_batch, true))) return fail(rc);
    const int cap = max_keypoints(c->levels);
    if ((rc = ensure(c, &c->d_out_kps, (size_t)params->max_batch * cap * sizeof(orbx_kp)))) return fail(rc);
    if ((rc = ensure(c, &c->d_out_desc, (size_t)params->max_batch * cap * 32))) return fail(rc);
    if ((rc = ensure(c, &c->d_out_n, (size_t)params->max_batch * 4))) return fail(rc);
    *out = c;
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->prm.device_id);
    if (c->stream) (void)sync_ctx(c);
    drop_graph(c);
#if defined(ORBX_STAMPS) || defined(ORBX_STAMPS_DESC)
    if (c->d_stamps.p) {
        std::vector<unsigned long long> h(1024 * 16);
        if (hipMemcpy(h.data(), c->d_stamps.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            for (int i = 0; i < 16; ++i) {
                unsigned long long sum = 0;
                for (int k = 0; k < 1024; ++k) sum += h[(size_t)k * 16 + i];
                std::fprintf(stderr, "stamp[%d] = %llu\n", i, sum);
            }
    }
#endif
    DevBuf* bufs[] = {&c->d_cells, &c->d_tiles, &c->d_pat, &c->d_pyr[0], &c->d_pyr[1], &c->d_blur[0], &c->d_blur[1],
                      &c->d_cand[0], &c->d_cand[1], &c->d_ccount[0], &c->d_ccount[1], &c->d_dense_pts, &c->d_dense_node, &c->d_oct_ws, &c->d_lvl_kps[0], &c->d_lvl_kps[1],
                      &c->d_lvl_count[0], &c->d_lvl_count[1], &c->d_out_kps,
                      &c->d_out_desc, &c->d_out_n, &c->d_stamps, &c->d_proj_minq, &c->d_proj_state, &c->d_proj_list, &c->d_proj_cache, &c->d_proj_ncache, &c->d_local_pick, &c->d_st_rows, &c->d_st_meta, &c->d_st_wmax,
                      &c->d_st_sad, &c->d_st_u, &c->d_st_depth};
    for (DevBuf* b : bufs) release(b);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_pad) (void)hipHostFree(c->h_pad);
    release(&c->d_pad);
    for (int i = 0; i < kPipeSlots; ++i) {
        if (c->h_up[i]) (void)hipHostFree(c->h_up[i]);
        if (c->h_res[i]) (void)hipHostFree(c->h_res[i]);
    }
    if (c->aux_made) {
        (void)hipEventDestroy(c->ev_start);
        (void)hipEventDestroy(c->ev_pyr_done);
        (void)hipEventDestroy(c->ev_blur_done);
        (void)hipEventDestroy(c->ev_fast_done);
        (void)hipEventDestroy(c->ev_oct_done);
        (void)hipEventDestroy(c->ev_desc[0]);
        (void)hipEventDestroy(c->ev_desc[1]);
        for (int i = 0; i < kArenaSets; ++i) (void)hipEventDestroy(c->ev_arena[i]);
        (void)hipEventDestroy(c->ev_oct_set[0]);
        (void)hipEventDestroy(c->ev_oct_set[1]);
        if (c->st_oct != c->st_desc) (void)hipStreamDestroy(c->st_oct);
        (void)hipStreamDestroy(c->st_desc);
        (void)hipStreamDestroy(c->st_aux);
    }
    if (c->pipe_made) {
        for (int i = 0; i < kPipeSlots; ++i) {
            (void)hipEventDestroy(c->ev_up[i]);
            (void)hipEventDestroy(c->ev_done[i]);
            (void)hipEventDestroy(c->ev_down[i]);
        }
        (void)hipStreamDestroy(c->st_up);
        (void)hipStreamDestroy(c->st_down);
    }
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) {
        release(&c->d_xt[l]);
        release(&c->d_yt[l]);
    }
    if (c->ev_made)
        for (int i = 0; i < kEvCount; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbx_extract_batch_device(orbx_ctx* c, int n, const uint8_t* d_imgs, int width, int height, int stride,
                                         size_t frame_stride, orbx_kp* d_kps, int kp_cap, uint8_t* d_desc,
                                         int32_t* d_n_out, int sync) {
    if (!d_imgs || !d_kps || !d_desc || !d_n_out || stride < width) return ORBX_E_INVALID;
    int rc = check_call(c, n, width, height, kp_cap);
    if (rc) return rc;
    const bool in_place = ((uintptr_t)d_imgs % 4 == 0) && (stride % 4 == 0) && (frame_stride % 4 == 0);
    // (Three arena sets in turn, so that a batch's pyramid and blur need not wait for the descriptors of the batch
    // before the previous one: 1.865 against 1.863 ms per 512 KITTI frames, 9.10 against 9.13 ms per 128 4K frames.)
    if ((rc = ensure_arenas(c, n, !in_place))) return rc;
    const int aset = c->arena_set ^ 1, kset = c->kp_set ^ 1;
    if (!in_place) {
        if ((rc = wait_arena_free(c, aset, c->stream))) return rc;  // level 0 of this set may still be read by descriptors
        c->main_touched = true;
        for (int i = 0; i < n; ++i)
            HIP_TRY(c, hipMemcpy2DAsync((uint8_t*)c->d_pyr[aset].p + (size_t)i * c->plane[0], c->pitch[0],
                                        d_imgs + (size_t)i * frame_stride, stride, width, height,
                                        hipMemcpyDeviceToDevice, c->stream));
        rc = run_batch(c, n, nullptr, 0, 0, d_kps, kp_cap, d_desc, d_n_out, aset, kset);
    } else {
        rc = run_batch(c, n, d_imgs, stride, frame_stride, d_kps, kp_cap, d_desc, d_n_out, aset, kset, -1, 0, false, true);
    }
    if (rc) return rc;
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_ctx* c, int n, const uint8_t* const* imgs, int width, int height,
                                  const int* strides, orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out) {
    if (!imgs || !strides || !kps || !desc || !n_out) return ORBX_E_INVALID;
    int rc = check_call(c, n, width, height, kp_cap);
    if (rc) return rc;
    for (int i = 0; i < n; ++i)
        if (!imgs[i] || strides[i] < width) return ORBX_E_INVALID;
    const bool trace = c->dbg_trace;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    if ((rc = ensure_arenas(c, n, true))) return rc;
    const int cap = max_keypoints(c->levels);
    if ((rc = ensure(c, &c->d_out_kps, (size_t)n * cap * sizeof(orbx_kp)))) return rc;
    if ((rc = ensure(c, &c->d_out_desc, (size_t)n * cap * 32))) return rc;
    if ((rc = ensure(c, &c->d_out_n, (size_t)n * 4))) return rc;
    if (n == 1 && !c->dbg_no_graph && !c->dbg_no_overlap && !c->profiling) {
        const GraphWant g = {width, height, cap, 1, false, 0.f, 0.f, c->pad_border};
        return extract_graph(c, g, imgs, strides, kps, kp_cap, desc, n_out, nullptr, nullptr);
    }
    c->pad_valid = false;
    if (n == 1 && c->pad_border && (rc = ensure_pad(c))) return rc;
    if (n >= kPipeMinFrames) {
        rc = extract_batch_pipelined(c, n, imgs, width, height, strides, kps, kp_cap, desc, n_out, cap);
        if (trace) std::fprintf(stderr, "orbx_extract_batch (pipelined) host ms: %.3f\n", now() - t_a);
        return rc;
    }
    const int aset = c->arena_set ^ 1, kset = c->kp_set ^ 1;
    if ((rc = upload_level0(c, aset, n, imgs, strides, width, height))) return rc;
    const double t_b = now();
    rc = run_batch(c, n, nullptr, 0, 0, (orbx_kp*)c->d_out_kps.p, cap, (uint8_t*)c->d_out_desc.p,
                   (int32_t*)c->d_out_n.p, aset, kset);
    if (rc) return rc;
    if ((rc = join(c))) return rc;
    if (n == 1 && c->pad_border) {  // the padded pyramid of a single image (plain launches: the graph has its own copy of this)
        if ((rc = enqueue_pad(c))) return rc;
    }
    const double t_c = now();
    std::vector<int32_t> cnt(n);
    HIP_TRY(c, hipMemcpyAsync(cnt.data(), c->d_out_n.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    const double t_d = now();
    for (int i = 0; i < n; ++i) {
        const int k = cnt[i];
        n_out[i] = k;
        if (k > 0) {
            HIP_TRY(c, hipMemcpyAsync(kps + (size_t)i * kp_cap, (orbx_kp*)c->d_out_kps.p + (size_t)i * cap,
                                      (size_t)k * sizeof(orbx_kp), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(desc + (size_t)i * kp_cap * 32, (uint8_t*)c->d_out_desc.p + (size_t)i * cap * 32,
                                      (size_t)k * 32, hipMemcpyDeviceToHost, c->stream));
        }
    }
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    c->pad_valid = n == 1 && c->pad_border != 0;  // copied before the synchronisation above returned
    if (trace)
        std::fprintf(stderr, "orbx_extract_batch host ms: upload-enqueue %.3f launch %.3f wait+counts %.3f download %.3f\n",
                     t_b - t_a, t_c - t_b, t_d - t_c, now() - t_d);
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_ctx* c, const uint8_t* img, int width, int height, int stride, orbx_kp* kps,
                            int kp_cap, uint8_t* desc, int* n_out) {
    const uint8_t* imgs[1] = {img};
    int strides[1] = {stride};
    return orbx_extract_batch(c, 1, imgs, width, height, strides, kps, kp_cap, desc, n_out);
}

extern "C" int orbx_extract_pair(orbx_ctx* c, const uint8_t* left, int stride_left, const uint8_t* right, int stride_right,
                                 int width, int height, orbx_kp* kps, int kp_cap, uint8_t* desc, int n_out[2], float mb,
                                 float mbf, float* u_right, float* depth) {
    if (!left || !right || !kps || !desc || !n_out || stride_left < width || stride_right < width) return ORBX_E_INVALID;
    const bool stereo = u_right != nullptr || depth != nullptr;
    if (stereo && (!u_right || !depth || !(mb > 0.0f) || !(mbf > 0.0f))) return ORBX_E_INVALID;
    int rc = check_call(c, 2, width, height, kp_cap);
    if (rc) return rc;
    if ((rc = ensure_arenas(c, 2, true))) return rc;
    const int cap = max_keypoints(c->levels);
    if (cap > 65535) return ORBX_E_CAPACITY;  // the stereo stage's limit
    if ((rc = ensure(c, &c->d_out_kps, (size_t)2 * cap * sizeof(orbx_kp)))) return rc;
    if ((rc = ensure(c, &c->d_out_desc, (size_t)2 * cap * 32))) return rc;
    if ((rc = ensure(c, &c->d_out_n, (size_t)2 * 4))) return rc;
    const uint8_t* imgs[2] = {left, right};
    const int strides[2] = {stride_left, stride_right};
    if (!c->dbg_no_graph && !c->dbg_no_overlap && !c->profiling) {
        const GraphWant g = {width, height, cap, 2, stereo, mb, mbf, 0};
        return extract_graph(c, g, imgs, strides, kps, kp_cap, desc, n_out, u_right, depth);
    }
    // plain launches (diagnostic switches): the batch entry point, then the stereo stage of the one pair
    if ((rc = orbx_extract_batch(c, 2, imgs, width, height, strides, kps, kp_cap, desc, n_out))) return rc;
    if (stereo) {
        int nl = 0;
        std::vector<float> u(cap), d(cap);
        if ((rc = orbx_stereo_match(c, 0, c, 1, mb, mbf, u.data(), d.data(), cap, &nl))) return rc;
        const int k = std::min(nl, n_out[0]);
        std::memcpy(u_right, u.data(), (size_t)k * 4);
        std::memcpy(depth, d.data(), (size_t)k * 4);
    }
    return ORBX_OK;
}

extern "C" int orbx_set_keep_pyramid(orbx_ctx* c, int border) {
    if (!c || border < 0 || border > 64) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    if (border != c->pad_border) drop_graph(c);  // the captured launch sequence includes the padded pyramid or not
    c->pad_border = border;
    c->pad_valid = false;
    return ORBX_OK;
}

extern "C" int orbx_padded_level(orbx_ctx* c, int level, const uint8_t** base, int* width, int* height, int* stride) {
    if (!c || !base) return ORBX_E_INVALID;
    if (!c->pad_valid || !c->have_last) return ORBX_E_STATE;
    if (level < 0 || level >= c->B.nlevels) return ORBX_E_INVALID;
    *base = (const uint8_t*)c->h_pad + c->pad_off[level];
    if (width) *width = c->B.lv[level].w;
    if (height) *height = c->B.lv[level].h;
    if (stride) *stride = c->B.lv[level].w + 2 * c->pad_border;
    return ORBX_OK;
}

extern "C" int orbx_stereo_match_batch_device(orbx_ctx* c, int n_pairs, const orbx_kp* d_kps, int kp_cap,
                                              const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                                              float* d_u_right, float* d_depth, int sync) {
    if (!c || !d_kps || !d_desc || !d_n || !d_u_right || !d_depth || n_pairs < 1 || kp_cap < 1) return ORBX_E_INVALID;
    if (!(mb > 0.0f) || !(mbf > 0.0f)) return ORBX_E_INVALID;
    if (!c->have_last) return ORBX_E_STATE;
    if (2 * n_pairs > c->B.nframes || kp_cap > 65535) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    int rc = join(c);  // the descriptors of the batch may still be running on the auxiliary stream
    if (rc) return rc;
    StereoScratch scr;
    if ((rc = stereo_scratch_ctx(c, n_pairs, kp_cap, c->B.lv[0].h, &scr))) return rc;
    // pair p = images (2p, 2p + 1) of the last batch; their results are 2 * kp_cap keypoints apart
    rc = run_stereo(c, c, n_pairs, 0, 1, 2, d_kps, d_kps + kp_cap, d_desc, d_desc + (size_t)kp_cap * 32, d_n, d_n + 1,
                        2 * (int64_t)kp_cap, 2, kp_cap, mb, mbf, d_u_right, d_depth, scr);
    if (rc) return rc;
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_stereo_match_batch_device_noalloc(orbx_ctx* c, int n_pairs, const orbx_kp* d_kps, int kp_cap,
                                                      const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                                                      float* d_u_right, float* d_depth, int sync) {
    if (!c || !d_kps || !d_desc || !d_n || !d_u_right || !d_depth || n_pairs < 1 || kp_cap < 1) return ORBX_E_INVALID;
    if (!(mb > 0.0f) || !(mbf > 0.0f)) return ORBX_E_INVALID;
    if (!c->have_last) return ORBX_E_STATE;
    if (2 * n_pairs > c->B.nframes || kp_cap > 16384) return ORBX_E_CAPACITY;
    const int rows = c->B.lv[0].h;
    if (rows > 16383) return ORBX_E_CAPACITY;
    // the per-pair scratch of the three kernels, carved out of the fixed array; sizes beyond it are refused
    const size_t rows_b = align_up((size_t)n_pairs * (rows + 1) * 4, 256);
    const size_t meta_b = align_up((size_t)n_pairs * kp_cap * sizeof(StereoMeta), 256);
    const size_t wmax_b = align_up((size_t)n_pairs * 4, 256);
    const size_t sad_b = (size_t)n_pairs * kp_cap * 4;
    if (rows_b + meta_b + wmax_b + sad_b > kScratchBytes) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    int rc = join(c);  // stream waits only: nothing is allocated on this path
    if (rc) return rc;
    uint8_t* base = c->scratch;  // the address hipGetSymbolAddress gave for this device, not the array's name
    StereoScratch scr;
    scr.row_start = (uint32_t*)base;
    scr.meta = (StereoMeta*)(base + rows_b);
    scr.wmax = (int32_t*)(base + rows_b + meta_b);
    scr.sad = (int32_t*)(base + rows_b + meta_b + wmax_b);
    {
        // one user of the array at a time on this GPU: behind the previous user's kernels, whichever context ran them
        const int dev = c->prm.device_id;
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (g_scratch_used[dev]) HIP_TRY(c, hipStreamWaitEvent(c->stream, g_scratch_event[dev], 0));
        rc = run_stereo(c, c, n_pairs, 0, 1, 2, d_kps, d_kps + kp_cap, d_desc, d_desc + (size_t)kp_cap * 32, d_n, d_n + 1,
                        2 * (int64_t)kp_cap, 2, kp_cap, mb, mbf, d_u_right, d_depth, scr);
        // recorded even after a failed launch: whatever did get enqueued must be waited for
        const hipError_t e_rec = hipEventRecord(g_scratch_event[dev], c->stream);
        if (e_rec == hipSuccess) g_scratch_used[dev] = true;
        if (rc == ORBX_OK && e_rec != hipSuccess) {
            c->last_hip = (int)e_rec;
            rc = ORBX_E_HIP;
        }
    }
    if (rc) return rc;
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_stereo_match(orbx_ctx* left, int img_left, orbx_ctx* right, int img_right, float mb, float mbf,
                                 float* u_right, float* depth, int cap, int* n_left) {
    if (!left || !right || !u_right || !depth || !n_left) return ORBX_E_INVALID;
    if (!(mb > 0.0f) || !(mbf > 0.0f)) return ORBX_E_INVALID;
    if (!left->have_last || !right->have_last || !left->last_kps || !right->last_kps) return ORBX_E_STATE;
    if (img_left < 0 || img_left >= left->B.nframes || img_right < 0 || img_right >= right->B.nframes) return ORBX_E_INVALID;
    if (left->prm.device_id != right->prm.device_id) return ORBX_E_INVALID;
    // both extractions must share the extractor parameters and the image size
    if (left->B.nlevels != right->B.nlevels || left->cur_w != right->cur_w || left->cur_h != right->cur_h ||
        left->prm.scale_factor != right->prm.scale_factor || left->last_kp_cap != right->last_kp_cap)
        return ORBX_E_INVALID;
    HIP_TRY(left, hipSetDevice(left->prm.device_id));
    { int rc_ = sync_ctx(right); if (rc_) return rc_; }  // the right extraction has finished
    // the left one may have been enqueued without waiting (orbx_extract_batch_device, sync = 0): its
    // descriptors run on their own stream, which the main stream must wait for before it reads them
    { int rc_ = join(left); if (rc_) return rc_; }
    const int kp_cap = left->last_kp_cap;
    if (kp_cap > 65535) return ORBX_E_CAPACITY;
    int32_t nl = 0;
    HIP_TRY(left, hipMemcpyAsync(&nl, left->last_n + img_left, 4, hipMemcpyDeviceToHost, left->stream));
    int rc;
    if ((rc = ensure(left, &left->d_st_u, (size_t)kp_cap * 4))) return rc;
    if ((rc = ensure(left, &left->d_st_depth, (size_t)kp_cap * 4))) return rc;
    StereoScratch scr;
    if ((rc = stereo_scratch_ctx(left, 1, kp_cap, left->B.lv[0].h, &scr))) return rc;
    rc = run_stereo(left, right, 1, img_left, img_right, 0, left->last_kps + (size_t)img_left * kp_cap,
                    right->last_kps + (size_t)img_right * kp_cap, left->last_desc + (size_t)img_left * kp_cap * 32,
                    right->last_desc + (size_t)img_right * kp_cap * 32, left->last_n + img_left, right->last_n + img_right, 0,
                    0, kp_cap, mb, mbf, (float*)left->d_st_u.p, (float*)left->d_st_depth.p, scr);
    if (rc) return rc;
    { int rc_ = sync_ctx(left); if (rc_) return rc_; }
    *n_left = nl;
    if (nl > cap) return ORBX_E_CAPACITY;
    if (nl > 0) {
        HIP_TRY(left, hipMemcpy(u_right, left->d_st_u.p, (size_t)nl * 4, hipMemcpyDeviceToHost));
        HIP_TRY(left, hipMemcpy(depth, left->d_st_depth.p, (size_t)nl * 4, hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}

extern "C" int orbx_gray_batch_device(orbx_ctx* c, int n, const uint8_t* d_src, int width, int height, int src_stride,
                                      size_t src_frame_stride, int channels, int blue_idx, uint8_t* d_dst,
                                      int dst_stride, size_t dst_frame_stride, int sync) {
    if (!c || !d_src || !d_dst || n < 1 || n > 65535 || width < 1 || height < 1 || height > 65535) return ORBX_E_INVALID;
    if ((channels != 3 && channels != 4) || (blue_idx != 0 && blue_idx != 2)) return ORBX_E_INVALID;
    if (src_stride < width * channels || dst_stride < width) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }  // the destination may be the input of a batch still being described
    const dim3 grid(((width + 3) / 4 + 255) / 256, height, n);
    if (channels == 3)
        hipLaunchKernelGGL(k_cvt_gray<3>, grid, dim3(256), 0, c->stream, d_src, src_stride, (uint64_t)src_frame_stride,
                           d_dst, dst_stride, (uint64_t)dst_frame_stride, width, blue_idx);
    else
        hipLaunchKernelGGL(k_cvt_gray<4>, grid, dim3(256), 0, c->stream, d_src, src_stride, (uint64_t)src_frame_stride,
                           d_dst, dst_stride, (uint64_t)dst_frame_stride, width, blue_idx);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_remap_batch_device(orbx_ctx* c, int n, const uint8_t* d_src, int src_width, int src_height,
                                       int src_stride, size_t src_frame_stride, const float* d_mapx, const float* d_mapy,
                                       int map_stride, uint8_t* d_dst, int dst_width, int dst_height, int dst_stride,
                                       size_t dst_frame_stride, int sync) {
    if (!c || !d_src || !d_mapx || !d_mapy || !d_dst || n < 1 || n > 65535) return ORBX_E_INVALID;
    if (src_width < 1 || src_height < 1 || dst_width < 1 || dst_height < 1 || dst_height > 65535) return ORBX_E_INVALID;
    if (src_stride < src_width || dst_stride < dst_width || map_stride < dst_width) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    const dim3 grid(((dst_width + 3) / 4 + 255) / 256, dst_height, n);
    hipLaunchKernelGGL(k_remap_linear, grid, dim3(256), 0, c->stream, d_src, src_width, src_height, src_stride,
                       (uint64_t)src_frame_stride, d_mapx, d_mapy, map_stride, d_dst, dst_width, dst_stride,
                       (uint64_t)dst_frame_stride);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_undistort_grid_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps, int kp_cap, const int32_t* d_n,
                                                const float K[4], const float dist[5], float min_x, float min_y,
                                                float inv_w, float inv_h, orbx_kp* d_kps_un, int32_t* d_cell_start,
                                                int32_t* d_cell_idx, int sync) {
    if (!c || !d_kps || !d_n || !K || !dist || !d_kps_un || !d_cell_start || !d_cell_idx || n < 1 || kp_cap < 1)
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    PostParams P = {K[0], K[1], K[2], K[3], dist[0], dist[1], dist[2], dist[3], dist[4], min_x, min_y, inv_w, inv_h};
    hipLaunchKernelGGL(k_undistort_grid, dim3(n), dim3(256), 0, c->stream, P, d_kps, d_n, kp_cap, d_kps_un, d_cell_start,
                       d_cell_idx);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

// ---- bag-of-words transform ------------------------------------------------------------------
struct orbx_voc {
    VocHost h;
    int device = -1;  // device the tree has been copied to
    void *d_rec = nullptr, *d_word = nullptr, *d_weight = nullptr;
};

extern "C" int orbx_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                               const uint8_t* is_leaf, const uint8_t* desc, const double* weight, orbx_voc** out) {
    if (!parent || !is_leaf || !desc || !weight || !out) return ORBX_E_INVALID;
    orbx_voc* v = new (std::nothrow) orbx_voc;
    if (!v) return ORBX_E_INVALID;
    const int rc = build_vocabulary(k, L, scoring, weighting, n_nodes, parent, is_leaf, desc, weight, &v->h);
    if (rc) {
        delete v;
        return rc;
    }
    *out = v;
    return ORBX_OK;
}

extern "C" int orbx_voc_load_text(const char* path, orbx_voc** out) {
    if (!path || !out) return ORBX_E_INVALID;
    orbx_voc* v = new (std::nothrow) orbx_voc;
    if (!v) return ORBX_E_INVALID;
    const int rc = load_vocabulary_text(path, &v->h);
    if (rc) {
        delete v;
        return rc;
    }
    *out = v;
    return ORBX_OK;
}

extern "C" void orbx_voc_destroy(orbx_voc* v) {
    if (!v) return;
    if (v->device >= 0) {
        (void)hipSetDevice(v->device);
        for (void* p : {v->d_rec, v->d_word, v->d_weight})
            if (p) (void)hipFree(p);
    }
    delete v;
}

extern "C" int orbx_voc_info(const orbx_voc* v, int* k, int* L, int* n_nodes, int* n_words) {
    if (!v) return ORBX_E_INVALID;
    if (k) *k = v->h.k;
    if (L) *L = v->h.L;
    if (n_nodes) *n_nodes = v->h.n_nodes();
    if (n_words) *n_words = v->h.n_words;
    return ORBX_OK;
}

extern "C" int orbx_voc_node(const orbx_voc* v, int node, int* n_children, uint32_t* first_child, uint8_t desc[32],
                             double* weight, uint32_t* word) {
    if (!v || node < 0 || node >= v->h.n_nodes()) return ORBX_E_INVALID;
    const uint32_t s = v->h.child_start[(size_t)node], e = v->h.child_start[(size_t)node + 1];
    if (n_children) *n_children = (int)(e - s);
    if (first_child) *first_child = e > s ? v->h.child[s] : 0;
    if (desc) std::memcpy(desc, v->h.desc.data() + (size_t)node * 32, 32);
    if (weight) *weight = v->h.weight[(size_t)node];
    if (word) *word = v->h.word[(size_t)node];
    return ORBX_OK;
}

extern "C" int orbx_bow_transform_batch_device(orbx_ctx* c, orbx_voc* v, int n, const uint8_t* d_desc, int kp_cap,
                                               const int32_t* d_n, int levelsup, uint32_t* d_word, uint32_t* d_node,
                                               double* d_weight, uint32_t* d_bow_id, double* d_bow_val, int32_t* d_bow_n,
                                               uint32_t* d_fv_node, uint32_t* d_fv_idx, int32_t* d_fv_n, int sync) {
    if (!c || !v || !d_desc || !d_n || !d_word || !d_node || !d_weight || !d_bow_id || !d_bow_val || !d_bow_n ||
        !d_fv_node || !d_fv_idx || !d_fv_n || n < 1 || kp_cap < 1 || levelsup < 0)
        return ORBX_E_INVALID;
    if (kp_cap > 16384) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    if (v->device >= 0 && v->device != c->prm.device_id) return ORBX_E_STATE;  // a tree lives on one device
    if (v->device < 0) {
        auto up = [&](void** dst, const void* src, size_t bytes) -> int {
            HIP_TRY(c, hipMalloc(dst, std::max<size_t>(bytes, 16)));
            HIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
            return ORBX_OK;
        };
        int rc;
        v->device = c->prm.device_id;
        // the children lists as records: centre, node id and the range of the child's own children
        std::vector<VocRec> rec((size_t)std::max(1, v->h.n_nodes() - 1));
        for (int pos = 0; pos + 1 < v->h.n_nodes(); ++pos) {
            const uint32_t id = v->h.child[(size_t)pos];
            VocRec& r = rec[(size_t)pos];
            std::memcpy(&r.d0, v->h.desc.data() + (size_t)id * 32, 32);
            r.id = id;
            r.begin = v->h.child_start[id];
            r.end = v->h.child_start[(size_t)id + 1];
            r.pad = 0;
        }
        if ((rc = up(&v->d_rec, rec.data(), rec.size() * sizeof(VocRec)))) return rc;
        if ((rc = up(&v->d_word, v->h.word.data(), v->h.word.size() * 4))) return rc;
        if ((rc = up(&v->d_weight, v->h.weight.data(), v->h.weight.size() * 8))) return rc;
    }
    { int rc_ = join(c); if (rc_) return rc_; }
    VocDev V;
    V.rec = (const VocRec*)v->d_rec;
    V.root_end = v->h.child_start[1];
    V.word = (const uint32_t*)v->d_word;
    V.weight = (const double*)v->d_weight;
    V.n_nodes = v->h.n_nodes();
    V.L = v->h.L;
    V.sums = (v->h.weighting == 0 || v->h.weighting == 1) ? 1 : 0;  // TF_IDF, TF (:1145) against IDF, BINARY (:1173)
    V.norm = v->h.scoring == 5 ? 0 : (v->h.scoring == 1 ? 2 : 1);   // ScoringObject.h:73-90
    hipLaunchKernelGGL(k_bow_words, dim3((kp_cap + kBowPerBlock - 1) / kBowPerBlock, n), dim3(256), 0, c->stream, V, d_desc,
                       d_n, kp_cap, v->h.L - levelsup, d_word, d_node, d_weight);
    int P = 1024;
    while (P < kp_cap) P <<= 1;
    const int dual = (size_t)P * 16 <= 128 * 1024 ? 1 : 0;
    const size_t lds = (size_t)P * (dual ? 16 : 8);
    hipLaunchKernelGGL(k_bow_vectors, dim3(n), dim3(1024), lds, c->stream, V, d_n, kp_cap, P, dual, (const uint32_t*)d_word,
                       (const uint32_t*)d_node, (const double*)d_weight, d_bow_id, d_bow_val, d_bow_n, d_fv_node, d_fv_idx,
                       d_fv_n);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_search_window_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un, int kp_cap,
                                               const int32_t* d_cell_start, const int32_t* d_cell_idx,
                                               const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_taken,
                                               float min_x, float min_y, float inv_w, float inv_h,
                                               const orbx_window_query* d_queries, int q_cap, const int32_t* d_nq,
                                               int32_t* d_best_idx, int32_t* d_best_dist, int sync) {
    static_assert(sizeof(orbx_window_query) == sizeof(WindowQuery), "query layout");
    if (!c || !d_kps_un || !d_cell_start || !d_cell_idx || !d_desc || !d_queries || !d_nq || !d_best_idx || !d_best_dist ||
        n < 1 || kp_cap < 1 || q_cap < 1)
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    WindowDev W = {d_kps_un, d_cell_start, d_cell_idx, d_desc, d_u_right, d_taken, min_x, min_y, inv_w, inv_h, kp_cap, q_cap};
    hipLaunchKernelGGL(k_search_window, dim3((q_cap + 15) / 16, n), dim3(256), 0, c->stream, W,
                       reinterpret_cast<const WindowQuery*>(d_queries), d_nq, d_best_idx, d_best_dist);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_bow_score_l1_batch_device(orbx_ctx* c, int n_pairs, const uint32_t* d_bow_id_1, const double* d_bow_val_1,
                                              const int32_t* d_bow_n_1, int cap_1, const uint32_t* d_bow_id_2,
                                              const double* d_bow_val_2, const int32_t* d_bow_n_2, int cap_2,
                                              const int32_t* d_pairs, double* d_score, int sync) {
    if (!c || !d_bow_id_1 || !d_bow_val_1 || !d_bow_n_1 || !d_bow_id_2 || !d_bow_val_2 || !d_bow_n_2 || !d_pairs || !d_score ||
        n_pairs < 1 || cap_1 < 1 || cap_2 < 1)
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    BowScoreDev D = {d_bow_id_1, d_bow_id_2, d_bow_val_1, d_bow_val_2, d_bow_n_1, d_bow_n_2, cap_1, cap_2, n_pairs, d_pairs, d_score};
    hipLaunchKernelGGL(k_bow_score_l1, dim3((n_pairs + 3) / 4), dim3(256), 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_fuse_search_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un, int kp_cap,
                                             const int32_t* d_cell_start, const int32_t* d_cell_idx, const uint8_t* d_desc,
                                             const float* d_u_right, float min_x, float min_y, float inv_w, float inv_h,
                                             const orbx_window_query* d_queries, int q_cap, const int32_t* d_nq,
                                             int32_t* d_best_idx, int32_t* d_best_dist, int sync) {
    if (!c || !d_kps_un || !d_cell_start || !d_cell_idx || !d_desc || !d_queries || !d_nq || !d_best_idx || !d_best_dist ||
        n < 1 || kp_cap < 1 || q_cap < 1)
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    WindowDev W = {d_kps_un, d_cell_start, d_cell_idx, d_desc, nullptr, nullptr, min_x, min_y, inv_w, inv_h, kp_cap, q_cap};
    FuseGate G;
    G.u_right = d_u_right;
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) G.inv_sigma2[i] = c->tab.inv_sigma2[std::min(i, c->prm.nlevels - 1)];
    hipLaunchKernelGGL(k_fuse_search, dim3((q_cap + 15) / 16, n), dim3(256), 0, c->stream, W, G,
                       reinterpret_cast<const WindowQuery*>(d_queries), d_nq, d_best_idx, d_best_dist);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un, int kp_cap,
                                                      const int32_t* d_cell_start, const int32_t* d_cell_idx,
                                                      const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_taken,
                                                      const orbx_projection_params* params, const float* d_poses,
                                                      const orbx_map_query* d_queries, int q_cap, const int32_t* d_nq,
                                                      int32_t* d_match, int32_t* d_query_match, int32_t* d_nmatches,
                                                      int sync) {
    static_assert(sizeof(orbx_map_query) == sizeof(MapQuery), "query layout");
    static_assert(sizeof(orbx_projection_params) == sizeof(ProjParams), "parameter layout");
    if (!c || !d_kps_un || !d_cell_start || !d_cell_idx || !d_desc || !params || !d_poses || !d_queries || !d_nq || !d_match ||
        !d_query_match || !d_nmatches || n < 1 || kp_cap < 1 || q_cap < 1)
        return ORBX_E_INVALID;
    if (q_cap >= (1 << 30)) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = ensure(c, &c->d_proj_minq, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_state, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_list, (size_t)n * q_cap * 8); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_cache, (size_t)n * q_cap * kProjCache * 8); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_ncache, (size_t)n * q_cap * 4); if (rc_) return rc_; }
    { int rc_ = join(c); if (rc_) return rc_; }
    ProjDev D;
    D.W = {d_kps_un, d_cell_start, d_cell_idx, d_desc, d_u_right, nullptr, params->min_x, params->min_y, params->inv_w,
           params->inv_h, kp_cap, q_cap};
    std::memcpy(&D.P, params, sizeof(ProjParams));
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) D.scale[i] = c->tab.scale[std::min(i, c->prm.nlevels - 1)];
    D.poses = d_poses;
    D.queries = reinterpret_cast<const MapQuery*>(d_queries);
    D.n_q = d_nq;
    D.taken0 = d_taken;
    D.match = d_match;
    D.query_match = d_query_match;
    D.n_matches = d_nmatches;
    D.minq = (int32_t*)c->d_proj_minq.p;
    D.held = (int32_t*)c->d_proj_state.p;
    D.list = (int32_t*)c->d_proj_list.p;
    D.cache = (uint2*)c->d_proj_cache.p;
    D.n_cache = (int32_t*)c->d_proj_ncache.p;
    const size_t lds = ((size_t)2 * kp_cap + (size_t)2 * q_cap) * 4;
    D.use_lds = lds <= 64 * 1024 ? 1 : 0;  // two workgroups per CU keep their round state in LDS
    hipLaunchKernelGGL(k_search_projection, dim3(n), dim3(kProjThreads), D.use_lds ? lds : 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_search_local_map_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un, int kp_cap,
                                                  const int32_t* d_cell_start, const int32_t* d_cell_idx,
                                                  const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_taken,
                                                  float min_x, float min_y, float inv_w, float inv_h, float th,
                                                  float nn_ratio, const orbx_local_query* d_queries, int q_cap,
                                                  const int32_t* d_nq, int32_t* d_match, int32_t* d_query_match,
                                                  int32_t* d_nmatches, int sync) {
    static_assert(sizeof(orbx_local_query) == sizeof(LocalQuery), "query layout");
    if (!c || !d_kps_un || !d_cell_start || !d_cell_idx || !d_desc || !d_queries || !d_nq || !d_match || !d_query_match ||
        !d_nmatches || n < 1 || kp_cap < 1 || q_cap < 1 || !(th > 0.0f) || !(nn_ratio > 0.0f))
        return ORBX_E_INVALID;
    if (q_cap >= (1 << 30) || kp_cap >= (1 << 20)) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = ensure(c, &c->d_proj_minq, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_state, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_list, (size_t)n * q_cap * 8); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_local_pick, (size_t)n * q_cap * 8); if (rc_) return rc_; }
    { int rc_ = join(c); if (rc_) return rc_; }
    LocalDev D;
    D.W = {d_kps_un, d_cell_start, d_cell_idx, d_desc, d_u_right, nullptr, min_x, min_y, inv_w, inv_h, kp_cap, q_cap};
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) D.scale[i] = c->tab.scale[std::min(i, c->prm.nlevels - 1)];
    D.th = th;
    D.nn_ratio = nn_ratio;
    D.queries = reinterpret_cast<const LocalQuery*>(d_queries);
    D.n_q = d_nq;
    D.taken0 = d_taken;
    D.match = d_match;
    D.query_match = d_query_match;
    D.n_matches = d_nmatches;
    D.minq = (int32_t*)c->d_proj_minq.p;
    D.held = (int32_t*)c->d_proj_state.p;
    D.list = (int32_t*)c->d_proj_list.p;
    D.pick = (int2*)c->d_local_pick.p;
    const size_t lds = ((size_t)2 * kp_cap + (size_t)2 * q_cap) * 4;
    D.use_lds = lds <= 64 * 1024 ? 1 : 0;
    D.max_dist = D.level_up = D.check_orientation = 0;
    hipLaunchKernelGGL(k_search_local<false>, dim3(n), dim3(kProjThreads), D.use_lds ? lds : 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_search_guided_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un, int kp_cap,
                                               const int32_t* d_cell_start, const int32_t* d_cell_idx, const uint8_t* d_desc,
                                               const uint8_t* d_taken, float min_x, float min_y, float inv_w, float inv_h,
                                               const orbx_guided_query* d_queries, int q_cap, const int32_t* d_nq, int level_up,
                                               int max_dist, int check_orientation, int32_t* d_match, int32_t* d_query_match,
                                               int32_t* d_nmatches, int sync) {
    static_assert(sizeof(orbx_guided_query) == sizeof(LocalQuery), "query layout");
    if (!c || !d_kps_un || !d_cell_start || !d_cell_idx || !d_desc || !d_queries || !d_nq || !d_match || !d_query_match ||
        !d_nmatches || n < 1 || kp_cap < 1 || q_cap < 1 || level_up < 0 || level_up > 1 || max_dist < 0 || max_dist > 256)
        return ORBX_E_INVALID;
    if (q_cap >= (1 << 30) || kp_cap >= (1 << 20)) return ORBX_E_CAPACITY;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = ensure(c, &c->d_proj_minq, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_state, (size_t)n * kp_cap * 4); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_proj_list, (size_t)n * q_cap * 8); if (rc_) return rc_; }
    { int rc_ = ensure(c, &c->d_local_pick, (size_t)n * q_cap * 8); if (rc_) return rc_; }
    { int rc_ = join(c); if (rc_) return rc_; }
    LocalDev D;
    D.W = {d_kps_un, d_cell_start, d_cell_idx, d_desc, nullptr, nullptr, min_x, min_y, inv_w, inv_h, kp_cap, q_cap};
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) D.scale[i] = c->tab.scale[std::min(i, c->prm.nlevels - 1)];
    D.th = 1.0f;
    D.nn_ratio = 1.0f;
    D.queries = reinterpret_cast<const LocalQuery*>(d_queries);
    D.n_q = d_nq;
    D.taken0 = d_taken;
    D.match = d_match;
    D.query_match = d_query_match;
    D.n_matches = d_nmatches;
    D.minq = (int32_t*)c->d_proj_minq.p;
    D.held = (int32_t*)c->d_proj_state.p;
    D.list = (int32_t*)c->d_proj_list.p;
    D.pick = (int2*)c->d_local_pick.p;
    D.max_dist = max_dist;
    D.level_up = level_up;
    D.check_orientation = check_orientation ? 1 : 0;
    const size_t lds = ((size_t)2 * kp_cap + (size_t)2 * q_cap) * 4;
    D.use_lds = lds <= 64 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(k_search_local<true>, dim3(n), dim3(kProjThreads), D.use_lds ? lds : 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

namespace {
// Both SearchByBoW overloads; d_valid_f != null selects the one between two keyframes.
int search_by_bow(orbx_ctx* c, int n, const uint32_t* d_fv_node_kf, const uint32_t* d_fv_idx_kf, const int32_t* d_fv_n_kf,
                  const uint8_t* d_desc_kf, const orbx_kp* d_kps_kf, const uint8_t* d_valid_kf, int kp_cap_kf,
                  const uint32_t* d_fv_node_f, const uint32_t* d_fv_idx_f, const int32_t* d_fv_n_f, const uint8_t* d_desc_f,
                  const orbx_kp* d_kps_f, const uint8_t* d_valid_f, int kp_cap_f, float nn_ratio, int check_orientation,
                  int32_t* d_match, int32_t* d_nmatches, int sync) {
    if (!c || !d_fv_node_kf || !d_fv_idx_kf || !d_fv_n_kf || !d_desc_kf || !d_kps_kf || !d_valid_kf || !d_fv_node_f ||
        !d_fv_idx_f || !d_fv_n_f || !d_desc_f || !d_kps_f || !d_match || !d_nmatches || n < 1 || kp_cap_kf < 1 ||
        kp_cap_f < 1 || !(nn_ratio > 0.0f))
        return ORBX_E_INVALID;
    if (kp_cap_kf > 16384 || kp_cap_f > 16384) return ORBX_E_CAPACITY;  // as the bag-of-words transform; 24 bits of a match entry hold the feature
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    BowMatchDev D;
    D.fv_node_kf = d_fv_node_kf;
    D.fv_idx_kf = d_fv_idx_kf;
    D.fv_n_kf = d_fv_n_kf;
    D.desc_kf = d_desc_kf;
    D.kps_kf = d_kps_kf;
    D.valid_kf = d_valid_kf;
    D.cap_kf = kp_cap_kf;
    D.fv_node_f = d_fv_node_f;
    D.fv_idx_f = d_fv_idx_f;
    D.fv_n_f = d_fv_n_f;
    D.desc_f = d_desc_f;
    D.kps_f = d_kps_f;
    D.cap_f = kp_cap_f;
    D.nn_ratio = nn_ratio;
    D.check_orientation = check_orientation ? 1 : 0;
    D.two_keyframes = d_valid_f ? 1 : 0;
    D.valid_f = d_valid_f;
    D.match = d_match;
    D.n_matches = d_nmatches;
    D.taken_words = ((kp_cap_f + 63) / 64 + 31) / 32;
    const size_t lds = (size_t)(kBowMatchThreads / 64) * D.taken_words * 64 * 4;
    hipLaunchKernelGGL(k_search_bow, dim3(n), dim3(kBowMatchThreads), lds, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}
}  // namespace

extern "C" int orbx_search_by_bow_batch_device(orbx_ctx* c, int n, const uint32_t* d_fv_node_kf, const uint32_t* d_fv_idx_kf,
                                               const int32_t* d_fv_n_kf, const uint8_t* d_desc_kf, const orbx_kp* d_kps_kf,
                                               const uint8_t* d_valid_kf, int kp_cap_kf, const uint32_t* d_fv_node_f,
                                               const uint32_t* d_fv_idx_f, const int32_t* d_fv_n_f, const uint8_t* d_desc_f,
                                               const orbx_kp* d_kps_f, int kp_cap_f, float nn_ratio, int check_orientation,
                                               int32_t* d_match, int32_t* d_nmatches, int sync) {
    return search_by_bow(c, n, d_fv_node_kf, d_fv_idx_kf, d_fv_n_kf, d_desc_kf, d_kps_kf, d_valid_kf, kp_cap_kf, d_fv_node_f,
                         d_fv_idx_f, d_fv_n_f, d_desc_f, d_kps_f, nullptr, kp_cap_f, nn_ratio, check_orientation, d_match,
                         d_nmatches, sync);
}

extern "C" int orbx_search_by_bow_keyframes_batch_device(orbx_ctx* c, int n, const uint32_t* d_fv_node_1,
                                                         const uint32_t* d_fv_idx_1, const int32_t* d_fv_n_1,
                                                         const uint8_t* d_desc_1, const orbx_kp* d_kps_1,
                                                         const uint8_t* d_valid_1, int kp_cap_1, const uint32_t* d_fv_node_2,
                                                         const uint32_t* d_fv_idx_2, const int32_t* d_fv_n_2,
                                                         const uint8_t* d_desc_2, const orbx_kp* d_kps_2,
                                                         const uint8_t* d_valid_2, int kp_cap_2, float nn_ratio,
                                                         int check_orientation, int32_t* d_match12, int32_t* d_nmatches,
                                                         int sync) {
    if (!d_valid_2) return ORBX_E_INVALID;
    return search_by_bow(c, n, d_fv_node_1, d_fv_idx_1, d_fv_n_1, d_desc_1, d_kps_1, d_valid_1, kp_cap_1, d_fv_node_2, d_fv_idx_2,
                         d_fv_n_2, d_desc_2, d_kps_2, d_valid_2, kp_cap_2, nn_ratio, check_orientation, d_match12, d_nmatches,
                         sync);
}

extern "C" int orbx_search_for_triangulation_batch_device(
    orbx_ctx* c, int n, const uint32_t* d_fv_node_1, const uint32_t* d_fv_idx_1, const int32_t* d_fv_n_1, const uint8_t* d_desc_1,
    const orbx_kp* d_kps_un_1, const uint8_t* d_has_mp_1, const float* d_u_right_1, int kp_cap_1, const uint32_t* d_fv_node_2,
    const uint32_t* d_fv_idx_2, const int32_t* d_fv_n_2, const uint8_t* d_desc_2, const orbx_kp* d_kps_un_2,
    const uint8_t* d_has_mp_2, const float* d_u_right_2, int kp_cap_2, const float* d_f12, const float* d_epipole,
    int only_stereo, int check_orientation, int32_t* d_match12, int32_t* d_nmatches, int sync) {
    if (!c || !d_fv_node_1 || !d_fv_idx_1 || !d_fv_n_1 || !d_desc_1 || !d_kps_un_1 || !d_has_mp_1 || !d_fv_node_2 ||
        !d_fv_idx_2 || !d_fv_n_2 || !d_desc_2 || !d_kps_un_2 || !d_has_mp_2 || !d_f12 || !d_epipole || !d_match12 ||
        !d_nmatches || n < 1 || kp_cap_1 < 1 || kp_cap_2 < 1 || ((uintptr_t)d_desc_1 | (uintptr_t)d_desc_2) % 16)
        return ORBX_E_INVALID;
    if (kp_cap_1 > 16384 || kp_cap_2 > 16384) return ORBX_E_CAPACITY;  // as the bag-of-words transform
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    TriDev D;
    D.fv_node_1 = d_fv_node_1;
    D.fv_idx_1 = d_fv_idx_1;
    D.fv_n_1 = d_fv_n_1;
    D.desc_1 = d_desc_1;
    D.kps_1 = d_kps_un_1;
    D.has_mp_1 = d_has_mp_1;
    D.u_right_1 = d_u_right_1;
    D.cap_1 = kp_cap_1;
    D.fv_node_2 = d_fv_node_2;
    D.fv_idx_2 = d_fv_idx_2;
    D.fv_n_2 = d_fv_n_2;
    D.desc_2 = d_desc_2;
    D.kps_2 = d_kps_un_2;
    D.has_mp_2 = d_has_mp_2;
    D.u_right_2 = d_u_right_2;
    D.cap_2 = kp_cap_2;
    D.f12 = d_f12;
    D.epipole = d_epipole;
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) {
        D.scale[i] = c->tab.scale[std::min(i, c->prm.nlevels - 1)];
        D.sigma2[i] = c->tab.sigma2[std::min(i, c->prm.nlevels - 1)];
    }
    D.only_stereo = only_stereo ? 1 : 0;
    D.check_orientation = check_orientation ? 1 : 0;
    D.match12 = d_match12;
    D.n_matches = d_nmatches;
    hipLaunchKernelGGL(k_search_triangulation, dim3(n), dim3(kBowMatchThreads), 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_search_for_initialization_batch_device(orbx_ctx* c, int n, const orbx_kp* d_kps_un_1,
                                                           const uint8_t* d_desc_1, const int32_t* d_n_1, int kp_cap_1,
                                                           float* d_prev_matched, const orbx_kp* d_kps_un_2,
                                                           const int32_t* d_cell_start_2, const int32_t* d_cell_idx_2,
                                                           const uint8_t* d_desc_2, int kp_cap_2, float min_x, float min_y,
                                                           float inv_w, float inv_h, int window_size, float nn_ratio,
                                                           int check_orientation, int32_t* d_match12, int32_t* d_nmatches,
                                                           int sync) {
    if (!c || !d_kps_un_1 || !d_desc_1 || !d_n_1 || !d_prev_matched || !d_kps_un_2 || !d_cell_start_2 || !d_cell_idx_2 ||
        !d_desc_2 || !d_match12 || !d_nmatches || n < 1 || kp_cap_1 < 1 || kp_cap_2 < 1 || window_size < 0 ||
        !(nn_ratio > 0.0f) || ((uintptr_t)d_desc_1 | (uintptr_t)d_desc_2) % 16 || (uintptr_t)d_prev_matched % 8)
        return ORBX_E_INVALID;
    if (kp_cap_1 > kInitMaxCap || kp_cap_2 > kInitMaxCap) return ORBX_E_CAPACITY;  // 16-bit state in LDS
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    InitDev D;
    D.W = {d_kps_un_2, d_cell_start_2, d_cell_idx_2, d_desc_2, nullptr, nullptr, min_x, min_y, inv_w, inv_h, kp_cap_2, 0};
    D.kps_un_1 = d_kps_un_1;
    D.desc_1 = d_desc_1;
    D.n_1 = d_n_1;
    D.prev_matched = reinterpret_cast<float2*>(d_prev_matched);
    D.match12 = d_match12;
    D.n_matches = d_nmatches;
    D.window = (float)window_size;
    D.nn_ratio = nn_ratio;
    D.cap_1 = kp_cap_1;
    D.check_orientation = check_orientation ? 1 : 0;
    D.list_cap = c->dbg_init_list > 0 ? c->dbg_init_list : kInitListCap;
    const size_t lds = init_match_lds_bytes(kp_cap_1, kp_cap_2, D.list_cap);
    hipLaunchKernelGGL(k_search_init, dim3(n), dim3(kInitThreads), lds, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

// Shared body of the map-point projections (mode: kProjFrustum .. kProjReloc).
static int project_points(orbx_ctx* c, int n, const orbx_frustum_params* p, const float* d_poses, const float* d_centres,
                          const orbx_frustum_point* d_points, const float* d_angles, int q_cap, const int32_t* d_nq,
                          void* d_out, int32_t* d_count, int mode, int sync) {
    static_assert(sizeof(orbx_frustum_point) == 68, "point layout");
    if (!c || !p || !d_poses || (!d_centres && mode != kProjSim3) || !d_points || !d_nq || !d_out || !d_count || n < 1 ||
        q_cap < 1)
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    ProjectDev D;
    D.points = d_points;
    D.nq = d_nq;
    D.poses = d_poses;
    D.centres = d_centres;
    D.angles = d_angles;
    D.out = d_out;
    D.n_out = d_count;
    D.q_cap = q_cap;
    D.nlevels = c->prm.nlevels;
    D.fx = p->fx; D.fy = p->fy; D.cx = p->cx; D.cy = p->cy; D.mbf = p->mbf;
    D.min_x = p->min_x; D.max_x = p->max_x; D.min_y = p->min_y; D.max_y = p->max_y;
    D.view_cos_limit = p->view_cos_limit;
    D.log_scale_factor = p->log_scale_factor;
    D.th = p->th;
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) D.scale[i] = c->tab.scale[std::min(i, c->prm.nlevels - 1)];
    HIP_TRY(c, hipMemsetAsync(d_count, 0, (size_t)n * sizeof(int32_t), c->stream));
    const dim3 grid((q_cap + 255) / 256, n);
    switch (mode) {
        case kProjFrustum: hipLaunchKernelGGL(k_project_points<kProjFrustum>, grid, dim3(256), 0, c->stream, D); break;
        case kProjFuse: hipLaunchKernelGGL(k_project_points<kProjFuse>, grid, dim3(256), 0, c->stream, D); break;
        case kProjLoop: hipLaunchKernelGGL(k_project_points<kProjLoop>, grid, dim3(256), 0, c->stream, D); break;
        case kProjSim3: hipLaunchKernelGGL(k_project_points<kProjSim3>, grid, dim3(256), 0, c->stream, D); break;
        default: hipLaunchKernelGGL(k_project_points<kProjReloc>, grid, dim3(256), 0, c->stream, D); break;
    }
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_is_in_frustum_batch_device(orbx_ctx* c, int n, const orbx_frustum_params* params, const float* d_poses,
                                               const float* d_centres, const orbx_frustum_point* d_points, int q_cap,
                                               const int32_t* d_nq, orbx_local_query* d_queries_out, int32_t* d_n_in_view,
                                               int sync) {
    return project_points(c, n, params, d_poses, d_centres, d_points, nullptr, q_cap, d_nq, d_queries_out, d_n_in_view,
                          kProjFrustum, sync);
}

extern "C" int orbx_fuse_project_batch_device(orbx_ctx* c, int n, const orbx_frustum_params* params, const float* d_poses,
                                              const float* d_centres, const orbx_frustum_point* d_points, int q_cap,
                                              const int32_t* d_nq, orbx_window_query* d_queries_out, int32_t* d_n_projected,
                                              int sync) {
    return project_points(c, n, params, d_poses, d_centres, d_points, nullptr, q_cap, d_nq, d_queries_out, d_n_projected,
                          kProjFuse, sync);
}

extern "C" int orbx_sim3_project_batch_device(orbx_ctx* c, int n, const orbx_frustum_params* params, const float* d_poses,
                                              const orbx_frustum_point* d_points, int q_cap, const int32_t* d_nq,
                                              orbx_window_query* d_queries_out, int32_t* d_n_projected, int sync) {
    return project_points(c, n, params, d_poses, nullptr, d_points, nullptr, q_cap, d_nq, d_queries_out, d_n_projected, kProjSim3,
                          sync);
}

extern "C" int orbx_project_guided_batch_device(orbx_ctx* c, int n, const orbx_frustum_params* params, const float* d_poses,
                                                const float* d_centres, const orbx_frustum_point* d_points,
                                                const float* d_angles, int q_cap, const int32_t* d_nq, int relocalisation,
                                                orbx_guided_query* d_queries_out, int32_t* d_n_projected, int sync) {
    return project_points(c, n, params, d_poses, d_centres, d_points, d_angles, q_cap, d_nq, d_queries_out, d_n_projected,
                          relocalisation ? kProjReloc : kProjLoop, sync);
}

extern "C" int orbx_rgbd_depth_batch_device(orbx_ctx* c, int n, const void* d_depth, int depth_is_f32, int width, int height,
                                            int stride_bytes, size_t frame_stride_bytes, float depth_map_factor,
                                            const orbx_kp* d_kps, const orbx_kp* d_kps_un, int kp_cap, const int32_t* d_n,
                                            float mbf, float* d_u_right, float* d_depth_out, int sync) {
    const int sample = depth_is_f32 ? 4 : 2;
    if (!c || !d_depth || !d_kps || !d_kps_un || !d_n || !d_u_right || !d_depth_out || n < 1 || kp_cap < 1 || width < 1 ||
        height < 1 || stride_bytes < width * sample || stride_bytes % sample || frame_stride_bytes % (size_t)sample ||
        ((uintptr_t)d_depth % (uintptr_t)sample) || (n > 1 && frame_stride_bytes < (size_t)height * stride_bytes))
        return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = join(c); if (rc_) return rc_; }
    RgbdDev D;
    D.depth = d_depth;
    D.frame_stride = frame_stride_bytes;
    D.is_f32 = depth_is_f32 ? 1 : 0;
    D.width = width;
    D.height = height;
    D.stride = stride_bytes;
    D.factor = depth_map_factor;
    D.mbf = mbf;
    D.kps = d_kps;
    D.kps_un = d_kps_un;
    D.n_kp = d_n;
    D.kp_cap = kp_cap;
    D.u_right = d_u_right;
    D.depth_out = d_depth_out;
    hipLaunchKernelGGL(k_rgbd_depth, dim3((kp_cap + 255) / 256, n), dim3(256), 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
    if (sync) { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_synchronize(orbx_ctx* c) {
    if (!c) return ORBX_E_INVALID;
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    return ORBX_OK;
}

extern "C" int orbx_pyramid_level(orbx_ctx* c, int img_idx, int level, uint8_t* dst, int dst_stride, int* width,
                                  int* height) {
    if (!c || !dst) return ORBX_E_INVALID;
    if (!c->have_last) return ORBX_E_STATE;
    if (img_idx < 0 || img_idx >= c->B.nframes || level < 0 || level >= c->B.nlevels) return ORBX_E_INVALID;
    const LevelDev& D = c->B.lv[level];
    if (dst_stride < D.w) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    // one contiguous copy into pinned memory, rows re-pitched on the host
    const size_t bytes = (size_t)(D.h - 1) * D.pitch + D.w;
    int rc = ensure_stage(c, bytes);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_stage, D.img + (size_t)img_idx * D.plane, bytes, hipMemcpyDeviceToHost, c->stream));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    for (int y = 0; y < D.h; ++y)
        std::memcpy(dst + (size_t)y * dst_stride, (const uint8_t*)c->h_stage + (size_t)y * D.pitch, (size_t)D.w);
    if (width) *width = D.w;
    if (height) *height = D.h;
    return ORBX_OK;
}

extern "C" int orbx_debug_read(orbx_ctx* c, int what, int img_idx, int level, void* dst, size_t cap_bytes,
                               size_t* n_bytes) {
    if (!c || !dst || !n_bytes) return ORBX_E_INVALID;
    if (!c->have_last) return ORBX_E_STATE;
    if (img_idx < 0 || img_idx >= c->B.nframes || level < 0 || level >= c->B.nlevels) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }
    const LevelDev& D = c->B.lv[level];
    const BatchDev& B = c->B;
    if (what != ORBX_DBG_BLURRED && c->last_chunked) return ORBX_E_STATE;  // the lists were reused chunk by chunk
    if (what == ORBX_DBG_BLURRED) {
        *n_bytes = (size_t)D.w * D.h;
        if (cap_bytes < *n_bytes) return ORBX_E_CAPACITY;
        HIP_TRY(c, hipMemcpy2D(dst, D.w, D.blur + (size_t)img_idx * D.bplane, D.bpitch, D.w, D.h, hipMemcpyDeviceToHost));
        return ORBX_OK;
    }
    if (what == ORBX_DBG_CANDIDATES) {
        std::vector<uint32_t> cnt(D.n_cells);
        const size_t slot0 = (size_t)img_idx * B.total_cells + D.cell_base;
        HIP_TRY(c, hipMemcpy(cnt.data(), (uint32_t*)c->d_ccount[c->kp_set].p + slot0, (size_t)D.n_cells * 4, hipMemcpyDeviceToHost));
        size_t total = 0;
        for (uint32_t v : cnt) total += v;
        *n_bytes = total * 12;
        if (cap_bytes < *n_bytes) return ORBX_E_CAPACITY;
        std::vector<uint32_t> all((size_t)D.n_cells * B.cell_cap);
        HIP_TRY(c, hipMemcpy(all.data(), (uint32_t*)c->d_cand[c->kp_set].p + slot0 * B.cell_cap, all.size() * 4, hipMemcpyDeviceToHost));
        int32_t* o = (int32_t*)dst;
        for (int k = 0; k < D.n_cells; ++k)
            for (uint32_t j = 0; j < cnt[k]; ++j) {
                const uint32_t v = all[(size_t)k * B.cell_cap + j];
                *o++ = pt_x(v);
                *o++ = pt_y(v);
                *o++ = pt_score(v);
            }
        return ORBX_OK;
    }
    if (what == ORBX_DBG_LEVEL_KPS) {
        int32_t n = 0;
        HIP_TRY(c, hipMemcpy(&n, (int32_t*)c->d_lvl_count[c->kp_set].p + (size_t)img_idx * B.nlevels + level, 4, hipMemcpyDeviceToHost));
        *n_bytes = (size_t)n * 12;
        if (cap_bytes < *n_bytes) return ORBX_E_CAPACITY;
        std::vector<uint32_t> v(n);
        if (n)
            HIP_TRY(c, hipMemcpy(v.data(), (uint32_t*)c->d_lvl_kps[c->kp_set].p + (size_t)img_idx * B.kp_total + D.kp_off,
                                 (size_t)n * 4, hipMemcpyDeviceToHost));
        int32_t* o = (int32_t*)dst;
        for (int i = 0; i < n; ++i) {
            *o++ = pt_x(v[i]) + kBorder;
            *o++ = pt_y(v[i]) + kBorder;
            *o++ = pt_score(v[i]);
        }
        return ORBX_OK;
    }
    return ORBX_E_INVALID;
}

extern "C" uint32_t orbx_host_blur_rim_selector(int target_col, int hi_col, int lo_col, int width) {
    return blur_rim_selector(target_col, hi_col, lo_col, width);
}

extern "C" int orbx_set_debug_flag(orbx_ctx* c, int flag, int value) {
    if (!c) return ORBX_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->prm.device_id));
    { int rc_ = sync_ctx(c); if (rc_) return rc_; }  // a switch never changes under work in flight
    switch (flag) {
        case ORBX_FLAG_NO_OVERLAP: c->dbg_no_overlap = value != 0; break;
        case ORBX_FLAG_RESIZE_TILES: c->dbg_resize_tiles = value != 0; break;
        case ORBX_FLAG_STEREO_BORDER_PATH: c->dbg_stereo_border = value != 0; break;
        case ORBX_FLAG_TRACE: c->dbg_trace = value != 0; break;
        case ORBX_FLAG_NO_GRAPH: c->dbg_no_graph = value != 0; break;
        case ORBX_FLAG_PIPE_CHUNK: c->dbg_pipe_chunk = std::max(0, value); break;
        case ORBX_FLAG_INIT_LIST: c->dbg_init_list = std::max(0, std::min(value, kInitListCap)); break;
        case ORBX_FLAG_ORDERED: c->dbg_ordered = value != 0; break;
        case ORBX_FLAG_POISON: c->dbg_poison = (value >= 0 && value <= 3) ? value : 0; break;
        default: return ORBX_E_INVALID;
    }
    drop_graph(c);  // captured with the old switches
    return ORBX_OK;
}

extern "C" int orbx_set_profiling(orbx_ctx* c, int enable) {
    if (!c) return ORBX_E_INVALID;
    c->profiling = enable != 0;
    c->ev_valid = false;
    return ORBX_OK;
}

extern "C" int orbx_get_stage_ms(orbx_ctx* c, float ms[ORBX_NUM_STAGES]) {
    if (!c || !ms) return ORBX_E_INVALID;
    if (!c->ev_valid) return ORBX_E_STATE;
    // each stage's own launches, start to end on the stream they ran on; with the two-stream
    // schedule stages overlap, so the values can add up to more than the batch's wall time
    HIP_TRY(c, hipEventSynchronize(c->ev[kEvDesc1]));
    auto span = [&](int a, int b, float* out) -> int {
        HIP_TRY(c, hipEventElapsedTime(out, c->ev[a], c->ev[b]));
        return ORBX_OK;
    };
    float fa = 0.f, fb = 0.f;
    int rc;
    if ((rc = span(kEvPyr0, kEvPyr1, &ms[0]))) return rc;
    if ((rc = span(kEvFastA0, kEvFastA1, &fa))) return rc;
    if ((rc = span(kEvFastB0, kEvFastB1, &fb))) return rc;
    ms[1] = fa + fb;
    if ((rc = span(kEvOct0, kEvOct1, &ms[2]))) return rc;
    if ((rc = span(kEvBlur0, kEvBlur1, &ms[3]))) return rc;
    if ((rc = span(kEvDesc0, kEvDesc1, &ms[4]))) return rc;
    return ORBX_OK;
}

extern "C" const char* orbx_stage_name(int stage) {
    static const char* names[ORBX_NUM_STAGES] = {"pyramid", "fast", "quadtree", "blur", "orient_describe"};
    return (stage >= 0 && stage < ORBX_NUM_STAGES) ? names[stage] : "?";
}

extern "C" const char* orbx_strerror(int code) {
    switch (code) {
        case ORBX_OK: return "ok";
        case ORBX_E_INVALID: return "invalid argument";
        case ORBX_E_HIP: return "HIP runtime error";
        case ORBX_E_NO_DEVICE: return "no usable GPU";
        case ORBX_E_CAPACITY: return "capacity exceeded";
        case ORBX_E_GEOMETRY: return "image geometry not supported";
        case ORBX_E_STATE: return "no extraction has run";
        case ORBX_E_DEVICE_DATA: return "a count or coordinate read on the device was out of bounds (results of that batch are invalid)";
        default: return "unknown error";
    }
}

extern "C" int orbx_last_hip_error(const orbx_ctx* c) { return c ? c->last_hip : 0; }
extern "C" const char* orbx_version(void) { return "orbx 0.1 (gfx950)"; }
