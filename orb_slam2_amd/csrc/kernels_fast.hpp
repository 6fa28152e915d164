// Per-cell FAST-9/16 detection with cell-local non-max suppression and the per-cell threshold
// fallback (ComputeKeyPointsOctTree, reference src/ORBextractor.cc:880-920, calling cv::FAST).
#ifndef ORBX_KERNELS_FAST_HPP
#define ORBX_KERNELS_FAST_HPP

#include "orbx_device.h"

namespace orbx {

// Block-wide exclusive scan of one int per thread (256 threads = 4 waves).
// Returns the exclusive prefix; *total receives the block sum.  Contains two barriers.
__device__ __forceinline__ int block_excl_scan_256(int v, int* s_wave4, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    __syncthreads();  // protects s_wave4 against a previous use
    if (lane == 63) s_wave4[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = s_wave4[k];
        if (k < wave) base += t;
    }
    *total = s_wave4[0] + s_wave4[1] + s_wave4[2] + s_wave4[3];
    return base + incl - v;
}

// The FAST segment-test score of a pixel, independent of any threshold:
//   S = max over the 16 cyclic arcs A of 9 ring pixels of max(min_{k in A} d_k, min_{k in A} -d_k),
//   d_k = centre - ring_k.
// A pixel is a corner at threshold t  <=>  S > t, and cv::FAST's cornerScore is S - 1 for every
// corner (OpenCV 3.4.5 cornerScore<16> returns max(t, S) - 1).  So the score of a corner does not
// depend on the threshold that found it.
// arc_max returns max over arcs of min over the arc of x_k: the "ring darker" half of S for
// x = centre - ring and the "ring brighter" half for x = ring - centre.  Window minima by
// doubling (2, 4, 8, then 9) with two-operand min/max only: on gfx950 v_min_i32/v_max_i32 issue
// faster than the three-operand forms.
__device__ __forceinline__ int arc_max(const int x[16]) {
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = min(x[k], x[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m8[k] = min(m4[k], m4[(k + 4) & 15]);
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) best = max(best, min(m8[k], x[(k + 8) & 15]));
    return best;
}

typedef unsigned short fast_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}

constexpr int kFastMaxCell = 66;  // largest sub-image side: wCell <= 59, + 6, + 1

// LDS carve-up of k_fast_tiles for one tile geometry (bytes).
struct FastLds {
    int tile, plane, queue;
};
// Tile layout: LDS column of detection column u is u + 4, so u = 0 sits on a dword boundary and
// the 3-pixel halo on its left occupies columns 1..3.  Four spare columns on the right keep the
// reads of the (discarded) pixels between wd and the next multiple of 4 inside the row.
__host__ __device__ inline int fast_tile_pitch(int wd) { return (wd + 14 + 3) & ~3; }
__host__ __device__ inline int fast_plane_pitch(int wd, int nc) { return (wd + nc + 1 + 3) & ~3; }
__host__ __device__ inline FastLds fast_lds_need(int wd, int hd, int nc) {
    FastLds n;
    n.tile = fast_tile_pitch(wd) * (hd + 6) + 16;
    n.plane = fast_plane_pitch(wd, nc) * (hd + 2);
    n.queue = (((wd + 3) & ~3) * hd) * 2;
    return n;
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// Steps 1-3 of one pass over a tile (see k_fast_tiles).  RETRY = false: threshold iniTh, every
// cell; RETRY = true: threshold minTh, only the cells whose s_th entry is >= 0.
template <bool RETRY>
__device__ __forceinline__ void fast_pass(const uint8_t* s_tile, uint8_t* s_score, uint8_t* s_keep, uint16_t* s_queue,
                                          int* s_qn, int* s_kept, const int* s_th, int TP, int SP, int wd, int hd,
                                          int nc, int w_cell, float inv_wcell, int thr) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) *s_qn = 0;
    __syncthreads();
    // ---- 1. rejection test, four pixels per work item, two pixels per instruction ----
    // Any arc of 9 holds one pixel of each opposite pair, so a corner needs the two compass pairs
    // (up/down, left/right) to each hold a pixel beyond the threshold on the same side.  The
    // dword's bytes are split into even and odd pixels as two 16-bit lanes; with the centre biased
    // by 256 a plain 32-bit subtract yields both lanes' differences, packed u16 min/max combine
    // the pairs, and adding a constant moves "beyond the threshold" into bit 15 of each lane.
    const int ndwd = (wd + 3) >> 2, n_items = hd * ndwd;
    const uint32_t kM = 0x00FF00FFu, kBias = 0x01000100u;
    const uint32_t add_bright = (uint32_t)(0x8000 - 257 - thr) * 0x00010001u;
    const uint32_t sub_dark = (uint32_t)(0x8000 + 255 - thr) * 0x00010001u;
    // (row, dword) of item tid, advanced by 256 items per iteration without divisions
    int v = (int)(((float)tid + 0.5f) * (1.0f / (float)ndwd));
    int j = tid - v * ndwd;
    const int dv = 256 / ndwd, dj = 256 - dv * ndwd;
    for (int i0 = 0; i0 < n_items; i0 += 256) {
        uint32_t pe = 0, po = 0;  // bit 15 / 31 of pe: pixels 0 / 2 pass; of po: pixels 1 / 3
        if (v < hd) {
            const uint8_t* pc = s_tile + (v + 3) * TP + 4 * j + 4;
            const uint32_t Cd = *reinterpret_cast<const uint32_t*>(pc);
            const uint32_t Ud = *reinterpret_cast<const uint32_t*>(pc + 3 * TP);
            const uint32_t Dd = *reinterpret_cast<const uint32_t*>(pc - 3 * TP);
            const uint32_t Rd = *reinterpret_cast<const u32_unaligned*>(pc + 3);
            const uint32_t Ld = *reinterpret_cast<const u32_unaligned*>(pc - 3);
            uint32_t ps[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {  // 0: pixels 0,2   1: pixels 1,3
                const int sh = 8 * h2;
                const uint32_t c = ((Cd >> sh) & kM) | kBias;
                const uint32_t d0 = c - ((Ud >> sh) & kM), d8 = c - ((Dd >> sh) & kM);
                const uint32_t d4 = c - ((Rd >> sh) & kM), d12 = c - ((Ld >> sh) & kM);
                const uint32_t br = pk_min_u16(pk_max_u16(d0, d8), pk_max_u16(d4, d12));
                const uint32_t dk = pk_max_u16(pk_min_u16(d0, d8), pk_min_u16(d4, d12));
                ps[h2] = ((br + add_bright) | (sub_dark - dk)) & 0x80008000u;
            }
            pe = ps[0];
            po = ps[1];
            if (RETRY) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int u = min(4 * j + b, wd - 1);
                    const bool on = s_th[min((int)(((float)u + 0.5f) * inv_wcell), nc - 1)] >= 0;
                    if (!on) {
                        if (b & 1) po &= ~(0x8000u << (8 * (b & 2)));
                        else pe &= ~(0x8000u << (8 * (b & 2)));
                    }
                }
            }
        }
        // wave-level append; entry = v << 8 | u
        const uint64_t m0 = __ballot((pe & 0x8000u) != 0), m1 = __ballot((po & 0x8000u) != 0),
                       m2 = __ballot((int)pe < 0), m3 = __ballot((int)po < 0);
        const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
        const int tot = n0 + n1 + n2 + n3;
        if (tot) {
            int base = 0;
            if (lane == 0) base = atomicAdd(s_qn, tot);
            base = __builtin_amdgcn_readfirstlane(base);
            const uint64_t below = (1ull << lane) - 1ull;
            const int e0 = (v << 8) + 4 * j;
            if (pe & 0x8000u) s_queue[base + __popcll(m0 & below)] = (uint16_t)(e0 + 0);
            if (po & 0x8000u) s_queue[base + n0 + __popcll(m1 & below)] = (uint16_t)(e0 + 1);
            if ((int)pe < 0) s_queue[base + n0 + n1 + __popcll(m2 & below)] = (uint16_t)(e0 + 2);
            if ((int)po < 0) s_queue[base + n0 + n1 + n2 + __popcll(m3 & below)] = (uint16_t)(e0 + 3);
        }
        v += dv;
        j += dj;
        if (j >= ndwd) {
            j -= ndwd;
            ++v;
        }
    }
    __syncthreads();
    const int qn = *s_qn;
    // ---- 2. full score for the queued pixels ----
    for (int q = tid; q < qn; q += 256) {
        const int e = s_queue[q], u = e & 0xFF, vq = e >> 8;
        if (u >= wd) continue;  // padding pixels of the last dword of a row
        const uint8_t* p = s_tile + (vq + 3) * TP + 4 + u;
        const int c0 = p[0];
        const int r0 = p[3 * TP], r8 = p[-3 * TP], r4 = p[3], r12 = p[-3];
        // side of the test this pixel passed: ring darker than the centre (bright corner) or
        // brighter (dark corner, evaluated on inverted pixels); both is possible but rare
        const bool bright = min(max(c0 - r0, c0 - r8), max(c0 - r4, c0 - r12)) > thr;
        const bool dark = max(min(c0 - r0, c0 - r8), min(c0 - r4, c0 - r12)) < -thr;
        int S = -256;
        for (int side = 0; side < 2; ++side) {
            if (side == 0 ? !bright : !dark) continue;
            const int flip = side ? 0xFF : 0;
            const int cv = c0 ^ flip;
            int d[16];
            d[0] = cv - (r0 ^ flip);
            d[1] = cv - (p[3 * TP + 1] ^ flip);
            d[2] = cv - (p[2 * TP + 2] ^ flip);
            d[3] = cv - (p[1 * TP + 3] ^ flip);
            d[4] = cv - (r4 ^ flip);
            d[5] = cv - (p[-1 * TP + 3] ^ flip);
            d[6] = cv - (p[-2 * TP + 2] ^ flip);
            d[7] = cv - (p[-3 * TP + 1] ^ flip);
            d[8] = cv - (r8 ^ flip);
            d[9] = cv - (p[-3 * TP - 1] ^ flip);
            d[10] = cv - (p[-2 * TP - 2] ^ flip);
            d[11] = cv - (p[-1 * TP - 3] ^ flip);
            d[12] = cv - (r12 ^ flip);
            d[13] = cv - (p[1 * TP - 3] ^ flip);
            d[14] = cv - (p[2 * TP - 2] ^ flip);
            d[15] = cv - (p[3 * TP - 1] ^ flip);
            S = max(S, arc_max(d));
        }
        if (S > thr) {
            const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
            if (!RETRY || s_th[c] >= 0) s_score[(vq + 1) * SP + 1 + u + c] = (uint8_t)(S - 1);
        }
    }
    __syncthreads();
    // ---- 3. non-max suppression of the queued pixels ----
    for (int q = tid; q < qn; q += 256) {
        const int e = s_queue[q], u = e & 0xFF, vq = e >> 8;
        if (u >= wd) continue;
        const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
        const uint8_t* sp = s_score + (vq + 1) * SP + 1 + u + c;
        const int sc = sp[0];
        if (sc > 0) {
            const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                              max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
            if (sc > m) {
                s_keep[(vq + 1) * SP + 1 + u + c] = 1;
                atomicAdd(&s_kept[c], 1);
            }
        }
    }
    __syncthreads();
}

// One workgroup per tile (a run of up to kFastTileCells cells of one cell row).
// grid = (tiles, frames), block = 256, dynamic LDS = tile + 2 planes + queue.
//  pass 0, threshold iniTh, all cells of the tile:
//   1. every detection pixel takes the compass-pair rejection test; survivors go to an LDS queue;
//   2. queued pixels get the full segment-test score S; S > t stores score S-1 in a score plane
//      in which the cells are separated by a zero column, so that
//   3. strict 3x3 non-max suppression over the plane is cell-local exactly as when cv::FAST runs
//      on the cell's sub-image; survivors are flagged and counted per cell;
//  pass 1, threshold minTh, only the cells whose pass-0 count is zero (the retry of
//      reference :903-907), same three steps;
//  then each wave compacts the flags of its cells in row-major order into the cell's slot.
__global__ __launch_bounds__(256) void k_fast_tiles(BatchDev B, const TileDesc* __restrict__ tiles,
                                                    uint32_t* __restrict__ cand, uint32_t* __restrict__ ccount,
                                                    int tile_bytes, int plane_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    __shared__ int s_qn;
    __shared__ int s_kept[kFastTileCells];
    __shared__ int s_th[kFastTileCells];
    __shared__ int s_retry;

    uint8_t* s_tile = s_mem;
    uint8_t* s_score = s_mem + tile_bytes;
    uint8_t* s_keep = s_score + plane_bytes;
    uint16_t* s_queue = reinterpret_cast<uint16_t*>(s_keep + plane_bytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileDesc td = tiles[blockIdx.x];
    const LevelDev& L = B.lv[td.level];
    const uint8_t* img = L.img + (uint64_t)blockIdx.y * L.plane;
    const int wd = td.wd, hd = td.hd, nc = td.nc, w_cell = td.w_cell;
    const int TP = fast_tile_pitch(wd), ndw = TP >> 2;
    const int SP = fast_plane_pitch(wd, nc);
    const int rows = hd + 6, y_top = td.y_first - 3, x_left = td.x_first - 4;
    const float inv_ndw = 1.0f / (float)ndw;
    const float inv_wcell = 1.0f / (float)w_cell;

    // stage the tile: LDS dword c of a row <- the (unaligned) global dword at x_left + 4c
    const int row_end = L.pitch - 4;  // last dword that still starts inside the row
    for (int i = tid; i < rows * ndw; i += 256) {
        const int r = (int)(((float)i + 0.5f) * inv_ndw), c = i - r * ndw;
        const int col = min(x_left + 4 * c, row_end);
        const uint32_t v = *reinterpret_cast<const u32_unaligned*>(img + (uint64_t)(y_top + r) * L.pitch + col);
        *reinterpret_cast<uint32_t*>(&s_tile[r * TP + 4 * c]) = v;
    }
    for (int i = tid; i < (2 * plane_bytes) >> 2; i += 256) reinterpret_cast<uint32_t*>(s_score)[i] = 0;  // score + keep
    if (tid < kFastTileCells) {
        s_kept[tid] = 0;
        s_th[tid] = 0;
    }
    if (tid == 0) s_retry = 0;
    __syncthreads();

    fast_pass<false>(s_tile, s_score, s_keep, s_queue, &s_qn, s_kept, s_th, TP, SP, wd, hd, nc, w_cell, inv_wcell,
                     B.ini_th);
    if (tid < nc) {
        const bool retry = s_kept[tid] == 0;
        s_th[tid] = retry ? 0 : -1;
        if (retry) s_retry = 1;
    }
    __syncthreads();
    if (s_retry)
        fast_pass<true>(s_tile, s_score, s_keep, s_queue, &s_qn, s_kept, s_th, TP, SP, wd, hd, nc, w_cell, inv_wcell,
                        B.min_th);

    // ---- ordered compaction: wave w takes cells w, w+4, ... ----
    for (int c = wave; c < nc; c += 4) {
        const int cwd = c < nc - 1 ? w_cell : wd - (nc - 1) * w_cell;
        const int npx = cwd * hd;
        const float inv_cwd = 1.0f / (float)cwd;
        const uint64_t slot = (uint64_t)blockIdx.y * B.total_cells + td.slot0 + c;
        uint32_t* out = cand + slot * B.cell_cap;
        int base = 0;
        for (int i0 = 0; i0 < npx; i0 += 64) {
            const int i = i0 + lane;
            int keep = 0, px = 0, py = 0, pidx = 0;
            if (i < npx) {
                py = (int)(((float)i + 0.5f) * inv_cwd);
                px = i - py * cwd;
                pidx = (py + 1) * SP + 1 + c * (w_cell + 1) + px;
                keep = s_keep[pidx];
            }
            const uint64_t m = __ballot(keep);
            if (keep) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < B.cell_cap)
                    out[pos] = pack_pt(td.off_x + c * w_cell + px, td.off_y + py, s_score[pidx]);
            }
            base += __popcll(m);
        }
        if (lane == 0) ccount[slot] = (uint32_t)min(base, B.cell_cap);
    }
}

}  // namespace orbx
#endif
