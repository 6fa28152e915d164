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
// corner (OpenCV 3.4.5 cornerScore<16> returns max(t, S) - 1).  So one evaluation of S serves
// both the initial and the minimum threshold.
__device__ __forceinline__ int fast_S(const int d[16]) {
    // windows of 3, then windows of 9 = three windows of 3 (v_min3/v_max3)
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo3[k] = min(d[k], min(d[(k + 1) & 15], d[(k + 2) & 15]));
        hi3[k] = max(d[k], max(d[(k + 1) & 15], d[(k + 2) & 15]));
    }
    int a = -256, b = 256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        a = max(a, min(lo3[k], min(lo3[(k + 3) & 15], lo3[(k + 6) & 15])));
        b = min(b, max(hi3[k], max(hi3[(k + 3) & 15], hi3[(k + 6) & 15])));
    }
    return max(a, -b);
}

constexpr int kFastMaxCell = 66;  // largest sub-image side: wCell <= 59, + 6, + 1

// LDS carve-up of k_fast_tiles for one tile geometry (bytes).
struct FastLds {
    int tile, plane, queue;
};
__host__ __device__ inline int fast_tile_pitch(int x_first, int wd) {
    const int xs = x_first - 3, xa = xs & ~3;
    return (((xs + wd + 6 + 3) >> 2) - (xa >> 2)) * 4;
}
__host__ __device__ inline int fast_plane_pitch(int wd, int nc) { return (wd + nc + 1 + 3) & ~3; }
__host__ __device__ inline FastLds fast_lds_need(int x_first, int wd, int hd, int nc) {
    FastLds n;
    n.tile = fast_tile_pitch(x_first, wd) * (hd + 6) + 16;
    n.plane = fast_plane_pitch(wd, nc) * (hd + 2);
    n.queue = ((wd * hd + 1) & ~1) * 2;
    return n;
}

// One workgroup per tile (a run of up to kFastTileCells cells of one cell row).
// grid = (tiles, frames), block = 256, dynamic LDS = tile + 2 planes + queue.
//  pass 0, threshold iniTh, all cells of the tile:
//   1. every detection pixel takes the two-pair rejection test on dwords of the staged tile
//      (any arc of 9 holds one pixel of each opposite pair); survivors go to an LDS queue;
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
    const int xs = td.x_first - 3, xa = xs & ~3, lead3 = xs - xa + 3;  // tile column of u = 0
    const int TP = fast_tile_pitch(td.x_first, wd), ndw = TP >> 2;
    const int SP = fast_plane_pitch(wd, nc);
    const int rows = hd + 6, y_top = td.y_first - 3;
    const float inv_ndw = 1.0f / (float)ndw;
    const float inv_wcell = 1.0f / (float)w_cell;

    for (int i = tid; i < rows * ndw; i += 256) {
        const int r = (int)(((float)i + 0.5f) * inv_ndw), c = i - r * ndw;
        const uint32_t v = *reinterpret_cast<const uint32_t*>(img + (uint64_t)(y_top + r) * L.pitch + xa + 4 * c);
        *reinterpret_cast<uint32_t*>(&s_tile[r * TP + 4 * c]) = v;
    }
    for (int i = tid; i < (2 * plane_bytes) >> 2; i += 256) reinterpret_cast<uint32_t*>(s_score)[i] = 0;  // score + keep
    if (tid < kFastTileCells) {
        s_kept[tid] = 0;
        s_th[tid] = B.ini_th;
    }
    if (tid == 0) s_retry = 0;
    __syncthreads();

    // detection dwords of a tile row: k0 .. k1
    const int k0 = lead3 >> 2, k1 = (lead3 + wd - 1) >> 2, ndwd = k1 - k0 + 1;
    const float inv_ndwd = 1.0f / (float)ndwd;

    for (int pass = 0; pass < 2; ++pass) {
        if (tid == 0) s_qn = 0;
        __syncthreads();
        // ---- 1. rejection test, four pixels per work item ----
        const int n_items = hd * ndwd;
        for (int i0 = 0; i0 < n_items; i0 += 256) {
            const int i = i0 + tid;
            uint32_t passbits = 0;
            int v = 0, ubase = 0;
            if (i < n_items) {
                v = (int)(((float)i + 0.5f) * inv_ndwd);
                const int k = k0 + (i - v * ndwd);
                ubase = 4 * k - lead3;
                const uint32_t* rc = reinterpret_cast<const uint32_t*>(s_tile + (v + 3) * TP);
                const uint32_t Cd = rc[k], Ld = rc[k - 1], Rd = rc[k + 1];
                const uint32_t Ud = reinterpret_cast<const uint32_t*>(s_tile + (v + 6) * TP)[k];
                const uint32_t Dd = reinterpret_cast<const uint32_t*>(s_tile + v * TP)[k];
                const uint32_t R4 = __builtin_amdgcn_alignbyte(Rd, Cd, 3);
                const uint32_t L12 = __builtin_amdgcn_alignbyte(Cd, Ld, 1);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int u = ubase + b;
                    int t = B.ini_th;
                    bool ok = u >= 0 && u < wd;
                    if (pass) {
                        const int c = min((int)(((float)max(u, 0) + 0.5f) * inv_wcell), nc - 1);
                        t = s_th[c];
                        ok = ok && t >= 0;
                    }
                    const int cv = (Cd >> (8 * b)) & 0xFF;
                    const int d0 = cv - (int)((Ud >> (8 * b)) & 0xFF), d8 = cv - (int)((Dd >> (8 * b)) & 0xFF);
                    const int d4 = cv - (int)((R4 >> (8 * b)) & 0xFF), d12 = cv - (int)((L12 >> (8 * b)) & 0xFF);
                    const bool bright = min(max(d0, d8), max(d4, d12)) > t;
                    const bool dark = max(min(d0, d8), min(d4, d12)) < -t;
                    if (ok && (bright || dark)) passbits |= 1u << b;
                }
            }
            // wave-level append
            const uint64_t m0 = __ballot(passbits & 1u), m1 = __ballot(passbits & 2u), m2 = __ballot(passbits & 4u),
                           m3 = __ballot(passbits & 8u);
            const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
            const int tot = n0 + n1 + n2 + n3;
            if (tot) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_qn, tot);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint64_t below = (1ull << lane) - 1ull;
                const uint32_t e = (uint32_t)(v << 8);
                if (passbits & 1u) s_queue[base + __popcll(m0 & below)] = (uint16_t)(e | (uint32_t)(ubase + 0));
                if (passbits & 2u) s_queue[base + n0 + __popcll(m1 & below)] = (uint16_t)(e | (uint32_t)(ubase + 1));
                if (passbits & 4u) s_queue[base + n0 + n1 + __popcll(m2 & below)] = (uint16_t)(e | (uint32_t)(ubase + 2));
                if (passbits & 8u) s_queue[base + n0 + n1 + n2 + __popcll(m3 & below)] = (uint16_t)(e | (uint32_t)(ubase + 3));
            }
        }
        __syncthreads();
        const int qn = s_qn;
        const int t_full = pass ? B.min_th : B.ini_th;
        // ---- 2. full score for the queued pixels ----
        for (int q = tid; q < qn; q += 256) {
            const int e = s_queue[q], u = e & 0xFF, v = e >> 8;
            const uint8_t* p = s_tile + (v + 3) * TP + lead3 + u;
            const int cv = p[0];
            int d[16];
            d[0] = cv - p[3 * TP];
            d[1] = cv - p[3 * TP + 1];
            d[2] = cv - p[2 * TP + 2];
            d[3] = cv - p[1 * TP + 3];
            d[4] = cv - p[3];
            d[5] = cv - p[-1 * TP + 3];
            d[6] = cv - p[-2 * TP + 2];
            d[7] = cv - p[-3 * TP + 1];
            d[8] = cv - p[-3 * TP];
            d[9] = cv - p[-3 * TP - 1];
            d[10] = cv - p[-2 * TP - 2];
            d[11] = cv - p[-1 * TP - 3];
            d[12] = cv - p[-3];
            d[13] = cv - p[1 * TP - 3];
            d[14] = cv - p[2 * TP - 2];
            d[15] = cv - p[3 * TP - 1];
            const int S = fast_S(d);
            if (S > t_full) {
                const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
                s_score[(v + 1) * SP + 1 + u + c] = (uint8_t)(S - 1);
            }
        }
        __syncthreads();
        // ---- 3. non-max suppression of the queued pixels ----
        for (int q = tid; q < qn; q += 256) {
            const int e = s_queue[q], u = e & 0xFF, v = e >> 8;
            const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
            const uint8_t* sp = s_score + (v + 1) * SP + 1 + u + c;
            const int sc = sp[0];
            if (sc > 0) {
                const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                                  max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
                if (sc > m) {
                    s_keep[(v + 1) * SP + 1 + u + c] = 1;
                    atomicAdd(&s_kept[c], 1);
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (tid < nc) {
                const bool retry = s_kept[tid] == 0;
                s_th[tid] = retry ? B.min_th : -1;
                if (retry) s_retry = 1;
            }
            __syncthreads();
            if (!s_retry) break;
        }
    }

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
