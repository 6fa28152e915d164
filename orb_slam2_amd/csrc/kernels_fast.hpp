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

constexpr int kFastMaxCell = 66;                 // largest sub-image side: wCell <= 59, + 6, + 1
constexpr int kFastTilePitch = 76;               // 66 + up to 3 bytes of left alignment, dword multiple
constexpr int kFastScorePitch = 64;              // (66 - 4) scores + rim

// One workgroup per visited cell.  grid = (total_cells, frames), block = 256.
//  1. stage the sub-image in LDS with aligned dword loads;
//  2. S for every interior pixel; score = S-1 where S > minTh, else 0;
//  3. strict 3x3 non-max suppression on the score tile (scores outside the interior are 0,
//     so the suppression is cell-local exactly as when cv::FAST runs on the sub-image);
//  4. a maximum survives the initial threshold iff its score >= iniTh (lower-scored
//     neighbours cannot suppress it); if no maximum of the cell does, every maximum is kept
//     (the minThFAST retry, reference :903-907);
//  5. ordered (row-major) compaction into the cell's candidate slot.
__global__ __launch_bounds__(256) void k_fast_cells(BatchDev B, const CellDesc* __restrict__ cells,
                                                    uint32_t* __restrict__ cand, uint32_t* __restrict__ ccount) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[kFastMaxCell * kFastTilePitch];
    __shared__ uint8_t s_score[(kFastMaxCell - 4) * kFastScorePitch];
    __shared__ int s_wave4[4];
    __shared__ int s_cnt_ini;

    const int tid = threadIdx.x;
    const CellDesc cd = cells[blockIdx.x];
    const LevelDev& L = B.lv[cd.level];
    const uint8_t* img = L.img + (uint64_t)blockIdx.y * L.plane;
    const int cw = cd.cw, ch = cd.ch;
    const int xa = cd.x0 & ~3;               // aligned start column
    const int lead = cd.x0 - xa;             // tile column of sub-image column 0
    const int ndw = ((cd.x0 + cw + 3) >> 2) - (xa >> 2);

    if (tid == 0) s_cnt_ini = 0;
    for (int i = tid; i < ch * ndw; i += 256) {
        const int r = i / ndw, c = i - r * ndw;
        const uint32_t v = *reinterpret_cast<const uint32_t*>(img + (uint64_t)(cd.y0 + r) * L.pitch + xa + 4 * c);
        *reinterpret_cast<uint32_t*>(&s_tile[r * kFastTilePitch + 4 * c]) = v;
    }
    for (int i = tid; i < (kFastMaxCell - 4) * kFastScorePitch / 4; i += 256) reinterpret_cast<uint32_t*>(s_score)[i] = 0;
    __syncthreads();

    const int dw = cw - 6, dh = ch - 6;  // detection area
    const int npx = dw > 0 && dh > 0 ? dw * dh : 0;
    const float inv_dw = 1.0f / (float)max(dw, 1);
    const int min_th = B.min_th, ini_th = B.ini_th;

    for (int i = tid; i < npx; i += 256) {
        const int py = (int)(((float)i + 0.5f) * inv_dw);
        const int px = i - py * dw;
        const uint8_t* p = &s_tile[(py + 3) * kFastTilePitch + px + 3 + lead];
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * kFastTilePitch];
        d[8] = v - p[-3 * kFastTilePitch];
        d[4] = v - p[3];
        d[12] = v - p[-3];
        // any arc of 9 holds one pixel of each opposite pair: reject early on two pairs
        const bool brightish = (max(d[0], d[8]) > min_th) && (max(d[4], d[12]) > min_th);
        const bool darkish = (min(d[0], d[8]) < -min_th) && (min(d[4], d[12]) < -min_th);
        if (brightish || darkish) {
            d[1] = v - p[3 * kFastTilePitch + 1];
            d[2] = v - p[2 * kFastTilePitch + 2];
            d[3] = v - p[1 * kFastTilePitch + 3];
            d[5] = v - p[-1 * kFastTilePitch + 3];
            d[6] = v - p[-2 * kFastTilePitch + 2];
            d[7] = v - p[-3 * kFastTilePitch + 1];
            d[9] = v - p[-3 * kFastTilePitch - 1];
            d[10] = v - p[-2 * kFastTilePitch - 2];
            d[11] = v - p[-1 * kFastTilePitch - 3];
            d[13] = v - p[1 * kFastTilePitch - 3];
            d[14] = v - p[2 * kFastTilePitch - 2];
            d[15] = v - p[3 * kFastTilePitch - 1];
            const int S = fast_S(d);
            if (S > min_th) s_score[(py + 1) * kFastScorePitch + px + 1] = (uint8_t)(S - 1);
        }
    }
    __syncthreads();

    // Non-max suppression; flags overwrite the (no longer needed) image tile, one byte per pixel:
    // bit0 = local maximum (kept at minTh), bit1 = also kept at iniTh.
    uint8_t* s_flag = s_tile;
    int my_ini = 0;
    for (int i = tid; i < npx; i += 256) {
        const int py = (int)(((float)i + 0.5f) * inv_dw);
        const int px = i - py * dw;
        const uint8_t* s = &s_score[(py + 1) * kFastScorePitch + px + 1];
        const int sc = s[0];
        int f = 0;
        if (sc > 0) {
            const int m = max(max(max((int)s[-1], (int)s[1]), max((int)s[-kFastScorePitch - 1], (int)s[-kFastScorePitch])),
                              max(max((int)s[-kFastScorePitch + 1], (int)s[kFastScorePitch - 1]),
                                  max((int)s[kFastScorePitch], (int)s[kFastScorePitch + 1])));
            if (sc > m) {
                f = 1;
                if (sc >= ini_th) {
                    f = 3;
                    my_ini++;
                }
            }
        }
        s_flag[i] = (uint8_t)f;
    }
    if (my_ini) atomicAdd(&s_cnt_ini, my_ini);
    __syncthreads();
    const int want = s_cnt_ini > 0 ? 2 : 1;

    // ordered compaction: thread t owns the row-major run [t*per, (t+1)*per)
    const int per = (npx + 255) >> 8;
    const int beg = min(tid * per, npx), end = min(beg + per, npx);
    int mine = 0;
    for (int i = beg; i < end; ++i) mine += (s_flag[i] & want) ? 1 : 0;
    int total;
    int pos = block_excl_scan_256(mine, s_wave4, &total);
    const uint64_t slot = (uint64_t)blockIdx.y * B.total_cells + cd.slot;
    uint32_t* out = cand + slot * B.cell_cap;
    for (int i = beg; i < end; ++i) {
        if (s_flag[i] & want) {
            const int py = (int)(((float)i + 0.5f) * inv_dw);
            const int px = i - py * dw;
            const int sc = s_score[(py + 1) * kFastScorePitch + px + 1];
            if (pos < B.cell_cap) out[pos] = pack_pt(px + 3 + cd.off_x, py + 3 + cd.off_y, sc);
            pos++;
        }
    }
    if (tid == 0) ccount[slot] = (uint32_t)min(total, B.cell_cap);
}

}  // namespace orbx
#endif
