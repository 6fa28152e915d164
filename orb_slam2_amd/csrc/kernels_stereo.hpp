// Stereo association of the two extractions of a stereo pair (Frame::ComputeStereoMatches,
// reference src/Frame.cc:485-659) on the pyramids, keypoints and descriptors the extractor left
// in device memory.  Three launches over all pairs of a batch:
//   k_stereo_rows    buckets the right keypoints by the first image row of their row band
//                    (the row table of :494-512, as a counting sort);
//   k_stereo_search  one wave per left keypoint: Hamming search over the buckets that can
//                    hold its row (:523-568), then the 11x11 sliding-window distance over
//                    11 shifts and the parabola fit (:570-642);
//   k_stereo_cut     median of the accepted window distances and the 1.5*1.4*median cut
//                    (:645-658).
// Every float statement is evaluated as written in the reference (no contraction), so the
// results equal the CPU restatement's bit for bit.
#ifndef ORBX_KERNELS_STEREO_HPP
#define ORBX_KERNELS_STEREO_HPP

#include "kernels_fast.hpp"
#include "orbx_device.h"

namespace orbx {

struct StereoMeta {  // a right keypoint, in bucket order
    float x;         // pt.x
    uint32_t rows;   // (minr & 0xFFFF) | maxr << 16, both as int16
    uint32_t id;     // index in the right keypoint array | octave << 16
};

struct StereoDev {
    const uint8_t* img_l[ORBX_MAX_LEVELS];  // pyramid level of left image 0 (un-blurred)
    const uint8_t* img_r[ORBX_MAX_LEVELS];
    uint64_t plane_l[ORBX_MAX_LEVELS], plane_r[ORBX_MAX_LEVELS];  // bytes between images
    int32_t pitch_l[ORBX_MAX_LEVELS], pitch_r[ORBX_MAX_LEVELS];
    int32_t w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS];
    int32_t nlevels;
    int32_t first_l, first_r, step;  // image index of pair p: first_x + p * step
    const orbx_kp* kps_l;            // keypoints of pair p: kps_x + p * kp_step
    const orbx_kp* kps_r;
    const uint8_t* desc_l;           // descriptors of pair p: desc_x + p * kp_step * 32
    const uint8_t* desc_r;
    const int32_t* n_l;              // counts of pair p: n_x[p * n_step]
    const int32_t* n_r;
    int64_t kp_step;
    int32_t n_step;
    int32_t kp_cap;                  // capacity of every per-pair scratch / output array
    int32_t rows;                    // level-0 image rows (:492)
    float max_d;                     // mbf / mb (:517)
    float mbf;
    // scratch and outputs, per pair
    uint32_t* row_start;             // [rows + 1] first bucket entry of each row, [rows] = count
    StereoMeta* meta;                // [kp_cap]
    int32_t* wmax;                   // widest row band, in rows
    int32_t* sad;                    // [kp_cap] window distance of an accepted match, else -1
    float* u_right;                  // [kp_cap]
    float* depth;                    // [kp_cap]
};

constexpr int kStereoKpPerBlock = 32;  // left keypoints per workgroup of k_stereo_search (8 per wave)

// grid = pairs, block = 256, dynamic LDS = (rows + 1) * 4 bytes.
__global__ __launch_bounds__(256) void k_stereo_rows(StereoDev S) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // rows + 1
    __shared__ int s_wave4[4];
    __shared__ int s_wmax;
    const int tid = threadIdx.x, p = blockIdx.x;
    const int rows = S.rows;
    const orbx_kp* kps = S.kps_r + (int64_t)p * S.kp_step;
    const int n = min(S.n_r[(int64_t)p * S.n_step], S.kp_cap);
    uint32_t* row_start = S.row_start + (size_t)p * (rows + 1);
    StereoMeta* meta = S.meta + (size_t)p * S.kp_cap;
    for (int i = tid; i <= rows; i += 256) s_hist[i] = 0;
    if (tid == 0) s_wmax = 0;
    __syncthreads();
    // row band of a right keypoint: [floor(y - r), ceil(y + r)], r = 2 * scale[octave] (:504-508)
    auto band = [&](const orbx_kp& k, int* minr, int* maxr) {
        const float r = __fmul_rn(2.0f, S.scale[k.octave]);
        *maxr = (int)ceilf(__fadd_rn(k.y, r));
        *minr = (int)floorf(__fsub_rn(k.y, r));
    };
    for (int i = tid; i < n; i += 256) {
        const orbx_kp k = kps[i];
        int minr, maxr;
        band(k, &minr, &maxr);
        if (maxr < 0 || minr >= rows) continue;  // touches no existing row
        const int b = max(minr, 0);
        atomicAdd(&s_hist[b], 1u);
        atomicMax(&s_wmax, maxr - b);
    }
    __syncthreads();
    // exclusive scan of the histogram: thread t owns a run of consecutive rows
    {
        const int per = (rows + 1 + 255) >> 8;
        const int beg = min(tid * per, rows + 1), end = min(beg + per, rows + 1);
        uint32_t sum = 0;
        for (int i = beg; i < end; ++i) sum += s_hist[i];
        int total;
        uint32_t run = (uint32_t)block_excl_scan_256((int)sum, s_wave4, &total);
        for (int i = beg; i < end; ++i) {
            const uint32_t v = s_hist[i];
            s_hist[i] = run;
            row_start[i] = run;
            run += v;
        }
    }
    if (tid == 0) S.wmax[p] = s_wmax;
    __syncthreads();
    // scatter; s_hist now serves as the per-row cursor (order inside a row is irrelevant: the
    // search keeps the minimum of distance << 16 | index)
    for (int i = tid; i < n; i += 256) {
        const orbx_kp k = kps[i];
        int minr, maxr;
        band(k, &minr, &maxr);
        if (maxr < 0 || minr >= rows) continue;
        const int b = max(minr, 0);
        const uint32_t pos = atomicAdd(&s_hist[b], 1u);
        StereoMeta m;
        m.x = k.x;
        m.rows = ((uint32_t)max(minr, -32768) & 0xFFFFu) | ((uint32_t)min(maxr, 32767) << 16);
        m.id = (uint32_t)i | ((uint32_t)k.octave << 16);
        meta[pos] = m;
    }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

__device__ __forceinline__ int stereo_px(const uint8_t* img, int pitch, int w, int h, int x, int y) {
    x = x < 0 ? -x : x;
    x = x >= w ? 2 * w - 2 - x : x;
    y = y < 0 ? -y : y;
    y = y >= h ? 2 * h - 2 - y : y;
    return img[(int64_t)y * pitch + x];
}

// grid = (ceil(kp_cap / kStereoKpPerBlock), pairs), block = 256.
__global__ __launch_bounds__(256) void k_stereo_search(StereoDev S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = blockIdx.y;
    const int n_l = min(S.n_l[(int64_t)p * S.n_step], S.kp_cap);
    const int first = blockIdx.x * kStereoKpPerBlock + wave * (kStereoKpPerBlock / 4);
    if (first >= n_l) return;
    const orbx_kp* kps_l = S.kps_l + (int64_t)p * S.kp_step;
    const orbx_kp* kps_r = S.kps_r + (int64_t)p * S.kp_step;
    const uint8_t* desc_l = S.desc_l + (int64_t)p * S.kp_step * 32;
    const uint8_t* desc_r = S.desc_r + (int64_t)p * S.kp_step * 32;
    const uint32_t* row_start = S.row_start + (size_t)p * (S.rows + 1);
    const StereoMeta* meta = S.meta + (size_t)p * S.kp_cap;
    const int wmax = S.wmax[p];
    const int img_l = S.first_l + p * S.step, img_r = S.first_r + p * S.step;

    for (int q = 0; q < kStereoKpPerBlock / 4; ++q) {
        const int iL = first + q;
        if (iL >= n_l) break;
        float out_u = -1.0f, out_d = -1.0f;
        int out_sad = -1;
        const orbx_kp kl = kps_l[iL];  // same address in every lane
        const int level = kl.octave;
        const float uL = kl.x, vL = kl.y;
        const int row = (int)vL;  // :530
        const float minU = __fsub_rn(uL, S.max_d), maxU = uL;  // minD = 0
        uint32_t key = 0xFFFFFFFFu;
        if (row >= 0 && row < S.rows && !(maxU < 0)) {
            const uint4* dl = reinterpret_cast<const uint4*>(desc_l + (int64_t)iL * 32);
            const uint4 dl0 = dl[0], dl1 = dl[1];
            const int lo = (int)row_start[max(0, row - wmax)], hi = (int)row_start[row + 1];
            for (int k = lo + lane; k < hi; k += 64) {
                const StereoMeta m = meta[k];
                const int minr = (int)(int16_t)(m.rows & 0xFFFFu), maxr = (int)(int16_t)(m.rows >> 16);
                const int oct = (int)(m.id >> 16), iR = (int)(m.id & 0xFFFFu);
                if (row < minr || row > maxr) continue;
                if (oct < level - 1 || oct > level + 1) continue;  // :552
                if (!(m.x >= minU && m.x <= maxU)) continue;       // :557
                const uint4* dr = reinterpret_cast<const uint4*>(desc_r + (int64_t)iR * 32);
                const uint4 r0 = dr[0], r1 = dr[1];
                const int dist = __popc(dl0.x ^ r0.x) + __popc(dl0.y ^ r0.y) + __popc(dl0.z ^ r0.z) + __popc(dl0.w ^ r0.w) +
                                 __popc(dl1.x ^ r1.x) + __popc(dl1.y ^ r1.y) + __popc(dl1.z ^ r1.z) + __popc(dl1.w ^ r1.w);
                // first strict minimum in index order (:562) == minimum of (distance, index)
                if (dist < 100) key = min(key, ((uint32_t)dist << 16) | (uint32_t)iR);  // TH_HIGH
            }
        }
        key = wave_min_u32(key);
        const int best_dist = (int)(key >> 16);
        if (key != 0xFFFFFFFFu && best_dist < 75) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2, :571
            const int best_r = (int)(key & 0xFFFFu);
            const float uR0 = kps_r[best_r].x;
            const float sf = S.inv_scale[level];
            const float scaled_ul = roundf(__fmul_rn(kl.x, sf));
            const float scaled_vl = roundf(__fmul_rn(kl.y, sf));
            const float scaled_ur0 = roundf(__fmul_rn(uR0, sf));
            const int lw = S.w[level], lh = S.h[level];
            // iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1 with L = w = 5 (:592-595)
            if (!(scaled_ur0 < 0 || __fadd_rn(scaled_ur0, 11.0f) >= (float)lw)) {
                const uint8_t* IL = S.img_l[level] + (uint64_t)img_l * S.plane_l[level];
                const uint8_t* IR = S.img_r[level] + (uint64_t)img_r * S.plane_r[level];
                const int pl = S.pitch_l[level], pr = S.pitch_r[level];
                const int xl = (int)scaled_ul - 5, yl = (int)scaled_vl - 5;
                // lane = g * 11 + s: shift s (incR = s - 5), rows g, g + 5, g + 10 of the window
                const int s = lane % 11, g = lane / 11;
                const int xr = (int)scaled_ur0 - 5 + (s - 5);
                uint32_t acc = 0;
                if (lane < 55) {
                    const bool inside = xl >= 0 && xl + 12 <= lw && xr >= 0 && xr + 12 <= lw && yl >= 0 && yl + 11 <= lh;
                    if (inside) {
                        const int cl = IL[(int64_t)(yl + 5) * pl + xl + 5], cr = IR[(int64_t)(yl + 5) * pr + xr + 5];
                        const int dc = cl - cr;
                        for (int r = g; r < 11; r += 5) {
                            const uint8_t* rl = IL + (int64_t)(yl + r) * pl + xl;
                            const uint8_t* rr = IR + (int64_t)(yl + r) * pr + xr;
                            uint32_t a[3], b[3];
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                a[d] = *reinterpret_cast<const u32_unaligned*>(rl + 4 * d);
                                b[d] = *reinterpret_cast<const u32_unaligned*>(rr + 4 * d);
                            }
#pragma unroll
                            for (int xx = 0; xx < 11; ++xx) {
                                const int pa = (int)((a[xx >> 2] >> (8 * (xx & 3))) & 0xFFu);
                                const int pb = (int)((b[xx >> 2] >> (8 * (xx & 3))) & 0xFFu);
                                // |(pa - cl) - (pb - cr)|; v_sad_u32 would do it in one instruction but issues
                                // at a quarter of the rate of sub/max
                                const int t = pa - pb - dc;
                                acc += (uint32_t)max(t, -t);
                            }
                        }
                    } else {
                        // windows reaching into the pyramid border: reflect-101 continuation (:1239-1245)
                        const int cl = stereo_px(IL, pl, lw, lh, xl + 5, yl + 5), cr = stereo_px(IR, pr, lw, lh, xr + 5, yl + 5);
                        for (int r = g; r < 11; r += 5)
                            for (int xx = 0; xx < 11; ++xx) {
                                const int da = stereo_px(IL, pl, lw, lh, xl + xx, yl + r) - cl;
                                const int db = stereo_px(IR, pr, lw, lh, xr + xx, yl + r) - cr;
                                acc += (uint32_t)abs(da - db);
                            }
                    }
                }
                // sum the five row groups of each shift into lanes 0..10
                uint32_t tot = acc;
#pragma unroll
                for (int gg = 1; gg < 5; ++gg) tot += (uint32_t)__shfl((int)acc, (lane + 11 * gg) & 63, 64);
                // first strict minimum over incR = -5..5 (:604) == minimum of (distance, shift)
                const uint32_t k2 = wave_min_u32(lane < 11 ? (tot << 8) | (uint32_t)lane : 0xFFFFFFFFu);
                const int sb = (int)(k2 & 0xFFu);
                const int best_w = (int)(k2 >> 8);
                const float d1 = (float)(uint32_t)__shfl((int)tot, (sb + 63) & 63, 64);
                const float d2 = (float)(uint32_t)__shfl((int)tot, sb, 64);
                const float d3 = (float)(uint32_t)__shfl((int)tot, (sb + 1) & 63, 64);
                if (sb != 0 && sb != 10) {  // :613
                    // :621, float arithmetic in the reference's order
                    const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
                    const float delta = __fdiv_rn(__fsub_rn(d1, d3), den);
                    if (!(delta < -1 || delta > 1)) {
                        float best_ur = __fmul_rn(S.scale[level], __fadd_rn(__fadd_rn(scaled_ur0, (float)(sb - 5)), delta));  // :627
                        float disparity = __fsub_rn(uL, best_ur);
                        if (disparity >= 0 && disparity < S.max_d) {  // :631
                            if (disparity <= 0) {  // :633-637 (double constants)
                                disparity = (float)0.01;
                                best_ur = (float)((double)uL - 0.01);
                            }
                            out_d = __fdiv_rn(S.mbf, disparity);
                            out_u = best_ur;
                            out_sad = best_w;
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            const size_t o = (size_t)p * S.kp_cap + iL;
            S.u_right[o] = out_u;
            S.depth[o] = out_d;
            S.sad[o] = out_sad;
        }
    }
}

// grid = pairs, block = 1024, dynamic LDS = align4(kp_cap) * 4 bytes.
__global__ __launch_bounds__(1024) void k_stereo_cut(StereoDev S) {
    extern __shared__ __attribute__((aligned(16))) int32_t s_sad[];
    __shared__ int s_count, s_median;
    const int tid = threadIdx.x, p = blockIdx.x;
    const int n = min(S.n_l[(int64_t)p * S.n_step], S.kp_cap);
    const int n4 = (n + 3) & ~3;
    const int32_t* sad = S.sad + (size_t)p * S.kp_cap;
    if (tid == 0) {
        s_count = 0;
        s_median = -1;
    }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n4; i += 1024) {
        const int v = i < n ? sad[i] : -1;
        s_sad[i] = v;
        mine += v >= 0 ? 1 : 0;
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    const int m = s_count;
    if (m == 0) return;  // nothing was matched
    const int k = m / 2;  // vDistIdx[size / 2] of the sorted list (:646)
    const int4* s4 = reinterpret_cast<const int4*>(s_sad);
    for (int i = tid; i < n; i += 1024) {
        const int v = s_sad[i];
        if (v < 0) continue;
        int less = 0, leq = 0;
        for (int j = 0; j < n4 >> 2; ++j) {
            const int4 t = s4[j];
            // unmatched entries are -1: never >= 0, so they are masked by the (t >= 0) terms
            less += (t.x >= 0 && t.x < v) + (t.y >= 0 && t.y < v) + (t.z >= 0 && t.z < v) + (t.w >= 0 && t.w < v);
            leq += (t.x >= 0 && t.x <= v) + (t.y >= 0 && t.y <= v) + (t.z >= 0 && t.z <= v) + (t.w >= 0 && t.w <= v);
        }
        if (less <= k && k < leq) s_median = v;  // every writer holds the same value
    }
    __syncthreads();
    const float th = __fmul_rn(1.5f * 1.4f, (float)s_median);  // :647
    for (int i = tid; i < n; i += 1024) {
        const int v = s_sad[i];
        if (v >= 0 && !((float)v < th)) {  // :649-657
            S.u_right[(size_t)p * S.kp_cap + i] = -1.0f;
            S.depth[(size_t)p * S.kp_cap + i] = -1.0f;
        }
    }
}

}  // namespace orbx
#endif
