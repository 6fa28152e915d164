// Stereo association of the two extractions of a stereo pair (Frame::ComputeStereoMatches,
// reference src/Frame.cc:485-659) on the pyramids, keypoints and descriptors the extractor left
// in device memory.  Three launches over all pairs of a batch:
//   k_stereo_rows    buckets the right keypoints by the first image row of their row band
//                    (the row table of :494-512, as a counting sort);
//   k_stereo_search  eight lanes per left keypoint: Hamming search over the buckets that can
//                    hold its row (:523-568), then the 11x11 sliding-window distance over
//                    11 shifts and the parabola fit (:570-642);
//   k_stereo_cut     median of the accepted window distances (radix select) and the 1.5*1.4*median cut
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
    int32_t border_only;             // diagnostic: no window takes the dword loader
    // scratch and outputs, per pair
    uint32_t* row_start;             // [rows + 1] first bucket entry of each row, [rows] = count
    StereoMeta* meta;                // [kp_cap]
    int32_t* wmax;                   // widest row band, in rows
    int32_t* sad;                    // [kp_cap] window distance of an accepted match, else -1
    float* u_right;                  // [kp_cap]
    float* depth;                    // [kp_cap]
};

constexpr int kStereoKpPerBlock = 32;  // left keypoints per workgroup of k_stereo_search (8 per wave)

// Minimum over an aligned group of eight lanes, left in all of them, through DPP (lane pairs, quads, the two quads
// of a half row): no LDS round trip.  The eight lanes of a group are active together.
__device__ __forceinline__ uint32_t min_over_8(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

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

__device__ __forceinline__ int stereo_px(const uint8_t* img, int pitch, int w, int h, int x, int y) {
    x = x < 0 ? -x : x;
    x = x >= w ? 2 * w - 2 - x : x;
    y = y < 0 ? -y : y;
    y = y >= h ? 2 * h - 2 - y : y;
    x = min(max(x, 0), w - 1);  // only bytes the distance never reads can lie this far out
    y = min(max(y, 0), h - 1);
    return img[(int64_t)y * pitch + x];
}

constexpr int kStereoWinL = 11 * 16;              // left window rows, pitch 16
constexpr int kStereoWin = kStereoWinL + 11 * 32;  // + right strip rows, pitch 32

// grid = (ceil(kp_cap / kStereoKpPerBlock), pairs), block = 256.
// A wave works on eight left keypoints at once, eight lanes each (group = lane >> 3), so that the
// dependent global loads of the eight (keypoint -> bucket bounds -> bucket entries -> descriptors ->
// matched right keypoint -> windows) overlap instead of queueing behind one another.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_stereo_search(StereoDev S) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][8][kStereoWin];
    __shared__ uint16_t s_acc[4][8][64];  // window distances: at most 121 * 510 < 2^16
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = blockIdx.y;
    const int grp = lane >> 3, j = lane & 7;
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

    const int iL = first + grp;
    const bool have = iL < n_l;
    const orbx_kp kl = kps_l[have ? iL : first];
    const int level = kl.octave;
    const float uL = kl.x, vL = kl.y;
    const int row = (int)vL;  // :530
    const float minU = __fsub_rn(uL, S.max_d), maxU = uL;  // minD = 0

    // ---- Hamming search over the buckets that can hold this row (:523-568)
    uint32_t key = 0xFFFFFFFFu;
    {
        int k = 0, hi = 0;
        if (have && row >= 0 && row < S.rows && !(maxU < 0)) {
            k = (int)row_start[max(0, row - wmax)] + j;
            hi = (int)row_start[row + 1];
        }
        const uint4* dl = reinterpret_cast<const uint4*>(desc_l + (int64_t)(have ? iL : first) * 32);
        const uint4 dl0 = dl[0], dl1 = dl[1];
        StereoMeta m = {0.0f, 0u, 0u};
        if (k < hi) m = meta[k];
        while (k < hi) {
            const int kn = k + 8;
            StereoMeta mn = m;
            if (kn < hi) mn = meta[kn];  // in flight while this entry's descriptor is fetched
            const int minr = (int)(int16_t)(m.rows & 0xFFFFu), maxr = (int)(int16_t)(m.rows >> 16);
            const int oct = (int)(m.id >> 16), iR = (int)(m.id & 0xFFFFu);
            if (row >= minr && row <= maxr && oct >= level - 1 && oct <= level + 1 &&  // :552
                m.x >= minU && m.x <= maxU) {                                          // :557
                const uint4* dr = reinterpret_cast<const uint4*>(desc_r + (int64_t)iR * 32);
                const uint4 r0 = dr[0], r1 = dr[1];
                const int dist = __popc(dl0.x ^ r0.x) + __popc(dl0.y ^ r0.y) + __popc(dl0.z ^ r0.z) + __popc(dl0.w ^ r0.w) +
                                 __popc(dl1.x ^ r1.x) + __popc(dl1.y ^ r1.y) + __popc(dl1.z ^ r1.z) + __popc(dl1.w ^ r1.w);
                // first strict minimum in index order (:562) == minimum of (distance, index)
                if (dist < 100) key = min(key, ((uint32_t)dist << 16) | (uint32_t)iR);  // TH_HIGH
            }
            m = mn;
            k = kn;
        }
        key = min_over_8(key);
    }

    // ---- the group stages its two windows in LDS (:570-600)
    float scaled_ur0 = 0.0f;
    bool refine = false;
    uint8_t* wl = s_win[wave][grp];
    uint8_t* wr = wl + kStereoWinL;
    if (key != 0xFFFFFFFFu && (int)(key >> 16) < 75) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2, :571
        const int best_r = (int)(key & 0xFFFFu);
        const float uR0 = kps_r[best_r].x;
        const float sf = S.inv_scale[level];
        const float scaled_ul = roundf(__fmul_rn(kl.x, sf));
        const float scaled_vl = roundf(__fmul_rn(kl.y, sf));
        scaled_ur0 = roundf(__fmul_rn(uR0, sf));
        const int lw = S.w[level], lh = S.h[level];
        // iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1 with L = w = 5 (:592-595)
        if (!(scaled_ur0 < 0 || __fadd_rn(scaled_ur0, 11.0f) >= (float)lw)) {
            refine = true;
            const uint8_t* IL = S.img_l[level] + (uint64_t)img_l * S.plane_l[level];
            const uint8_t* IR = S.img_r[level] + (uint64_t)img_r * S.plane_r[level];
            const int pl = S.pitch_l[level], pr = S.pitch_r[level];
            const int xl = (int)scaled_ul - 5, yl = (int)scaled_vl - 5;
            const int xr0 = (int)scaled_ur0 - 10;  // left end of the window of shift incR = -5
            // 11 rows of 12 (left) and 24 (right, all 11 shifts) bytes, fetched as dwords
            if (!S.border_only && xl >= 0 && xl + 12 <= lw && xr0 >= 0 && xr0 + 24 <= lw && yl >= 0 && yl + 11 <= lh) {
                // lanes 0-2: the left row's dwords; lanes 3-7: right dwords 0-4; right dword 5 of
                // rows j and j + 8 afterwards
                const bool is_l = j < 3;
                const int pitch = is_l ? pl : pr;
                const uint8_t* src = is_l ? IL + (int64_t)yl * pl + xl + 4 * j : IR + (int64_t)yl * pr + xr0 + 4 * (j - 3);
                uint8_t* dst = is_l ? wl + 4 * j : wr + 4 * (j - 3);
                const int dpitch = is_l ? 16 : 32;
                uint32_t v[11];
#pragma unroll
                for (int r = 0; r < 11; ++r) v[r] = *reinterpret_cast<const u32_unaligned*>(src + (int64_t)r * pitch);
                const uint8_t* src5 = IR + (int64_t)(yl + j) * pr + xr0 + 20;
                const uint32_t e0 = *reinterpret_cast<const u32_unaligned*>(src5);
                uint32_t e1 = 0;
                if (j < 3) e1 = *reinterpret_cast<const u32_unaligned*>(src5 + (int64_t)8 * pr);
#pragma unroll
                for (int r = 0; r < 11; ++r) *reinterpret_cast<uint32_t*>(dst + r * dpitch) = v[r];
                *reinterpret_cast<uint32_t*>(wr + j * 32 + 20) = e0;
                if (j < 3) *reinterpret_cast<uint32_t*>(wr + (j + 8) * 32 + 20) = e1;
            } else {
                // windows reaching into the pyramid border: reflect-101 continuation (:1239-1245)
                for (int i = j; i < 11 * 12 + 11 * 24; i += 8) {
                    if (i < 11 * 12) {
                        const int r = i / 12, c = i - 12 * r;
                        wl[r * 16 + c] = (uint8_t)stereo_px(IL, pl, lw, lh, xl + c, yl + r);
                    } else {
                        const int t = i - 11 * 12, r = t / 24, c = t - 24 * r;
                        wr[r * 32 + c] = (uint8_t)stereo_px(IR, pr, lw, lh, xr0 + c, yl + r);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- window distances, one keypoint after the other with 55 lanes: lane = g * 11 + s is
    // shift s (incR = s - 5) over rows g, g + 5, g + 10 (:597-611)
    const uint64_t todo = __ballot(refine);
    {
        const int s = lane % 11, g = lane / 11;
        const int q4 = s & ~3;
        const uint32_t sh = (uint32_t)(s & 3);
        for (int q = 0; q < 8; ++q) {
            if (!((todo >> (8 * q)) & 1)) continue;  // wave-uniform
            const uint8_t* ql = s_win[wave][q];
            const uint8_t* qr = ql + kStereoWinL;
            if (lane < 55) {
                const int dc = (int)ql[5 * 16 + 5] - (int)qr[5 * 32 + s + 5];
                // |(pa - cl) - (pb - cr)| = |pa - pb - dc| for two pixels per instruction: v_perm_b32 widens two
                // bytes to the halves of a dword, packed 16-bit subtract / negate / maximum / add do the rest
                // (7 instructions per pixel pair instead of 6 per pixel; v_sad_u32 would take one per pixel but
                // issues at a quarter of the rate).  The twelfth pixel of a row is not part of the window: its
                // halves are zero, and so is its half of dc.
                const uint32_t dc2 = (uint32_t)(dc & 0xFFFF) * 0x00010001u, dc1 = (uint32_t)(dc & 0xFFFF);
                uint32_t acc2 = 0;  // two partial sums, each at most 3 rows * 6 pixels * 510
                for (int r = g; r < 11; r += 5) {
                    const uint4 av = *reinterpret_cast<const uint4*>(ql + r * 16);
                    const uint32_t* rq = reinterpret_cast<const uint32_t*>(qr + r * 32 + q4);
                    const uint32_t r0 = rq[0], r1 = rq[1], r2 = rq[2], r3 = rq[3];
                    const uint32_t a[3] = {av.x, av.y, av.z};
                    const uint32_t b[3] = {__builtin_amdgcn_alignbyte(r1, r0, sh), __builtin_amdgcn_alignbyte(r2, r1, sh),
                                           __builtin_amdgcn_alignbyte(r3, r2, sh)};
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) {
                        const uint32_t sel = pp == 5 ? 0x0C0C0C02u : (pp & 1) ? 0x0C030C02u : 0x0C010C00u;
                        const uint32_t A = __builtin_amdgcn_perm(0u, a[pp >> 1], sel), Bq = __builtin_amdgcn_perm(0u, b[pp >> 1], sel);
                        const uint32_t t = pk_sub_i16(pk_sub_i16(A, Bq), pp == 5 ? dc1 : dc2);
                        acc2 = pk_add_u16(acc2, pk_max_i16(t, pk_sub_i16(0u, t)));
                    }
                }
                s_acc[wave][q][lane] = (uint16_t)((acc2 & 0xFFFFu) + (acc2 >> 16));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- best shift and the parabola through its neighbours, per group (:604-642)
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    if (refine) {
        uint16_t* qa = s_acc[wave][grp];
        const uint32_t t0 = (uint32_t)qa[j] + qa[11 + j] + qa[22 + j] + qa[33 + j] + qa[44 + j];  // shift j
        uint32_t t1 = 0;                                                               // shift 8 + j
        if (j < 3) t1 = (uint32_t)qa[8 + j] + qa[19 + j] + qa[30 + j] + qa[41 + j] + qa[52 + j];
        // first strict minimum over incR = -5..5 (:604) == minimum of (distance, shift)
        uint32_t k2 = (t0 << 8) | (uint32_t)j;
        if (j < 3) k2 = min(k2, (t1 << 8) | (uint32_t)(8 + j));
        __builtin_amdgcn_wave_barrier();  // every partial sum was read
        qa[j] = (uint16_t)t0;
        if (j < 3) qa[8 + j] = (uint16_t)t1;
        k2 = min_over_8(k2);
        __builtin_amdgcn_wave_barrier();
        const int sb = (int)(k2 & 0xFFu);
        const int best_w = (int)(k2 >> 8);
        if (sb != 0 && sb != 10) {  // :613
            const float d1 = (float)qa[sb - 1], d2 = (float)qa[sb], d3 = (float)qa[sb + 1];
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
    if (have && j == 0) {
        const size_t o = (size_t)p * S.kp_cap + iL;
        S.u_right[o] = out_u;
        S.depth[o] = out_d;
        S.sad[o] = out_sad;
    }
}

// grid = pairs, block = 1024, dynamic LDS = align4(kp_cap) * 4 bytes.
__global__ __launch_bounds__(1024) void k_stereo_cut(StereoDev S) {
    extern __shared__ __attribute__((aligned(16))) int32_t s_sad[];
    __shared__ int s_count, s_median;
    __shared__ int s_hist[16][16];
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
    __syncthreads();  // every thread has read the count before thread 0 re-uses the variable
    if (m == 0) return;  // nothing was matched
    // vDistIdx[size / 2] of the sorted list (:646): radix select, most significant digit first, four
    // passes of four bits (a window distance is at most 121 * 510 < 2^16).  Each pass counts, per
    // digit value, the accepted distances that match the digits chosen so far (one histogram per
    // wave, LDS atomics), then thread 0 walks the 16 totals to the digit that holds rank k.
    if (tid == 0) {
        s_count = m / 2;  // rank still to be skipped inside the current prefix
        s_median = 0;     // digits chosen so far
    }
    for (int shift = 12; shift >= 0; shift -= 4) {
        if (tid < 256) s_hist[tid >> 4][tid & 15] = 0;
        __syncthreads();
        const int prefix = s_median;
        for (int i = tid; i < n; i += 1024) {
            const int v = s_sad[i];
            if (v >= 0 && (v >> (shift + 4)) == prefix) atomicAdd(&s_hist[tid >> 6][(v >> shift) & 15], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int k = s_count, d = 0;
            for (; d < 15; ++d) {
                int c = 0;
                for (int w = 0; w < 16; ++w) c += s_hist[w][d];
                if (k < c) break;
                k -= c;
            }
            s_count = k;
            s_median = (prefix << 4) | d;
        }
        __syncthreads();
    }
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
