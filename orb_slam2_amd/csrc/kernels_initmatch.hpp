// ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549): the matcher of the monocular
// initialisation (Tracking::MonocularInitialization, src/Tracking.cc:667).  Every level-0 keypoint of the
// first frame searches a square window of the second frame around the place it was matched at last time,
// keeps the two smallest distances among the features that no earlier keypoint holds with a distance at
// least as small (vMatchedDistance, :465), and on success takes its best feature away from whoever held
// it (:486-490).  That makes the loop sequential in the first frame's keypoint index.
//
// One workgroup serves one pair of frames.  The expensive part of an iteration -- walking the window and
// the 256-bit distances -- does not depend on earlier iterations, so 64 keypoints do it at once, sixteen
// lanes each, and leave (key, feature | rotation bin << 16) lists in LDS; sixteen lanes then replay the 64 decisions in index order
// against vMatchedDistance / vnMatches21 / vnMatches12, which live in LDS as 16-bit arrays.  Keypoints above
// level 0 do nothing in the reference's loop (:438), so the chunks are cut from the list of level-0 keypoints.
#ifndef ORBX_KERNELS_INITMATCH_HPP
#define ORBX_KERNELS_INITMATCH_HPP

#include "kernels_bow.hpp"

namespace orbx {

constexpr int kInitThreads = 1024;
constexpr int kInitGroups = kInitThreads / 16;  // keypoints of the first frame whose windows are walked together
constexpr int kInitListCap = 96;                // candidates a keypoint's list holds; longer ones are walked again
constexpr int kInitMaxCap = 8192;               // features per frame: the state arrays are 16-bit and live in LDS
constexpr int kInitThLow = 50;                  // ORBmatcher::TH_LOW, ORBmatcher.cc:38

struct InitDev {
    WindowDev W;               // the second frame; u_right and taken are null, q_cap is not used
    const orbx_kp* kps_un_1;   // [n][cap_1]
    const uint8_t* desc_1;     // [n][cap_1][32]
    const int32_t* n_1;        // [n]
    float2* prev_matched;      // [n][cap_1], read and updated
    int32_t* match12;          // [n][cap_1]
    int32_t* n_matches;        // [n]
    float window, nn_ratio;
    int32_t cap_1, check_orientation, list_cap;
};

inline size_t init_match_lds_bytes(int cap_1, int cap_2, int list_cap) {
    return (size_t)kInitGroups * list_cap * 8 + (size_t)cap_2 * 4 + (size_t)cap_1 * 4;
}

// grid = pairs of frames, block = 1024, dynamic LDS = init_match_lds_bytes(cap_1, cap_2, list_cap).
__global__ __launch_bounds__(kInitThreads, 8) void k_search_init(InitDev D) {
    __shared__ int s_cnt[kInitGroups], s_hist[kHistoLength], s_keep[3], s_nm, s_wave[kInitThreads / 64];
    extern __shared__ __attribute__((aligned(16))) int32_t s_dyn[];  // the lists are read and written as 8-byte entries
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, sub = tid & 15, grp = tid >> 4;
    const int cap_1 = D.cap_1, cap_2 = D.W.kp_cap, lc = D.list_cap;
    uint2* s_list = reinterpret_cast<uint2*>(s_dyn);                                     // [64][lc] (key, feature | bin << 16)
    uint16_t* s_vmd = reinterpret_cast<uint16_t*>(s_list + (size_t)kInitGroups * lc);    // vMatchedDistance, 0xFFFF: INT_MAX
    int16_t* s_vn21 = reinterpret_cast<int16_t*>(s_vmd + cap_2);                         // vnMatches21
    int16_t* s_vn12 = s_vn21 + cap_2;                                                    // vnMatches12
    int16_t* s_l0 = s_vn12 + cap_1;                                                      // the level-0 keypoints, ascending
    const orbx_kp* kps1 = D.kps_un_1 + (size_t)f * cap_1;
    const orbx_kp* kps2 = D.W.kps_un + (size_t)f * cap_2;
    const uint8_t* desc1 = D.desc_1 + (size_t)f * cap_1 * 32;
    float2* pm = D.prev_matched + (size_t)f * cap_1;
    int32_t* match12 = D.match12 + (size_t)f * cap_1;
    const int n1 = max(0, min(D.n_1[f], cap_1));

    for (int i = tid; i < cap_2; i += kInitThreads) {
        s_vmd[i] = 0xFFFFu;
        s_vn21[i] = -1;
    }
    for (int i = tid; i < cap_1; i += kInitThreads) s_vn12[i] = -1;
    if (tid < kHistoLength) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    // the keypoints that search (:438), in index order
    int n0 = 0;
    for (int b = 0; b < n1; b += kInitThreads) {
        const int i = b + tid;
        const bool searches = i < n1 && !(kps1[i].octave > 0);
        const unsigned long long m = __ballot(searches);
        if (lane == 0) s_wave[tid >> 6] = __popcll(m);
        __syncthreads();
        int before = n0, all = n0;
        for (int w = 0; w < kInitThreads / 64; ++w) {
            const int c = s_wave[w];
            before += w < (tid >> 6) ? c : 0;
            all += c;
        }
        if (searches) s_l0[before + __popcll(m & ((1ull << lane) - 1ull))] = (int16_t)i;
        n0 = all;
        __syncthreads();
    }

    const auto nobody = [](int, const orbx_kp&) { return false; };
    for (int c0 = 0; c0 < n0; c0 += kInitGroups) {
        // ---- all windows of the chunk at once: lists of (key, feature), key = distance << 23 | visiting order ----
        if (sub == 0) s_cnt[grp] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (c0 + grp < n0) {  // the sixteen lanes of a group decide alike
            const int q = s_l0[c0 + grp];
            const float2 at = pm[q];
            const uint4 d0 = *reinterpret_cast<const uint4*>(desc1 + (size_t)q * 32);
            const uint4 d1 = *reinterpret_cast<const uint4*>(desc1 + (size_t)q * 32 + 16);
            const int level1 = kps1[q].octave;
            const float angle1 = kps1[q].angle;
            uint2* mine = s_list + (size_t)grp * lc;
            int win;
            uint32_t kmin;
            window_best(D.W, f, at.x, at.y, D.window, 0.0f, level1, level1, d0, d1, sub, nobody,
                        [&](int i2, uint32_t, uint32_t kc, const orbx_kp& kp) __attribute__((always_inline)) {
                            const int s = atomicAdd(&s_cnt[grp], 1);
                            if (s < lc) mine[s] = make_uint2(kc, (uint32_t)(i2 | (proj_rot_bin(angle1, kp.angle) << 16)));
                        },
                        &win, &kmin);
        }
        __syncthreads();
        // ---- the chunk's decisions in index order, by the first sixteen lanes (one DPP row) ----
        if (tid < 16) {
            const int nk = min(kInitGroups, n0 - c0);
            for (int k = 0; k < nk; ++k) {
                const int cnt = s_cnt[k];
                if (cnt == 0) continue;  // an empty window (:446)
                const int i1 = s_l0[c0 + k];
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                int v1 = -1, v2 = -1;
                if (cnt <= lc) {
                    const uint2* L = s_list + (size_t)k * lc;
                    for (int e = tid; e < cnt; e += 16) {
                        const uint2 c = L[e];
                        if ((uint32_t)s_vmd[c.y & 0xFFFFu] > (c.x >> 23))  // :465
                            top2_merge(k1, k2, v1, v2, c.x, 0xFFFFFFFFu, (int)c.y, -1);
                    }
                } else {
                    // the list overflowed: walk the window again and filter on the way
                    const float2 at = pm[i1];
                    const uint4 d0 = *reinterpret_cast<const uint4*>(desc1 + (size_t)i1 * 32);
                    const uint4 d1 = *reinterpret_cast<const uint4*>(desc1 + (size_t)i1 * 32 + 16);
                    const int level1 = kps1[i1].octave;
                    const float angle1 = kps1[i1].angle;
                    int win;
                    uint32_t kmin;
                    window_best(D.W, f, at.x, at.y, D.window, 0.0f, level1, level1, d0, d1, tid, nobody,
                                [&](int i2, uint32_t dist, uint32_t kc, const orbx_kp& kp) __attribute__((always_inline)) {
                                    if ((uint32_t)s_vmd[i2] > dist)
                                        top2_merge(k1, k2, v1, v2, kc, 0xFFFFFFFFu, i2 | (proj_rot_bin(angle1, kp.angle) << 16), -1);
                                },
                                &win, &kmin);
                }
                top2_row_merge(k1, k2, v1, v2);
                if (k1 == 0xFFFFFFFFu) continue;
                const int bestDist = (int)(k1 >> 23), bestIdx2 = v1 & 0xFFFF, bin = v1 >> 16;
                const float second = k2 == 0xFFFFFFFFu ? 2147483648.0f : (float)(k2 >> 23);  // (float)INT_MAX
                if (bestDist <= kInitThLow && (float)bestDist < __fmul_rn(second, D.nn_ratio)) {  // :482-484
                    if (tid == 0) {
                        int nm = s_nm;
                        const int before = s_vn21[bestIdx2];
                        if (before >= 0) {  // :486-490
                            s_vn12[before] = -1;
                            --nm;
                        }
                        s_vn12[i1] = (int16_t)bestIdx2;  // :492-495
                        s_vn21[bestIdx2] = (int16_t)i1;
                        s_vmd[bestIdx2] = (uint16_t)bestDist;
                        s_nm = nm + 1;
                        // the histogram counts every accepted match, those replaced later too (:497-509)
                        s_hist[bin] += 1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
        }
        __syncthreads();
    }

    if (D.check_orientation) {  // :515-540
        if (tid == 0) rot_three_maxima(s_hist, s_keep);
        __syncthreads();
    }
    for (int i1 = tid; i1 < cap_1; i1 += kInitThreads) {
        int m = i1 < n1 ? (int)s_vn12[i1] : -1;
        if (m >= 0 && D.check_orientation) {
            const int bin = proj_rot_bin(kps1[i1].angle, kps2[m].angle);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {  // :532-536
                m = -1;
                atomicSub(&s_nm, 1);
            }
        }
        match12[i1] = m;
        if (m >= 0) pm[i1] = make_float2(kps2[m].x, kps2[m].y);  // :544-546
    }
    __syncthreads();
    if (tid == 0) D.n_matches[f] = s_nm;
}

}  // namespace orbx
#endif
