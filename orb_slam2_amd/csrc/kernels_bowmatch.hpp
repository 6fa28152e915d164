// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (reference src/ORBmatcher.cc:159-290):
// the features of a keyframe and of a frame that fall into the same vocabulary node are matched, node by
// node (called from Tracking::TrackReferenceKeyFrame, src/Tracking.cc:894, and Tracking::Relocalization, :1563),
// and ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (:551-684, called from
// LoopClosing::ComputeSim3, src/LoopClosing.cc:268), the same loop between two keyframes.
#ifndef ORBX_KERNELS_BOWMATCH_HPP
#define ORBX_KERNELS_BOWMATCH_HPP

#include "kernels_bow.hpp"

namespace orbx {

constexpr int kBowMatchThreads = 1024;
constexpr int kThLow = 50;  // ORBmatcher.cc:38

struct BowMatchDev {
    // the keyframe side of pair f: FeatureVector pairs [f*cap_kf .. + fv_n_kf[f]), descriptors, keypoints
    // (for the angle), and per feature whether it holds a map point that is not bad (:193-197)
    const uint32_t *fv_node_kf, *fv_idx_kf;
    const int32_t* fv_n_kf;
    const uint8_t* desc_kf;
    const orbx_kp* kps_kf;
    const uint8_t* valid_kf;
    int cap_kf;
    // the frame side
    const uint32_t *fv_node_f, *fv_idx_f;
    const int32_t* fv_n_f;
    const uint8_t* desc_f;
    const orbx_kp* kps_f;
    int cap_f;
    float nn_ratio;
    int check_orientation;
    // Two keyframes (:551-684) instead of keyframe and frame: the second side's features need a good map point
    // too (valid_f, :603-609), the best distance must be BELOW TH_LOW (:627), and the result is indexed by the
    // first keyframe's features (vpMatches12, :631).
    int two_keyframes;
    const uint8_t* valid_f;
    int32_t* match;      // [n][cap_f]: keyframe feature whose map point the frame feature ends with, -1: none;
                         // two keyframes: [n][cap_kf]: the second keyframe's feature per feature of the first
    int32_t* n_matches;  // [n]
    int taken_words;     // 32-bit words of "already matched" flags per lane: ceil(ceil(cap_f / 64) / 32)
};

// Two sorted pairs of keys (k1 <= k2, o1 <= o2) -> the two smallest of the four.
__device__ __forceinline__ void bow_top2_merge(uint32_t& k1, uint32_t& k2, uint32_t o1, uint32_t o2) {
    const uint32_t lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
}

template <int CTRL>
__device__ __forceinline__ void bow_top2_merge_dpp(uint32_t& k1, uint32_t& k2) {
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k1, CTRL, 0xF, 0xF, true);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k2, CTRL, 0xF, 0xF, true);
    bow_top2_merge(k1, k2, o1, o2);
}

// First index in [from, n) whose node differs from `node` (n if none), found by the wave 64 entries at a time.
__device__ __forceinline__ int bow_run_end(const uint32_t* nodes, int from, int n, uint32_t node) {
    const int lane = threadIdx.x & 63;
    for (int e = from;; e += 64) {
        const int c = e + lane;
        const bool differs = c >= n || nodes[c] != node;
        const uint64_t m = __ballot(differs);
        if (m) return e + __builtin_ctzll(m);
    }
}

// One workgroup per (keyframe, frame) pair.  Nodes are independent of each other (a feature belongs to one
// node), so the waves share the keyframe's runs of equal node id; inside a run the reference's loop over the
// keyframe's features is kept as it is, because a frame feature taken by one of them is skipped by the next
// (:209-210).  For one keyframe feature the 64 lanes scan the frame's features of the node (the first 64 stay
// in registers for the whole run), each keeps its two smallest (distance, list position) keys, a butterfly
// merge leaves the wave's two smallest: the reference's bestDist1 / bestIdxF / bestDist2 (:216-225 takes the
// first of equal distances, and bestDist2 ends as the second smallest distance of the list).  The "taken"
// flags of the run's candidates are one bit per (lane, chunk of 64) in LDS.  A match is written as keyframe
// feature | rotation bin << 24; after all runs the 30-bin histogram decides which bins stay (:267-288,
// ComputeThreeMaxima :1638-1679) and a last pass over the frame's features strips or clears the entries.
// grid = pairs, block = 1024, dynamic LDS = 16 waves * taken_words * 64 * 4 bytes.
__global__ __launch_bounds__(kBowMatchThreads, 8) void k_search_bow(BowMatchDev D) {
    extern __shared__ uint32_t s_taken[];
    __shared__ int s_nm, s_hist[kHistoLength], s_keep[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.x;
    // the counts are cut to the arrays' capacities, as in the sibling kernels: a larger value would send the run
    // search and the descriptor loads into the next pair's slice
    const int nkf = max(0, min(D.fv_n_kf[f], D.cap_kf)), nf = max(0, min(D.fv_n_f[f], D.cap_f));
    const uint32_t* nodes_kf = D.fv_node_kf + (size_t)f * D.cap_kf;
    const uint32_t* idx_kf = D.fv_idx_kf + (size_t)f * D.cap_kf;
    const uint8_t* desc_kf = D.desc_kf + (size_t)f * D.cap_kf * 32;
    const orbx_kp* kps_kf = D.kps_kf + (size_t)f * D.cap_kf;
    const uint8_t* valid_kf = D.valid_kf + (size_t)f * D.cap_kf;
    const uint32_t* nodes_f = D.fv_node_f + (size_t)f * D.cap_f;
    const uint32_t* idx_f = D.fv_idx_f + (size_t)f * D.cap_f;
    const uint8_t* desc_f = D.desc_f + (size_t)f * D.cap_f * 32;
    const orbx_kp* kps_f = D.kps_f + (size_t)f * D.cap_f;
    const uint8_t* valid_f = D.two_keyframes ? D.valid_f + (size_t)f * D.cap_f : nullptr;
    const int cap_out = D.two_keyframes ? D.cap_kf : D.cap_f;
    int32_t* match = D.match + (size_t)f * cap_out;
    const int th_low = D.two_keyframes ? kThLow - 1 : kThLow;  // :228 "<= TH_LOW", :627 "< TH_LOW"

    for (int i = tid; i < cap_out; i += kBowMatchThreads) match[i] = -1;  // :163, :563
    if (tid < kHistoLength) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();

    uint32_t* my_taken = s_taken + (size_t)wave * D.taken_words * 64;
    constexpr uint32_t kNone = (256u << 16) | 0xFFFFu;  // no candidate: distance 256 (:201-203)
    // the wave's slice of the keyframe's pairs; it works on the runs whose first pair lies in the slice
    const int per = ((nkf + kBowMatchThreads - 1) / kBowMatchThreads) * 64;
    const int beg = wave * per, end = min(beg + per, nkf);
    for (int i0 = beg; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t node_i = i < end ? nodes_kf[i] : 0u;
        const bool head = i < end && (i == 0 || nodes_kf[i - 1] != node_i);
        uint64_t heads = __ballot(head);
        while (heads) {
            const int hl = __builtin_ctzll(heads);
            heads &= heads - 1;
            const int h = i0 + hl;
            const uint32_t node = (uint32_t)__builtin_amdgcn_readlane((int)node_i, hl);
            const int e = bow_run_end(nodes_kf, h + 1, nkf, node);
            // the frame's pairs of the same node: lower bound, then the end of the run (:182, :256-263)
            int lo = 0, hi = nf;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (nodes_f[mid] < node) lo = mid + 1; else hi = mid;
            }
            if (lo >= nf || nodes_f[lo] != node) continue;
            const int fe = bow_run_end(nodes_f, lo + 1, nf, node);
            const int chunks = (fe - lo + 63) >> 6;
            for (int w = 0; w < (chunks + 31) >> 5; ++w) my_taken[w * 64 + lane] = 0u;
            if (valid_f)  // a feature of the second keyframe without a good map point is never a candidate (:603-609)
                for (int j = 0; j < chunks; ++j) {
                    const int c = lo + lane + 64 * j;
                    if (c < fe && !valid_f[idx_f[c]]) my_taken[(j >> 5) * 64 + lane] |= 1u << (j & 31);
                }
            // the first 64 candidates stay in registers for the whole run
            const bool has0 = lo + lane < fe;
            uint4 c0a = make_uint4(0, 0, 0, 0), c0b = make_uint4(0, 0, 0, 0);
            if (has0) {
                const uint4* p = reinterpret_cast<const uint4*>(desc_f + (size_t)idx_f[lo + lane] * 32);
                c0a = p[0];
                c0b = p[1];
            }
            for (int k = h; k < e; ++k) {  // :187
                const uint32_t kf_idx = idx_kf[k];
                if (!valid_kf[kf_idx]) continue;  // :193-197
                const uint4* pk = reinterpret_cast<const uint4*>(desc_kf + (size_t)kf_idx * 32);
                const uint4 ka = pk[0], kb = pk[1];
                uint32_t k1 = kNone, k2 = kNone;
                for (int j = 0; j < chunks; ++j) {
                    const int c = lo + lane + 64 * j;
                    const uint32_t tw = my_taken[(j >> 5) * 64 + lane];
                    if (c < fe && !((tw >> (j & 31)) & 1u)) {  // :209-210
                        uint4 a = c0a, b = c0b;
                        if (j > 0) {
                            const uint4* p = reinterpret_cast<const uint4*>(desc_f + (size_t)idx_f[c] * 32);
                            a = p[0];
                            b = p[1];
                        }
                        const int dist = __popc(a.x ^ ka.x) + __popc(a.y ^ ka.y) + __popc(a.z ^ ka.z) + __popc(a.w ^ ka.w) +
                                         __popc(b.x ^ kb.x) + __popc(b.y ^ kb.y) + __popc(b.z ^ kb.z) + __popc(b.w ^ kb.w);
                        if (dist < 256) {  // a distance of 256 beats neither 256 (:216, :222)
                            const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)(c - lo);
                            if (key < k1) {
                                k2 = k1;
                                k1 = key;
                            } else if (key < k2) {
                                k2 = key;
                            }
                        }
                    }
                }
                // inside a row of sixteen lanes through DPP (pairs, quads, halves, the row), then across the four rows
                bow_top2_merge_dpp<0xB1>(k1, k2);
                bow_top2_merge_dpp<0x4E>(k1, k2);
                bow_top2_merge_dpp<0x141>(k1, k2);
                bow_top2_merge_dpp<0x140>(k1, k2);
#pragma unroll
                for (int off = 16; off < 64; off <<= 1) {
                    const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, off), o2 = (uint32_t)__shfl_xor((int)k2, off);
                    bow_top2_merge(k1, k2, o1, o2);
                }
                const int best1 = (int)(k1 >> 16), best2 = (int)(k2 >> 16);
                if (best1 <= th_low && (float)best1 < __fmul_rn(D.nn_ratio, (float)best2)) {  // :228-230
                    const int pos = (int)(k1 & 0xFFFFu);
                    const uint32_t idxF = idx_f[lo + pos];
                    if (lane == (pos & 63)) my_taken[((pos >> 6) >> 5) * 64 + lane] |= 1u << ((pos >> 6) & 31);
                    const int bin = D.check_orientation ? proj_rot_bin(kps_kf[kf_idx].angle, kps_f[idxF].angle) : 0;  // :236-248
                    if (lane == 0) {
                        if (D.two_keyframes)
                            match[kf_idx] = (int32_t)(idxF | ((uint32_t)bin << 24));  // :631
                        else
                            match[idxF] = (int32_t)(kf_idx | ((uint32_t)bin << 24));  // :232
                        atomicAdd(&s_nm, 1);
                        if (D.check_orientation) atomicAdd(&s_hist[bin], 1);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (D.check_orientation && tid == 0) rot_three_maxima(s_hist, s_keep);
    __syncthreads();
    for (int i = tid; i < cap_out; i += kBowMatchThreads) {
        const int m = __hip_atomic_load(&match[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m < 0) continue;
        const int bin = m >> 24;
        if (D.check_orientation && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {  // :276-286
            match[i] = -1;
            atomicSub(&s_nm, 1);
        } else {
            match[i] = m & 0xFFFFFF;
        }
    }
    __syncthreads();
    if (tid == 0) D.n_matches[f] = s_nm;
}

// ---- ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852, CheckDistEpipolarLine :140-157) ----
// The matcher of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:329): between the features of two keyframes
// that hold no map point yet, node by node, the smallest descriptor distance <= TH_LOW among the candidates far
// enough from the epipole (when neither feature has a stereo measurement) and close enough to the epipolar line.
// The reference never marks a feature of the second keyframe as matched (vbMatched2 is only initialised, :706), so
// every feature of the first keyframe decides on its own: "dist > bestDist -> continue" with bestDist lowered only
// by accepted candidates leaves the accepted candidate of smallest distance, the LAST of equals in list order.
struct TriDev {
    const uint32_t *fv_node_1, *fv_idx_1;
    const int32_t* fv_n_1;
    const uint8_t* desc_1;
    const orbx_kp* kps_1;      // mvKeysUn
    const uint8_t* has_mp_1;   // GetMapPoint(idx) != NULL
    const float* u_right_1;    // mvuRight, null: no feature has a stereo measurement
    int cap_1;
    const uint32_t *fv_node_2, *fv_idx_2;
    const int32_t* fv_n_2;
    const uint8_t* desc_2;
    const orbx_kp* kps_2;
    const uint8_t* has_mp_2;
    const float* u_right_2;
    int cap_2;
    const float* f12;          // [n][9], row-major
    const float* epipole;      // [n][2]: (ex, ey) in the second image (:693-699)
    float scale[ORBX_MAX_LEVELS], sigma2[ORBX_MAX_LEVELS];  // pKF2->mvScaleFactors, mvLevelSigma2
    int only_stereo, check_orientation;
    int32_t* match12;          // [n][cap_1]
    int32_t* n_matches;        // [n]
};

// grid = pairs, block = 1024: 64 groups of sixteen lanes, a group per feature of the first keyframe's FeatureVector.
__global__ __launch_bounds__(kBowMatchThreads, 8) void k_search_triangulation(TriDev D) {
    __shared__ int s_hist[kHistoLength], s_keep[3], s_nm;
    const int f = blockIdx.x, tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int cap_1 = D.cap_1, cap_2 = D.cap_2;
    const int n1 = max(0, min(D.fv_n_1[f], cap_1)), n2 = max(0, min(D.fv_n_2[f], cap_2));
    const uint32_t* node1 = D.fv_node_1 + (size_t)f * cap_1;
    const uint32_t* idx1s = D.fv_idx_1 + (size_t)f * cap_1;
    const uint32_t* node2 = D.fv_node_2 + (size_t)f * cap_2;
    const uint32_t* idx2s = D.fv_idx_2 + (size_t)f * cap_2;
    const orbx_kp* kps1 = D.kps_1 + (size_t)f * cap_1;
    const orbx_kp* kps2 = D.kps_2 + (size_t)f * cap_2;
    const uint8_t* mp1 = D.has_mp_1 + (size_t)f * cap_1;
    const uint8_t* mp2 = D.has_mp_2 + (size_t)f * cap_2;
    const float* ur1 = D.u_right_1 ? D.u_right_1 + (size_t)f * cap_1 : nullptr;
    const float* ur2 = D.u_right_2 ? D.u_right_2 + (size_t)f * cap_2 : nullptr;
    int32_t* match12 = D.match12 + (size_t)f * cap_1;
    float F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = D.f12[(size_t)f * 9 + i];
    const float ex = D.epipole[(size_t)f * 2], ey = D.epipole[(size_t)f * 2 + 1];
    for (int i = tid; i < cap_1; i += kBowMatchThreads) match12[i] = -1;  // :707
    if (tid < kHistoLength) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();

    for (int e = grp; e < n1; e += kBowMatchThreads / 16) {
        const uint32_t node = node1[e];
        const int i1 = (int)idx1s[e];
        if (mp1[i1]) continue;  // :728-732; the sixteen lanes of a group decide alike
        const bool stereo1 = ur1 && ur1[i1] >= 0.0f;  // :734
        if (D.only_stereo && !stereo1) continue;      // :736-738
        // the second keyframe's run of this node: [lo, hi)
        int lo = 0, hi = n2;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (node2[mid] < node) lo = mid + 1; else hi = mid;
        }
        int up = lo, top = n2;
        while (up < top) {
            const int mid = (up + top) >> 1;
            if (node2[mid] <= node) up = mid + 1; else top = mid;
        }
        const orbx_kp k1 = kps1[i1];
        // epipolar line in the second image, l = x1' F12 = [a b c] (:142-145)
        const float a = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[0]), __fmul_rn(k1.y, F[3])), F[6]);
        const float b = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[1]), __fmul_rn(k1.y, F[4])), F[7]);
        const float c = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[2]), __fmul_rn(k1.y, F[5])), F[8]);
        const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
        const uint4 d0 = *reinterpret_cast<const uint4*>(D.desc_1 + ((size_t)f * cap_1 + i1) * 32);
        const uint4 d1 = *reinterpret_cast<const uint4*>(D.desc_1 + ((size_t)f * cap_1 + i1) * 32 + 16);
        uint32_t key = 0xFFFFFFFFu;  // distance << 16 | (0xFFFF - list position): smallest distance, last of equals
        int best = -1;
        for (int j = lo + sub; j < up; j += 16) {
            const int i2 = (int)idx2s[j];
            if (mp2[i2]) continue;  // :751-755
            const bool stereo2 = ur2 && ur2[i2] >= 0.0f;  // :757
            if (D.only_stereo && !stereo2) continue;      // :759-761
            const uint4* dp = reinterpret_cast<const uint4*>(D.desc_2 + ((size_t)f * cap_2 + i2) * 32);
            const uint4 da = dp[0], db = dp[1];
            const uint32_t dist = __popc(d0.x ^ da.x) + __popc(d0.y ^ da.y) + __popc(d0.z ^ da.z) + __popc(d0.w ^ da.w) +
                                  __popc(d1.x ^ db.x) + __popc(d1.y ^ db.y) + __popc(d1.z ^ db.z) + __popc(d1.w ^ db.w);
            if (dist > (uint32_t)kThLow) continue;  // :767
            const orbx_kp k2 = kps2[i2];
            const int level2 = min(max(k2.octave, 0), ORBX_MAX_LEVELS - 1);
            if (!stereo1 && !stereo2) {  // :772-778
                const float distex = __fsub_rn(ex, k2.x), distey = __fsub_rn(ey, k2.y);
                if (__fadd_rn(__fmul_rn(distex, distex), __fmul_rn(distey, distey)) < __fmul_rn(100.0f, D.scale[level2])) continue;
            }
            if (den == 0.0f) continue;  // :151-152
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, k2.x), __fmul_rn(b, k2.y)), c);
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)D.sigma2[level2])) continue;  // :156
            const uint32_t kc = (dist << 16) | (uint32_t)(0xFFFF - (j - lo));
            if (kc < key) {
                key = kc;
                best = i2;
            }
        }
        window_min_dpp<0xB1>(key, best);
        window_min_dpp<0x4E>(key, best);
        window_min_dpp<0x141>(key, best);
        window_min_dpp<0x140>(key, best);
        if (sub == 0 && key != 0xFFFFFFFFu) {  // :787-804
            match12[i1] = best;
            atomicAdd(&s_nm, 1);
            if (D.check_orientation) atomicAdd(&s_hist[proj_rot_bin(k1.angle, kps2[best].angle)], 1);
        }
    }
    __syncthreads();
    if (D.check_orientation) {  // :820-839
        if (tid == 0) rot_three_maxima(s_hist, s_keep);
        __syncthreads();
        for (int i = tid; i < cap_1; i += kBowMatchThreads) {
            const int m = __hip_atomic_load(&match12[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m < 0) continue;
            const int bin = proj_rot_bin(kps1[i].angle, kps2[m].angle);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                match12[i] = -1;
                atomicSub(&s_nm, 1);
            }
        }
        __syncthreads();
    }
    if (tid == 0) D.n_matches[f] = s_nm;
}

}  // namespace orbx
#endif
