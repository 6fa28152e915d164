// Bag-of-words transform of a batch's descriptors (Frame::ComputeBoW, reference src/Frame.cc:411-418
// = TemplatedVocabulary::transform, reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194
// and :1218-1260) on the descriptors the extractor left in device memory.  Two launches:
//   k_bow_words    every descriptor goes down the vocabulary tree: 16 lanes share one descriptor,
//                  each takes children of the current node, the Hamming distances are reduced to
//                  the first minimum in child order;
//   k_bow_vectors  one workgroup per frame turns the (word, weight, node) triples into the
//                  BowVector (words ascending, weights of a word added in feature order, then the
//                  normalisation the scoring type asks for, in word order) and the FeatureVector
//                  ((node, feature) pairs, nodes ascending, features ascending inside a node).
// Sums and the norm are accumulated sequentially in double in the reference's order, so the
// values equal the CPU restatement's bit for bit.
#ifndef ORBX_KERNELS_BOW_HPP
#define ORBX_KERNELS_BOW_HPP

#include "orbx_device.h"

namespace orbx {

// A child as the descent reads it: its centre and where its own children are, in one 48-byte record,
// so that a level costs one dependent memory round trip.
struct VocRec {
    uint4 d0, d1;          // centre
    uint32_t id;           // node id
    uint32_t begin, end;   // records of this node's children: rec[begin .. end)
    uint32_t pad;
};

struct VocDev {
    const VocRec* rec;            // [n_nodes - 1] children of node 0, then of node 1, ... (each list in node
                                  // order, the reference's push_back order)
    uint32_t root_end;            // the root's children are rec[0 .. root_end)
    const uint32_t* word;         // [n_nodes] word id (0 unless the file flags the node as a leaf)
    const double* weight;         // [n_nodes]
    int32_t n_nodes, L;
    int32_t sums;                 // TF_IDF / TF: weights of a word are added; IDF / BINARY: first one kept
    int32_t norm;                 // 0 none (values divided by the word count when sums), 1 L1, 2 L2
};

constexpr int kBowGroup = 16;                 // lanes per descriptor
constexpr int kBowPerBlock = 256 / kBowGroup; // descriptors per workgroup

// grid = (ceil(kp_cap / kBowPerBlock), frames), block = 256.
__global__ __launch_bounds__(256) void k_bow_words(VocDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_kp,
                                                   int kp_cap, int nid_level, uint32_t* __restrict__ word,
                                                   uint32_t* __restrict__ node, double* __restrict__ weight) {
    const int f = blockIdx.y, sub = threadIdx.x & (kBowGroup - 1);
    const int i = blockIdx.x * kBowPerBlock + (threadIdx.x >> 4);
    const int n = min(n_kp[f], kp_cap);
    const bool have = i < n;
    const uint4* dp = reinterpret_cast<const uint4*>(desc + ((size_t)f * kp_cap + (have ? i : 0)) * 32);
    const uint4 d0 = dp[0], d1 = dp[1];
    uint32_t cur = 0, nid = 0;
    int level = 0;
    // a group leaves the loop when its node has no children; idle groups never enter it
    uint32_t cs = 0, ce = have ? V.root_end : 0;
    const int group_lane0 = (int)(threadIdx.x & 63u) & ~(kBowGroup - 1);
    while (__any(cs < ce)) {
        if (cs < ce) {
            uint32_t key = 0xFFFFFFFFu, b_id = 0, b_begin = 0, b_end = 0;
            for (uint32_t c = cs + sub; c < ce; c += kBowGroup) {
                const uint4* rp = reinterpret_cast<const uint4*>(V.rec + c);
                const uint4 a = rp[0], b = rp[1], m = rp[2];
                const uint32_t dist = __popc(d0.x ^ a.x) + __popc(d0.y ^ a.y) + __popc(d0.z ^ a.z) + __popc(d0.w ^ a.w) +
                                      __popc(d1.x ^ b.x) + __popc(d1.y ^ b.y) + __popc(d1.z ^ b.z) + __popc(d1.w ^ b.w);
                // strict "<" in child order (:1241-1250) == minimum of (distance, position)
                const uint32_t kc = (dist << 20) | (c - cs);
                if (kc < key) {
                    key = kc;
                    b_id = m.x;
                    b_begin = m.y;
                    b_end = m.z;
                }
            }
            uint32_t kmin = key;
#pragma unroll
            for (int d = kBowGroup / 2; d >= 1; d >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
            // position p of a list is read by lane p % 16 of the group: that lane holds the winner's record
            const int src = group_lane0 + (int)(kmin & (uint32_t)(kBowGroup - 1));
            cur = (uint32_t)__shfl((int)b_id, src, 64);
            cs = (uint32_t)__shfl((int)b_begin, src, 64);
            ce = (uint32_t)__shfl((int)b_end, src, 64);
            ++level;
            if (level == nid_level) nid = cur;  // :1252-1253
        }
    }
    if (have && sub == 0) {
        const size_t o = (size_t)f * kp_cap + i;
        word[o] = V.word[cur];
        node[o] = nid;
        weight[o] = V.weight[cur];
    }
}

// In-place bitonic sort of two arrays of P (a power of two) 64-bit keys in LDS by all threads of
// the block, stage by stage together (one barrier per stage for both); kb may be null.
__device__ __forceinline__ void bow_sort2(unsigned long long* ka, unsigned long long* kb, int P) {
    const int work = kb ? P : (P >> 1);
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < work; t += blockDim.x) {
                unsigned long long* keys = t < (P >> 1) ? ka : kb;
                const int u = t < (P >> 1) ? t : t - (P >> 1);
                const int lo = ((u & ~(j - 1)) << 1) | (u & (j - 1)), hi = lo | j;
                const unsigned long long a = keys[lo], b = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
}

// Exclusive scan of one int per thread over the block (1024 threads); *total gets the sum.
__device__ __forceinline__ int bow_block_scan(int v, int* s_part, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int base = 0, sum = 0;
    for (int k = 0; k < nw; ++k) {
        const int t = s_part[k];
        if (k < wave) base += t;
        sum += t;
    }
    *total = sum;
    return base + incl - v;
}

// grid = frames, block = 1024, dynamic LDS = (dual ? 2 : 1) * Pmax * 8 bytes with Pmax = the power of
// two >= kp_cap (at least 1024).  A frame sorts P = the power of two >= its own feature count; with
// `dual` the (word, feature) and (node, feature) keys are sorted side by side, otherwise one after
// the other in the same buffer (capacities whose two key arrays would not fit the LDS).
__global__ __launch_bounds__(1024) void k_bow_vectors(VocDev V, const int32_t* __restrict__ n_kp, int kp_cap, int Pmax, int dual,
                                                      const uint32_t* __restrict__ word, const uint32_t* __restrict__ node,
                                                      const double* __restrict__ weight, uint32_t* __restrict__ bow_id,
                                                      double* __restrict__ bow_val, int32_t* __restrict__ bow_n,
                                                      uint32_t* __restrict__ fv_node, uint32_t* __restrict__ fv_idx,
                                                      int32_t* __restrict__ fv_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    __shared__ int s_part[16];
    __shared__ double s_norm;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_kp[f], kp_cap);
    const size_t o = (size_t)f * kp_cap;
    const unsigned long long kNone = ~0ull;
    int P = 1024;
    while (P < n) P <<= 1;
    unsigned long long* kw = s_keys;         // (word, feature) keys
    unsigned long long* kn = dual ? s_keys + Pmax : s_keys;  // (node, feature) keys

    for (int i = tid; i < P; i += 1024) {
        const bool ok = i < n && weight[o + i] > 0;  // not a stopped word (:1157)
        kw[i] = ok ? (((unsigned long long)word[o + i] << 16) | (unsigned)i) : kNone;
        if (dual) kn[i] = ok ? (((unsigned long long)node[o + i] << 16) | (unsigned)i) : kNone;
    }
    __syncthreads();
    bow_sort2(kw, dual ? kn : nullptr, P);

    // ---- BowVector: a thread owns P / 1024 consecutive sorted positions; heads of runs of one word
    // become entries ----
    const int per = P >> 10, j0 = tid * per;
    int heads = 0, valid = 0;
    for (int j = j0; j < j0 + per; ++j) {
        const unsigned long long kj = kw[j];
        if (kj == kNone) continue;
        ++valid;
        if (j == 0 || (kw[j - 1] >> 16) != (kj >> 16)) ++heads;
    }
    int n_bow, n_valid;
    int pos = bow_block_scan(heads, s_part, &n_bow);
    (void)bow_block_scan(valid, s_part, &n_valid);
    for (int j = j0; j < j0 + per; ++j) {
        const unsigned long long kj = kw[j];
        if (kj == kNone) continue;
        if (j == 0 || (kw[j - 1] >> 16) != (kj >> 16)) {
            // BowVector::addWeight (BowVector.cpp:31-44): the weights of a word are added in feature order
            double acc = weight[o + (kj & 0xFFFFu)];
            if (V.sums)
                for (int t = j + 1; t < P && (kw[t] >> 16) == (kj >> 16); ++t) acc = __dadd_rn(acc, weight[o + (kw[t] & 0xFFFFu)]);
            bow_id[o + pos] = (uint32_t)(kj >> 16);
            bow_val[o + pos] = acc;
            ++pos;
        }
    }
    // ---- FeatureVector ----
    if (!dual) {
        __syncthreads();  // every thread is done with the word keys
        for (int i = tid; i < P; i += 1024)
            kn[i] = (i < n && weight[o + i] > 0) ? (((unsigned long long)node[o + i] << 16) | (unsigned)i) : kNone;
        __syncthreads();
        bow_sort2(kn, nullptr, P);
    }
    for (int j = tid; j < n_valid; j += 1024) {
        fv_node[o + j] = (uint32_t)(kn[j] >> 16);
        fv_idx[o + j] = (uint32_t)(kn[j] & 0xFFFFu);
    }
    __threadfence_block();
    __syncthreads();  // the entries written above are read back below by other threads of the block
    // The norm is a sum in word order (BowVector::normalize, BowVector.cpp:62-84), so one thread adds
    // the values one after the other; it reads them from LDS (the key arrays are free now), where a
    // dependent chain of 2000 loads costs microseconds instead of a third of a millisecond.
    double* s_val = reinterpret_cast<double*>(s_keys);
    for (int t = tid; t < n_bow; t += 1024) s_val[t] = bow_val[o + t];
    __syncthreads();
    if (tid == 0) {
        bow_n[f] = n_bow;
        fv_n[f] = n_valid;
        double norm = 0.0;
        if (V.norm == 1) {
            for (int t = 0; t < n_bow; ++t) norm = __dadd_rn(norm, fabs(s_val[t]));
        } else if (V.norm == 2) {
            for (int t = 0; t < n_bow; ++t) norm = __dadd_rn(norm, __dmul_rn(s_val[t], s_val[t]));
            norm = __dsqrt_rn(norm);
        } else {
            norm = (V.sums && n_bow > 0) ? (double)n_bow : 0.0;  // :1164-1170
        }
        s_norm = norm;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int t = tid; t < n_bow; t += 1024) bow_val[o + t] = __ddiv_rn(s_val[t], norm);
}


// ---- best match inside a window ----------------------------------------------------------------
// Frame::GetFeaturesInArea (reference src/Frame.cc:338-396) + the candidate loop of
// ORBmatcher::SearchByProjection (reference src/ORBmatcher.cc:1427-1455) on the undistorted
// keypoints, the CSR grid and the descriptors of a batch in device memory.
struct WindowQuery {     // one projected map point
    float u, v, radius;  // window centre and half-size
    int32_t min_level, max_level;
    float ur;            // predicted right coordinate (compared where the feature has one)
    uint8_t desc[32];    // the map point's descriptor
};
static_assert(sizeof(WindowQuery) == 56, "WindowQuery layout");

struct WindowDev {
    const orbx_kp* kps_un;       // [n][kp_cap]
    const int32_t* cell_start;   // [n][64 * 48 + 1]
    const int32_t* cell_idx;     // [n][kp_cap]
    const uint8_t* desc;         // [n][kp_cap][32]
    const float* u_right;        // [n][kp_cap] or null
    const uint8_t* taken;        // [n][kp_cap] or null
    float min_x, min_y, inv_w, inv_h;
    int32_t kp_cap, q_cap;
};

// Sixteen lanes share one query.  The grid is stored column by column (cell = column * 48 + row), so
// the cells a window covers in one grid column are one contiguous run of cell_idx, and the runs of
// the window's columns, one after the other, are exactly the order in which GetFeaturesInArea
// visits the features.  The lanes read the runs' bounds (one column per lane, sixteen at a time),
// number the entries of all runs consecutively and take entries s, s + 16, ... -- every lane has
// the same amount of work whatever the cells hold.  The first strict minimum in visiting order is
// the minimum of (distance, entry number).  All sixteen lanes of a group call this together; each
// receives the winner (-1: none) and its key (distance in bits 23..31, 0xFFFFFFFF: none).
// held(i2, keypoint) tells whether the query must pass over feature i2 (it holds a map point the query respects,
// or it fails a test of the caller's);
// on_cand(i2, dist, key, keypoint) runs for every candidate that passes the filters.
template <int CTRL>
__device__ __forceinline__ void window_min_dpp(uint32_t& k, int& v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, CTRL, 0xF, 0xF, true);
    const int w = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
    const bool take = o < k;
    k = take ? o : k;
    v = take ? w : v;
}

template <class Held, class OnCand>
__device__ __forceinline__ void window_best(const WindowDev& W, int f, float x, float y, float r, float ur, int minLevel,
                                            int maxLevel, const uint4& d0, const uint4& d1, int sub, Held held,
                                            OnCand on_cand, int* win_out, uint32_t* key_out) {
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, W.min_x), r), W.inv_w)));  // :343
    const int nMaxCellX = min(ORBX_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, W.min_x), r), W.inv_w)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, W.min_y), r), W.inv_h)));
    const int nMaxCellY = min(ORBX_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, W.min_y), r), W.inv_h)));
    uint32_t key = 0xFFFFFFFFu;
    int mine = -1;
    const int lane = (int)(threadIdx.x & 63u), gbase = lane & ~15;
    if (nMinCellX < ORBX_GRID_COLS && nMaxCellX >= 0 && nMinCellY < ORBX_GRID_ROWS && nMaxCellY >= 0 &&
        nMinCellX <= nMaxCellX && nMinCellY <= nMaxCellY) {
        const bool check_levels = minLevel > 0 || maxLevel >= 0;  // :359
        const int ncx = nMaxCellX - nMinCellX + 1;
        const orbx_kp* kps = W.kps_un + (size_t)f * W.kp_cap;
        const int32_t* cs = W.cell_start + (size_t)f * (ORBX_GRID_COLS * ORBX_GRID_ROWS + 1);
        const int32_t* ci = W.cell_idx + (size_t)f * W.kp_cap;
        uint32_t order0 = 0;  // entries of the columns handled so far
        for (int cx0 = 0; cx0 < ncx; cx0 += 16) {
            // lane s: the run of column cx0 + s
            int a = 0, len = 0;
            if (cx0 + sub < ncx) {
                const int c0 = (nMinCellX + cx0 + sub) * ORBX_GRID_ROWS;
                a = cs[c0 + nMinCellY];
                len = cs[c0 + nMaxCellY + 1] - a;
            }
            // inclusive prefix over the group's sixteen lanes (one DPP row: shifts by 1, 2, 4, 8, zeros shifted in)
            int incl = len;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
            const int total = __shfl(incl, gbase + 15, 64);
            const int shift = a - (incl - len);  // entry e of this lane's column is cell_idx[e + shift]
            // the sixteen lanes stay together through this loop: the shuffles read lanes of the whole group
            for (int e0 = 0; e0 < total; e0 += 16) {
                const int e = e0 + sub;
                // the column of entry e: the first lane whose inclusive prefix exceeds e
                int k = 0;
                k += __shfl(incl, gbase + k + 7, 64) <= e ? 8 : 0;
                k += __shfl(incl, gbase + k + 3, 64) <= e ? 4 : 0;
                k += __shfl(incl, gbase + k + 1, 64) <= e ? 2 : 0;
                k += __shfl(incl, gbase + k, 64) <= e ? 1 : 0;
                const int sh = __shfl(shift, gbase + min(k, 15), 64);
                if (e < total) {
                    const int i2 = ci[e + sh];
                    const orbx_kp kp = kps[i2];
                    bool ok = !check_levels || (kp.octave >= minLevel && (maxLevel < 0 || kp.octave <= maxLevel));
                    ok = ok && fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r;  // :389
                    ok = ok && !held(i2, kp);                                                     // ORBmatcher.cc:1440-1442
                    if (ok && W.u_right) {
                        const float urf = W.u_right[(size_t)f * W.kp_cap + i2];
                        if (urf > 0 && fabsf(__fsub_rn(ur, urf)) > r) ok = false;  // :1444-1450
                    }
                    if (ok) {
                        const uint4* dp = reinterpret_cast<const uint4*>(W.desc + ((size_t)f * W.kp_cap + i2) * 32);
                        const uint4 da = dp[0], db = dp[1];
                        const uint32_t dist = __popc(d0.x ^ da.x) + __popc(d0.y ^ da.y) + __popc(d0.z ^ da.z) +
                                              __popc(d0.w ^ da.w) + __popc(d1.x ^ db.x) + __popc(d1.y ^ db.y) +
                                              __popc(d1.z ^ db.z) + __popc(d1.w ^ db.w);
                        const uint32_t kc = (dist << 23) | (order0 + (uint32_t)e);
                        on_cand(i2, dist, kc, kp);
                        if (dist < 256u && kc < key) {  // bestDist starts at 256, strict "<" (:1456)
                            key = kc;
                            mine = i2;
                        }
                    }
                }
            }
            order0 += (uint32_t)total;
        }
    }
    // the group's smallest key with its feature, through the row's DPP network (keys are unique: the entry
    // number is part of them; lanes without a candidate hold 0xFFFFFFFF and -1)
    uint32_t kmin = key;
    int win = mine;
    window_min_dpp<0xB1>(kmin, win);   // quad_perm [1,0,3,2]
    window_min_dpp<0x4E>(kmin, win);   // quad_perm [2,3,0,1]
    window_min_dpp<0x141>(kmin, win);  // row_half_mirror
    window_min_dpp<0x140>(kmin, win);  // row_mirror
    *win_out = kmin == 0xFFFFFFFFu ? -1 : win;
    *key_out = kmin;
}

// grid = (ceil(q_cap / 16), frames), block = 256.
__global__ __launch_bounds__(256) void k_search_window(WindowDev W, const WindowQuery* __restrict__ queries,
                                                       const int32_t* __restrict__ n_q, int32_t* __restrict__ best_idx,
                                                       int32_t* __restrict__ best_dist) {
    const int f = blockIdx.y, sub = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (q >= min(n_q[f], W.q_cap)) return;  // whole groups leave together
    const WindowQuery* Q = queries + (size_t)f * W.q_cap + q;
    const uint4 d0 = *reinterpret_cast<const uint4_unaligned*>(Q->desc), d1 = *reinterpret_cast<const uint4_unaligned*>(Q->desc + 16);
    int win;
    uint32_t kmin;
    const uint8_t* taken = W.taken ? W.taken + (size_t)f * W.kp_cap : nullptr;
    window_best(W, f, Q->u, Q->v, Q->radius, Q->ur, Q->min_level, Q->max_level, d0, d1, sub,
                [&](int i2, const orbx_kp&) { return taken && taken[i2]; }, [](int, uint32_t, uint32_t, const orbx_kp&) {}, &win, &kmin);
    if (sub == 0) {
        const size_t o = (size_t)f * W.q_cap + q;
        best_idx[o] = win;
        best_dist[o] = kmin == 0xFFFFFFFFu ? 256 : (int)(kmin >> 23);
    }
}

// ---- ORBVocabulary::score for the L1 norm ----------------------------------------------------
// DBoW2 L1Scoring::score (reference Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68; called per candidate keyframe
// from KeyFrameDatabase::DetectLoopCandidates, src/KeyFrameDatabase.cc:133, DetectRelocalizationCandidates, :249,
// and LoopClosing::DetectLoop, src/LoopClosing.cc:137): the sum of |vi - wi| - |vi| - |wi| over the words both
// BowVectors hold, accumulated in ascending word order, then -sum / 2.  One wave per pair: the lanes look their
// 64 entries of the first vector up in the second (binary search), and the terms of the words found are added
// one after the other in lane order, which is the reference's order.
struct BowScoreDev {
    const uint32_t *id_1, *id_2;   // [m_k][cap_k] ascending word ids (what the bag-of-words transform leaves)
    const double *val_1, *val_2;   // [m_k][cap_k]
    const int32_t *n_1, *n_2;      // [m_k]
    int32_t cap_1, cap_2, n_pairs;
    const int32_t* pairs;          // [n_pairs][2]: vector of the first set, vector of the second set
    double* score;                 // [n_pairs]
};

// grid = ceil(n_pairs / 4), block = 256 (a wave per pair).
__global__ __launch_bounds__(256) void k_bow_score_l1(BowScoreDev D) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= D.n_pairs) return;  // whole waves leave together
    const int i = D.pairs[2 * p], j = D.pairs[2 * p + 1];
    const int n1 = max(0, min(D.n_1[i], D.cap_1)), n2 = max(0, min(D.n_2[j], D.cap_2));
    const uint32_t* id1 = D.id_1 + (size_t)i * D.cap_1;
    const uint32_t* id2 = D.id_2 + (size_t)j * D.cap_2;
    const double* v1 = D.val_1 + (size_t)i * D.cap_1;
    const double* v2 = D.val_2 + (size_t)j * D.cap_2;
    double score = 0.0;
    for (int c = 0; c < n1; c += 64) {
        const int e = c + lane;
        double term = 0.0;
        bool common = false;
        if (e < n1) {
            const uint32_t w = id1[e];
            int lo = 0, hi = n2;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (id2[mid] < w) lo = mid + 1; else hi = mid;
            }
            if (lo < n2 && id2[lo] == w) {
                const double vi = v1[e], wi = v2[lo];
                term = __dsub_rn(__dsub_rn(fabs(__dsub_rn(vi, wi)), fabs(vi)), fabs(wi));  // :41
                common = true;
            }
        }
        unsigned long long m = __ballot(common);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            score = __dadd_rn(score, __shfl(term, b, 64));
        }
    }
    if (lane == 0) D.score[p] = __ddiv_rn(-score, 2.0);  // :65
}

// ---- the search of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) --------------------
// (reference src/ORBmatcher.cc:929-986 with KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:575-614): a projected map
// point looks for the most similar keypoint of the keyframe in its window, on the predicted level or the one
// below (the query's level range), among the keypoints whose reprojection error passes the chi-square test --
// (ex^2 + ey^2) * mvInvLevelSigma2[level] <= 5.99, with a stereo measurement (ex^2 + ey^2 + er^2) * ... <= 7.8.
// The searches of different map points do not depend on each other; what the reference does with the result
// (replace / add an observation, :989-1008) changes the map, not the search, and stays with the caller.
struct FuseGate {
    const float* u_right;                  // [n][kp_cap] (mvuRight) or null: no keypoint has a stereo measurement
    float inv_sigma2[ORBX_MAX_LEVELS];     // mvInvLevelSigma2
};

// grid = (ceil(q_cap / 16), keyframes), block = 256.  W.u_right and W.taken are null.
__global__ __launch_bounds__(256) void k_fuse_search(WindowDev W, FuseGate G, const WindowQuery* __restrict__ queries,
                                                     const int32_t* __restrict__ n_q, int32_t* __restrict__ best_idx,
                                                     int32_t* __restrict__ best_dist) {
    const int f = blockIdx.y, sub = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (q >= min(n_q[f], W.q_cap)) return;  // whole groups leave together
    const WindowQuery* Q = queries + (size_t)f * W.q_cap + q;
    const uint4 d0 = *reinterpret_cast<const uint4_unaligned*>(Q->desc), d1 = *reinterpret_cast<const uint4_unaligned*>(Q->desc + 16);
    const float u = Q->u, v = Q->v, ur = Q->ur;
    const float* u_right = G.u_right ? G.u_right + (size_t)f * W.kp_cap : nullptr;
    int win;
    uint32_t kmin;
    window_best(W, f, u, v, Q->radius, ur, Q->min_level, Q->max_level, d0, d1, sub,
                [&](int i2, const orbx_kp& kp) {
                    const float inv = G.inv_sigma2[min(max(kp.octave, 0), ORBX_MAX_LEVELS - 1)];
                    const float ex = __fsub_rn(u, kp.x), ey = __fsub_rn(v, kp.y);
                    const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    const float kpr = u_right ? u_right[i2] : -1.0f;
                    if (kpr >= 0.0f) {  // :951-964
                        const float er = __fsub_rn(ur, kpr);
                        return (double)__fmul_rn(__fadd_rn(e2, __fmul_rn(er, er)), inv) > 7.8;
                    }
                    return (double)__fmul_rn(e2, inv) > 5.99;  // :965-975
                },
                [](int, uint32_t, uint32_t, const orbx_kp&) {}, &win, &kmin);
    if (sub == 0) {
        const size_t o = (size_t)f * W.q_cap + q;
        best_idx[o] = win;
        best_dist[o] = kmin == 0xFFFFFFFFu ? 256 : (int)(kmin >> 23);
    }
}

// ---- ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) as a whole -----------------
// (reference src/ORBmatcher.cc:1365-1507).  The reference walks the last frame's map points in
// order and writes each match into the current frame at once, so a later point skips the features
// that earlier points with observations hold.  Here all undecided points search at the same time,
// in rounds.  A feature records the index of the point that holds it, and a point respects only
// holders with a lower index than its own.  A point's choice stands as soon as no undecided
// earlier point with observations still has that feature among its acceptable candidates
// (distance <= TH_HIGH) -- then nothing the earlier points can still do changes what this point
// sees.  The lowest undecided point always qualifies, so the rounds end, with the result of the
// sequential loop.  Only the first round walks the windows: it leaves each point's acceptable
// candidates (at most kProjCache, with their preference keys) in scratch memory, and the later
// rounds -- one thread per undecided point -- pick the best of those not yet held by an earlier
// point.  A point with more acceptable candidates than the cache holds searches its window again
// in every round.
struct MapQuery {        // feature i of the last frame
    float x, y, z;       // world position of its map point
    int32_t octave;      // mvKeys[i].octave
    float angle;         // mvKeysUn[i].angle
    int32_t flags;       // bit 0: the map point has observations (it holds what it takes); bit 1: skip
    uint8_t desc[32];    // the map point's descriptor
};
static_assert(sizeof(MapQuery) == 56, "MapQuery layout");

struct ProjParams {
    float fx, fy, cx, cy, mbf, mb;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
    float th;
    int32_t mono, check_orientation;
};

struct ProjDev {
    WindowDev W;             // taken unused
    ProjParams P;
    float scale[ORBX_MAX_LEVELS];
    const float* poses;      // [n][24]: Tcw then Tlw, rows of [R|t]
    const MapQuery* queries; // [n][q_cap]
    const int32_t* n_q;      // [n]
    const uint8_t* taken0;   // [n][kp_cap] or null
    int32_t* match;          // [n][kp_cap]
    int32_t* query_match;    // [n][q_cap]
    int32_t* n_matches;      // [n]
    int32_t* minq;           // scratch [n][kp_cap]: lowest undecided holder that may still take the feature
    int32_t* list;           // scratch [n][2][q_cap]: the undecided points of this round / of the next
    uint2* cache;            // scratch [n][q_cap][kProjCache]: (preference key, feature) of the acceptable candidates
    int32_t* n_cache;        // scratch [n][q_cap]: how many there are (more than kProjCache: not all are cached)
    int32_t use_lds;         // minq, held and list live in the workgroup's dynamic LDS (2 * kp_cap + 2 * q_cap ints)
    int32_t* held;           // scratch [n][kp_cap]: index of the holding point (-1: before the call, INT_MAX: free)
};

constexpr int kProjThreads = 1024;
constexpr int kProjCache = 8;
constexpr int kProjLocal = 2;  // list entries per thread whose choice stays in a register between the phases of a round
constexpr int kHistoLength = 30, kThHigh = 100;  // ORBmatcher.cc:37-39

__device__ __forceinline__ int proj_rot_bin(float last_angle, float cur_angle) {
    float rot = __fsub_rn(last_angle, cur_angle);  // :1470-1475
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHistoLength));
    if (bin == kHistoLength) bin = 0;
    return min(max(bin, 0), kHistoLength - 1);  // angles outside [0, 360) would leave the table in the reference
}

// ComputeThreeMaxima (ORBmatcher.cc:1638-1679) over bin sizes: the bins that stay, -1 where fewer than three do.
__device__ __forceinline__ void rot_three_maxima(const int* hist, int* keep) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < kHistoLength; ++i) {
        const int s = hist[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        ind3 = -1;
    }
    keep[0] = ind1;
    keep[1] = ind2;
    keep[2] = ind3;
}

// The best cached candidate of a listed point that no earlier point holds (-1: none); a holder also
// marks every such candidate as still wanted by it.
__device__ __forceinline__ int proj_cached_choice(uint32_t ent, const uint2* cache, const int32_t* n_cache, const int32_t* held,
                                                  int32_t* minq) {
    const int q = (int)(ent & 0x3FFFFFFFu), nc = n_cache[q];
    const bool holds = (ent >> 30) & 1u;
    uint32_t best = 0xFFFFFFFFu;
    int choice = -1;
    // all cached candidates are requested before the first is looked at: one memory round trip
    uint4 cc[kProjCache / 2];
    const uint4* cp = reinterpret_cast<const uint4*>(cache + (size_t)q * kProjCache);
#pragma unroll
    for (int k = 0; k < kProjCache / 2; ++k) cc[k] = 2 * k < nc ? cp[k] : make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
#pragma unroll
    for (int k = 0; k < kProjCache; ++k) {
        const uint32_t ckey = (k & 1) ? cc[k / 2].z : cc[k / 2].x, cidx = (k & 1) ? cc[k / 2].w : cc[k / 2].y;
        if (k >= nc || held[cidx] < q) continue;
        if (holds) atomicMin(&minq[cidx], q);
        if (ckey < best) {
            best = ckey;
            choice = (int)cidx;
        }
    }
    return choice;
}

// grid = frames, block = 1024 (64 groups of sixteen lanes).
__global__ __launch_bounds__(kProjThreads, 8) void k_search_projection(ProjDev D) {
    __shared__ float s_T[12], s_scale[ORBX_MAX_LEVELS];  // the scale table goes through LDS: a per-lane index into a kernel argument would put it in scratch memory
    __shared__ int s_dir, s_left, s_nm, s_hist[kHistoLength], s_keep[3], s_cnt[kProjThreads / 16];
    const int f = blockIdx.x, tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int kp_cap = D.W.kp_cap, q_cap = D.W.q_cap;
    const int nq = min(D.n_q[f], q_cap);
    const MapQuery* Q = D.queries + (size_t)f * q_cap;
    int32_t* qm = D.query_match + (size_t)f * q_cap;
    int32_t* match = D.match + (size_t)f * kp_cap;
    // per-round state: in LDS when the frame's capacities fit, else in global scratch
    extern __shared__ int32_t s_dyn[];
    int32_t* minq = D.use_lds ? s_dyn : D.minq + (size_t)f * kp_cap;
    int32_t* held = D.use_lds ? s_dyn + kp_cap : D.held + (size_t)f * kp_cap;
    int32_t* lists = D.use_lds ? s_dyn + 2 * kp_cap : D.list + (size_t)f * 2 * q_cap;
    uint2* cache = D.cache + (size_t)f * q_cap * kProjCache;
    int32_t* n_cache = D.n_cache + (size_t)f * q_cap;
    const ProjParams& P = D.P;

    for (int i = tid; i < kp_cap; i += kProjThreads) {
        held[i] = (D.taken0 && D.taken0[(size_t)f * kp_cap + i]) ? -1 : 0x7FFFFFFF;
        match[i] = -1;
    }
    for (int q = tid; q < q_cap; q += kProjThreads) qm[q] = -1;
    if (tid < 12) s_T[tid] = D.poses[(size_t)f * 24 + tid];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < ORBX_MAX_LEVELS; ++i) s_scale[i] = D.scale[i];
    }
    if (tid < kHistoLength) s_hist[tid] = 0;
    if (tid < kProjThreads / 16) s_cnt[tid] = 0;
    if (tid == 0) {
        const float* Tc = D.poses + (size_t)f * 24;
        const float* Tl = Tc + 12;
        // twc = -Rcw.t()*tcw (:1378): sums in double, scaled by -1, rounded to float
        float twc[3];
        for (int k = 0; k < 3; ++k) {
            double s = 0.0;
            for (int j = 0; j < 3; ++j) s = __dadd_rn(s, __dmul_rn((double)Tc[4 * j + k], (double)Tc[4 * j + 3]));
            twc[k] = (float)__dmul_rn(s, -1.0);
        }
        // tlc.z of tlc = Rlw*twc + tlw (:1383): float products summed left to right, then the addend
        const float t = __fadd_rn(__fadd_rn(__fmul_rn(Tl[8], twc[0]), __fmul_rn(Tl[9], twc[1])), __fmul_rn(Tl[10], twc[2]));
        const float tz = __fadd_rn(t, Tl[11]);
        s_dir = P.mono ? 0 : (tz > P.mb ? 1 : (-tz > P.mb ? 2 : 0));  // 1 forward, 2 backward (:1385-1386)
        s_nm = 0;
    }
    __syncthreads();
    const int dir = s_dir;

    // The undecided points of a round are a dense list (in the first one: all of them), so that the sixteen-lane
    // groups share the searches of the later, sparser rounds evenly.
    int n_list = nq;
    for (int round = 0; round <= nq && n_list > 0; ++round) {  // every round decides at least the lowest undecided point
        const int32_t* cur = lists + (size_t)(round & 1) * q_cap;
        int32_t* next = lists + (size_t)((round + 1) & 1) * q_cap;
        for (int i = tid; i < kp_cap; i += kProjThreads) minq[i] = 0x7FFFFFFF;
        if (tid == 0) s_left = 0;
        __syncthreads();
        for (int e = grp; e < n_list; e += kProjThreads / 16) {
            const uint32_t ent = round == 0 ? (uint32_t)e : (uint32_t)cur[e];
            if (round > 0 && !(ent >> 31)) continue;  // decided from the cached candidates below
            const int q = (int)(ent & 0x3FFFFFFFu);
            const MapQuery* m = Q + q;
            int choice = -1;
            if (!(m->flags & 2)) {
                float c3[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {  // x3Dc = Rcw*x3Dw + tcw (:1398)
                    const float t = __fadd_rn(__fadd_rn(__fmul_rn(s_T[4 * r], m->x), __fmul_rn(s_T[4 * r + 1], m->y)),
                                              __fmul_rn(s_T[4 * r + 2], m->z));
                    c3[r] = __fadd_rn(t, s_T[4 * r + 3]);
                }
                const float invzc = (float)__ddiv_rn(1.0, (double)c3[2]);  // :1402
                const float u = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, c3[0]), invzc), P.cx);
                const float v = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, c3[1]), invzc), P.cy);
                if (!(invzc < 0) && !(u < P.min_x || u > P.max_x) && !(v < P.min_y || v > P.max_y)) {
                    const int oct = min(max(m->octave, 0), ORBX_MAX_LEVELS - 1);
                    const float radius = __fmul_rn(P.th, s_scale[oct]);  // :1418
                    const int lo = dir == 1 ? m->octave : (dir == 2 ? 0 : m->octave - 1);
                    const int hi = dir == 1 ? -1 : (dir == 2 ? m->octave : m->octave + 1);
                    const float ur = __fsub_rn(u, __fmul_rn(P.mbf, invzc));  // :1446
                    const uint4 d0 = *reinterpret_cast<const uint4_unaligned*>(m->desc);
                    const uint4 d1 = *reinterpret_cast<const uint4_unaligned*>(m->desc + 16);
                    const bool holds = m->flags & 1;
                    int win;
                    uint32_t key;
                    window_best(D.W, f, u, v, radius, ur, lo, hi, d0, d1, sub, [&](int i2, const orbx_kp&) { return held[i2] < q; },
                                [&](int i2, uint32_t dist, uint32_t kc, const orbx_kp&) {
                                    if (dist > (uint32_t)kThHigh) return;
                                    if (holds) atomicMin(&minq[i2], q);
                                    if (round == 0) {
                                        const int slot = atomicAdd(&s_cnt[grp], 1);
                                        if (slot < kProjCache) cache[(size_t)q * kProjCache + slot] = make_uint2(kc, (uint32_t)i2);
                                    }
                                },
                                &win, &key);
                    if (win >= 0 && (int)(key >> 23) <= kThHigh) choice = win;  // :1463
                }
            }
            if (sub == 0) {
                qm[q] = choice < 0 ? -1 : -3 - choice;  // <= -3: undecided with a choice
                if (round == 0) n_cache[q] = atomicExch(&s_cnt[grp], 0);  // after the group's additions, in wave order
            }
        }
        // later rounds, one thread per undecided point: the best cached candidate that no earlier point holds.
        // The choice stays in a register until the decision below (the first kProjLocal list entries of a
        // thread; further ones go through query_match like the searched points).
        int ch[kProjLocal];
#pragma unroll
        for (int i = 0; i < kProjLocal; ++i) ch[i] = -1;
        if (round > 0) {
#pragma unroll
            for (int i = 0; i < kProjLocal; ++i) {
                const int e = tid + i * kProjThreads;
                if (e < n_list && !((uint32_t)cur[e] >> 31)) ch[i] = proj_cached_choice((uint32_t)cur[e], cache, n_cache, held, minq);
            }
            for (int e = tid + kProjLocal * kProjThreads; e < n_list; e += kProjThreads) {
                const uint32_t ent = (uint32_t)cur[e];
                if (ent >> 31) continue;
                const int c = proj_cached_choice(ent, cache, n_cache, held, minq);
                qm[ent & 0x3FFFFFFFu] = c < 0 ? -1 : -3 - c;
            }
        }
        __syncthreads();
        for (int e = tid, i = 0; e < n_list; e += kProjThreads, ++i) {
            uint32_t ent;
            if (round == 0) {  // list entry: point | holds << 30 | "more candidates than the cache holds" << 31
                ent = (uint32_t)e | ((uint32_t)(Q[e].flags & 1) << 30) | (n_cache[e] > kProjCache ? 0x80000000u : 0u);
            } else {
                ent = (uint32_t)cur[e];
            }
            const int q = (int)(ent & 0x3FFFFFFFu);
            int win;
            if (round > 0 && !(ent >> 31) && i < kProjLocal) {
                win = ch[0];
#pragma unroll
                for (int k = 1; k < kProjLocal; ++k) win = i == k ? ch[k] : win;
                if (win < 0) {
                    qm[q] = -1;
                    continue;
                }
            } else {
                const int v = qm[q];
                if (v > -3) continue;
                win = -3 - v;
            }
            if (__hip_atomic_load(&minq[win], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= q) {
                qm[q] = win;
                if ((ent >> 30) & 1u) held[win] = q;  // the only holder this feature will ever have
            } else {
                next[atomicAdd(&s_left, 1)] = (int32_t)ent;
            }
        }
        __syncthreads();
        n_list = s_left;
        __syncthreads();  // everyone has read s_left before the next round clears it
    }

    // the feature ends with the point written last: the holder if there is one (nothing follows
    // it), else the highest index
    for (int q = tid; q < nq; q += kProjThreads) {
        const int v = qm[q];
        if (v < 0) continue;
        atomicMax(&match[v], (int)(((uint32_t)(Q[q].flags & 1) << 30) | (uint32_t)q));
        atomicAdd(&s_nm, 1);
        if (P.check_orientation) atomicAdd(&s_hist[proj_rot_bin(Q[q].angle, D.W.kps_un[(size_t)f * kp_cap + v].angle)], 1);
    }
    __syncthreads();
    if (P.check_orientation) {
        if (tid == 0) rot_three_maxima(s_hist, s_keep);
        __syncthreads();
        for (int q = tid; q < nq; q += kProjThreads) {
            const int v = qm[q];
            if (v < 0) continue;
            const int bin = proj_rot_bin(Q[q].angle, D.W.kps_un[(size_t)f * kp_cap + v].angle);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {  // :1493-1502
                held[v] = -2;  // who held it no longer matters
                atomicSub(&s_nm, 1);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < kp_cap; i += kProjThreads) {
        const int m = __hip_atomic_load(&match[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        match[i] = (m < 0 || held[i] == -2) ? -1 : (m & 0x3FFFFFFF);
    }
    if (tid == 0) D.n_matches[f] = s_nm;
}

// ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th): the local-map search ----
// (reference src/ORBmatcher.cc:45-129, called from Tracking::SearchLocalPoints).  Every map point in view
// comes with its projection (Frame::isInFrustum left it in the MapPoint); the function searches the window
// of each, keeps the best and the second best distance, applies the ratio test when both lie on one pyramid
// level, and writes the match into the frame at once, so that later points skip features held by a point
// with observations.  That loop-carried dependence is kept exactly with the rounds of k_search_projection:
// all undecided points search at the same time; a feature records the index of the point that holds it, a
// point respects only holders below itself, and holders mark every feature they may still take (distance
// <= TH_HIGH) with the lowest such index.  A point's outcome stands when no undecided holder below it may
// still take its best or its second best feature: taking any other feature away changes neither of the two
// and so neither the threshold test nor the ratio test.  The lowest undecided point always qualifies.
// The reference's update rule (:102-114) leaves in (bestDist2, bestLevel2) the candidate with the second
// smallest (distance, visiting order): sixteen lanes keep their two smallest keys and merge them.
struct LocalQuery {      // a map point of the local map
    float u, v, ur;      // mTrackProjX, mTrackProjY, mTrackProjXR
    int32_t level;       // mnTrackScaleLevel
    float view_cos;      // mTrackViewCos
    int32_t flags;       // bit 0: the map point has observations (it holds what it takes); bit 1: skip
    uint8_t desc[32];    // the map point's descriptor
};
static_assert(sizeof(LocalQuery) == 56, "LocalQuery layout");

struct LocalDev {
    WindowDev W;             // taken unused
    float scale[ORBX_MAX_LEVELS];
    float th, nn_ratio;
    const LocalQuery* queries;  // [n][q_cap]
    const int32_t* n_q;         // [n]
    const uint8_t* taken0;      // [n][kp_cap] or null
    int32_t* match;             // [n][kp_cap]
    int32_t* query_match;       // [n][q_cap]
    int32_t* n_matches;         // [n]
    int32_t* minq;              // scratch [n][kp_cap]
    int32_t* held;              // scratch [n][kp_cap]
    int32_t* list;              // scratch [n][2][q_cap]
    int2* pick;                 // scratch [n][q_cap]: (best, second best) feature of an undecided point, -1: none
    int32_t use_lds;            // minq, held and list live in the workgroup's dynamic LDS (2 * kp_cap + 2 * q_cap ints)
    // the guided searches (k_search_local<true>): accepted distance, levels above the predicted one, rotation check
    int32_t max_dist, level_up, check_orientation;
};

// Merges the two smallest keys of another lane into (k1, k2); the values ride along.
__device__ __forceinline__ void top2_merge(uint32_t& k1, uint32_t& k2, int& v1, int& v2, uint32_t o1, uint32_t o2, int w1, int w2) {
    // selects only: conditional stores through the references would keep the four values in memory
    const bool first = o1 < k1;        // the other lane's best is the new best
    const bool second_mine = k1 < o2;  // then the old best or the other lane's second comes next
    const bool middle = o1 < k2;       // else the other lane's best may still be the new second
    const uint32_t nk2 = first ? (second_mine ? k1 : o2) : (middle ? o1 : k2);
    const int nv2 = first ? (second_mine ? v1 : w2) : (middle ? w1 : v2);
    k1 = first ? o1 : k1;
    v1 = first ? w1 : v1;
    k2 = nk2;
    v2 = nv2;
}

// One step of the merge inside a row of sixteen lanes: the partner lane comes through DPP (no LDS round
// trip).  Every lane of the row must be active.
template <int CTRL>
__device__ __forceinline__ void top2_merge_dpp(uint32_t& k1, uint32_t& k2, int& v1, int& v2) {
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k1, CTRL, 0xF, 0xF, true);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k2, CTRL, 0xF, 0xF, true);
    const int w1 = __builtin_amdgcn_update_dpp(0, v1, CTRL, 0xF, 0xF, true);
    const int w2 = __builtin_amdgcn_update_dpp(0, v2, CTRL, 0xF, 0xF, true);
    top2_merge(k1, k2, v1, v2, o1, o2, w1, w2);
}

// Leaves the two smallest keys of a row of sixteen lanes (with their values) in all of its lanes.  The keys
// must be unique: every step joins two disjoint sets (lane pairs, quads, halves, the row).
__device__ __forceinline__ void top2_row_merge(uint32_t& k1, uint32_t& k2, int& v1, int& v2) {
    top2_merge_dpp<0xB1>(k1, k2, v1, v2);   // quad_perm [1,0,3,2]
    top2_merge_dpp<0x4E>(k1, k2, v1, v2);   // quad_perm [2,3,0,1]
    top2_merge_dpp<0x141>(k1, k2, v1, v2);  // row_half_mirror
    top2_merge_dpp<0x140>(k1, k2, v1, v2);  // row_mirror
}

// GUIDED: the same loop-carried dependence without the ratio test, for map points the caller has projected:
// ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (reference src/ORBmatcher.cc:1509-1636,
// relocalisation: levels predicted - 1 .. predicted + 1, distance <= ORBdist, rotation histogram) and
// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:290-403, loop closing: levels
// predicted - 1 .. predicted, distance <= TH_LOW).  Both skip features that hold a map point (:1578, :375) and
// write theirs at once (:1594, :396), so every point holds what it takes.  The query record is read as
// (u, v, angle of the map point's keypoint, predicted level, radius, flags, descriptor).
// grid = frames, block = 1024 (64 groups of sixteen lanes).
template <bool GUIDED>
__global__ __launch_bounds__(kProjThreads, 8) void k_search_local(LocalDev D) {
    __shared__ int s_left, s_nm, s_hist[kHistoLength], s_keep[3];
    __shared__ float s_scale[ORBX_MAX_LEVELS];  // through LDS: a per-lane index into a kernel argument would put the table in scratch memory
    const int f = blockIdx.x, tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int kp_cap = D.W.kp_cap, q_cap = D.W.q_cap;
    const int nq = min(D.n_q[f], q_cap);
    const LocalQuery* Q = D.queries + (size_t)f * q_cap;
    int32_t* qm = D.query_match + (size_t)f * q_cap;
    int32_t* match = D.match + (size_t)f * kp_cap;
    extern __shared__ int32_t s_dyn[];
    int32_t* minq = D.use_lds ? s_dyn : D.minq + (size_t)f * kp_cap;
    int32_t* held = D.use_lds ? s_dyn + kp_cap : D.held + (size_t)f * kp_cap;
    int32_t* lists = D.use_lds ? s_dyn + 2 * kp_cap : D.list + (size_t)f * 2 * q_cap;
    int2* pick = D.pick + (size_t)f * q_cap;
    const bool factor = (double)D.th != 1.0;  // :49

    for (int i = tid; i < kp_cap; i += kProjThreads) {
        held[i] = (D.taken0 && D.taken0[(size_t)f * kp_cap + i]) ? -1 : 0x7FFFFFFF;
        match[i] = -1;
    }
    for (int q = tid; q < q_cap; q += kProjThreads) qm[q] = -1;
    if (tid == 0) {
        s_nm = 0;
#pragma unroll
        for (int i = 0; i < ORBX_MAX_LEVELS; ++i) s_scale[i] = D.scale[i];
    }
    __syncthreads();

    int n_list = nq;
    for (int round = 0; round <= nq && n_list > 0; ++round) {  // every round decides at least the lowest undecided point
        const int32_t* cur = lists + (size_t)(round & 1) * q_cap;
        int32_t* next = lists + (size_t)((round + 1) & 1) * q_cap;
        for (int i = tid; i < kp_cap; i += kProjThreads) minq[i] = 0x7FFFFFFF;
        if (tid == 0) s_left = 0;
        __syncthreads();
        // every undecided point searches its window among the features no point below it holds
        for (int e = grp; e < n_list; e += kProjThreads / 16) {
            const int q = round == 0 ? e : cur[e];
            const LocalQuery* m = Q + q;
            int best = -1, second = -1, outcome = -1;
            if (!(m->flags & 2)) {  // :54-58
                const int level = min(max(m->level, 0), ORBX_MAX_LEVELS - 1);
                float r = (double)m->view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:131-137)
                if (factor) r = __fmul_rn(r, D.th);
                const float radius = GUIDED ? m->view_cos : __fmul_rn(r, s_scale[level]);  // :69; guided: the caller's (:1563, :360)
                const uint32_t may_take = GUIDED ? (uint32_t)D.max_dist : (uint32_t)kThHigh;
                const uint4 d0 = *reinterpret_cast<const uint4_unaligned*>(m->desc);
                const uint4 d1 = *reinterpret_cast<const uint4_unaligned*>(m->desc + 16);
                const bool holds = m->flags & 1;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                int v1 = -1, v2 = -1;  // feature | octave << 20
                int win;
                uint32_t key;
                window_best(D.W, f, m->u, m->v, radius, m->ur, m->level - 1, GUIDED ? m->level + D.level_up : m->level, d0, d1, sub,
                            [&](int i2, const orbx_kp&) { return held[i2] < q; },
                            [&](int i2, uint32_t dist, uint32_t kc, const orbx_kp& kp) {
                                if (holds && dist <= may_take) atomicMin(&minq[i2], q);
                                top2_merge(k1, k2, v1, v2, kc, 0xFFFFFFFFu, i2 | (kp.octave << 20), -1);
                            },
                            &win, &key);
                // the group's two smallest keys (keys are unique: the visiting order is part of them)
                top2_row_merge(k1, k2, v1, v2);
                if (k1 != 0xFFFFFFFFu) {
                    const int bestDist = (int)(k1 >> 23), bestLevel = v1 >> 20;
                    // a candidate at distance 256 is never recorded by the reference (dist < bestDist2 fails against the initial
                    // 256, :110-114): as a second best it leaves bestLevel2 at -1
                    const bool no2 = k2 == 0xFFFFFFFFu || (k2 >> 23) >= 256u;
                    const int bestDist2 = no2 ? 256 : (int)(k2 >> 23), bestLevel2 = no2 ? -1 : (v2 >> 20);
                    if (GUIDED) {
                        if (bestDist <= D.max_dist) outcome = best = v1 & 0xFFFFF;  // :1592, :394; the second best plays no part
                    } else if (bestDist <= kThHigh) {  // :118
                        best = v1 & 0xFFFFF;
                        second = k2 == 0xFFFFFFFFu ? -1 : (v2 & 0xFFFFF);
                        const bool rejected = bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(D.nn_ratio, (float)bestDist2);  // :120
                        outcome = rejected ? -1 : best;
                    }
                }
            }
            if (sub == 0) {
                pick[q] = make_int2(best, second);
                qm[q] = outcome;
            }
        }
        __syncthreads();
        // the outcome of a point stands when no undecided holder below it may still take its best or second best
        for (int e = tid; e < n_list; e += kProjThreads) {
            const int q = round == 0 ? e : cur[e];
            const int2 p = pick[q];
            const bool final_ = p.x < 0 || (__hip_atomic_load(&minq[p.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= q &&
                                            (p.y < 0 || __hip_atomic_load(&minq[p.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= q));
            if (final_) {
                const int win = qm[q];
                if (win >= 0 && (Q[q].flags & 1)) held[win] = q;  // the only holder this feature will ever have
            } else {
                qm[q] = -1;
                next[atomicAdd(&s_left, 1)] = q;
            }
        }
        __syncthreads();
        n_list = s_left;
        __syncthreads();  // everyone has read s_left before the next round clears it
    }
    if (GUIDED && D.check_orientation) {  // :1597-1607, :1614-1633; the record's third float is the keypoint's angle
        if (tid < kHistoLength) s_hist[tid] = 0;
        __syncthreads();
        for (int q = tid; q < nq; q += kProjThreads) {
            const int v = qm[q];
            if (v >= 0) atomicAdd(&s_hist[proj_rot_bin(Q[q].ur, D.W.kps_un[(size_t)f * kp_cap + v].angle)], 1);
        }
        __syncthreads();
        if (tid == 0) rot_three_maxima(s_hist, s_keep);
        __syncthreads();
        for (int q = tid; q < nq; q += kProjThreads) {
            const int v = qm[q];
            if (v < 0) continue;
            const int bin = proj_rot_bin(Q[q].ur, D.W.kps_un[(size_t)f * kp_cap + v].angle);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) qm[q] = -1;
        }
        __syncthreads();
    }
    // a feature ends with the point written last: the holder if there is one (nothing follows it), else the
    // highest index; every write counts (:123-124)
    for (int q = tid; q < nq; q += kProjThreads) {
        const int v = qm[q];
        if (v < 0) continue;
        atomicMax(&match[v], (int)(((uint32_t)(Q[q].flags & 1) << 30) | (uint32_t)q));
        atomicAdd(&s_nm, 1);
    }
    __syncthreads();
    for (int i = tid; i < kp_cap; i += kProjThreads) {
        const int m = __hip_atomic_load(&match[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        match[i] = m < 0 ? -1 : (m & 0x3FFFFFFF);
    }
    if (tid == 0) D.n_matches[f] = s_nm;
}

}  // namespace orbx
#endif
