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
    if (tid == 0) {
        bow_n[f] = n_bow;
        fv_n[f] = n_valid;
        double norm = 0.0;
        if (V.norm == 1) {  // BowVector::normalize (BowVector.cpp:62-84): accumulated in word order
            for (int t = 0; t < n_bow; ++t) norm = __dadd_rn(norm, fabs(bow_val[o + t]));
        } else if (V.norm == 2) {
            for (int t = 0; t < n_bow; ++t) norm = __dadd_rn(norm, __dmul_rn(bow_val[o + t], bow_val[o + t]));
            norm = __dsqrt_rn(norm);
        } else {
            norm = (V.sums && n_bow > 0) ? (double)n_bow : 0.0;  // :1164-1170
        }
        s_norm = norm;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int t = tid; t < n_bow; t += 1024) bow_val[o + t] = __ddiv_rn(bow_val[o + t], norm);
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

// Sixteen lanes share one query: lane s takes the window's cells s, s + 16, ... in the order
// GetFeaturesInArea visits them (column by column); the first strict minimum in visiting order
// is the minimum of (distance, cell order, position in the cell).
// grid = (ceil(q_cap / 16), frames), block = 256.
__global__ __launch_bounds__(256) void k_search_window(WindowDev W, const WindowQuery* __restrict__ queries,
                                                       const int32_t* __restrict__ n_q, int32_t* __restrict__ best_idx,
                                                       int32_t* __restrict__ best_dist) {
    const int f = blockIdx.y, sub = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (q >= min(n_q[f], W.q_cap)) return;  // whole groups leave together
    const WindowQuery* Q = queries + (size_t)f * W.q_cap + q;
    const float x = Q->u, y = Q->v, r = Q->radius, ur = Q->ur;
    const int minLevel = Q->min_level, maxLevel = Q->max_level;
    const uint4 d0 = *reinterpret_cast<const uint4_unaligned*>(Q->desc), d1 = *reinterpret_cast<const uint4_unaligned*>(Q->desc + 16);
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, W.min_x), r), W.inv_w)));  // :343
    const int nMaxCellX = min(ORBX_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, W.min_x), r), W.inv_w)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, W.min_y), r), W.inv_h)));
    const int nMaxCellY = min(ORBX_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, W.min_y), r), W.inv_h)));
    uint32_t key = 0xFFFFFFFFu;
    int mine = -1;
    if (nMinCellX < ORBX_GRID_COLS && nMaxCellX >= 0 && nMinCellY < ORBX_GRID_ROWS && nMaxCellY >= 0 &&
        nMinCellX <= nMaxCellX && nMinCellY <= nMaxCellY) {
        const bool check_levels = minLevel > 0 || maxLevel >= 0;  // :359
        const int ncy = nMaxCellY - nMinCellY + 1, nc = (nMaxCellX - nMinCellX + 1) * ncy;
        const orbx_kp* kps = W.kps_un + (size_t)f * W.kp_cap;
        const int32_t* cs = W.cell_start + (size_t)f * (ORBX_GRID_COLS * ORBX_GRID_ROWS + 1);
        const int32_t* ci = W.cell_idx + (size_t)f * W.kp_cap;
        for (int t = sub; t < nc; t += 16) {
            const int cx = t / ncy, c = (nMinCellX + cx) * ORBX_GRID_ROWS + nMinCellY + (t - cx * ncy);
            const int j0 = cs[c], j1 = cs[c + 1];
            for (int j = j0; j < j1; ++j) {
                const int i2 = ci[j];
                const orbx_kp kp = kps[i2];
                if (check_levels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (!(fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r)) continue;  // :389
                if (W.taken && W.taken[(size_t)f * W.kp_cap + i2]) continue;                      // ORBmatcher.cc:1435-1437
                if (W.u_right) {
                    const float urf = W.u_right[(size_t)f * W.kp_cap + i2];
                    if (urf > 0 && fabsf(__fsub_rn(ur, urf)) > r) continue;  // :1439-1445
                }
                const uint4* dp = reinterpret_cast<const uint4*>(W.desc + ((size_t)f * W.kp_cap + i2) * 32);
                const uint4 a = dp[0], b = dp[1];
                const uint32_t dist = __popc(d0.x ^ a.x) + __popc(d0.y ^ a.y) + __popc(d0.z ^ a.z) + __popc(d0.w ^ a.w) +
                                      __popc(d1.x ^ b.x) + __popc(d1.y ^ b.y) + __popc(d1.z ^ b.z) + __popc(d1.w ^ b.w);
                const uint32_t kc = (dist << 23) | ((uint32_t)t << 11) | (uint32_t)min(j - j0, 2047);
                if (dist < 256u && kc < key) {  // bestDist starts at 256, strict "<" (:1451)
                    key = kc;
                    mine = i2;
                }
            }
        }
    }
    uint32_t kmin = key;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
    const int src = ((int)(threadIdx.x & 63u) & ~15) + (int)((kmin >> 11) & 15u);  // cell order t is handled by lane t % 16
    const int win = __shfl(mine, src, 64);
    if (sub == 0) {
        const size_t o = (size_t)f * W.q_cap + q;
        best_idx[o] = kmin == 0xFFFFFFFFu ? -1 : win;
        best_dist[o] = kmin == 0xFFFFFFFFu ? 256 : (int)(kmin >> 23);
    }
}

}  // namespace orbx
#endif
