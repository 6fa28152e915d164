// Quadtree retention (DistributeOctTree + ExtractorNode::DivideNode, reference
// src/ORBextractor.cc:507-852) as a data-parallel, list-free algorithm: one workgroup per
// (frame, level).
//
// Reformulation (the result is identical to the std::list version):
//  * a node is a box; its points are the candidates inside it in candidate order, so the point
//    kept for a node is the one with the largest response, lowest candidate index on ties;
//  * children are always pushed at the front of the list, so list order is reverse creation
//    order; an `order` array (front -> back) of node slots replaces the list;
//  * every pass splits nodes holding > 1 point.  A sweep (:667-741) takes them in list order and
//    splits all; the one-at-a-time phase (:757-823) takes them by (size descending, newest
//    first -- the pinned tie-break for the reference's pointer comparison) and stops as soon as
//    the node count reaches N;
//  * points never move: each carries the slot of its node; a pass re-labels the points of split
//    nodes and counts, with LDS atomics, the points per child quadrant for the next pass;
//  * the first non-empty child re-uses its parent's slot, so at most max_kp + n_ini slots exist.
#ifndef ORBX_KERNELS_OCTREE_HPP
#define ORBX_KERNELS_OCTREE_HPP

#include "kernels_fast.hpp"
#include "orbx_device.h"

namespace orbx {

constexpr uint32_t kNoSplit = 0xFFFFFFFFu;
constexpr int kPtBatch = 8;  // global loads in flight per thread in the loops over the points
constexpr int kOctWideFrom = 1000;  // levels that keep more features than this get 1024 threads
constexpr int kOctWideBelowBlocks = 64;  // (frame, level) pairs of a launch up to which the wide workgroup is used whatever the level size

// Block-wide exclusive scan of one int per thread (NT threads = NT / 64 waves).
// Returns the exclusive prefix; *total receives the block sum.  Contains two barriers.
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave, int* total) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    __syncthreads();  // protects s_wave against a previous use
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int t = s_wave[k];
        if (k < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

// In-place exclusive scan of a[0..n) in LDS by NT threads; returns the total.
// Thread t owns the run [t*per, (t+1)*per).
template <int NT>
__device__ __forceinline__ uint32_t lds_excl_scan(uint32_t* a, int n, int* s_wave4) {
    const int per = (n + NT - 1) / NT;
    const int beg = min((int)threadIdx.x * per, n), end = min(beg + per, n);
    uint32_t sum = 0;
    for (int i = beg; i < end; ++i) sum += a[i];
    int total;
    uint32_t run = (uint32_t)block_excl_scan<NT>((int)sum, s_wave4, &total);
    for (int i = beg; i < end; ++i) {
        const uint32_t v = a[i];
        a[i] = run;
        run += v;
    }
    __syncthreads();
    return (uint32_t)total;
}

struct OctLds {
    uint16_t *bx0, *bx1, *by0, *by1;  // node boxes, per slot
    uint32_t* cnt;                    // points per slot
    uint32_t* cc;                     // points per child quadrant, 4 per slot
    uint16_t* child;                  // slot of each child of a node split in this pass, 4 per slot
    uint32_t* mid;                    // split point (mx | my << 16) of a node split in this pass, else kNoSplit
    uint32_t* best;                   // final arg-max key per slot
    uint16_t *ordA, *ordB;            // list order, ping-pong
    uint32_t* key;                    // per order index: count << 16 | (0xFFFF - index) if expandable else 0
    uint16_t* rank;                   // per order index: processing rank among expandable nodes
    uint16_t* proc;                   // processing rank -> order index
    uint32_t* scanA;                  // per rank: created | growth << 16 (then its exclusive scan)
    uint32_t* scanB;                  // per order index: stays-in-place flag (then its exclusive scan)
};

__host__ __device__ inline size_t oct_lds_bytes(int max_slots, int max_kp, int n_cells) {
    const size_t ms = (size_t)((max_slots + 3) & ~3), mk = (size_t)((max_kp + 3) & ~3);
    size_t node = ms * (4 * 2 + 4 + 16 + 8 + 4 + 4) + mk * (2 + 2 + 4 + 2 + 2 + 4 + 4);
    size_t pro = (size_t)(n_cells + 1) * 4;  // cell prefix, only used before the node tables
    return (node > pro ? node : pro) + 64;
}

// grid = (frames, nlevels), block = NT (256, or 1024 for levels of thousands of features), dynamic LDS = max
// over levels of oct_lds_bytes.
// GT (global tables): quotas whose tables do not fit the LDS (above about 2500 features per level) keep them in a
// workspace in device memory instead, ws_stride bytes per (frame, level), dynamic LDS = 0.  The code is the same:
// the tables are reached through pointers, the atomics become global atomics, and the barriers order the
// workgroup's global accesses as they order its LDS accesses.  Slower, and only there so that such quotas run.
template <int NT, bool GT>
__global__ __launch_bounds__(NT) void k_octree(BatchDev B, const uint32_t* __restrict__ cand,
                                                const uint32_t* __restrict__ ccount,
                                                uint32_t* __restrict__ dense_pts, uint16_t* __restrict__ dense_node,
                                                uint32_t* __restrict__ lvl_kps, int32_t* __restrict__ lvl_count,
                                                uint8_t* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    uint8_t* s_raw;
    if constexpr (GT)
        s_raw = ws + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * ws_stride;
    else
        s_raw = s_lds;
    __shared__ int s_wave4[NT / 64];
    __shared__ int s_i[8];  // 0: jcut, 1: nToExpand, 2: finished, 3: careful

    const int tid = threadIdx.x;
    const int level = blockIdx.y, frame = blockIdx.x;  // level-major launch order: the levels with the most work start first
    const LevelDev& L = B.lv[level];
    const int N = L.quota;
    unsigned long long t0 = 0;
#ifdef ORBX_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- prologue: gather the level's candidates into a dense array, in cell order ----------
    uint32_t* s_prefix = reinterpret_cast<uint32_t*>(s_raw);
    const uint64_t slot0 = (uint64_t)frame * B.total_cells + L.cell_base;
    // A count is trusted no further than the slot it describes: a value above the slot's capacity (memory the
    // library does not own was written, or a producer failed) is cut to it and reported, so that the sum stays within
    // the dense arrays (n_cells * cell_cap entries per level) and every read of `cand` stays inside a slot.
    for (int c = tid; c < L.n_cells; c += NT) {
        const uint32_t v = ccount[slot0 + c];
        if (v > (uint32_t)B.cell_cap && B.status) B.status[kStatusCellCount] = 1u;
        s_prefix[c] = min(v, (uint32_t)B.cell_cap);
    }
    if (tid == 0) s_prefix[L.n_cells] = 0;
    __syncthreads();
    const int C = (int)lds_excl_scan<NT>(s_prefix, L.n_cells + 1, s_wave4);
    uint32_t* pts = dense_pts + (uint64_t)frame * B.dense_total + L.dense_off;
    uint16_t* node_of = dense_node + (uint64_t)frame * B.dense_total + L.dense_off;
    int32_t* out_count = lvl_count + (uint64_t)frame * B.nlevels + level;
    if (C == 0) {
        if (tid == 0) *out_count = 0;
        return;
    }
    const float h_x = L.h_x;
    // Global loads are issued kPtBatch at a time before their first use: the loops over the points
    // are latency-bound, and independent loads in flight divide the number of round trips.
    for (int p0 = tid; p0 < C; p0 += NT * kPtBatch) {
        uint32_t v[kPtBatch];
        uint64_t src[kPtBatch];
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) {
            const int p = min(p0 + NT * k, C - 1);
            int lo = 0, hi = L.n_cells;  // last cell with prefix <= p
            while (hi - lo > 1) {
                const int m = (lo + hi) >> 1;
                if ((int)s_prefix[m] <= p) lo = m; else hi = m;
            }
            src[k] = (slot0 + lo) * B.cell_cap + (p - (int)s_prefix[lo]);
        }
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) v[k] = cand[src[k]];
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) {
            const int p = p0 + NT * k;
            if (p < C) {
                // the detector leaves a 3-pixel rim of the detection rectangle free (cv::FAST's own border); a point
                // outside [3, rect - 3) cannot come from it: it is moved inside (later stages read image patches around
                // the retained points) and reported
                if ((unsigned)(pt_x(v[k]) - 3) >= (unsigned)(L.rect_w - 6) || (unsigned)(pt_y(v[k]) - 3) >= (unsigned)(L.rect_h - 6)) {
                    if (B.status) B.status[kStatusPoint] = 1u;
                    v[k] = pack_pt(max(3, min(pt_x(v[k]), L.rect_w - 4)), max(3, min(pt_y(v[k]), L.rect_h - 4)), pt_score(v[k]));
                }
                pts[p] = v[k];
                int r = (int)((float)pt_x(v[k]) / h_x);  // root of the point (:620)
                r = min(r, L.n_ini - 1);
                node_of[p] = (uint16_t)r;
            }
        }
    }
    __syncthreads();  // s_prefix is dead from here on; the node tables take its place
    ORBX_STAMP(B, 8, t0);

    // ---- carve the node tables -----------------------------------------------------------------
    OctLds T;
    {
        const size_t ms = (size_t)((L.max_slots + 3) & ~3), mk = (size_t)((L.max_kp + 3) & ~3);
        uint8_t* q = s_raw;
        T.cnt = (uint32_t*)q; q += ms * 4;
        T.cc = (uint32_t*)q; q += ms * 16;
        T.mid = (uint32_t*)q; q += ms * 4;
        T.best = (uint32_t*)q; q += ms * 4;
        T.key = (uint32_t*)q; q += mk * 4;
        T.scanA = (uint32_t*)q; q += mk * 4;
        T.scanB = (uint32_t*)q; q += mk * 4;
        T.bx0 = (uint16_t*)q; q += ms * 2;
        T.bx1 = (uint16_t*)q; q += ms * 2;
        T.by0 = (uint16_t*)q; q += ms * 2;
        T.by1 = (uint16_t*)q; q += ms * 2;
        T.child = (uint16_t*)q; q += ms * 8;
        T.ordA = (uint16_t*)q; q += mk * 2;
        T.ordB = (uint16_t*)q; q += mk * 2;
        T.rank = (uint16_t*)q; q += mk * 2;
        T.proc = (uint16_t*)q; q += mk * 2;
    }
    for (int s = tid; s < L.max_slots; s += NT) {
        T.cnt[s] = 0;
        T.mid[s] = kNoSplit;
        T.best[s] = 0;
        T.cc[4 * s] = T.cc[4 * s + 1] = T.cc[4 * s + 2] = T.cc[4 * s + 3] = 0;
    }
    __syncthreads();
    // roots (:585-610): box i = [int(hX*i), int(hX*(i+1))] x [0, H]
    for (int i = tid; i < L.n_ini; i += NT) {
        T.bx0[i] = (uint16_t)(int)(h_x * (float)i);
        T.bx1[i] = (uint16_t)(int)(h_x * (float)(i + 1));
        T.by0[i] = 0;
        T.by1[i] = (uint16_t)L.rect_h;
    }
    for (int p0 = tid; p0 < C; p0 += NT * kPtBatch) {
        int s[kPtBatch];
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) s[k] = p0 + NT * k < C ? (int)node_of[p0 + NT * k] : -1;
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k)
            if (s[k] >= 0) atomicAdd(&T.cnt[s[k]], 1u);
    }
    __syncthreads();
    int n_alive = 0;
    {
        // non-empty roots in order (:627-639); n_ini is small
        if (tid == 0) {
            int n = 0;
            for (int i = 0; i < L.n_ini; ++i)
                if (T.cnt[i] > 0) T.ordA[n++] = (uint16_t)i;
            s_i[4] = n;
            s_i[2] = 0;
            s_i[3] = 0;
        }
        __syncthreads();
        n_alive = s_i[4];
    }
    int n_slots = L.n_ini;
    uint16_t* ord = T.ordA;
    uint16_t* ord_new = T.ordB;

    // child-quadrant counts of every node with > 1 point
    for (int p0 = tid; p0 < C; p0 += NT * kPtBatch) {
        int sn[kPtBatch];
        uint32_t v[kPtBatch];
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) {
            const bool ok = p0 + NT * k < C;
            sn[k] = ok ? (int)node_of[p0 + NT * k] : -1;
            v[k] = ok ? pts[p0 + NT * k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kPtBatch; ++k) {
            const int s = sn[k];
            if (s >= 0 && T.cnt[s] > 1) {
                const int x0 = T.bx0[s], x1 = T.bx1[s], y0 = T.by0[s], y1 = T.by1[s];
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);  // ceil halves (:512-513)
                const int q = (pt_x(v[k]) >= mx ? 1 : 0) + (pt_y(v[k]) >= my ? 2 : 0);
                atomicAdd(&T.cc[4 * s + q], 1u);
            }
        }
    }
    __syncthreads();

    ORBX_STAMP(B, 9, t0);
    bool careful = false;
    for (;;) {
        // ---------------- node step ----------------
        const int n = n_alive;
        // sort key of an expandable node: size, then list position (earlier first); unique, 0 = not expandable.
        // The size field holds 16 bits: larger nodes all carry 0xFFFF there and are told apart by their
        // real counts where the order matters (the ranking of the one-at-a-time phase below).
        for (int i = tid; i < ((n + 3) & ~3); i += NT) {
            const uint32_t c = i < n ? T.cnt[ord[i]] : 0u;
            T.key[i] = c > 1 ? (min(c, 0xFFFFu) << 16) | (0xFFFFu - (uint32_t)i) : 0u;
        }
        if (tid == 0) {
            s_i[0] = 0x7FFFFFFF;
            s_i[1] = 0;
        }
        __syncthreads();
        int n_exp;
        if (!careful) {
            for (int i = tid; i < n; i += NT) T.scanB[i] = T.key[i] ? 1u : 0u;
            __syncthreads();
            n_exp = (int)lds_excl_scan<NT>(T.scanB, n, s_wave4);
            for (int i = tid; i < n; i += NT)
                if (T.key[i]) T.rank[i] = (uint16_t)T.scanB[i];
        } else {
            // rank by (size descending, list position ascending): counting rank, O(n^2 / NT)
            int cnt_exp = 0;
            for (int i = tid; i < n; i += NT) {
                const uint32_t ki = T.key[i];
                if (ki) {
                    int r = 0;
                    if ((ki >> 16) != 0xFFFFu) {
                        const uint4* k4 = reinterpret_cast<const uint4*>(T.key);
                        for (int j = 0; j < (n + 3) >> 2; ++j) {
                            const uint4 kj = k4[j];
                            r += (kj.x > ki ? 1 : 0) + (kj.y > ki ? 1 : 0) + (kj.z > ki ? 1 : 0) + (kj.w > ki ? 1 : 0);
                        }
                    } else {
                        // a node of 65535 points or more: only nodes of that class can precede it, by their
                        // real sizes, earlier list position first among equals
                        const uint32_t ci = T.cnt[ord[i]];
                        for (int j = 0; j < n; ++j) {
                            if ((T.key[j] >> 16) != 0xFFFFu) continue;
                            const uint32_t cj = T.cnt[ord[j]];
                            r += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
                        }
                    }
                    T.rank[i] = (uint16_t)r;
                    cnt_exp++;
                }
            }
            int tot;
            block_excl_scan<NT>(cnt_exp, s_wave4, &tot);
            n_exp = tot;
        }
        __syncthreads();
        // per rank: created children | growth << 16
        for (int i = tid; i < n; i += NT) {
            if (T.key[i]) {
                const int s = ord[i];
                const int k = (T.cc[4 * s] ? 1 : 0) + (T.cc[4 * s + 1] ? 1 : 0) + (T.cc[4 * s + 2] ? 1 : 0) +
                              (T.cc[4 * s + 3] ? 1 : 0);
                const int r = T.rank[i];
                T.proc[r] = (uint16_t)i;
                T.scanA[r] = (uint32_t)k | ((uint32_t)(k - 1) << 16);
            }
        }
        __syncthreads();
        // keep inclusive values reachable: scanA becomes exclusive, the element value is re-derived
        // from cc when needed
        uint32_t tot_all = 0;
        if (n_exp > 0) tot_all = lds_excl_scan<NT>(T.scanA, n_exp, s_wave4);
        // cut: first rank whose inclusive growth brings the node count to N (careful phase only)
        if (careful) {
            for (int r = tid; r < n_exp; r += NT) {
                const uint32_t incl = (r + 1 < n_exp) ? T.scanA[r + 1] : tot_all;
                if (n + (int)(incl >> 16) >= N) atomicMin(&s_i[0], r);
            }
        }
        __syncthreads();
        const int jcut = min(s_i[0], n_exp - 1);  // last rank that is split (-1 if none)
        const uint32_t tot_cut = (jcut + 1 < n_exp) ? T.scanA[jcut + 1] : tot_all;
        const int created = n_exp > 0 ? (int)(tot_cut & 0xFFFF) : 0;
        const int growth = n_exp > 0 ? (int)(tot_cut >> 16) : 0;
        // stays-in-place flags in list order
        for (int i = tid; i < n; i += NT) {
            const bool split = T.key[i] && (int)T.rank[i] <= jcut;
            T.scanB[i] = split ? 0u : 1u;
        }
        __syncthreads();
        // split the chosen nodes
        for (int r = tid; r <= jcut; r += NT) {
            const int i = T.proc[r];
            const int s = ord[i];
            const uint32_t ex = T.scanA[r];
            const int c_off = (int)(ex & 0xFFFF), g_off = (int)(ex >> 16);
            const int x0 = T.bx0[s], x1 = T.bx1[s], y0 = T.by0[s], y1 = T.by1[s];
            const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
            uint32_t c4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c4[q] = T.cc[4 * s + q];
            T.mid[s] = (uint32_t)mx | ((uint32_t)my << 16);
            int m = 0, n_big = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // n1 = UL, n2 = UR, n3 = BL, n4 = BR (:517-540)
                if (c4[q] == 0) {
                    T.child[4 * s + q] = 0xFFFF;
                    continue;
                }
                const int slot = m == 0 ? s : n_slots + g_off + (m - 1);
                T.child[4 * s + q] = (uint16_t)slot;
                T.bx0[slot] = (uint16_t)((q & 1) ? mx : x0);
                T.bx1[slot] = (uint16_t)((q & 1) ? x1 : mx);
                T.by0[slot] = (uint16_t)((q & 2) ? my : y0);
                T.by1[slot] = (uint16_t)((q & 2) ? y1 : my);
                T.cnt[slot] = c4[q];
                T.cc[4 * slot] = T.cc[4 * slot + 1] = T.cc[4 * slot + 2] = T.cc[4 * slot + 3] = 0;
                ord_new[created - 1 - (c_off + m)] = (uint16_t)slot;  // newest first
                n_big += c4[q] > 1 ? 1 : 0;
                ++m;
            }
            if (n_big) atomicAdd(&s_i[1], n_big);
        }
        __syncthreads();
        // nodes that were not split keep their relative order behind the new children.
        // (slots are read from `ord`; a re-used slot id in `ord` belongs to a split node and is skipped)
        {
            // flags must be captured before the scan overwrites them
            const int per = (n + NT - 1) / NT;
            const int beg = min(tid * per, n), end = min(beg + per, n);
            uint32_t keep_mask = 0;  // per <= 32 guaranteed by max_kp <= 8192
            for (int i = beg; i < end; ++i) keep_mask |= (T.scanB[i] ? 1u : 0u) << (i - beg);
            __syncthreads();
            lds_excl_scan<NT>(T.scanB, n, s_wave4);
            for (int i = beg; i < end; ++i)
                if ((keep_mask >> (i - beg)) & 1u) ord_new[created + (int)T.scanB[i]] = ord[i];
        }
        const int n_split = jcut + 1;
        n_alive = n - n_split + created;
        n_slots += growth;
        __syncthreads();
        if (tid == 0) {
            const int n_to_expand = s_i[1];
            const bool fin = n_alive >= N || n_alive == n;
            s_i[2] = fin ? 1 : 0;
            if (!fin && !careful && n_alive + 3 * n_to_expand > N) s_i[3] = 1;
        }
        __syncthreads();
        const bool finished = s_i[2] != 0;
        careful = s_i[3] != 0;
        {
            uint16_t* t = ord;
            ord = ord_new;
            ord_new = t;
        }

        ORBX_STAMP(B, careful ? 11 : 10, t0);
        // ---------------- point step ----------------
        for (int p0 = tid; p0 < C; p0 += NT * kPtBatch) {
            int sn[kPtBatch];
            uint32_t vv[kPtBatch];
#pragma unroll
            for (int k = 0; k < kPtBatch; ++k) {
                const bool ok = p0 + NT * k < C;
                sn[k] = ok ? (int)node_of[p0 + NT * k] : -1;
                vv[k] = ok ? pts[p0 + NT * k] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kPtBatch; ++k) {
                int s = sn[k];
                if (s < 0) continue;
                const int p = p0 + NT * k;
                const uint32_t v = vv[k];
                const uint32_t m = T.mid[s];
                if (m != kNoSplit) {
                    const int q = (pt_x(v) >= (int)(m & 0xFFFF) ? 1 : 0) + (pt_y(v) >= (int)(m >> 16) ? 2 : 0);
                    s = T.child[4 * s + q];
                    node_of[p] = (uint16_t)s;
                    if (!finished && T.cnt[s] > 1) {
                        const int x0 = T.bx0[s], x1 = T.bx1[s], y0 = T.by0[s], y1 = T.by1[s];
                        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                        const int q2 = (pt_x(v) >= mx ? 1 : 0) + (pt_y(v) >= my ? 2 : 0);
                        atomicAdd(&T.cc[4 * s + q2], 1u);
                    }
                }
                if (finished) {
                    // largest response, lowest candidate index on ties (:836-846)
                    atomicMax(&T.best[s], ((uint32_t)pt_score(v) << 24) | (0xFFFFFFu - (uint32_t)p));
                }
            }
        }
        __syncthreads();
        ORBX_STAMP(B, 12, t0);
        if (finished) break;
        for (int s = tid; s < n_slots; s += NT) T.mid[s] = kNoSplit;
        __syncthreads();
    }

    // ---- output in list order ------------------------------------------------------------------
    uint32_t* out = lvl_kps + (uint64_t)frame * B.kp_total + L.kp_off;
    for (int i = tid; i < n_alive; i += NT) {
        const uint32_t b = T.best[ord[i]];
        const uint32_t p = 0xFFFFFFu - (b & 0xFFFFFFu);
        out[i] = pts[p];
    }
    if (tid == 0) *out_count = n_alive;
    ORBX_STAMP(B, 13, t0);
}

}  // namespace orbx
#endif
