// Per-cell FAST-9/16 detection with cell-local non-max suppression and the per-cell threshold
// fallback (ComputeKeyPointsOctTree, reference src/ORBextractor.cc:880-920, calling cv::FAST).
#ifndef ORBX_KERNELS_FAST_HPP
#define ORBX_KERNELS_FAST_HPP

#include "orbx_device.h"

namespace orbx {

// Inclusive prefix sum over the 64 lanes of a wave with DPP adds (row shifts 1, 2, 4, 8 inside the rows of
// 16 lanes, then the row broadcasts 15 and 31); lanes a shift does not reach add 0.
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}

// Block-wide exclusive scan of one int per thread (256 threads = 4 waves).
// Returns the exclusive prefix; *total receives the block sum.  Contains two barriers.
__device__ __forceinline__ int block_excl_scan_256(int v, int* s_wave4, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
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

// The same on pairs of ring positions packed as two 16-bit lanes: X[j] holds (x[2j], x[2j+1]).
// Only the first window step needs a lane shift (v_alignbit); the others combine whole pairs.
typedef short fast_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(fast_i16x2, a), __builtin_bit_cast(fast_i16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(fast_i16x2, a), __builtin_bit_cast(fast_i16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {  // wrapping, per 16-bit lane
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(fast_i16x2, a) - __builtin_bit_cast(fast_i16x2, b));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {  // wrapping, per 16-bit lane
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(fast_i16x2, a) + __builtin_bit_cast(fast_i16x2, b));
}
__device__ __forceinline__ int arc_max_pk(const uint32_t X[8]) {
    uint32_t m2[8], m4[8], m8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m2[j] = pk_min_i16(X[j], __builtin_amdgcn_alignbit(X[(j + 1) & 7], X[j], 16));  // (x[2j+1], x[2j+2])
#pragma unroll
    for (int j = 0; j < 8; ++j) m4[j] = pk_min_i16(m2[j], m2[(j + 1) & 7]);
#pragma unroll
    for (int j = 0; j < 8; ++j) m8[j] = pk_min_i16(m4[j], m4[(j + 2) & 7]);
    uint32_t best = pk_min_i16(m8[0], X[4]);
#pragma unroll
    for (int j = 1; j < 8; ++j) best = pk_max_i16(best, pk_min_i16(m8[j], X[(j + 4) & 7]));
    return max((int)(short)(best & 0xFFFFu), (int)(short)(best >> 16));
}

typedef unsigned short fast_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}

__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add_sat_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(fast_u16x2, a), __builtin_bit_cast(fast_u16x2, b)));
}

// (a & mask) | (b & ~mask) as one v_bfi_b32.  Written as an instruction because the compiler, knowing that only
// some result bits are read later, splits the expression into more instructions than it saves.
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}

// 1 / n for a small positive integer n as one v_rcp_f32 (1 ulp) instead of an IEEE division (11 instructions).
// Every use is floor((i + 0.5) / n) = (int)((i + 0.5f) * fast_rcp(n)) with i < 2^13 and n < 2^13: (i + 0.5) / n
// stays at least 0.5 / n away from an integer, thousands of times the rounding error, so the floor is exact.
__device__ __forceinline__ float fast_rcp(int n) { return __builtin_amdgcn_rcpf((float)n); }

constexpr int kFastMaxCell = 66;  // largest sub-image side: wCell <= 59, + 6, + 1

// LDS carve-up of k_fast_tiles for one tile geometry (bytes).
struct FastLds {
    int tile, plane, queue, kept, verdict;
};
// Tile layout: LDS column of detection column u is u + 4, so u = 0 sits on a dword boundary and
// the 3-pixel halo on its left occupies columns 1..3.  Four spare columns on the right keep the
// reads of the (discarded) pixels between wd and the next multiple of 4 inside the row.
__host__ __device__ inline int fast_tile_pitch(int wd) { return (wd + 14 + 15) & ~15; }
__host__ __device__ inline int fast_plane_pitch(int wd, int nc) { return (wd + nc + 1 + 3) & ~3; }
// Queue entries for both ends together.  A quarter of a tile's pixels is far above what textured images
// produce (about 6 %); a tile that overflows falls back to scoring every pixel.
__host__ __device__ inline int fast_queue_cap(int wd, int hd) { return (((wd + 3) & ~3) * hd) / 4; }
__host__ __device__ inline int fast_kept_cap(int wd, int hd) { return (((wd + 3) & ~3) * hd) / 8 + 64; }
// Verdict plane of step 1: one byte per work item (a dword of four pixels), rows of a multiple of four items.
__host__ __device__ inline int fast_verdict_pitch(int wd) { return (((wd + 3) >> 2) + 3) & ~3; }
__host__ __device__ inline FastLds fast_lds_need(int wd, int hd, int nc) {
    FastLds n;
    n.tile = fast_tile_pitch(wd) * (hd + 6) + 16;
    n.plane = fast_plane_pitch(wd, nc) * (hd + 2);
    n.queue = fast_queue_cap(wd, hd) * 2;
    n.kept = fast_kept_cap(wd, hd) * 2;
    n.verdict = fast_verdict_pitch(wd) * hd;
    return n;
}

constexpr int kFastFixedPitch = 160;  // fits tiles up to 131 detection columns (four cells of up to 32 pixels, or three of up to 43)
#ifndef ORBX_FAST_STAGE_DEPTH
#define ORBX_FAST_STAGE_DEPTH 2  // 16-byte loads in flight per thread while staging a tile: a tile of four cells is 380 chunks, two per thread at most (3: 1.3 % slower, the third slot is address arithmetic for nothing; 4: 0.5 % slower)
#endif
constexpr int kFastStageDepth = ORBX_FAST_STAGE_DEPTH;  // 16-byte loads in flight per thread while staging a tile

// Diagnostic build only (-DORBX_STAMPS): wave 0 adds the cycles between phase boundaries into
// B.stamps[phase] (1024 copies, spread by block id, to keep the atomics off one address); the product build compiles these calls away.
#ifdef ORBX_STAMPS
#define ORBX_STAMP(B, slot, t0)                                                                   \
    do {                                                                                          \
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                               \
        if (threadIdx.x == 0 && (B).stamps) atomicAdd(&(B).stamps[(slot) + 16 * ((blockIdx.x + 37 * blockIdx.y) & 1023)], t1_ - (t0));              \
        (t0) = __builtin_amdgcn_s_memtime();                                                       \
    } while (0)
#else
#define ORBX_STAMP(B, slot, t0) do { (void)(t0); } while (0)
#endif

struct FastShared {  // small per-workgroup state in static LDS
    int q_n;  // survivors queued so far (entries: a pixel that passes the rejection test on both sides has two)
    int n_kept, retry;
    int n_corner;  // pixels scored as corners in step 2 of the current pass
    int kept[kFastTileCells];
    int th[kFastTileCells];
};

// Queue entry of a survivor of the rejection test: row << 9 | column << 1 | side, side 1 = ring darker than the
// centre (a "bright" corner), 0 = ring brighter; step 2 sets bit 15 of the entries it scores as corners.
// Full segment-test half score of the 64 queue entries [q0, q0 + 64) below q_hi, by one wave.  The side is a lane's
// own: d_k = (ring_k - centre) * s with s = -1 (ring darker) or +1, one packed multiply-add per pair of ring
// pixels, so that a wave scores entries of both sides without diverging.
template <bool RETRY>
__device__ __forceinline__ void fast_score_wave(const uint8_t* s_tile, uint8_t* s_score, uint16_t* s_queue,
                                                FastShared* sh, int q0, int q_hi, int TP, int SP, int wd, int nc,
                                                float inv_wcell, int thr, uint16_t* s_corner_top, int corner_room) {
    const int lane = threadIdx.x & 63;
    {
        const int q = q0 + lane;
        bool hit = false;
        int ke = 0;
        if (q < q_hi) {
            const int e = s_queue[q], u = (e >> 1) & 0xFF, vq = e >> 9;
            if (u < wd) {  // not a padding pixel of the last dword of a row
                // w = upper left corner of the pixel's 7x7 window: every ring pixel is w[(3 + dy) * TP + 3 + dx], a
                // constant offset when the pitch is a compile-time constant
                const uint8_t* w = s_tile + vq * TP + 1 + u;
#define ORBX_RING(dx, dy) ((uint32_t)w[(3 + (dy)) * TP + 3 + (dx)])
                const fast_i16x2 sgn = __builtin_bit_cast(fast_i16x2, (e & 1) ? 0xFFFFFFFFu : 0x00010001u);
                // -centre * s in both lanes
                const fast_i16x2 cv2 = __builtin_bit_cast(fast_i16x2, ORBX_RING(0, 0) * 0x00010001u);
                const fast_i16x2 cs = (e & 1) ? cv2 : -cv2;
                // ring positions 2j and 2j+1 share a register
                uint32_t P[8];
                P[0] = ORBX_RING(0, 3) | (ORBX_RING(1, 3) << 16);
                P[1] = ORBX_RING(2, 2) | (ORBX_RING(3, 1) << 16);
                P[2] = ORBX_RING(3, 0) | (ORBX_RING(3, -1) << 16);
                P[3] = ORBX_RING(2, -2) | (ORBX_RING(1, -3) << 16);
                P[4] = ORBX_RING(0, -3) | (ORBX_RING(-1, -3) << 16);
                P[5] = ORBX_RING(-2, -2) | (ORBX_RING(-3, -1) << 16);
                P[6] = ORBX_RING(-3, 0) | (ORBX_RING(-3, 1) << 16);
                P[7] = ORBX_RING(-2, 2) | (ORBX_RING(-1, 3) << 16);
#undef ORBX_RING
                uint32_t D[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) D[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(fast_i16x2, P[j]) * sgn + cs);
                const int S = arc_max_pk(D);
                if (S > thr) {
                    const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
                    if (!RETRY || sh->th[c] >= 0) {
                        s_score[(vq + 1) * SP + 1 + u + c] = (uint8_t)(S - 1);
                        hit = true;
                        ke = (c << 14) | (e >> 1);
                        s_queue[q] = (uint16_t)(e | 0x8000);  // marks the entry as a corner (a pixel is one on one side at most)
                    }
                }
            }
        }
        // the corners are listed (downwards from the top of the survivor list's memory) so that step 3
        // visits them alone
        const uint64_t m = __ballot(hit);
        if (m) {
            int base = 0;
            if (lane == 0) {  // a plain LDS add by one lane for the wave
                uint32_t got;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(got)
                             : "v"((uint32_t)(uintptr_t)&sh->n_corner), "v"((uint32_t)__popcll(m))
                             : "memory");
                base = (int)got;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) {
                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (pos < corner_room) s_corner_top[-pos] = (uint16_t)ke;
            }
        }
    }
}

// Steps 1-3 of one pass over a tile (see k_fast_tiles).  RETRY = false: threshold iniTh, every
// cell; RETRY = true: threshold minTh, only the cells whose th entry is >= 0.  The rejection test visits the
// dword columns [j_lo, j_lo + ndw) of every row (the whole tile, or the columns that hold the cells to retry).
template <bool RETRY>
__device__ __forceinline__ void fast_pass(const uint8_t* s_tile, uint8_t* s_score, uint16_t* s_queue, uint16_t* s_keptl,
                                          uint8_t* s_verdict, FastShared* sh, int TP, int SP, int wd, int hd, int nc,
                                          float inv_wcell, int thr, int q_cap, int k_cap, const BatchDev& B, int stamp_base,
                                          int j_lo, int ndw) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long t0 = 0;
#ifdef ORBX_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0) {
        sh->q_n = 0;
        sh->n_corner = 0;
    }
    const int kept_before = sh->n_kept;  // survivors of the first pass (read before anyone adds to it: step 3 is two barriers away)
    if (RETRY)  // the verdicts of the first pass outside the columns this pass visits must not be queued again
        for (int i = tid; i < (fast_verdict_pitch(wd) * hd) >> 2; i += 256) reinterpret_cast<uint32_t*>(s_verdict)[i] = 0;
    __syncthreads();
    // ---- 1. rejection test, four pixels per work item, two pixels per instruction ----
    // Any arc of 9 holds one pixel of each opposite pair, so a corner needs the two compass pairs
    // (up/down, left/right) and the two diagonal pairs ((+-2, +-2)) to each hold a pixel beyond the threshold
    // on the same side.  (The diagonal pairs halve the survivors, 11.5 % -> 5.7 % of the pixels of a textured
    // frame against 2.8 % corners, for a third more instructions in the test.)
    // The test is a filter: it must pass every corner and may pass a few pixels more (step 2 scores the survivors
    // exactly).  That allows it to run on the dwords as they are, as two 16-bit lanes whose HIGH bytes are the
    // pixels under test (pixels 1 and 3 of the dword; pixels 0 and 2 after a shift by one byte) and whose low bytes
    // are whatever pixel lies there: packed u16 min/max order the lanes by the high byte first, so the high byte
    // of a chain of minima and maxima is that chain on the high bytes, and the low bytes move a difference of
    // two lanes by less than 256.  centre - max(min(u,d), min(l,r), ...) > t is tested as
    // lane difference >= 256 t + 1, which holds whenever the exact test does.  Two saturating packed operations
    // per side turn the comparison into bit 15 of each lane.
    // The eight verdicts of a work item (four pixels, two sides) go to a byte plane; step 1b turns the
    // plane into the queue.  (Appending to the queue here, with one ballot per verdict, cost as many vector
    // instructions again as the test itself.)
    const int n_items = hd * ndw, vp = fast_verdict_pitch(wd);
    // bias that moves "difference >= 256 t + 1" into a lane's bit 15 (t above 127 is tested as 127: still a filter)
    const uint32_t k_bias = (uint32_t)(0x8000 - (256 * min(thr, 127) + 1)) * 0x00010001u;
    // The test of one work item: pw = the tile's row v - 3 at the dword left of the item's, pv = the item's verdict
    // byte, jd = the item's dword column in the tile (only the retry reads it).
    auto test_item = [&](const uint8_t* pw, uint8_t* pv, int jd) {
        (void)jd;
        uint32_t pb[2], pd[2];  // [0]: bits 15/31 = pixels 0/2, [1]: pixels 1/3
        // the dwords of the item's columns (c), of the four columns before (p) and after (n), rows -3 .. +3
        const uint32_t Cc = *reinterpret_cast<const uint32_t*>(pw + 3 * TP + 4), Cp = *reinterpret_cast<const uint32_t*>(pw + 3 * TP),
                       Cn = *reinterpret_cast<const uint32_t*>(pw + 3 * TP + 8);
        const uint32_t Ud = *reinterpret_cast<const uint32_t*>(pw + 6 * TP + 4), Dd = *reinterpret_cast<const uint32_t*>(pw + 4);
        const uint32_t P2c = *reinterpret_cast<const uint32_t*>(pw + 5 * TP + 4), P2p = *reinterpret_cast<const uint32_t*>(pw + 5 * TP),
                       P2n = *reinterpret_cast<const uint32_t*>(pw + 5 * TP + 8);
        const uint32_t M2c = *reinterpret_cast<const uint32_t*>(pw + TP + 4), M2p = *reinterpret_cast<const uint32_t*>(pw + TP),
                       M2n = *reinterpret_cast<const uint32_t*>(pw + TP + 8);
        uint32_t on[2] = {0x80008000u, 0x80008000u};
        if (RETRY) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int u = min(4 * jd + b, wd - 1);
                if (sh->th[min((int)(((float)u + 0.5f) * inv_wcell), nc - 1)] < 0) on[b & 1] &= ~(0x8000u << (8 * (b & 2)));
            }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {  // 0: pixels 0, 2   1: pixels 1, 3
            // every operand is a dword whose bytes 1 and 3 are the ring pixels of the two pixels under test:
            // v_alignbyte_b32 of two neighbouring dwords (pixels +-3 on the row, +-2 on the rows +-2), the
            // dword itself or the dword shifted by one byte (centre, rows +-3)
            uint32_t c, u, d, r, l, pp, mp, pm, mm;
            if (h2) {
                c = Cc, u = Ud, d = Dd;
                r = __builtin_amdgcn_alignbyte(Cn, Cc, 3), l = __builtin_amdgcn_alignbyte(Cc, Cp, 1);
                pp = __builtin_amdgcn_alignbyte(P2n, P2c, 2), mp = __builtin_amdgcn_alignbyte(P2c, P2p, 2);
                pm = __builtin_amdgcn_alignbyte(M2n, M2c, 2), mm = __builtin_amdgcn_alignbyte(M2c, M2p, 2);
            } else {
                c = Cc << 8, u = Ud << 8, d = Dd << 8;
                r = __builtin_amdgcn_alignbyte(Cn, Cc, 2), l = Cp;
                pp = __builtin_amdgcn_alignbyte(P2n, P2c, 1), mp = __builtin_amdgcn_alignbyte(P2c, P2p, 1);
                pm = __builtin_amdgcn_alignbyte(M2n, M2c, 1), mm = __builtin_amdgcn_alignbyte(M2c, M2p, 1);
            }
            // ring darker on all four pairs: centre - max(min(u,d), min(l,r), min(pp,mm), min(pm,mp)) > t;
            // ring brighter: min(max(u,d), ...) - centre > t
            uint32_t hi = pk_max_u16(pk_min_u16(u, d), pk_min_u16(l, r));
            uint32_t lo = pk_min_u16(pk_max_u16(u, d), pk_max_u16(l, r));
            hi = pk_max_u16(hi, pk_max_u16(pk_min_u16(pp, mm), pk_min_u16(pm, mp)));
            lo = pk_min_u16(lo, pk_min_u16(pk_max_u16(pp, mm), pk_max_u16(pm, mp)));
            pb[h2] = pk_add_sat_u16(pk_sub_sat_u16(c, hi), k_bias);  // only bits 15 and 31 are read
            pd[h2] = pk_add_sat_u16(pk_sub_sat_u16(lo, c), k_bias);
            if (RETRY) {
                pb[h2] &= on[h2];
                pd[h2] &= on[h2];
            }
        }
        // Verdict byte of the work item: bit 2p + 1 = ring darker, bit 2p = ring brighter for pixel p of the
        // dword.  One v_perm_b32 per side gathers the four flag bytes, pixel 3 first; three bit-field inserts
        // fold the four bit pairs into the low byte (the other bits of the operands are never read).
        const uint32_t Bm = __builtin_amdgcn_perm(pb[1], pb[0], 0x01050307u);
        const uint32_t Dm = __builtin_amdgcn_perm(pd[1], pd[0], 0x01050307u);
        uint32_t t = bfi(0x80808080u, Bm, Dm >> 1);  // bits 7, 6 of every byte
        t = bfi(0xC0C0C0C0u, t, t >> 10);          // bytes 0 and 2: pairs at bits 7..4
        t = bfi(0x000000F0u, t, t >> 20);          // byte 0: all four pairs
        *pv = (uint8_t)t;
    };
    if (ndw <= 32) {
        // A tile row of at most 32 work items (four cells of up to 31 pixels: most tiles): each half of a wave takes
        // one row, lane by lane, and the four waves go down the tile eight rows at a time.  The item's places then
        // move by constants and a row's 32 dword reads fall on 32 different banks; nothing wraps.
        const int col = lane & 31, r0 = 2 * (tid >> 6) + (lane >> 5);
        if (col < ndw) {
            const uint8_t* pw = s_tile + r0 * TP + 4 * (col + j_lo);
            uint8_t* pv = s_verdict + r0 * vp + col + j_lo;
            for (int r = r0; r < hd; r += 8, pw += 8 * TP, pv += 8 * vp) test_item(pw, pv, col + j_lo);
        }
    } else {
        // (row, dword) of item tid, advanced by 256 items per iteration without divisions; the item's place in
        // the tile (row v - 3, the dword left of the item's) and in the verdict plane move along by additions
        const float inv_ndw = fast_rcp(ndw);
        const int v0 = (int)(((float)tid + 0.5f) * inv_ndw);
        int j = tid - v0 * ndw;  // relative to j_lo
        const int dv = (int)(256.5f * inv_ndw), dj = 256 - dv * ndw;  // 256 / ndw and 256 % ndw
        const uint8_t* pw = s_tile + v0 * TP + 4 * (j + j_lo);
        uint8_t* pv = s_verdict + v0 * vp + j + j_lo;
        // steps of the two places: without and with a wrap into the next row (selected per lane, no branch)
        const int da_t = dv * TP + 4 * dj, da_v = dv * vp + dj, da_tw = da_t + TP - 4 * ndw, da_vw = da_v + vp - ndw;
        for (int i0 = 0; i0 < n_items; i0 += 256) {
            if (tid < n_items - i0) test_item(pw, pv, j + j_lo);  // items past the end (last iteration only) do nothing
            j += dj;
            const bool wrap = j >= ndw;
            j -= wrap ? ndw : 0;
            pw += wrap ? da_tw : da_t;
            pv += wrap ? da_vw : da_v;
        }
    }
    __syncthreads();
    // ---- 1b. the verdict plane becomes the queue ----
    // A thread takes the verdicts of four neighbouring work items (16 pixels of one row, two sides) as one
    // dword; the wave's prefix sum of the hit counts (DPP scan) and ONE LDS add per wave give every lane its
    // first slot, and the lane then writes its hits one per iteration.  A wave whose reservation does not fit
    // skips its writes; the overflow is detected after the pass.
    {
        const int vp4 = vp >> 2, n_groups = hd * vp4;
        const float inv_vp4 = fast_rcp(vp4);
        const uint32_t* s_verdict32 = reinterpret_cast<const uint32_t*>(s_verdict);
        for (int g0 = 0; g0 < n_groups; g0 += 256) {
            const int g = g0 + tid;
            uint32_t w = 0;
            int ebase = 0;
            if (g < n_groups) {
                w = s_verdict32[g];
                const int vg = (int)(((float)g + 0.5f) * inv_vp4);
                ebase = (vg << 9) + 32 * (g - vg * vp4);  // bit k of w: pixel k >> 1 of the 16, side k & 1
            }
            const int cnt = __popc(w);
            const int incl = wave_incl_scan(cnt);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot == 0) continue;  // wave-uniform
            int base = 0;
            if (lane == 0) {
                uint32_t got;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(got)
                             : "v"((uint32_t)(uintptr_t)&sh->q_n), "v"((uint32_t)tot)
                             : "memory");
                base = (int)got;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + tot > q_cap) continue;  // wave-uniform
            uint16_t* q_p = s_queue + base + (incl - cnt);
            // eight hits per lane with the slot as an instruction offset, then whatever is left
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (!__builtin_amdgcn_ballot_w64(w != 0)) break;
                if (w) {
                    q_p[n] = (uint16_t)(ebase + __builtin_ctz(w));
                    w &= w - 1;
                }
            }
            q_p += 8;
            while (__builtin_amdgcn_ballot_w64(w != 0)) {
                if (w) {
                    *q_p++ = (uint16_t)(ebase + __builtin_ctz(w));
                    w &= w - 1;
                }
            }
        }
    }
    __syncthreads();
    ORBX_STAMP(B, stamp_base + 0, t0);
#ifdef ORBX_FAST_STOP
    if (ORBX_FAST_STOP == 2) return;
#endif
    const int qn = sh->q_n;
    if (qn > q_cap) {
        // queue overflow (a tile of dense noise): score every pixel on both sides, no survivors list;
        // the output stage then finds the maxima by scanning the score plane
        for (int i = tid; i < wd * hd; i += 256) {
            const int vq = (int)(((float)i + 0.5f) * fast_rcp(wd)), u = i - vq * wd;
            const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
            if (RETRY && sh->th[c] < 0) continue;
            const uint8_t* p = s_tile + (vq + 3) * TP + 4 + u;
            const int cv = p[0];
            const int r[16] = {p[3 * TP], p[3 * TP + 1], p[2 * TP + 2], p[TP + 3], p[3], p[-TP + 3], p[-2 * TP + 2], p[-3 * TP + 1],
                               p[-3 * TP], p[-3 * TP - 1], p[-2 * TP - 2], p[-TP - 3], p[-3], p[TP - 3], p[2 * TP - 2], p[3 * TP - 1]};
            int d[16], g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                d[k] = cv - r[k];
                g[k] = r[k] - cv;
            }
            const int S = max(arc_max(d), arc_max(g));
            if (S > thr) s_score[(vq + 1) * SP + 1 + u + c] = (uint8_t)(S - 1);
        }
        __syncthreads();
        // count the maxima per cell (the retry decision needs the counts)
        for (int i = tid; i < wd * hd; i += 256) {
            const int vq = (int)(((float)i + 0.5f) * fast_rcp(wd)), u = i - vq * wd;
            const int c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
            if (RETRY && sh->th[c] < 0) continue;
            const uint8_t* sp = s_score + (vq + 1) * SP + 1 + u + c;
            const int sc = sp[0];
            if (sc > 0) {
                const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                                  max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
                if (sc > m) atomicAdd(&sh->kept[c], 1);
            }
        }
        if (tid == 0) sh->n_kept = 1 << 24;  // forces the scanning output path
        __syncthreads();
        return;
    }
    // ---- 2. full score of the queued pixels ----
    // The queue is cut into runs of 64 entries and the runs dealt to the four waves.
    const int corner_room = max(0, min(k_cap, k_cap - kept_before));
    for (int q0 = 64 * (tid >> 6); q0 < qn; q0 += 256)
        fast_score_wave<RETRY>(s_tile, s_score, s_queue, sh, q0, qn, TP, SP, wd, nc, inv_wcell, thr, s_keptl + k_cap - 1, corner_room);
    __syncthreads();
    ORBX_STAMP(B, stamp_base + 1, t0);
#ifdef ORBX_FAST_STOP
    if (ORBX_FAST_STOP == 3) return;
#endif
    // ---- 3. non-max suppression; survivors are listed and counted per cell ----
    // A pixel is a corner on one side at most (two arcs of 9 do not fit a ring of 16), so the corner
    // list holds no pixel twice.  It is used when it and the survivors it can add both fit beside the
    // survivors already listed; else step 3 walks the whole queue.
    const int nco = sh->n_corner;
    if (kept_before + 2 * nco <= k_cap) {
        for (int q0 = 0; q0 < nco; q0 += 256) {
            const int q = q0 + tid;
            bool keep = false;
            int ke = 0, c = 0;
            if (q < nco) {
                ke = s_keptl[k_cap - 1 - q];
                c = ke >> 14;
                const uint8_t* sp = s_score + ((ke >> 8 & 0x3F) + 1) * SP + 1 + (ke & 0xFF) + c;
                const int sc = sp[0];
                const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                                  max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
                keep = sc > m;
            }
            const uint64_t m = __ballot(keep);
            if (m) {
                int base = 0;
                if (lane == 0) {  // a plain LDS add by one lane for the wave
                    uint32_t got;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(got)
                                 : "v"((uint32_t)(uintptr_t)&sh->n_kept), "v"((uint32_t)__popcll(m))
                                 : "memory");
                    base = (int)got;
                }
                base = __builtin_amdgcn_readfirstlane(base);
                if (keep) {
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    s_keptl[pos] = (uint16_t)ke;  // below the corner list: pos < kept_before + nco
                    atomicAdd(&sh->kept[c], 1);
                }
            }
        }
        __syncthreads();
        ORBX_STAMP(B, stamp_base + 2, t0);
        return;
    }
    for (int q0 = 0; q0 < qn; q0 += 256) {
        const int q = q0 + tid;
        bool keep = false;
        int ke = 0, c = 0;
        if (q < qn) {
            const int e = s_queue[q], u = (e >> 1) & 0xFF, vq = (e >> 9) & 0x3F;
            if (e & 0x8000) {  // step 2 found a corner on this entry's side: no pixel is listed twice
                c = min((int)(((float)u + 0.5f) * inv_wcell), nc - 1);
                const uint8_t* sp = s_score + (vq + 1) * SP + 1 + u + c;
                const int sc = sp[0];
                const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                                  max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
                keep = sc > m;
                ke = (c << 14) | ((e & 0x7FFF) >> 1);
            }
        }
        const uint64_t m = __ballot(keep);
        if (m) {
            int base = 0;
            if (lane == 0) {  // a plain LDS add by one lane for the wave
                uint32_t got;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(got)
                             : "v"((uint32_t)(uintptr_t)&sh->n_kept), "v"((uint32_t)__popcll(m))
                             : "memory");
                base = (int)got;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            if (keep) {
                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (pos < k_cap) s_keptl[pos] = (uint16_t)ke;
                atomicAdd(&sh->kept[c], 1);
            }
        }
    }
    __syncthreads();
    ORBX_STAMP(B, stamp_base + 2, t0);
}

// One workgroup per tile (a run of up to kFastTileCells cells of one cell row).
// grid = (tiles, frames), block = 256, dynamic LDS = tile + score plane + queue + kept list + verdict plane.
//  pass 0, threshold iniTh, all cells of the tile:
//   1. every detection pixel takes the compass-pair rejection test; the verdicts go to a byte plane, which
//      a second sweep (1b) compacts into an LDS queue of survivors;
//   2. queued pixels get the full segment-test score S; S > t stores score S-1 in a score plane
//      in which the cells are separated by a zero column, so that
//   3. strict 3x3 non-max suppression over the plane is cell-local exactly as when cv::FAST runs
//      on the cell's sub-image; survivors are listed and counted per cell;
//  pass 1, threshold minTh, only the cells whose pass-0 count is zero (the retry of
//      reference :903-907), same three steps;
//  then the survivors are written to their cell's slot in row-major order: each one's position
//  is its rank among the listed survivors (the list key is cell | row | column).
// TPC: the LDS row pitch of the tile as a compile-time constant (every tile of the launch then uses it, and the
// row offsets of the ring and of the rejection test's neighbours are instruction immediates), or 0: each tile
// uses the smallest pitch that fits it.
template <int TPC>
__global__ __launch_bounds__(256) void k_fast_tiles(BatchDev B, const TileDesc* __restrict__ tiles,
                                                    uint32_t* __restrict__ cand, uint32_t* __restrict__ ccount,
                                                    int tile_bytes, int plane_bytes, int queue_bytes, int kept_bytes,
                                                    int verdict_bytes, int frame_major) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    __shared__ FastShared sh;

    uint8_t* s_tile = s_mem;
    uint8_t* s_score = s_mem + tile_bytes;
    uint16_t* s_queue = reinterpret_cast<uint16_t*>(s_score + plane_bytes);
    uint16_t* s_keptl = reinterpret_cast<uint16_t*>(s_score + plane_bytes + queue_bytes);
    uint8_t* s_verdict = s_score + plane_bytes + queue_bytes + kept_bytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid = (tiles, frames), or (frames of a group of 8, tiles, groups) when the frames are dealt over
    // the XCDs (workgroups go to the XCDs round-robin in x-fastest order: all tiles of a frame then meet in one L2)
    const int tile_i = frame_major ? blockIdx.y : blockIdx.x;
    const int frame = frame_major ? blockIdx.z * gridDim.x + blockIdx.x : blockIdx.y;
    if (frame >= B.nframes) return;
    const TileDesc td = tiles[tile_i];
    const LevelDev& L = B.lv[td.level];
#ifdef ORBX_DIAG_FAST_READS_FRAME0  // diagnostic build: every frame's tiles are staged from frame 0 (cache hits)
    const uint8_t* img = L.img;
#else
    const uint8_t* img = L.img + (uint64_t)frame * L.plane;
#endif
    const int wd = td.wd, hd = td.hd, nc = td.nc, w_cell = td.w_cell;
    const int TP = TPC ? TPC : fast_tile_pitch(wd);
    const int SP = fast_plane_pitch(wd, nc);
    const int rows = hd + 6, y_top = td.y_first - 3, x_left = td.x_first - 4;
    const float inv_wcell = fast_rcp(w_cell);
    const int q_cap = fast_queue_cap(wd, hd), k_cap = fast_kept_cap(wd, hd);
    unsigned long long t0 = 0;
#ifdef ORBX_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif

    // stage the tile: 16-byte chunk c of an LDS row <- the 16 (unaligned) global bytes at
    // x_left + 16c.  All of a thread's loads are issued before the first LDS write, so a tile
    // costs one memory round trip, not one per chunk.  The loads go through a buffer resource over the frame's
    // level: a 32-bit offset per lane (24-bit multiply-adds) instead of 64-bit address arithmetic.
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, L.h * L.pitch, 0x00020000);
        const int nch = fast_tile_pitch(wd) >> 4, total = rows * nch;  // the chunks the tile needs, whatever the pitch
        const int row_end = L.pitch - 16;  // last chunk start that keeps the load inside the row
        const float inv_nch = fast_rcp(nch);
        for (int i0 = 0; i0 < total; i0 += 256 * kFastStageDepth) {
            u32x4 reg[kFastStageDepth];
            int dst[kFastStageDepth];
#pragma unroll
            for (int k = 0; k < kFastStageDepth; ++k) {
                const int i = i0 + 256 * k + tid;
                dst[k] = -1;
                if (i < total) {
                    const int r = (int)(((float)i + 0.5f) * inv_nch), c = i - __mul24(r, nch);
                    const int col = min(x_left + 16 * c, row_end);
                    reg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, __mul24(y_top + r, L.pitch) + col, 0, 0);
                    dst[k] = __mul24(r, TP) + 16 * c;
                }
            }
#pragma unroll
            for (int k = 0; k < kFastStageDepth; ++k)
                if (dst[k] >= 0) *reinterpret_cast<u32x4*>(&s_tile[dst[k]]) = reg[k];
        }
    }
    // score plane and verdict plane start at zero (16 bytes per store: every size of the carve-up is a multiple of 16).
    // The padding items at the end of each verdict row are never written by step 1 and must read as "no hit".
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        for (int i = tid; i < plane_bytes >> 4; i += 256) reinterpret_cast<u32x4*>(s_score)[i] = u32x4{0, 0, 0, 0};
        for (int i = tid; i < verdict_bytes >> 4; i += 256) reinterpret_cast<u32x4*>(s_verdict)[i] = u32x4{0, 0, 0, 0};
    }
    if (tid < kFastTileCells) {
        sh.kept[tid] = 0;
        sh.th[tid] = 0;
    }
    if (tid == 0) {
        sh.retry = 0;
        sh.n_kept = 0;
    }
    __syncthreads();
    ORBX_STAMP(B, 0, t0);
#ifdef ORBX_FAST_STOP  // diagnostic build: the kernel ends after phase ORBX_FAST_STOP (cost breakdown by differences)
#define ORBX_FAST_STOP_AT(k)                                                                               \
    if (ORBX_FAST_STOP == (k)) {                                                                           \
        if (tid < nc) ccount[(uint64_t)frame * B.total_cells + td.slot0 + tid] = 0;                        \
        return;                                                                                            \
    }
#else
#define ORBX_FAST_STOP_AT(k)
#endif
    ORBX_FAST_STOP_AT(1)

    fast_pass<false>(s_tile, s_score, s_queue, s_keptl, s_verdict, &sh, TP, SP, wd, hd, nc, inv_wcell, B.ini_th, q_cap, k_cap, B, 1, 0,
                     (wd + 3) >> 2);
#ifdef ORBX_FAST_STOP
    if (ORBX_FAST_STOP == 2 || ORBX_FAST_STOP == 3) {
        if (tid < nc) ccount[(uint64_t)frame * B.total_cells + td.slot0 + tid] = 0;
        return;
    }
#endif
    if (tid < nc) {
        const bool retry = sh.kept[tid] == 0;
        sh.th[tid] = retry ? 0 : -1;
        if (retry) sh.retry = 1;
    }
    __syncthreads();
    ORBX_FAST_STOP_AT(4)
    if (sh.retry) {
        // the dword columns from the first to the last cell to retry
        int c_first = nc, c_last = 0;
        for (int c = nc - 1; c >= 0; --c)
            if (sh.th[c] >= 0) c_first = c;
        for (int c = 0; c < nc; ++c)
            if (sh.th[c] >= 0) c_last = c;
        const int j_lo = (c_first * w_cell) >> 2, j_hi = (min((c_last + 1) * w_cell, wd) + 3) >> 2;
        fast_pass<true>(s_tile, s_score, s_queue, s_keptl, s_verdict, &sh, TP, SP, wd, hd, nc, inv_wcell, B.min_th, q_cap, k_cap, B, 4,
                        j_lo, j_hi - j_lo);
    }
    ORBX_FAST_STOP_AT(5)

    // ---- ordered output ----
#ifdef ORBX_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif
    const uint64_t slot0 = (uint64_t)frame * B.total_cells + td.slot0;
    const int nk = sh.n_kept;
    const int nb = nc * hd;  // (cell, row) pairs of the tile
    if (nk <= 256 && nk <= k_cap && nb <= 256 && nb * 12 + 16 <= queue_bytes) {
        // A survivor's place in its cell's slot is its rank in (cell, row, column) order.  Every (cell, row) pair keeps a
        // 64-bit mask of the columns (cell-local, below 64) that hold a survivor, in the memory of the queue, which nobody
        // reads any more: the survivors set their bits, a block-wide prefix sum over the masks' bit counts gives every
        // pair its first rank, and a survivor adds the bits below its own.  (Counting, for every survivor, the listed
        // keys below its own was 2.5 vector instructions per pair of survivors: a third of the kernel's instructions
        // at 135 survivors per tile.)
        unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_queue);
        int* s_start = reinterpret_cast<int*>(s_mask + nb);
        int* s_wave4 = s_start + nb;
        if (tid < nb) s_mask[tid] = 0ull;
        __syncthreads();
        int e = 0, c = 0, b = 0, ucell = 0;
        if (tid < nk) {
            e = s_keptl[tid];
            c = e >> 14;
            ucell = (e & 0xFF) - c * w_cell;
            b = c * hd + ((e >> 8) & 0x3F);
            atomicOr(&s_mask[b], 1ull << ucell);
        }
        __syncthreads();
        int total;
        const int excl = block_excl_scan_256(tid < nb ? __popcll(s_mask[tid]) : 0, s_wave4, &total);
        if (tid < nb) s_start[tid] = excl;
        __syncthreads();
        if (tid < nk) {
            const int vq = (e >> 8) & 0x3F, u = e & 0xFF;
            const int pos = s_start[b] - s_start[c * hd] + __popcll(s_mask[b] & ((1ull << ucell) - 1ull));
            if (pos < B.cell_cap)
                cand[(slot0 + c) * B.cell_cap + pos] = pack_pt(td.off_x + u, td.off_y + vq, s_score[(vq + 1) * SP + 1 + u + c]);
        }
        if (tid < nc) ccount[slot0 + tid] = (uint32_t)min(sh.kept[tid], B.cell_cap);
    } else if (nk <= 256 && nk <= k_cap) {
        // (tiles whose queue memory cannot hold the masks: a lone narrow cell at the end of a cell row)
        // rank among the listed survivors = position in (cell, row, column) order
        if (tid < nk) {
            const int e = s_keptl[tid];
            int rank = 0;
            for (int jx = 0; jx < nk; ++jx) rank += (int)s_keptl[jx] < e ? 1 : 0;
            const int c = e >> 14, vq = (e >> 8) & 0x3F, u = e & 0xFF;
            int start = 0;
            for (int cc = 0; cc < c; ++cc) start += sh.kept[cc];
            const int pos = rank - start;
            if (pos < B.cell_cap)
                cand[(slot0 + c) * B.cell_cap + pos] = pack_pt(td.off_x + u, td.off_y + vq, s_score[(vq + 1) * SP + 1 + u + c]);
        }
        if (tid < nc) ccount[slot0 + tid] = (uint32_t)min(sh.kept[tid], B.cell_cap);
    } else {
        // dense tiles: wave w scans cells w, w+4, ... in row-major order and repeats the suppression test
        for (int c = wave; c < nc; c += 4) {
            const int cwd = c < nc - 1 ? w_cell : wd - (nc - 1) * w_cell;
            const int npx = cwd * hd;
            const float inv_cwd = fast_rcp(cwd);
            uint32_t* out = cand + (slot0 + c) * B.cell_cap;
            int base = 0;
            for (int i0 = 0; i0 < npx; i0 += 64) {
                const int i = i0 + lane;
                bool keep = false;
                int px = 0, py = 0, sc = 0;
                if (i < npx) {
                    py = (int)(((float)i + 0.5f) * inv_cwd);
                    px = i - py * cwd;
                    const uint8_t* sp = s_score + (py + 1) * SP + 1 + c * (w_cell + 1) + px;
                    sc = sp[0];
                    if (sc > 0) {
                        const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-SP - 1], (int)sp[-SP])),
                                          max(max((int)sp[-SP + 1], (int)sp[SP - 1]), max((int)sp[SP], (int)sp[SP + 1])));
                        keep = sc > m;
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < B.cell_cap) out[pos] = pack_pt(td.off_x + c * w_cell + px, td.off_y + py, sc);
                }
                base += __popcll(m);
            }
            if (lane == 0) ccount[slot0 + c] = (uint32_t)min(base, B.cell_cap);
        }
    }
    ORBX_STAMP(B, 7, t0);
}

}  // namespace orbx
#endif
