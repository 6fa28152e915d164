// Pyramid kernels: bilinear down-scale (ComputePyramid, reference src/ORBextractor.cc:1213-1250)
// and the 7x7 Gaussian (reference :1188).  Integer arithmetic only; results are bit-exact.
#ifndef ORBX_KERNELS_PYRAMID_HPP
#define ORBX_KERNELS_PYRAMID_HPP

#include "orbx_device.h"

namespace orbx {

// cv::resize(level l-1 -> level l, INTER_LINEAR), 8UC1 fixed-point path of OpenCV 3.4.5:
//   T(dx)  = S[sx]*a0 + S[sx+1]*a1                      (11-bit weights, int32)
//   D      = (((b0*(T0>>4))>>16) + ((b1*(T1>>4))>>16) + 2) >> 2
// A workgroup produces a 256 x 16 destination tile.  The source rows and columns the tile's taps
// touch are staged in LDS with 16-byte loads (one vector-memory instruction moves 1 KiB), the tap
// tables of the tile's columns and rows go to LDS as well, and the per-pixel gathers
// (S[sx], S[sx+1] of two rows) are LDS byte reads.  A thread produces four horizontally adjacent
// destination pixels of four rows and stores one dword per row.
// grid = (ceil(dw/256), ceil(dh/16), frames), block = 256,
// dynamic LDS = resize_lds_bytes(...) (source tile + 256 column taps + 16 row taps).
constexpr int kRsTW = 256, kRsTH = 16;

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t taps, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, taps), c, false);
}


__host__ __device__ inline int resize_src_pitch(int max_col_span) { return ((max_col_span + 15 + 16) & ~15); }

__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, int spitch, uint64_t splane,
                                                int sw, int sh, uint8_t* __restrict__ dst, int dpitch,
                                                uint64_t dplane, int dw, int dh, const ResizeTap* __restrict__ xt,
                                                const ResizeTap* __restrict__ yt, int lds_pitch, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rs[];
    uint8_t* s_src = s_rs;
    ResizeTap* s_xt = reinterpret_cast<ResizeTap*>(s_rs + lds_pitch * lds_rows);
    ResizeTap* s_yt = s_xt + kRsTW;
    const int tid = threadIdx.x;
    const int dx_first = blockIdx.x * kRsTW, dy_first = blockIdx.y * kRsTH;
    const int ncol = min(kRsTW, dw - dx_first), nrow = min(kRsTH, dh - dy_first);
    // source window of the tile: taps are monotone in the destination index
    const int c_lo = xt[dx_first].ofs & ~15, c_hi = xt[dx_first + ncol - 1].ofs1;
    const int r_lo = yt[dy_first].ofs, r_hi = yt[dy_first + nrow - 1].ofs1;
    const int nchunk = ((c_hi - c_lo) >> 4) + 1, nr = r_hi - r_lo + 1;
    const uint8_t* S = src + (uint64_t)blockIdx.z * splane;
    const bool wide = ((spitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    const float inv_chunk = 1.0f / (float)nchunk;
    if (wide) {
        for (int i = tid; i < nr * nchunk; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv_chunk), c = i - r * nchunk;
            const uint4 v = *reinterpret_cast<const uint4*>(S + (uint64_t)(r_lo + r) * spitch + c_lo + 16 * c);
            *reinterpret_cast<uint4*>(s_src + r * lds_pitch + 16 * c) = v;
        }
    } else {
        // rows are only 4-byte aligned (a caller's level-0 buffer used in place)
        for (int i = tid; i < nr * nchunk * 4; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv_chunk * 0.25f), c = i - r * nchunk * 4;
            const int col = min(c_lo + 4 * c, spitch - 4);
            const uint32_t v = *reinterpret_cast<const uint32_t*>(S + (uint64_t)(r_lo + r) * spitch + col);
            *reinterpret_cast<uint32_t*>(s_src + r * lds_pitch + 4 * c) = v;
        }
    }
    if (tid < ncol) {
        ResizeTap t = xt[dx_first + tid];
        t.ofs = (int16_t)(t.ofs - c_lo);
        t.ofs1 = (int16_t)(t.ofs1 - c_lo);
        s_xt[tid] = t;
    }
    if (tid < nrow) {
        ResizeTap t = yt[dy_first + tid];
        t.ofs = (int16_t)((t.ofs - r_lo) * lds_pitch);
        t.ofs1 = (int16_t)((t.ofs1 - r_lo) * lds_pitch);
        s_yt[tid] = t;
    }
    __syncthreads();
    const int lx = (tid & 63) * 4;
    if (lx >= ncol) return;
    ResizeTap tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tx[k] = s_xt[min(lx + k, ncol - 1)];
    uint8_t* D = dst + (uint64_t)blockIdx.z * dplane + dx_first + lx;
    // The eight source bytes a row contributes to the thread's four pixels lie within 8 bytes of the
    // first tap for every scale factor up to 2 (span 3*scale + 2).  They are then fetched as three
    // aligned dwords and shifted into two; one v_perm_b32 per pixel picks its two taps into the halves of a dword and
    // one v_dot2_u32_u16 against the packed weights gives S[sx]*a0 + S[sx+1]*a1.
    const int base = tx[0].ofs;
    const bool fits = (int)tx[3].ofs1 - base <= 7;
    const int base_al = base & ~3;
    const uint32_t base_sh = (uint32_t)(base & 3);
    uint32_t sel[4], coef[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sel[k] = (uint32_t)(tx[k].ofs - base) | 0x0C00u | ((uint32_t)(tx[k].ofs1 - base) << 16) | 0x0C000000u;
        coef[k] = (uint32_t)(uint16_t)tx[k].c0 | ((uint32_t)(uint16_t)tx[k].c1 << 16);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int ly = (tid >> 6) + 4 * rr;
        if (ly >= nrow) break;
        const ResizeTap ty = s_yt[ly];
        const uint8_t* r0 = s_src + ty.ofs;
        const uint8_t* r1 = s_src + ty.ofs1;
        uint32_t out = 0;
        if (fits) {
            // three aligned dwords per row, shifted into place (unaligned LDS reads are much slower)
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(r0 + base_al);
            const uint32_t* q1 = reinterpret_cast<const uint32_t*>(r1 + base_al);
            const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
            const uint32_t a_lo = __builtin_amdgcn_alignbyte(a1, a0, base_sh), a_hi = __builtin_amdgcn_alignbyte(a2, a1, base_sh);
            const uint32_t b_lo = __builtin_amdgcn_alignbyte(b1, b0, base_sh), b_hi = __builtin_amdgcn_alignbyte(b2, b1, base_sh);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t0 = (int)udot2(__builtin_amdgcn_perm(a_hi, a_lo, sel[k]), coef[k], 0u);
                const int t1 = (int)udot2(__builtin_amdgcn_perm(b_hi, b_lo, sel[k]), coef[k], 0u);
                // both terms are below 2^24; the result is in [0, 255] (the weights of an axis sum to 2048)
                const int v = (int)(__umul24((uint32_t)ty.c0, (uint32_t)(t0 >> 4)) >> 16) +
                              (int)(__umul24((uint32_t)ty.c1, (uint32_t)(t1 >> 4)) >> 16);
                out |= (uint32_t)((v + 2) >> 2) << (8 * k);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t0 = (int)__umul24(r0[tx[k].ofs], tx[k].c0) + (int)__umul24(r0[tx[k].ofs1], tx[k].c1);
                const int t1 = (int)__umul24(r1[tx[k].ofs], tx[k].c0) + (int)__umul24(r1[tx[k].ofs1], tx[k].c1);
                int v = (((ty.c0 * (t0 >> 4)) >> 16) + ((ty.c1 * (t1 >> 4)) >> 16) + 2) >> 2;
                v = min(max(v, 0), 255);
                out |= (uint32_t)v << (8 * k);
            }
        }
        // the last dword of a row is stored whole, as in k_resize_rows: the spill lands in the row's padding
        *reinterpret_cast<uint32_t*>(D + (uint64_t)(dy_first + ly) * dpitch) = out;
    }
}

// Streaming form of the same resize, used when every run of four destination columns reads at
// most eight consecutive source bytes (any scale factor up to 2) and the row taps never move
// backwards.  No LDS: a lane owns four adjacent destination columns of one frame and walks down
// a band of destination rows.  Per SOURCE row it loads its eight bytes once (one unaligned
// 8-byte buffer load, kRsPF rows ahead of their use), forms the four horizontal sums
// (v_perm_b32 + v_dot2_u32_u16 with the weights pre-multiplied by 16, masked to the value
// (T >> 4) << 8), and keeps them for the next source row; a destination row is emitted when its
// second source row has been formed, so a source row shared by two destination rows is
// filtered once.  (b * (T >> 4)) >> 16 is one v_mul_hi_u32_u24 of (b << 8) and ((T >> 4) << 8).
// The lanes of a row band are dealt over (frame, dword column) pairs in one sequence, so waves
// are full whatever the level width; all lanes of a workgroup share the band and with it the
// source-row schedule, which keeps the control flow scalar.
// grid = (ceil(frames * ceil(dw/4) / 256), ceil(dh / rows_per_item)), block = 256.
constexpr int kRsPF = 2;  // source rows in flight per lane (4: 3 % slower alone, 0.5 % in the overlapped step; 1, 3, 6, 8: slower as well)

__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b) {  // both below 2^24; a is wave-uniform (a scalar register operand)
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

// Row tap y as two scalars: x = ofs | ofs1 << 16, y = c0 | c1 << 16.  The index is wave-uniform;
// loading the entry as one 8-byte word lets the compiler use a scalar load.
__device__ __forceinline__ uint2 resize_row_tap(const ResizeTap* __restrict__ yt, int y) {
    const uint2 t = reinterpret_cast<const uint2*>(yt)[y];
    return make_uint2(__builtin_amdgcn_readfirstlane(t.x), __builtin_amdgcn_readfirstlane(t.y));
}

__global__ __launch_bounds__(256) void k_resize_rows(const uint8_t* __restrict__ src, int spitch, uint64_t splane,
                                                     uint8_t* __restrict__ dst, int dpitch, uint64_t dplane, int dw, int dh,
                                                     const ResizeTap* __restrict__ xt, const ResizeTap* __restrict__ yt,
                                                     int n_frames, int rows_per_item) {
    // 64 registers allocated although 29 are used: the waves of FAST, the blur and the descriptor kernel, which share
    // the SIMDs with this one, take 64 each, and a retiring 32-register wave leaves a hole none of them fits into
    // (0.5 % of the step; the occupancy limit of 8 waves per SIMD is the same either way).
    asm volatile("" ::: "v63");
    const int nd = (dw + 3) >> 2;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const bool live = item < n_frames * nd;
    const int f = live ? item / nd : 0;
    const int x0 = live ? 4 * (item - f * nd) : 0;
    ResizeTap tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tx[k] = xt[min(x0 + k, dw - 1)];
    // the eight bytes start at the first tap, or earlier when that would run over the row's end
    const int base = min((int)tx[0].ofs, spitch - 8);
    uint32_t sel[4], coef[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sel[k] = (uint32_t)(tx[k].ofs - base) | 0x0C00u | ((uint32_t)(tx[k].ofs1 - base) << 16) | 0x0C000000u;
        coef[k] = ((uint32_t)(uint16_t)tx[k].c0 << 4) | ((uint32_t)(uint16_t)tx[k].c1 << 20);
    }
    const int kOut = 0x7FFFFFF0;  // beyond the buffer: the hardware drops the access
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (int)(splane * (uint64_t)n_frames), 0x00020000);
    const int vsrc = live ? (int)((uint64_t)f * splane) + base : kOut;
    // stores through a buffer resource as well: the lane's offset is fixed, the row's offset is scalar (no 64-bit address
    // arithmetic per row), and a lane without work is out of range instead of masked off
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(dplane * (uint64_t)n_frames), 0x00020000);
    const int vdst = live ? (int)((uint64_t)f * dplane) + x0 : kOut;

    const int y0 = blockIdx.y * rows_per_item, y_end = min(y0 + rows_per_item, dh);
    // taps of the next two destination rows, fetched ahead of their use
    uint2 t_cur = resize_row_tap(yt, y0), t_nxt = resize_row_tap(yt, min(y0 + 1, dh - 1));
    const int s0 = (int)(t_cur.x & 0xFFFFu);
    const int s_last = (int)(resize_row_tap(yt, y_end - 1).x >> 16);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 q[kRsPF];
#pragma unroll
    for (int k = 0; k < kRsPF; ++k) q[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, vsrc, min(s0 + k, s_last) * spitch, 0);
    uint32_t Xp[4] = {0, 0, 0, 0};
    int y = y0;
    for (int sb = s0; sb <= s_last; sb += kRsPF) {
#pragma unroll
        for (int k = 0; k < kRsPF; ++k) {
            const int sr = sb + k;
            if (sr > s_last) break;
            const u32x2 v = q[k];
            q[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, vsrc, min(sr + kRsPF, s_last) * spitch, 0);
            uint32_t X[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) X[px] = udot2(__builtin_amdgcn_perm(v.y, v.x, sel[px]), coef[px], 0u) & 0x00FFFF00u;
            // The destination rows whose second source row this is: one at most, or two where the taps are clamped
            // to the image's last row (the launcher checks it).  Two conditional blocks and not a loop: vector-memory
            // loads and stores are counted by one in-order counter, and behind a loop with an unknown number of
            // stores the compiler must wait for every outstanding load before it may read the oldest one, which
            // empties the load pipeline once per source row.
            auto emit = [&]() {
                if (__builtin_expect((int)(t_cur.x & 0xFFFFu) == sr, 0)) {  // both taps on one row (the clamped last rows)
                    asm volatile("" ::: "memory");  // a scalar branch around four moves, not four selects on every row
#pragma unroll
                    for (int px = 0; px < 4; ++px) Xp[px] = X[px];
                }
                const uint32_t b0 = (t_cur.y & 0xFFFFu) << 8, b1 = (t_cur.y >> 16) << 8;
                uint32_t out = 0;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    // both products are below 2^24 >> 8; the sum is in [0, 1020 + 2]
                    const uint32_t t = mul_hi_u24(b0, Xp[px]) + mul_hi_u24(b1, X[px]) + 2u;
                    out |= (t >> 2) << (8 * px);
                }
                // the last dword of a row is stored whole: the row pitch is a multiple of 64, the columns beyond the
                // level's width are padding that nothing reads (they receive the last column's value), and a byte
                // store is a partial write of its own to the memory system
                __builtin_amdgcn_raw_buffer_store_b32(out, rd, vdst, y * dpitch, 0);
                ++y;
                t_cur = t_nxt;
                t_nxt = resize_row_tap(yt, min(y + 1, dh - 1));
            };
            if (y < y_end && (int)(t_cur.x >> 16) == sr) emit();
            if (y < y_end && (int)(t_cur.x >> 16) == sr) emit();
#pragma unroll
            for (int px = 0; px < 4; ++px) Xp[px] = X[px];
        }
    }
}

// Up to three pyramid levels in ONE launch, for a launch of a few frames, where the pyramid is seven kernels that
// wait for one another and do microseconds of work each.  A level's pixel is a function of four pixels of the level
// below, so level l+2 (and l+3) can be evaluated from level l directly by evaluating the pixels in between on the
// way -- the same arithmetic in the same order, hence the same bytes -- at 4x (16x) the loads, which a launch of one
// frame does not notice.  Thread = one destination pixel of one of the chain's levels.
struct ResizeChain {
    const uint8_t* src;           // level l, frames splane apart
    int spitch;
    uint64_t splane;
    int nlev;                     // levels produced: l+1 .. l+nlev (1..3)
    uint8_t* dst[3];
    int dpitch[3];
    uint64_t dplane[3];
    int dw[3], dh[3];
    const ResizeTap* xt[3];       // taps of level l+1+k into level l+k
    const ResizeTap* yt[3];
};

template <int K>
__device__ __forceinline__ int resize_chain_px(const ResizeChain& C, const uint8_t* S, int x, int y) {
    if constexpr (K < 0) {
        return S[(uint64_t)y * C.spitch + x];
    } else {
        const ResizeTap tx = C.xt[K][x], ty = C.yt[K][y];
        const int a = resize_chain_px<K - 1>(C, S, tx.ofs, ty.ofs), b = resize_chain_px<K - 1>(C, S, tx.ofs1, ty.ofs);
        const int c = resize_chain_px<K - 1>(C, S, tx.ofs, ty.ofs1), d = resize_chain_px<K - 1>(C, S, tx.ofs1, ty.ofs1);
        const int t0 = a * tx.c0 + b * tx.c1, t1 = c * tx.c0 + d * tx.c1;
        const int v = (((ty.c0 * (t0 >> 4)) >> 16) + ((ty.c1 * (t1 >> 4)) >> 16) + 2) >> 2;
        return min(max(v, 0), 255);
    }
}

// grid = (ceil(max pixels of a level / 256), nlev, frames), block = 256.
__global__ __launch_bounds__(256) void k_resize_chain(ResizeChain C) {
    const int k = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int w = C.dw[k];
    if (i >= w * C.dh[k]) return;
    const int y = i / w, x = i - y * w;
    const uint8_t* S = C.src + (uint64_t)blockIdx.z * C.splane;
    const int v = k == 0 ? resize_chain_px<0>(C, S, x, y) : k == 1 ? resize_chain_px<1>(C, S, x, y) : resize_chain_px<2>(C, S, x, y);
    C.dst[k][(uint64_t)blockIdx.z * C.dplane[k] + (uint64_t)y * C.dpitch[k] + x] = (uint8_t)v;
}

__host__ __device__ __forceinline__ int reflect101(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

// The pyramid as ComputePyramid leaves it in mvImagePyramid (reference src/ORBextractor.cc:1227-1245): every level
// inside a buffer that carries a `border`-pixel BORDER_REFLECT_101 rim (cv::copyMakeBorder), rows w + 2 * border
// bytes apart, the levels of one frame one after the other at dst + off[level].  The extractor itself never reads the
// rim; it exists for the callers that read the public member.  Thread = one byte of a padded level.
// grid = ceil(off[nlevels] / 256), block = 256.
struct PadDev {
    const uint8_t* src[ORBX_MAX_LEVELS];
    int32_t spitch[ORBX_MAX_LEVELS];
    int32_t w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS];
    uint32_t off[ORBX_MAX_LEVELS + 1];  // first byte of each padded level; [nlevels] = total
    uint8_t* dst;
    int32_t nlevels, border;
};

__global__ __launch_bounds__(256) void k_pad_levels(PadDev P) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P.off[P.nlevels]) return;
    int l = 0;
    while (l + 1 < P.nlevels && i >= P.off[l + 1]) ++l;
    const int pw = P.w[l] + 2 * P.border;
    const uint32_t r = i - P.off[l];
    const int y = (int)(r / (uint32_t)pw), x = (int)(r - (uint32_t)y * (uint32_t)pw);
    const int sy = reflect101(y - P.border, P.h[l]), sx = reflect101(x - P.border, P.w[l]);
    P.dst[i] = P.src[l][(uint64_t)sy * P.spitch[l] + sx];
}

// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8UC1, OpenCV 3.4.5 fixed-point path:
// taps 18,34,49,55,49,34,18 (/256, sum 257); row pass exact (<= 65535); column pass
// min(255, (sum + 32768) >> 16).
//
// Streaming form, all levels and frames in one launch: no LDS, register sliding window.
// A level is cut into row tiles of kBlurRows rows; the (row tile, dword column) pairs of a level
// are numbered linearly and dealt to lanes, so no lane idles at a strip end.  A lane owns four
// adjacent columns of its row tile and walks down it: per source row it loads its own dword
// (kBlurPF rows ahead of its use, through a buffer resource), takes the dwords to its left and
// right (pixels x0-4 .. x0-1 and x0+4 .. x0+7) from the neighbouring lanes, forms
// the four row sums with v_alignbyte_b32 + v_dot4_u32_u8 (taps 18,34,49,55 | 49,34,18,0), keeps
// the row sums of the last six rows packed in pairs and evaluates the column pass with
// v_dot2_u32_u16, storing one dword per row.  Rows are reflected by index.  In the first and the
// last two dword columns of a row, whose 7-tap windows leave the image, the neighbour dwords (and the
// last dword's bytes beyond the width) are rebuilt from the reflected columns by byte permutations
// (blur_rim_selector); only the waves that hold such a lane run that variant of the loop.
// grid = (B.blur_first[nlevels], frames), block = 256; or, frame_major, (frames of a group of 8,
// B.blur_first[nlevels], groups): workgroups go to the XCDs round-robin in x-fastest order, so the
// blocks of a frame, which share halo rows and neighbour columns, then meet in one L2.
#ifndef ORBX_BLUR_ROWS
#define ORBX_BLUR_ROWS 24
#endif
constexpr int kBlurRows = ORBX_BLUR_ROWS;
#ifndef ORBX_BLUR_PF
#define ORBX_BLUR_PF 8
#endif
constexpr int kBlurPF = ORBX_BLUR_PF;
constexpr int kBlurItems = 4 * 62;  // (row tile, dword column) pairs per streaming block: 62 per wave

// Column pass for one pixel at row r.  pr[j % 6] holds the row sums of rows (j, j+1) packed as
// two u16; the window rows r-6 .. r are pr[(r-6)%6], pr[(r-4)%6], pr[(r-2)%6] and the fresh sum.
__device__ __forceinline__ uint32_t blur_col(const uint32_t (&pr)[6][4], uint32_t w_r, int r, int k) {
    uint32_t v = __umul24(18u, w_r) + 32768u;
    v = udot2(pr[(r + 4) % 6][k], 49u | (34u << 16), v);  // rows r-2, r-1
    v = udot2(pr[(r + 2) % 6][k], 49u | (55u << 16), v);  // rows r-4, r-3
    v = udot2(pr[r % 6][k], 18u | (34u << 16), v);        // rows r-6, r-5
    return v;  // the output pixel is min(255, v >> 16): see blur_pack
}

// min(255, v >> 16) of four column sums as the four bytes of a dword: one v_perm_b32 puts the high halves of
// two sums side by side, v_sat_pk_u8_i16 saturates both to bytes (the halves are in [0, 257]), a last
// v_perm_b32 joins the two byte pairs -- 5 instructions instead of a shift, a minimum and an insert per pixel.
__device__ __forceinline__ uint32_t blur_pack(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t s01, s23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(s01) : "v"(__builtin_amdgcn_perm(v1, v0, 0x07060302u)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(s23) : "v"(__builtin_amdgcn_perm(v3, v2, 0x07060302u)));
    return __builtin_amdgcn_perm(s23, s01, 0x05040100u);
}

// Blocks of one level.
__host__ __device__ inline int blur_main_blocks(int w, int h) {
    const int nd = (w + 3) >> 2, nty = (h + kBlurRows - 1) / kBlurRows;
    return (nd * nty + kBlurItems - 1) / kBlurItems;
}


// Selector of a v_perm_b32 that builds the dword of columns tb .. tb+3 of a row of width w, reflected at the image's
// sides (BORDER_REFLECT_101), from two dwords: `hi` holds columns hb .. hb+3 (selector values 4..7), `lo` columns
// lb .. lb+3 (values 0..3).  Columns no tap can reach (more than 3 outside the image) get byte 0 of `lo`.
__host__ __device__ __forceinline__ uint32_t blur_rim_selector(int tb, int hb, int lb, int w) {
    uint32_t sel = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = tb + k;
        uint32_t idx = 0;
        if (col >= -3 && col <= w + 2) {
            const int rc = reflect101(col, w);
            idx = (rc >= hb && rc < hb + 4) ? (uint32_t)(4 + rc - hb) : (uint32_t)((rc - lb) & 3);
        }
        sel |= idx << (8 * k);
    }
    return sel;
}

// The streaming part of k_blur7_all for one lane: walks down the lane's row tile.  INNER: every source row of
// the tile lies inside the image (see the call).
// RIM: some lane of the wave owns a dword column at the left or right rim of the image; its neighbour dwords (and, in
// the last column, the bytes beyond the image's width) are then rebuilt from the reflected columns by byte
// permutations with per-lane selectors (blur_rim_selectors; identity in every other lane).
template <bool INNER, bool RIM>
__device__ __forceinline__ void blur_stream(const __amdgpu_buffer_rsrc_t rs, const __amdgpu_buffer_rsrc_t rd, int y0, int h,
                                            int spitch, int dpitch, int vcol, int x0, bool store_ok, uint32_t sel_l,
                                            uint32_t sel_c, uint32_t sel_r, bool last_col) {
    const int kOut = 0x7FFFFFF0;
    // loads in flight: two fewer where both the row reflection and the rim selectors are live, to stay within 64 registers
    constexpr int PF = (!INNER && RIM) ? kBlurPF - 2 : kBlurPF;
    // INNER: offsets of the tile's first source row and first output dword (unsigned: a lane with nothing to
    // load or store starts at kOut and stays out of the buffer's range)
    const uint32_t vbase = (uint32_t)((y0 - 3) * spitch) + (uint32_t)vcol;
    const uint32_t sbase = store_ok ? (uint32_t)(y0 * dpitch + x0) : (uint32_t)kOut;
    uint32_t qC[PF];
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const int vo = INNER ? (int)(vbase + (uint32_t)(r * spitch)) : reflect101(min(y0 + r - 3, h + 2), h) * spitch + vcol;
        qC[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, 0);
    }
    uint32_t pr[6][4], wp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < kBlurRows + 6; ++r) {
        uint32_t C = qC[r % PF];
        uint32_t Lw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x138, 0xF, 0xF, true);  // lane - 1
        uint32_t R = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x130, 0xF, 0xF, true);   // lane + 1
        if (RIM) {
            const uint32_t z = last_col ? Lw : R;  // the last column's right neighbour is made of its own and its left dword
            const uint32_t l2 = __builtin_amdgcn_perm(C, Lw, sel_l), c2 = __builtin_amdgcn_perm(C, Lw, sel_c);
            R = __builtin_amdgcn_perm(z, C, sel_r);
            Lw = l2;
            C = c2;
        }
        if (r + PF < kBlurRows + 6) {
            const int vo = INNER ? (int)(vbase + (uint32_t)((r + PF) * spitch)) : reflect101(min(y0 + r + PF - 3, h + 2), h) * spitch + vcol;
            qC[r % PF] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, 0);
        }
        uint32_t wr[4];
#ifndef ORBX_BLUR_ALIGNBYTE  // (the other form: -DORBX_BLUR_ALIGNBYTE, measured 0.4 % slower in the step)
        // the seven taps of pixel k against the three dwords as they are, with the weights shifted instead of the
        // pixels: ten v_dot4_u32_u8 and no v_alignbyte_b32 (eight and six the other way)
        wr[0] = __builtin_amdgcn_udot4(C, 0x12223137u, __builtin_amdgcn_udot4(Lw, 0x31221200u, 0u, false), false);
        wr[1] = __builtin_amdgcn_udot4(R, 0x00000012u, __builtin_amdgcn_udot4(C, 0x22313731u, __builtin_amdgcn_udot4(Lw, 0x22120000u, 0u, false), false), false);
        wr[2] = __builtin_amdgcn_udot4(R, 0x00001222u, __builtin_amdgcn_udot4(C, 0x31373122u, __builtin_amdgcn_udot4(Lw, 0x12000000u, 0u, false), false), false);
        wr[3] = __builtin_amdgcn_udot4(R, 0x00122231u, __builtin_amdgcn_udot4(C, 0x37312212u, 0u, false), false);
#else
        const uint32_t a0 = __builtin_amdgcn_alignbyte(C, Lw, 1), b0 = __builtin_amdgcn_alignbyte(R, C, 1);
        const uint32_t a1 = __builtin_amdgcn_alignbyte(C, Lw, 2), b1 = __builtin_amdgcn_alignbyte(R, C, 2);
        const uint32_t a2 = __builtin_amdgcn_alignbyte(C, Lw, 3), b2 = __builtin_amdgcn_alignbyte(R, C, 3);
        wr[0] = __builtin_amdgcn_udot4(b0, 0x00122231u, __builtin_amdgcn_udot4(a0, 0x37312212u, 0u, false), false);
        wr[1] = __builtin_amdgcn_udot4(b1, 0x00122231u, __builtin_amdgcn_udot4(a1, 0x37312212u, 0u, false), false);
        wr[2] = __builtin_amdgcn_udot4(b2, 0x00122231u, __builtin_amdgcn_udot4(a2, 0x37312212u, 0u, false), false);
        wr[3] = __builtin_amdgcn_udot4(R, 0x00122231u, __builtin_amdgcn_udot4(C, 0x37312212u, 0u, false), false);
#endif
        if (r >= 6) {
            const int y = y0 + r - 6;
            const uint32_t o = blur_pack(blur_col(pr, wr[0], r, 0), blur_col(pr, wr[1], r, 1), blur_col(pr, wr[2], r, 2),
                                         blur_col(pr, wr[3], r, 3));
            const int vs = INNER ? (int)(sbase + (uint32_t)((r - 6) * dpitch)) : ((store_ok && y < h) ? y * dpitch + x0 : kOut);
            __builtin_amdgcn_raw_buffer_store_b32(o, rd, vs, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (r >= 1) pr[(r + 5) % 6][k] = __builtin_amdgcn_perm(wr[k], wp[k], 0x05040100u);  // rows (r-1, r)
            wp[k] = wr[k];
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur7_all(BatchDev B, int frame_major) {
    const int blk_all = frame_major ? blockIdx.y : blockIdx.x;
    const int frame = frame_major ? blockIdx.z * gridDim.x + blockIdx.x : blockIdx.y;
    if (frame >= B.nframes) return;
    int level = 0;
    while (level + 1 < B.nlevels && blk_all >= B.blur_first[level + 1]) ++level;
    const LevelDev& L = B.lv[level];
    const int blk = blk_all - B.blur_first[level];
    const int w = L.w, h = L.h, spitch = L.pitch, dpitch = L.bpitch;
#ifdef ORBX_DIAG_BLUR_READS_FRAME0  // diagnostic build: every frame's blur reads frame 0 (cache hits), an upper bound on what sharing a staged tile could save
    const uint8_t* S8 = L.img;
#else
    const uint8_t* S8 = L.img + (uint64_t)frame * L.plane;
#endif
    uint8_t* D8 = L.blur + (uint64_t)frame * L.bplane;
    const int tid = threadIdx.x;
    const int nd = (w + 3) >> 2, nty = (h + kBlurRows - 1) / kBlurRows;
    // A wave covers 62 consecutive pairs with lanes 1..62; lanes 0 and 63 load the pairs before and
    // after them and only serve as the left / right neighbour of lanes 1 and 62, so that every lane
    // loads one dword per row and takes the two dwords beside it from the adjacent lanes
    // (v_mov_b32_dpp wave_shr / wave_shl) instead of from memory.
    const int lane = tid & 63;
    const int g = blk * kBlurItems + (tid >> 6) * 62 + lane - 1;
    const int ty = g < 0 ? nty : (int)(((float)g + 0.5f) * (1.0f / (float)nd));  // g / nd, exact for g < 2^22
    const int dcol = g - ty * nd;
    const bool live = ty < nty;
    const int x0 = dcol * 4;
    const int y0 = ty * kBlurRows;
    // Buffer addressing: out-of-range accesses (idle lanes, lanes that own no full dword) are
    // dropped by the hardware range check.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(S8), 0, h * spitch, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(D8, 0, h * dpitch, 0x00020000);
    const int kOut = 0x7FFFFFF0;
    const bool store_ok = live && lane >= 1 && lane <= 62;
    const int vcol = live ? x0 : kOut;
    // Rim columns.  In the first dword column the left neighbour is columns 3, 2, 1 of the lane's own dword; in the last
    // one (which may reach into the row's padding: the pitch is a multiple of 64 and nothing reads the padding) the
    // bytes beyond the width and the right neighbour are reflected columns of the lane's own and its left dword; in
    // the one before, the right neighbour's bytes beyond the width come from the lane's own dword.
    const bool last_col = live && dcol == nd - 1;
    const bool rim = live && (dcol == 0 || dcol >= nd - 2);
    uint32_t sel_l = 0, sel_c = 0, sel_r = 0;
    // Row tiles whose 38 source rows all lie inside the image need no reflection and no row checks: their row
    // offsets are the lane's first offset plus a scalar multiple of the pitch (about a fifth fewer vector
    // instructions).  The choice is made per wave; idle lanes go along with either path.
    const bool inner = !live || (y0 >= 3 && y0 + kBlurRows + 2 < h);
    const bool all_inner = __builtin_amdgcn_ballot_w64(!inner) == 0, any_rim = __builtin_amdgcn_ballot_w64(rim) != 0;
    if (any_rim) {  // three waves in five hold no rim lane and skip the selectors
        sel_l = blur_rim_selector(x0 - 4, x0, x0 - 4, w);
        sel_c = blur_rim_selector(x0, x0, x0 - 4, w);
        sel_r = blur_rim_selector(x0 + 4, last_col ? x0 - 4 : x0 + 4, x0, w);
    }
    if (all_inner && !any_rim)
        blur_stream<true, false>(rs, rd, y0, h, spitch, dpitch, vcol, x0, store_ok, 0, 0, 0, false);
    else if (all_inner)
        blur_stream<true, true>(rs, rd, y0, h, spitch, dpitch, vcol, x0, store_ok, sel_l, sel_c, sel_r, last_col);
    else if (!any_rim)
        blur_stream<false, false>(rs, rd, y0, h, spitch, dpitch, vcol, x0, store_ok, 0, 0, 0, false);
    else
        blur_stream<false, true>(rs, rd, y0, h, spitch, dpitch, vcol, x0, store_ok, sel_l, sel_c, sel_r, last_col);
}

}  // namespace orbx
#endif
