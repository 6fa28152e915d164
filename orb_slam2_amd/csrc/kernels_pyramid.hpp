// Pyramid kernels: bilinear down-scale (ComputePyramid, reference src/ORBextractor.cc:1213-1250)
// and the 7x7 Gaussian (reference :1188).  Integer arithmetic only; results are bit-exact.
#ifndef ORBX_KERNELS_PYRAMID_HPP
#define ORBX_KERNELS_PYRAMID_HPP

#include "orbx_device.h"

namespace orbx {

// cv::resize(level l-1 -> level l, INTER_LINEAR), 8UC1 fixed-point path of OpenCV 3.4.5:
//   T(dx)  = S[sx]*a0 + S[sx+1]*a1                      (11-bit weights, int32)
//   D      = (((b0*(T0>>4))>>16) + ((b1*(T1>>4))>>16) + 2) >> 2
// One thread produces four horizontally adjacent destination pixels and stores one dword.
// grid = (ceil(dw/256), ceil(dh/4), frames), block = (64, 4).
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, int spitch, uint64_t splane,
                                                uint8_t* __restrict__ dst, int dpitch, uint64_t dplane,
                                                int dw, int dh, const ResizeTap* __restrict__ xt,
                                                const ResizeTap* __restrict__ yt) {
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= dh || dx0 >= dw) return;
    const ResizeTap ty = yt[dy];
    const uint8_t* r0 = src + (uint64_t)blockIdx.z * splane + (uint64_t)ty.ofs * spitch;
    const uint8_t* r1 = src + (uint64_t)blockIdx.z * splane + (uint64_t)ty.ofs1 * spitch;
    const int b0 = ty.c0, b1 = ty.c1;
    uint32_t out = 0;
    const int nvalid = min(4, dw - dx0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nvalid) {
            const ResizeTap tx = xt[dx0 + k];
            const int t0 = r0[tx.ofs] * tx.c0 + r0[tx.ofs1] * tx.c1;
            const int t1 = r1[tx.ofs] * tx.c0 + r1[tx.ofs1] * tx.c1;
            int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            out |= (uint32_t)v << (8 * k);
        }
    }
    uint8_t* d = dst + (uint64_t)blockIdx.z * dplane + (uint64_t)dy * dpitch + dx0;
    if (nvalid == 4) {
        *reinterpret_cast<uint32_t*>(d) = out;
    } else {
        for (int k = 0; k < nvalid; ++k) d[k] = (uint8_t)(out >> (8 * k));
    }
}

__device__ __forceinline__ int reflect101(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8UC1, OpenCV 3.4.5 fixed-point path:
// taps 18,34,49,55,49,34,18 (/256, sum 257); row pass exact (<= 65535); column pass
// min(255, (sum + 32768) >> 16).
//
// Streaming form: no LDS, register sliding window.  A thread owns four adjacent columns and walks
// down kBlurRows output rows.  Per source row it loads three aligned dwords (pixels x0-4 .. x0+7,
// kBlurPF rows ahead of their use), forms the four row sums with v_alignbyte_b32 + v_dot4_u32_u8
// (taps 18,34,49,55 | 49,34,18,0), keeps the last seven row sums per column in registers and
// stores one dword per row.  Rows are reflected by index (wave-uniform, scalar arithmetic).
// Dwords holding a column whose 7-tap window leaves the image (columns [0,4) and the last 3..6)
// are not stored by the streaming blocks; the extra "edge" block column
// (blockIdx.x == gridDim.x-1) computes them through a small LDS tile of row sums.
// grid = (ceil(w/256) + 1, ceil(h / (4*kBlurRows)), frames), block = (64, 4).
constexpr int kBlurRows = 32;
constexpr int kBlurPF = 4;

__device__ __forceinline__ uint32_t blur_col(const uint32_t (&wn)[7][4], int r, int k) {
    // window rows r-6 .. r live in slots (r-6)%7 .. r%7
    // 24-bit multiplies (full rate): every operand is below 2^18
    const uint32_t v = __umul24(18u, wn[(r + 1) % 7][k] + wn[r % 7][k]) + __umul24(34u, wn[(r + 2) % 7][k] + wn[(r + 6) % 7][k]) +
                       __umul24(49u, wn[(r + 3) % 7][k] + wn[(r + 5) % 7][k]) + __umul24(55u, wn[(r + 4) % 7][k]);
    return min(255u, (v + 32768u) >> 16);
}

__device__ __forceinline__ uint32_t gauss_tap(int j) {
    const int k = j < 4 ? j : 6 - j;
    return k == 0 ? 18u : k == 1 ? 34u : k == 2 ? 49u : 55u;
}

__global__ __launch_bounds__(256) void k_blur7_stream(const uint8_t* __restrict__ src, int spitch, uint64_t splane,
                                                      uint8_t* __restrict__ dst, int dpitch, uint64_t dplane,
                                                      int w, int h) {
    const uint8_t* S8 = src + (uint64_t)blockIdx.z * splane;
    uint8_t* D8 = dst + (uint64_t)blockIdx.z * dplane;
    // streaming lanes store dwords x0 in [4, last_full]; the edge block owns [0,4) and [xs, w)
    const int last_full = ((w - 7) >> 2) << 2;
    const int xs = last_full + 4;
    if (blockIdx.x == gridDim.x - 1) {
        constexpr int kRows = 4 * kBlurRows;                 // output rows per block
        __shared__ uint16_t s_sum[kRows + 6][10];
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const int yb = blockIdx.y * kRows;
        const int nright = w - xs;                           // 3..6
        if (tid < kRows + 6) {
            const int yy = reflect101(min(yb + tid - 3, h + 2), h);
            const uint8_t* row = S8 + (uint64_t)yy * spitch;
            for (int e = 0; e < 10; ++e) {
                const int x = e < 4 ? e : xs + (e - 4);
                uint32_t rs = 0;
                if (e < 4 + nright) {
#pragma unroll
                    for (int t = 0; t < 7; ++t) rs += gauss_tap(t) * row[reflect101(x + t - 3, w)];
                }
                s_sum[tid][e] = (uint16_t)rs;
            }
        }
        __syncthreads();
        if (tid < kRows && yb + tid < h) {
            uint8_t* drow = D8 + (uint64_t)(yb + tid) * dpitch;
            for (int e = 0; e < 4 + nright; ++e) {
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 7; ++j) acc += gauss_tap(j) * s_sum[tid + j][e];
                drow[e < 4 ? e : xs + (e - 4)] = (uint8_t)min(255u, (acc + 32768u) >> 16);
            }
        }
        return;
    }
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * kBlurRows;
    if (y0 >= h) return;  // wave-uniform
    const int ndw = spitch >> 2;
    const int c1 = min(x0 >> 2, ndw - 1), c0 = max(c1 - 1, 0), c2 = min(c1 + 1, ndw - 1);
    const uint32_t* S = reinterpret_cast<const uint32_t*>(S8);
    const bool store_ok = x0 >= 4 && x0 <= last_full;
    uint32_t* D = reinterpret_cast<uint32_t*>(D8) + (x0 >> 2);
    const int ddw = dpitch >> 2;
    uint32_t qL[kBlurPF], qC[kBlurPF], qR[kBlurPF];
#pragma unroll
    for (int r = 0; r < kBlurPF; ++r) {
        const uint32_t* row = S + (uint64_t)reflect101(min(y0 + r - 3, h + 2), h) * ndw;
        qL[r] = row[c0];
        qC[r] = row[c1];
        qR[r] = row[c2];
    }
    uint32_t wn[7][4];
#pragma unroll
    for (int r = 0; r < kBlurRows + 6; ++r) {
        const uint32_t L = qL[r % kBlurPF], C = qC[r % kBlurPF], R = qR[r % kBlurPF];
        if (r + kBlurPF < kBlurRows + 6) {
            const uint32_t* row = S + (uint64_t)reflect101(min(y0 + r + kBlurPF - 3, h + 2), h) * ndw;
            qL[r % kBlurPF] = row[c0];
            qC[r % kBlurPF] = row[c1];
            qR[r % kBlurPF] = row[c2];
        }
        const uint32_t a0 = __builtin_amdgcn_alignbyte(C, L, 1), b0 = __builtin_amdgcn_alignbyte(R, C, 1);
        const uint32_t a1 = __builtin_amdgcn_alignbyte(C, L, 2), b1 = __builtin_amdgcn_alignbyte(R, C, 2);
        const uint32_t a2 = __builtin_amdgcn_alignbyte(C, L, 3), b2 = __builtin_amdgcn_alignbyte(R, C, 3);
        wn[r % 7][0] = __builtin_amdgcn_udot4(b0, 0x00122231u, __builtin_amdgcn_udot4(a0, 0x37312212u, 0u, false), false);
        wn[r % 7][1] = __builtin_amdgcn_udot4(b1, 0x00122231u, __builtin_amdgcn_udot4(a1, 0x37312212u, 0u, false), false);
        wn[r % 7][2] = __builtin_amdgcn_udot4(b2, 0x00122231u, __builtin_amdgcn_udot4(a2, 0x37312212u, 0u, false), false);
        wn[r % 7][3] = __builtin_amdgcn_udot4(R, 0x00122231u, __builtin_amdgcn_udot4(C, 0x37312212u, 0u, false), false);
        if (r >= 6) {
            const int y = y0 + r - 6;  // wave-uniform
            const uint32_t o = blur_col(wn, r, 0) | (blur_col(wn, r, 1) << 8) | (blur_col(wn, r, 2) << 16) |
                               (blur_col(wn, r, 3) << 24);
            if (y < h && store_ok) D[(uint64_t)y * ddw] = o;
        }
    }
}

}  // namespace orbx
#endif
