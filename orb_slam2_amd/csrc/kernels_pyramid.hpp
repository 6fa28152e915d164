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
// taps 18,34,49,55,49,34,18 (/256, sum 257); row pass exact in u16; column pass
// min(255, (sum + 32768) >> 16).
// One workgroup blurs a 64x32 tile: stage 70x38 source bytes (reflected at the level's own
// borders), row pass into a 64x38 u16 tile, column pass from LDS, dword stores.
// grid = (ceil(w/64), ceil(h/32), frames), block = 256.
constexpr int kBlurTW = 64, kBlurTH = 32;
__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ src, int spitch, uint64_t splane,
                                               uint8_t* __restrict__ dst, int dpitch, uint64_t dplane,
                                               int w, int h) {
    __shared__ uint8_t s_in[kBlurTH + 6][kBlurTW + 8];     // 38 x 72
    __shared__ uint16_t s_row[kBlurTH + 6][kBlurTW];       // 38 x 64
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kBlurTW, y0 = blockIdx.y * kBlurTH;
    const uint8_t* S = src + (uint64_t)blockIdx.z * splane;
    for (int i = tid; i < (kBlurTH + 6) * (kBlurTW + 6); i += 256) {
        const int r = i / (kBlurTW + 6), c = i - r * (kBlurTW + 6);
        const int yy = reflect101(min(y0 + r - 3, h + 2), h);
        const int xx = reflect101(min(x0 + c - 3, w + 2), w);
        s_in[r][c] = S[(uint64_t)yy * spitch + xx];
    }
    __syncthreads();
    for (int i = tid; i < (kBlurTH + 6) * kBlurTW; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t* p = &s_in[r][c];
        const int v = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3];
        s_row[r][c] = (uint16_t)v;
    }
    __syncthreads();
    uint8_t* D = dst + (uint64_t)blockIdx.z * dplane;
    for (int i = tid; i < kBlurTH * (kBlurTW / 4); i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        const int y = y0 + r, x = x0 + c;
        if (y >= h || x >= w) continue;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t v = 18u * (s_row[r][c + k] + s_row[r + 6][c + k]) +
                               34u * (s_row[r + 1][c + k] + s_row[r + 5][c + k]) +
                               49u * (s_row[r + 2][c + k] + s_row[r + 4][c + k]) + 55u * s_row[r + 3][c + k];
            out |= min(255u, (v + 32768u) >> 16) << (8 * k);
        }
        uint8_t* d = D + (uint64_t)y * dpitch + x;
        if (x + 3 < w) {
            *reinterpret_cast<uint32_t*>(d) = out;
        } else {
            for (int k = 0; x + k < w; ++k) d[k] = (uint8_t)(out >> (8 * k));
        }
    }
}

}  // namespace orbx
#endif
