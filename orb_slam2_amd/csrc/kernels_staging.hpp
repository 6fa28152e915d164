// Input staging in front of the extractor: colour -> gray (cv::cvtColor in Tracking::GrabImage*,
// reference src/Tracking.cc:175-199, 213-231, 250-268) and rectification (cv::remap with float maps,
// reference Examples/Stereo/stereo_euroc.cc:137-138), OpenCV 3.4.5 8-bit fixed-point arithmetic.
// Both are streaming per-pixel kernels over a batch of equally sized frames in HBM.
#ifndef ORBX_KERNELS_STAGING_HPP
#define ORBX_KERNELS_STAGING_HPP

#include "orbx_device.h"

namespace orbx {

// gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14.  A thread converts four pixels and stores one dword.
// grid = (ceil(ceil(w/4)/256), h, frames), block = 256.
template <int CH>
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, int sstride, uint64_t splane,
                                                  uint8_t* __restrict__ dst, int dstride, uint64_t dplane, int w, int blue_idx) {
    const int x0 = 4 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y;
    if (x0 >= w) return;
    const uint8_t* S = src + (uint64_t)blockIdx.z * splane + (uint64_t)y * sstride + (uint64_t)x0 * CH;
    uint8_t* D = dst + (uint64_t)blockIdx.z * dplane + (uint64_t)y * dstride + x0;
    const int nv = min(4, w - x0);
    uint32_t out = 0;
    if (nv == 4) {
        // the 12 or 16 source bytes of four pixels as (unaligned) dwords
        uint32_t q[CH];
        if (CH == 4) {
            const uint4 v = *reinterpret_cast<const uint4_unaligned*>(S);
            q[0] = v.x, q[1] = v.y, q[2] = v.z, q[CH - 1] = v.w;
        } else {
#pragma unroll
            for (int d = 0; d < CH; ++d) q[d] = *reinterpret_cast<const u32_unaligned*>(S + 4 * d);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto byte_at = [&](int i) { return (int)((q[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
            const int b = byte_at(k * CH + blue_idx), g = byte_at(k * CH + 1), r = byte_at(k * CH + (blue_idx ^ 2));
            out |= (uint32_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14) << (8 * k);
        }
    } else {
        for (int k = 0; k < nv; ++k) {
            const int b = S[k * CH + blue_idx], g = S[k * CH + 1], r = S[k * CH + (blue_idx ^ 2)];
            out |= (uint32_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14) << (8 * k);
        }
    }
    if (nv == 4 && (dstride & 3) == 0 && (reinterpret_cast<uintptr_t>(D) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(D) = out;
    } else {
        for (int k = 0; k < nv; ++k) D[k] = (uint8_t)(out >> (8 * k));
    }
}

// cv::remap INTER_LINEAR, BORDER_CONSTANT 0: map coordinates rounded to 1/32 px (round half to
// even), weights (32-fx)(32-fy)*32 .. summing to 32768, result (sum + 16384) >> 15.
// The maps are shared by all frames.  A thread produces four destination pixels of one row.
// grid = (ceil(ceil(dw/4)/256), dh, frames), block = 256.
__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                                      uint64_t splane, const float* __restrict__ mapx,
                                                      const float* __restrict__ mapy, int map_stride,
                                                      uint8_t* __restrict__ dst, int dw, int dstride, uint64_t dplane) {
    const int x0 = 4 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y;
    if (x0 >= dw) return;
    const uint8_t* S = src + (uint64_t)blockIdx.z * splane;
    uint8_t* D = dst + (uint64_t)blockIdx.z * dplane + (uint64_t)y * dstride + x0;
    const float* MX = mapx + (uint64_t)y * map_stride + x0;
    const float* MY = mapy + (uint64_t)y * map_stride + x0;
    const int nv = min(4, dw - x0);
    float mx[4] = {0.f, 0.f, 0.f, 0.f}, my[4] = {0.f, 0.f, 0.f, 0.f};
    if (nv == 4) {
        const uint4 a = *reinterpret_cast<const uint4_unaligned*>(MX), b = *reinterpret_cast<const uint4_unaligned*>(MY);
        mx[0] = __uint_as_float(a.x), mx[1] = __uint_as_float(a.y), mx[2] = __uint_as_float(a.z), mx[3] = __uint_as_float(a.w);
        my[0] = __uint_as_float(b.x), my[1] = __uint_as_float(b.y), my[2] = __uint_as_float(b.z), my[3] = __uint_as_float(b.w);
    } else {
        for (int k = 0; k < nv; ++k) mx[k] = MX[k], my[k] = MY[k];
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nv) break;
        const int fsx = (int)rintf(__fmul_rn(mx[k], 32.0f)), fsy = (int)rintf(__fmul_rn(my[k], 32.0f));
        const int sx = min(max(fsx >> 5, -32768), 32767), sy = min(max(fsy >> 5, -32768), 32767);
        const int fx = fsx & 31, fy = fsy & 31;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int v;
        if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
            // the two taps of a row as one (unaligned) 16-bit load
            const uint8_t* p = S + (uint64_t)sy * sstride + sx;
            const uint32_t t0 = *reinterpret_cast<const u16_unaligned*>(p), t1 = *reinterpret_cast<const u16_unaligned*>(p + sstride);
            v = (int)(t0 & 0xFFu) * w00 + (int)(t0 >> 8) * w01 + (int)(t1 & 0xFFu) * w10 + (int)(t1 >> 8) * w11;
        } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
            continue;  // constant border 0
        } else {
            const bool bx0 = (unsigned)sx < (unsigned)sw, bx1 = (unsigned)(sx + 1) < (unsigned)sw;
            const bool by0 = (unsigned)sy < (unsigned)sh, by1 = (unsigned)(sy + 1) < (unsigned)sh;
            const int v0 = (bx0 && by0) ? S[(int64_t)sy * sstride + sx] : 0;
            const int v1 = (bx1 && by0) ? S[(int64_t)sy * sstride + sx + 1] : 0;
            const int v2 = (bx0 && by1) ? S[(int64_t)(sy + 1) * sstride + sx] : 0;
            const int v3 = (bx1 && by1) ? S[(int64_t)(sy + 1) * sstride + sx + 1] : 0;
            v = v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11;
        }
        v = (v + (1 << 14)) >> 15;
        out |= (uint32_t)min(max(v, 0), 255) << (8 * k);
    }
    if (nv == 4 && (dstride & 3) == 0 && (reinterpret_cast<uintptr_t>(D) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(D) = out;
    } else {
        for (int k = 0; k < nv; ++k) D[k] = (uint8_t)(out >> (8 * k));
    }
}

}  // namespace orbx
#endif
