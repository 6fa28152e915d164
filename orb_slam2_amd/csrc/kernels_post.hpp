// Right after the extractor: undistorted keypoints (Frame::UndistortKeyPoints, reference
// src/Frame.cc:421-451, calling cv::undistortPoints) and the 64 x 48 frame grid
// (Frame::AssignFeaturesToGrid + PosInGrid, reference src/Frame.cc:232-247, 398-408).
#ifndef ORBX_KERNELS_POST_HPP
#define ORBX_KERNELS_POST_HPP

#include "kernels_fast.hpp"
#include "orbx_device.h"

namespace orbx {

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;  // include/Frame.h:37-38

struct PostParams {
    float fx, fy, cx, cy;       // mK
    float k1, k2, p1, p2, k3;   // mDistCoef
    float min_x, min_y, inv_w, inv_h;  // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
};

// cv::undistortPoints(pts, pts, K, D, Mat(), K) of OpenCV 3.4.5: five iterations of the inverse
// distortion model in double precision, statements in OpenCV's order (the zero-valued rational
// and thin-prism terms are kept so that every intermediate matches).
__device__ __forceinline__ void undistort_point(const PostParams& P, float xin, float yin, float* xo, float* yo) {
    const double fx = P.fx, fy = P.fy, cx = P.cx, cy = P.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = P.k1, k1 = P.k2, k2 = P.p1, k3 = P.p2, k4 = P.k3;
    double x = xin, y = yin;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx;
    const double yy = 0. * x + fy * y + cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

__device__ __forceinline__ int grid_cell(const PostParams& P, float x, float y) {
    const int px = (int)roundf(__fmul_rn(__fsub_rn(x, P.min_x), P.inv_w));  // :400-401
    const int py = (int)roundf(__fmul_rn(__fsub_rn(y, P.min_y), P.inv_h));
    if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) return -1;
    return px * kGridRows + py;
}

// One workgroup per frame.  kps_un receives the keypoints with undistorted coordinates
// (mvKeysUn); cell_start[kGridCells + 1] / cell_idx[] are the grid in CSR form, indices of a cell
// in ascending order (the reference's push_back order).
// grid = frames, block = 256.
__global__ __launch_bounds__(256) void k_undistort_grid(PostParams P, const orbx_kp* __restrict__ kps,
                                                        const int32_t* __restrict__ counts, int kp_cap,
                                                        orbx_kp* __restrict__ kps_un, int32_t* __restrict__ cell_start,
                                                        int32_t* __restrict__ cell_idx) {
    __shared__ uint32_t s_cnt[kGridCells + 1];
    __shared__ int s_wave4[4];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n = min(counts[f], kp_cap);
    const orbx_kp* in = kps + (size_t)f * kp_cap;
    orbx_kp* out = kps_un + (size_t)f * kp_cap;
    int32_t* cs = cell_start + (size_t)f * (kGridCells + 1);
    int32_t* ci = cell_idx + (size_t)f * kp_cap;
    for (int i = tid; i <= kGridCells; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const bool distorted = P.k1 != 0.0f;  // src/Frame.cc:423
    for (int i = tid; i < n; i += 256) {
        orbx_kp k = in[i];
        if (distorted) undistort_point(P, k.x, k.y, &k.x, &k.y);
        out[i] = k;
        const int c = grid_cell(P, k.x, k.y);
        if (c >= 0) atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    {
        const int per = (kGridCells + 1 + 255) >> 8;
        const int beg = min(tid * per, kGridCells + 1), end = min(beg + per, kGridCells + 1);
        uint32_t sum = 0;
        for (int i = beg; i < end; ++i) sum += s_cnt[i];
        int total;
        uint32_t run = (uint32_t)block_excl_scan_256((int)sum, s_wave4, &total);
        for (int i = beg; i < end; ++i) {
            const uint32_t v = s_cnt[i];
            s_cnt[i] = run;  // becomes the cell's write cursor
            cs[i] = (int32_t)run;
            run += v;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = grid_cell(P, out[i].x, out[i].y);
        if (c >= 0) ci[atomicAdd(&s_cnt[c], 1u)] = i;
    }
    __syncthreads();
    // a cell's entries were appended in arbitrary order: sort each (short) cell list ascending
    for (int c = tid; c < kGridCells; c += 256) {
        const int b = cs[c], e = cs[c + 1];
        for (int i = b + 1; i < e; ++i) {
            const int v = ci[i];
            int j = i - 1;
            while (j >= b && ci[j] > v) {
                ci[j + 1] = ci[j];
                --j;
            }
            ci[j + 1] = v;
        }
    }
}

// Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682) with the depth map's conversion of
// Tracking::GrabImageRGBD (src/Tracking.cc:230-231: imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor), one float32
// multiply per sample) applied to the sampled values only: feature i reads the depth at (int)v, (int)u of its
// keypoint (cv::Mat::at<float>(float, float) truncates), d > 0 gives mvDepth[i] = d and
// mvuRight[i] = mvKeysUn[i].pt.x - mbf / d, else both are -1.
struct RgbdDev {
    const void* depth;      // [n] frames of `height` rows, `stride` bytes apart; 16-bit unsigned or float32 samples
    size_t frame_stride;    // bytes
    int is_f32, width, height, stride;
    float factor, mbf;
    const orbx_kp *kps, *kps_un;  // [n][kp_cap]
    const int32_t* n_kp;          // [n]
    int kp_cap;
    float *u_right, *depth_out;   // [n][kp_cap]
};

// grid = (ceil(kp_cap / 256), frames), block = 256: one thread per feature slot.
__global__ __launch_bounds__(256) void k_rgbd_depth(RgbdDev D) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D.kp_cap) return;
    float ur = -1.0f, z = -1.0f;  // :664-665
    if (i < D.n_kp[f]) {
        const orbx_kp kp = D.kps[(size_t)f * D.kp_cap + i];
        const int x = (int)kp.x, y = (int)kp.y;  // :672-675
        if (x >= 0 && x < D.width && y >= 0 && y < D.height) {  // (every keypoint lies inside its image)
            const uint8_t* row = (const uint8_t*)D.depth + (size_t)f * D.frame_stride + (size_t)y * D.stride;
            const float raw = D.is_f32 ? reinterpret_cast<const float*>(row)[x] : (float)reinterpret_cast<const uint16_t*>(row)[x];
            const float d = __fmul_rn(raw, D.factor);
            if (d > 0.0f) {  // :677-681
                z = d;
                ur = __fsub_rn(D.kps_un[(size_t)f * D.kp_cap + i].x, __fdiv_rn(D.mbf, d));
            }
        }
    }
    D.u_right[(size_t)f * D.kp_cap + i] = ur;
    D.depth_out[(size_t)f * D.kp_cap + i] = z;
}

}  // namespace orbx
#endif
