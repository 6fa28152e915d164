// Orientation (IC_Angle, reference src/ORBextractor.cc:78-106), rBRIEF descriptor
// (computeOrbDescriptor, :112-152) and output assembly (operator(), :1172-1210): one wave per
// retained keypoint.
#ifndef ORBX_KERNELS_DESCRIBE_HPP
#define ORBX_KERNELS_DESCRIBE_HPP

#include "orbx_device.h"

namespace orbx {

// cv::fastAtan2 (OpenCV 3.4.5 scalar form), degrees in [0, 360).  IEEE float32 with separate
// multiply and add (this file is compiled with -ffp-contract=off).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float r2d = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * r2d, p3 = -0.3258083974640975f * r2d;
    const float p5 = 0.1555786518463281f * r2d, p7 = -0.04432655554792128f * r2d;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cos and sin of a float32 radian angle in [0, ~2 pi]: double-precision evaluation with a fixed
// operation sequence (range reduction by multiples of pi/2 with a two-term constant, then the
// fdlibm kernel polynomials on |r| <= pi/4, all through explicit fma), rounded once to float32.
// The sequence uses only IEEE-754 basic operations, so it is reproducible bit for bit.
__device__ __forceinline__ void sincos_pinned(float angf, float* c_out, float* s_out) {
    const double x = (double)angf;
    const double kd = floor(fma(x, 6.36619772367581382433e-01, 0.5));
    double r = fma(-kd, 1.57079632673412561417e+00, x);
    r = fma(-kd, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int k = (int)kd & 3;
    const double ss = (k == 0) ? s : (k == 1) ? c : (k == 2) ? -s : -c;
    const double cc = (k == 0) ? c : (k == 1) ? -s : (k == 2) ? -c : s;
    *c_out = (float)cc;
    *s_out = (float)ss;
}

constexpr int kIcCount = 749;  // pixels of the radius-15 disc

// grid = 8 * ceil(frames/8) * nlevels * kp_blocks (1-D), block = 256 (one wave per keypoint).
// Workgroups are dealt round-robin to the 8 XCDs, so block ids are decoded as
// (id % 8, id / 8) = (XCD lane, work index) and every frame is given to one XCD lane: all
// keypoints of a frame then gather from the same L2 instead of pulling the frame's pyramid into
// eight of them.  This is a locality hint only; correctness does not depend on placement.
__global__ __launch_bounds__(256) void k_orient_describe(BatchDev B, const uint32_t* __restrict__ lvl_kps,
                                                         const int32_t* __restrict__ lvl_count,
                                                         const int8_t* __restrict__ ic_uv,   // 749 (u, v) pairs
                                                         const int8_t* __restrict__ pattern, // 512 (x, y) pairs
                                                         orbx_kp* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                         int32_t* __restrict__ out_n, int kp_cap, int kp_blocks) {
    __shared__ int8_t s_pat[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_frame = B.nlevels * kp_blocks;
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int frame = (j / per_frame) * 8 + xl;
    if (frame >= B.nframes) return;
    const int rem = j % per_frame;
    const int level = rem / kp_blocks, kblk = rem % kp_blocks;
    const int32_t* counts = lvl_count + (uint64_t)frame * B.nlevels;
    const int n = counts[level];
    if (kblk == 0 && level == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < B.nlevels; ++l) tot += counts[l];
        out_n[frame] = min(tot, kp_cap);
    }
    if (kblk * 4 >= n) return;  // whole block idle (uniform)
    reinterpret_cast<uint32_t*>(s_pat)[tid] = reinterpret_cast<const uint32_t*>(pattern)[tid];
    __syncthreads();
    const int i = kblk * 4 + wave;
    if (i >= n) return;
    const LevelDev& L = B.lv[level];
    int base = 0;
    for (int l = 0; l < level; ++l) base += counts[l];
    const int o = base + i;
    if (o >= kp_cap) return;

    const uint32_t pk = lvl_kps[(uint64_t)frame * B.kp_total + L.kp_off + i];
    const int x = pt_x(pk) + kBorder, y = pt_y(pk) + kBorder;  // level coordinates (:936-937)

    // intensity centroid over the disc, on the un-blurred level
    const uint8_t* c = L.img + (uint64_t)frame * L.plane + (uint64_t)y * L.pitch + x;
    int m10 = 0, m01 = 0;
    int uu[12], vv[12], val[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {  // 749 = 11*64 + 45: all loads issued before the first use
        const int jj = min(lane + 64 * t, kIcCount - 1);
        const int16_t uv = reinterpret_cast<const int16_t*>(ic_uv)[jj];
        uu[t] = (int8_t)(uv & 0xFF);
        vv[t] = (int8_t)(uv >> 8);
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) val[t] = c[vv[t] * L.pitch + uu[t]];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const int ok = (lane + 64 * t) < kIcCount ? 1 : 0;
        m10 += ok * uu[t] * val[t];
        m01 += ok * vv[t] * val[t];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m10 += __shfl_xor(m10, d);
        m01 += __shfl_xor(m01, d);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // rotated BRIEF tests on the blurred level; lane handles tests 4*lane .. 4*lane+3
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    sincos_pinned(__fmul_rn(angle, factor_pi), &a, &b);
    const uint8_t* cb = L.blur + (uint64_t)frame * L.bplane + (uint64_t)y * L.bpitch + x;
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int8_t* pp = &s_pat[(4 * lane + t) * 4];
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float px = (float)pp[2 * e], py = (float)pp[2 * e + 1];
            // the reference's -march=native build contracts x*b + y*a and x*a - y*b into one fma
            const int row = (int)rintf(__fmaf_rn(px, b, __fmul_rn(py, a)));
            const int col = (int)rintf(__fmaf_rn(px, a, -__fmul_rn(py, b)));
            val[e] = cb[row * L.bpitch + col];
        }
        bits |= (val[0] < val[1] ? 1u : 0u) << t;
    }
    const uint32_t other = __shfl_xor(bits, 1);
    uint8_t* dsc = out_desc + ((uint64_t)frame * kp_cap + o) * 32;
    if ((lane & 1) == 0) dsc[lane >> 1] = (uint8_t)(bits | (other << 4));
    if (lane == 0) {
        orbx_kp k;
        const float fx = (float)x, fy = (float)y;
        k.x = level ? __fmul_rn(fx, L.scale) : fx;  // :1198-1207
        k.y = level ? __fmul_rn(fy, L.scale) : fy;
        k.size = L.size;
        k.angle = angle;
        k.response = (float)pt_score(pk);
        k.octave = level;
        k.class_id = -1;
        out_kps[(uint64_t)frame * kp_cap + o] = k;
    }
}

}  // namespace orbx
#endif
