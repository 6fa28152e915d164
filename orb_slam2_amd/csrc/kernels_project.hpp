// The projections in front of the windowed searches: Frame::isInFrustum (reference src/Frame.cc:271-327, called
// per local map point from Tracking::SearchLocalPoints, src/Tracking.cc:1340-1353) with MapPoint::PredictScale
// (src/MapPoint.cc:402-417); the first half of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:881-927),
// which Fuse(pKF, Scw, ...) (:1046-1086) and the loop closer's SearchByProjection(pKF, Scw, ...) (:321-360) repeat
// with a pose taken out of a similarity; and the projection of the relocalisation's SearchByProjection
// (:1534-1563); and the two-step transform of SearchBySim3 (:1195-1226, :1275-1306).  One thread per map point: 68 bytes in, 56 bytes out, which are the records the local-map search,
// the Fuse search and the guided search take, so a batch of poses and points goes from the map to matches without
// a round trip through the host.
//
// Arithmetic as the reference's cv::Mat expressions evaluate it: Rcw * P + tcw in float, products summed left to
// right; cv::norm and cv::Mat::dot of three floats accumulate in double; 1.0f / z, the pixel coordinates and
// P - Ow in float; the viewing cosine is the double quotient narrowed to float.
#ifndef ORBX_KERNELS_PROJECT_HPP
#define ORBX_KERNELS_PROJECT_HPP

#include "kernels_fast.hpp"

namespace orbx {

// log(x) for a positive finite float, computed in double from IEEE operations only (no library call), so that
// the host restatement, which performs the same operations in the same order, agrees bit for bit:
// x = m * 2^e with m in (sqrt(1/2), sqrt(2)], s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + s^4/5 + ...),
// thirteen terms (s^2 <= 0.0295, the first dropped term is below 1e-20).  The error is a few 1e-16, so the
// float it is narrowed to is the correctly rounded logarithm except within 1e-8 ulp of a rounding boundary.
__device__ __forceinline__ double log_pinned(float x) {
    const double xd = (double)x;
    long long bits = __double_as_longlong(xd);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll;
    double m = __longlong_as_double(bits);  // [1, 2)
    if (m > 1.4142135623730951) {
        m = __dmul_rn(m, 0.5);
        e += 1;
    }
    const double s = __ddiv_rn(__dsub_rn(m, 1.0), __dadd_rn(m, 1.0));
    const double z = __dmul_rn(s, s);
    double p = 1.0 / 25.0;
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 23.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 21.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 19.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 17.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 15.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 13.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 11.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 9.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 7.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 5.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0 / 3.0);
    p = __dadd_rn(__dmul_rn(p, z), 1.0);
    return __dadd_rn(__dmul_rn((double)e, 0.6931471805599453), __dmul_rn(__dmul_rn(2.0, s), p));
}

// MapPoint::PredictScale (src/MapPoint.cc:402-417): ceil(log(mfMaxDistance / dist) / mfLogScaleFactor) in float,
// clamped to the pyramid.  A ratio that is not positive and finite gives a logarithm the reference's conversion to
// int turns into INT_MIN on x86-64, hence level 0.
__device__ __forceinline__ int predict_scale(float max_dist, float dist, float log_scale_factor, int nlevels) {
    const float ratio = __fdiv_rn(max_dist, dist);
    if (!(ratio > 0.0f) || ratio > 3.4028234663852886e38f) return 0;
    const float lg = (float)log_pinned(ratio);
    const float q = ceilf(__fdiv_rn(lg, log_scale_factor));
    if (!(q >= 0.0f) || q >= 2147483648.0f) return 0;  // negative, not a number, or beyond int (INT_MIN again)
    if (q >= (float)nlevels) return nlevels - 1;
    return (int)q;
}

struct ProjectDev {
    const orbx_frustum_point* points;  // [n][q_cap]
    const int32_t* nq;                 // [n]
    const float* poses;                // [n][12]: rows of [Rcw | tcw]; kProjSim3: [n][24], [R1w | t1w] then [sR21 | t21]
    const float* centres;              // [n][3]: mOw / Ow (not read by kProjSim3)
    const float* angles;               // [n][q_cap] or null: the angle a guided record carries
    void* out;                         // [n][q_cap] orbx_local_query, orbx_window_query or orbx_guided_query
    int32_t* n_out;                    // [n], zeroed before the launch
    int q_cap, nlevels;
    float fx, fy, cx, cy, mbf, min_x, max_x, min_y, max_y;
    float view_cos_limit, log_scale_factor, th;
    float scale[ORBX_MAX_LEVELS];
};

enum : int {
    kProjFrustum = 0,  // Frame::isInFrustum -> orbx_local_query (flags bit 1 set for a point skipped or out of view)
    kProjFuse = 1,     // ORBmatcher.cc:881-927 -> orbx_window_query (radius -1 for a point skipped or dropped: an empty window)
    kProjLoop = 2,     // ORBmatcher.cc:321-360, the same tests -> orbx_guided_query (flags 1, or 3 for a point skipped or dropped)
    kProjReloc = 3,    // ORBmatcher.cc:1534-1563: no depth and no viewing test, the Frame's closed bounds -> orbx_guided_query
    kProjSim3 = 4      // ORBmatcher.cc:1195-1226: through the first keyframe's pose, then the similarity between the cameras;
                       // the distance is that of the camera-frame point, no viewing test -> orbx_window_query
};

// grid = (ceil(q_cap / 256), frames), block = 256.
// Every lane reads and writes its own record; staging the records through LDS for dword-consecutive global
// accesses measured slower (0.048 against 0.041 ms for 512 x 2000 points): the lines are shared in the cache.
template <int MODE>
__global__ __launch_bounds__(256) void k_project_points(ProjectDev D) {
    constexpr bool FUSE = MODE == kProjFuse || MODE == kProjLoop;  // the tests of ORBmatcher.cc:885-921
    constexpr bool KF = FUSE || MODE == kProjSim3;                 // x / z first, KeyFrame::IsInImage
    __shared__ float s_scale[ORBX_MAX_LEVELS];
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < ORBX_MAX_LEVELS) s_scale[threadIdx.x] = D.scale[threadIdx.x];
    __syncthreads();
    const int nq = max(0, min(D.nq[f], D.q_cap));
    bool in_view = false;
    if (i < nq) {
        const size_t at = (size_t)f * D.q_cap + i;
        const orbx_frustum_point pt = D.points[at];
        const float* T = D.poses + (size_t)f * (MODE == kProjSim3 ? 24 : 12);
        float u = 0.0f, v = 0.0f, ur = 0.0f, view_cos = 0.0f;
        int level = 0;
        in_view = !(pt.flags & 2);
        float pc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float t = __fadd_rn(__fadd_rn(__fmul_rn(T[4 * r], pt.x), __fmul_rn(T[4 * r + 1], pt.y)), __fmul_rn(T[4 * r + 2], pt.z));
            pc[r] = __fadd_rn(t, T[4 * r + 3]);
        }
        if (MODE == kProjSim3) {  // p3Dc2 = sR21 * p3Dc1 + t21 (:1197)
            const float* S = T + 12;
            float p2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float t = __fadd_rn(__fadd_rn(__fmul_rn(S[4 * r], pc[0]), __fmul_rn(S[4 * r + 1], pc[1])), __fmul_rn(S[4 * r + 2], pc[2]));
                p2[r] = __fadd_rn(t, S[4 * r + 3]);
            }
            pc[0] = p2[0], pc[1] = p2[1], pc[2] = p2[2];
        }
        if (MODE != kProjReloc && pc[2] < 0.0f) in_view = false;  // Frame.cc:285, ORBmatcher.cc:885
        // ORBmatcher.cc:1539 and :1056 divide in double and narrow: the same float (53 >= 2 * 24 + 2 bits)
        const float invz = __fdiv_rn(1.0f, pc[2]);
        if (KF) {  // ORBmatcher.cc:888-897, KeyFrame::IsInImage (src/KeyFrame.cc:616-619)
            u = __fadd_rn(__fmul_rn(D.fx, __fmul_rn(pc[0], invz)), D.cx);
            v = __fadd_rn(__fmul_rn(D.fy, __fmul_rn(pc[1], invz)), D.cy);
            if (!(u >= D.min_x && u < D.max_x && v >= D.min_y && v < D.max_y)) in_view = false;
        } else {  // Frame.cc:289-296, ORBmatcher.cc:1541-1547
            u = __fadd_rn(__fmul_rn(__fmul_rn(D.fx, pc[0]), invz), D.cx);
            v = __fadd_rn(__fmul_rn(__fmul_rn(D.fy, pc[1]), invz), D.cy);
            if (u < D.min_x || u > D.max_x) in_view = false;
            if (v < D.min_y || v > D.max_y) in_view = false;
        }
        ur = __fsub_rn(u, __fmul_rn(D.mbf, invz));  // Frame.cc:321, ORBmatcher.cc:899
        const float max_d = __fmul_rn(1.2f, pt.max_dist), min_d = __fmul_rn(0.8f, pt.min_dist);  // MapPoint.cc:373-383
        float po[3] = {pc[0], pc[1], pc[2]};  // kProjSim3: cv::norm(p3Dc2) (:1216)
        if (MODE != kProjSim3) {
            const float* O = D.centres + (size_t)f * 3;
            po[0] = __fsub_rn(pt.x, O[0]), po[1] = __fsub_rn(pt.y, O[1]), po[2] = __fsub_rn(pt.z, O[2]);
        }
        double n2 = __dmul_rn((double)po[0], (double)po[0]);
        n2 = __dadd_rn(n2, __dmul_rn((double)po[1], (double)po[1]));
        n2 = __dadd_rn(n2, __dmul_rn((double)po[2], (double)po[2]));
        const float dist = (float)__dsqrt_rn(n2);
        if (dist < min_d || dist > max_d) in_view = false;  // Frame.cc:304, ORBmatcher.cc:915
        double dot = __dmul_rn((double)po[0], (double)pt.nx);
        dot = __dadd_rn(dot, __dmul_rn((double)po[1], (double)pt.ny));
        dot = __dadd_rn(dot, __dmul_rn((double)po[2], (double)pt.nz));
        if (FUSE) {
            if (dot < __dmul_rn(0.5, (double)dist)) in_view = false;  // ORBmatcher.cc:921
        } else if (MODE == kProjFrustum) {
            view_cos = (float)__ddiv_rn(dot, (double)dist);  // Frame.cc:310
            if (view_cos < D.view_cos_limit) in_view = false;
        }
        if (in_view) level = predict_scale(pt.max_dist, dist, D.log_scale_factor, D.nlevels);
        if (MODE == kProjLoop || MODE == kProjReloc) {
            orbx_guided_query q;
            q.u = in_view ? u : 0.0f;
            q.v = in_view ? v : 0.0f;
            q.angle = D.angles ? D.angles[at] : 0.0f;
            q.level = level;
            q.radius = in_view ? __fmul_rn(D.th, s_scale[level]) : -1.0f;  // ORBmatcher.cc:360, :1563
            q.flags = in_view ? 1 : 3;
            memcpy(q.desc, pt.desc, 32);
            reinterpret_cast<orbx_guided_query*>(D.out)[at] = q;
        } else if (MODE == kProjFuse || MODE == kProjSim3) {
            orbx_window_query q;
            q.u = in_view ? u : 0.0f;
            q.v = in_view ? v : 0.0f;
            q.radius = in_view ? __fmul_rn(D.th, s_scale[level]) : -1.0f;  // ORBmatcher.cc:927
            q.min_level = level - 1;                                       // :948
            q.max_level = level;
            q.ur = in_view ? ur : 0.0f;
            memcpy(q.desc, pt.desc, 32);
            reinterpret_cast<orbx_window_query*>(D.out)[at] = q;
        } else {
            orbx_local_query q;
            q.u = in_view ? u : 0.0f;
            q.v = in_view ? v : 0.0f;
            q.ur = in_view ? ur : 0.0f;
            q.level = level;
            q.view_cos = in_view ? view_cos : 0.0f;
            q.flags = (pt.flags & 1) | (in_view ? 0 : 2);
            memcpy(q.desc, pt.desc, 32);
            reinterpret_cast<orbx_local_query*>(D.out)[at] = q;
        }
    }
    const unsigned long long m = __ballot(in_view);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&D.n_out[f], __popcll(m));
}

}  // namespace orbx
#endif
