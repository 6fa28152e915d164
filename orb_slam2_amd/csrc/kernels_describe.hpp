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

constexpr int kPatchRows = 37, kPatchDw = 10, kPatchPitch = 40;  // blurred neighbourhood of a keypoint in LDS
// Dwords per lane.  Only the dwords the rotated pattern can reach are loaded: its farthest point is (13, 13), radius
// sqrt(338), and a sample that rounds to row n lies at least |n| - 0.5 from the centre row, so its column is at most
// rint(sqrt(338 - (|n| - 0.5)^2)): 13 columns in the outermost rows, all 37 in the middle ones, 310 dwords in all
// (build_describe_lane_table lists them, with 1 % added to the squared radius) instead of 370.
constexpr int kPatchLoads = 5;
constexpr int kDescPerWave = 4;              // keypoints a wave handles in turn
constexpr int kDescKp = 4 * kDescPerWave;    // keypoints per workgroup

// Wave-wide integer sum through DPP (no LDS traffic); the total lands in lane 63.
__device__ __forceinline__ int wave_sum_to_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return v;
}

typedef u32_unaligned desc_u32_unaligned;

// Lane constants of the kernel, the same for every workgroup: built once on the host and loaded as
// kDescTabVec 16-byte vectors per lane (vector v of lane l at tab[v * 64 + l]).
//   v0  wu[4]   phase A weight dwords u + 16 inside the disc, 0 outside      (lane = (row, half))
//   v1  w1[4]   phase A weight dwords 1 inside the disc, 0 outside
//   v2  v, column offset 16 * half - 15, 0, 0                                (v = row - 15)
//   v3, v4      x of the lane's eight pattern points (float32 bits)          (tests 4l .. 4l + 3)
//   v5, v6      y of the same points
//   v7, v8      kPatchLoads patch-staging entries: LDS offset | (row - 18) << 16 | (4 * dword - 18) << 24
constexpr int kDescTabVec = 9;
inline void build_describe_lane_table(const int8_t* pattern /* 512 (x, y) pairs */, uint32_t* tab /* [9][64][4] */) {
    // Row half-widths of the radius-15 disc (reference src/ORBextractor.cc:473-493).
    const int umax_tab[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int lane = 0; lane < 64; ++lane) {
        uint32_t v[kDescTabVec][4] = {};
        const int row = lane >> 1, half = lane & 1, vv = row - 15;
        if (row < 31) {
            const int um = umax_tab[vv < 0 ? -vv : vv];
            for (int d = 0; d < 4; ++d)
                for (int bb = 0; bb < 4; ++bb) {
                    const int u = 16 * half + 4 * d + bb - 15;
                    if ((u < 0 ? -u : u) <= um) {
                        v[0][d] |= (uint32_t)(u + 16) << (8 * bb);
                        v[1][d] |= 1u << (8 * bb);
                    }
                }
            v[2][0] = (uint32_t)vv;
            v[2][1] = (uint32_t)(16 * half - 15);
        }
        for (int e = 0; e < 8; ++e) {
            const float fx = (float)pattern[16 * lane + 2 * e], fy = (float)pattern[16 * lane + 2 * e + 1];
            std::memcpy(&v[3 + e / 4][e % 4], &fx, 4);
            std::memcpy(&v[5 + e / 4][e % 4], &fy, 4);
        }
        for (int t = 0; t < kPatchLoads; ++t) {
            // entry lane + 64 t of the row-major list of the dwords inside the pattern's reach (the spare entries at the
            // end repeat the last one)
            int want = lane + 64 * t, seen = 0, r = 0, k = 0;
            for (int rr = 0; rr < kPatchRows; ++rr) {
                const double m = std::max(std::abs(rr - 18) - 0.5, 0.0);
                const int c = std::min(18, (int)std::floor(std::sqrt(338.0 * 1.01 - m * m) + 0.5));
                for (int kk = (18 - c) / 4; kk <= (18 + c) / 4; ++kk, ++seen)
                    if (seen <= want) r = rr, k = kk;
            }
            v[7 + t / 4][t % 4] = (uint32_t)(r * 40 + 4 * k) | ((uint32_t)((r - 18) & 0xFF) << 16) | ((uint32_t)((4 * k - 18) & 0xFF) << 24);
        }
        for (int q = 0; q < kDescTabVec; ++q)
            for (int d = 0; d < 4; ++d) tab[(q * 64 + lane) * 4 + d] = v[q][d];
    }
}

// Diagnostic build only (-DORBX_STAMPS_DESC): wave 0 adds the cycles between phase boundaries into
// B.stamps[slot] (1024 copies spread by block id); the product build compiles these calls away.
#ifdef ORBX_STAMPS_DESC
#define ORBX_STAMP_D(B, slot, t0)                                                                  \
    do {                                                                                           \
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                                \
        if (threadIdx.x == 0 && (B).stamps) atomicAdd(&(B).stamps[(slot) + 16 * (blockIdx.x & 1023)], t1_ - (t0)); \
        (t0) = __builtin_amdgcn_s_memtime();                                                        \
    } while (0)
#else
#define ORBX_STAMP_D(B, slot, t0) do { (void)(t0); } while (0)
#endif

// One workgroup handles up to kDescKp retained keypoints of one (frame, level) in three phases:
//  A. orientation moments: each wave takes its keypoints in turn; the 31x31 patch is read as 31
//     rows of 8 (unaligned) dwords, 4 dwords per lane, and v_dot4_u32_u8 against per-lane constant
//     weight dwords (u+16 inside the disc, 0 outside; 1 inside, 0 outside) yields
//     m10 = sum(u*I) and m01 = sum(v*I) exactly; a DPP reduction leaves the sums in LDS;
//  B. one thread per keypoint evaluates fastAtan2, the pinned sin/cos, and writes the 28-byte
//     keypoint record (so the double-precision work is done once per keypoint, not per lane);
//  C. each wave takes its keypoints again; lane l evaluates tests 4l..4l+3 (their eight
//     pattern points are lane constants held in registers) on the blurred level.
// Workgroups are dealt round-robin to the 8 XCDs, so block ids are decoded as
// (id % 8, id / 8) = (XCD lane, work index) and every frame is given to one XCD lane: all
// keypoints of a frame then gather from the same L2.  A locality hint only.
// grid = 8 * ceil(frames/8) * nlevels * kp_blocks (1-D), block = 256.
__global__ __launch_bounds__(256) void k_orient_describe(BatchDev B, const uint32_t* __restrict__ lvl_kps,
                                                         const int32_t* __restrict__ lvl_count,
                                                         const uint4* __restrict__ lane_tab,  // build_describe_lane_table
                                                         orbx_kp* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                         int32_t* __restrict__ out_n, int kp_cap, int kp_blocks) {
    __shared__ uint32_t s_pk[kDescKp];
    __shared__ int s_m10[kDescKp], s_m01[kDescKp];
    __shared__ float s_a[kDescKp], s_b[kDescKp];
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][kPatchRows * kPatchPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // This kernel allocates exactly 64 registers per lane, like FAST and the blur, whose waves it shares the SIMDs
    // with: a retiring wave then leaves a slice of the register file that fits a wave of either of them.  (With the
    // 56 registers it needs, the extraction step got 6 % slower although this kernel alone got faster.)
    asm volatile("" ::: "v63");
    const int per_frame = B.nlevels * kp_blocks;
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int frame = (j / per_frame) * 8 + xl;
    if (frame >= B.nframes) return;
    const int rem = j % per_frame;
    const int level = rem / kp_blocks, kblk = rem % kp_blocks;
    const int32_t* counts = lvl_count + (uint64_t)frame * B.nlevels;
    // a level's count is trusted no further than the level's slot in the retained-point array (max_kp entries): a
    // value outside it is cut to the slot and reported, so no point is read from beyond the slot
    auto level_count = [&](int l) {
        const int v = counts[l];
        if ((unsigned)v > (unsigned)B.lv[l].max_kp && B.status) B.status[kStatusLevelCount] = 1u;
        return max(0, min(v, B.lv[l].max_kp));
    };
    const int n = level_count(level);
    if (kblk == 0 && level == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < B.nlevels; ++l) tot += level_count(l);
        out_n[frame] = min(tot, kp_cap);
    }
    const int first = kblk * kDescKp;
    if (first >= n) return;  // whole block idle (uniform)
    const int nk = min(kDescKp, n - first);
    const LevelDev& L = B.lv[level];
    int base_out = 0;
    for (int l = 0; l < level; ++l) base_out += level_count(l);
    base_out += first;
    unsigned long long t0 = 0;
#ifdef ORBX_STAMPS_DESC
    t0 = __builtin_amdgcn_s_memtime();
#endif
    if (tid < nk) {
        // the patches of phases A and C are read around the point: it must lie where the detector can put one
        uint32_t pk = lvl_kps[(uint64_t)frame * B.kp_total + L.kp_off + first + tid];
        if ((unsigned)(pt_x(pk) - 3) >= (unsigned)(L.rect_w - 6) || (unsigned)(pt_y(pk) - 3) >= (unsigned)(L.rect_h - 6)) {
            if (B.status) B.status[kStatusPoint] = 1u;
            pk = pack_pt(max(3, min(pt_x(pk), L.rect_w - 4)), max(3, min(pt_y(pk), L.rect_h - 4)), pt_score(pk));
        }
        s_pk[tid] = pk;
    }

    // ---- lane constants (see build_describe_lane_table) ----
    uint4 tv[kDescTabVec];
#pragma unroll
    for (int q = 0; q < kDescTabVec; ++q) tv[q] = lane_tab[q * 64 + lane];
    const uint32_t ic_wu[4] = {tv[0].x, tv[0].y, tv[0].z, tv[0].w}, ic_w1[4] = {tv[1].x, tv[1].y, tv[1].z, tv[1].w};
    // the patch offset is computed, not loaded, so that the patch loads do not wait for the table
    const int ic_v = lane < 62 ? (lane >> 1) - 15 : 0;
    const int ic_off = lane < 62 ? ic_v * L.pitch + 16 * (lane & 1) - 15 : 0;  // lanes 62, 63: no row 31, weights 0
    const float pat_x[8] = {__builtin_bit_cast(float, tv[3].x), __builtin_bit_cast(float, tv[3].y), __builtin_bit_cast(float, tv[3].z),
                            __builtin_bit_cast(float, tv[3].w), __builtin_bit_cast(float, tv[4].x), __builtin_bit_cast(float, tv[4].y),
                            __builtin_bit_cast(float, tv[4].z), __builtin_bit_cast(float, tv[4].w)};
    const float pat_y[8] = {__builtin_bit_cast(float, tv[5].x), __builtin_bit_cast(float, tv[5].y), __builtin_bit_cast(float, tv[5].z),
                            __builtin_bit_cast(float, tv[5].w), __builtin_bit_cast(float, tv[6].x), __builtin_bit_cast(float, tv[6].y),
                            __builtin_bit_cast(float, tv[6].z), __builtin_bit_cast(float, tv[6].w)};
    const uint32_t pr_tab[8] = {tv[7].x, tv[7].y, tv[7].z, tv[7].w, tv[8].x, tv[8].y, tv[8].z, tv[8].w};
    __syncthreads();
    ORBX_STAMP_D(B, 0, t0);

    // ---- A. orientation moments (the next keypoint's patch is in flight while this one is reduced) ----
    const uint8_t* img = L.img + (uint64_t)frame * L.plane;
    // wave w takes keypoints w, w + 4, w + 8, ... of the workgroup's run, so a short run is shared evenly
    const int cnt = max(0, (nk - wave + 3) >> 2);  // wave-uniform
    auto patch_ptr = [&](int kp) {
        const uint32_t pk = __builtin_amdgcn_readfirstlane(s_pk[kp]);
        return img + (uint64_t)(pt_y(pk) + kBorder) * L.pitch + (pt_x(pk) + kBorder);  // level coordinates (:936-937)
    };
    // eight patches are requested before the first one is reduced: the phase then costs one memory
    // round trip per eight keypoints, not one per keypoint
    constexpr int kABatch = kDescPerWave < 8 ? kDescPerWave : 8;
    for (int q0 = 0; q0 < cnt; q0 += kABatch) {
        uint4 px4[kABatch];
#pragma unroll
        for (int q = 0; q < kABatch; ++q) {
            px4[q] = make_uint4(0, 0, 0, 0);
            if (q0 + q < cnt) px4[q] = *reinterpret_cast<const uint4_unaligned*>(patch_ptr(wave + 4 * (q0 + q)) + ic_off);
        }
#pragma unroll
        for (int q = 0; q < kABatch; ++q) {
            if (q0 + q >= cnt) break;
            const uint32_t px[4] = {px4[q].x, px4[q].y, px4[q].z, px4[q].w};
            int su = 0, s1 = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s1 = (int)__builtin_amdgcn_udot4(px[d], ic_w1[d], (uint32_t)s1, false);
                su = (int)__builtin_amdgcn_udot4(px[d], ic_wu[d], (uint32_t)su, false);
            }
            int m10 = su - 16 * s1, m01 = ic_v * s1;
            m10 = wave_sum_to_lane63(m10);
            m01 = wave_sum_to_lane63(m01);
            if (lane == 63) {
                s_m10[wave + 4 * (q0 + q)] = m10;
                s_m01[wave + 4 * (q0 + q)] = m01;
            }
        }
    }
    __syncthreads();
    ORBX_STAMP_D(B, 1, t0);

    // ---- B. angle, rotation, keypoint record: one thread per keypoint ----
    if (tid < nk) {
        const uint32_t pk = s_pk[tid];
        const float angle = fast_atan2_deg((float)s_m01[tid], (float)s_m10[tid]);
        const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
        float a, b;
        sincos_pinned(__fmul_rn(angle, factor_pi), &a, &b);
        s_a[tid] = a;
        s_b[tid] = b;
        const int o = base_out + tid;
        if (o < kp_cap) {
            orbx_kp k;
            const float fx = (float)(pt_x(pk) + kBorder), fy = (float)(pt_y(pk) + kBorder);
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
    __syncthreads();
    ORBX_STAMP_D(B, 2, t0);

    // ---- C. rotated BRIEF tests on the blurred level ----
    // The 37x37 neighbourhood the rotated pattern can reach (|coordinate| <= 18) is copied to LDS
    // with coalesced row loads (the 310 dwords of its 37 rows that the pattern can reach, 5 per lane, the next keypoint's rows in flight
    // while this one is evaluated); the 512 scattered samples are then LDS byte reads, so the
    // vector cache sees each patch line once instead of once per gather instruction.
    const uint8_t* blur = L.blur + (uint64_t)frame * L.bplane;
    const int bp = L.bpitch;
    uint8_t* my_patch = s_patch[wave];
    int pr_off[kPatchLoads], pr_lds[kPatchLoads];
#pragma unroll
    for (int t = 0; t < kPatchLoads; ++t) {
        const uint32_t e = pr_tab[t];
        pr_off[t] = ((int)(e << 8) >> 24) * bp + ((int)e >> 24);  // (row - 18) * pitch + 4 * dword - 18
        pr_lds[t] = (int)(e & 0xFFFFu);
    }
    auto fetch = [&](int kp, uint32_t (&reg)[kPatchLoads]) {
        const uint32_t pk = __builtin_amdgcn_readfirstlane(s_pk[kp]);
        const uint8_t* cb = blur + (uint64_t)(pt_y(pk) + kBorder) * bp + (pt_x(pk) + kBorder);
#pragma unroll
        for (int t = 0; t < kPatchLoads; ++t) reg[t] = *reinterpret_cast<const desc_u32_unaligned*>(cb + pr_off[t]);
    };
    const int cnt_c = max(0, (min(nk, kp_cap - base_out) - wave + 3) >> 2);  // keypoints beyond the caller's capacity are dropped
    uint32_t nreg[kPatchLoads];
    if (cnt_c > 0) fetch(wave, nreg);
    for (int q = 0; q < cnt_c; ++q) {
        const int kp = wave + 4 * q;
        __builtin_amdgcn_wave_barrier();  // the previous keypoint's LDS reads are done (same wave, in order)
#pragma unroll
        for (int t = 0; t < kPatchLoads; ++t) *reinterpret_cast<uint32_t*>(my_patch + pr_lds[t]) = nreg[t];
        if (q + 1 < cnt_c) fetch(kp + 4, nreg);
        __builtin_amdgcn_wave_barrier();
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_a[kp])));
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_b[kp])));
        // cvRound = round half to even.  Adding 1.5 * 2^23 rounds a float of magnitude below 2^22 to
        // that integer in its low mantissa bits: the sum's bit pattern is 0x4B400000 + n, so
        // row * pitch + col is one 24-bit multiply-add of the two bit patterns plus a folded constant.
        const float kMagic = 12582912.0f;
        // (unsigned arithmetic: the intermediate terms wrap, the final index is the small true one)
        const uint32_t kFold = (uint32_t)(18 * kPatchPitch + 18) - (0x400000u * (uint32_t)kPatchPitch + 0x4B400000u);
        int val[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // the reference's -march=native build contracts x*b + y*a and x*a - y*b into one fma
            const float fr = __fadd_rn(__fmaf_rn(pat_x[e], b, __fmul_rn(pat_y[e], a)), kMagic);
            const float fc = __fadd_rn(__fmaf_rn(pat_x[e], a, -__fmul_rn(pat_y[e], b)), kMagic);
            val[e] = my_patch[(uint32_t)__mul24(__builtin_bit_cast(int, fr), kPatchPitch) + __builtin_bit_cast(uint32_t, fc) + kFold];
        }
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) bits |= (val[2 * t] < val[2 * t + 1] ? 1u : 0u) << t;
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0xB1, 0xF, 0xF, true);  // lane ^ 1
        uint8_t* dsc = out_desc + ((uint64_t)frame * kp_cap + base_out + kp) * 32;
        if ((lane & 1) == 0) dsc[lane >> 1] = (uint8_t)(bits | (other << 4));
    }
    ORBX_STAMP_D(B, 3, t0);
}

}  // namespace orbx
#endif
