// Micro-benchmark: issue rate of the integer VALU instructions the kernels lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t b = seed * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + a[(i + 1) & 7];                               // v_add_u32 (not foldable over iterations)
            if (OP == 1) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);        // v_dot4_u32_u8
            if (OP == 2) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a[i]), __builtin_bit_cast(u16x2, b), a[i], false);
            if (OP == 3) a[i] = __umul24(a[i], b) + a[i];                            // v_mad_u32_u24
            if (OP == 4) a[i] = __builtin_amdgcn_alignbyte(a[i], b, 1);              // v_alignbyte_b32
            if (OP == 5) a[i] = a[i] * b;                                            // v_mul_lo_u32
            if (OP == 6) a[i] = min(a[i], min(b, a[(i + 1) & 7]));                   // v_min3_u32
            if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], b, 0x05040100u);         // v_perm_b32
            if (OP == 8) a[i] = min((a[i] >> 16) & 0xFFFF, b);                        // sdwa candidate
            if (OP == 9) a[i] = __builtin_amdgcn_update_dpp(a[i], b + a[i], 0x138, 0xF, 0xF, false);
            if (OP == 10) a[i] = (uint32_t)max((int)a[i] - (int)(b & 0xFF), (int)a[(i + 1) & 7]);
            if (OP == 11) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a[i]), __builtin_bit_cast(u16x2, b)));  // v_pk_max_u16
            if (OP == 12) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(i16x2, a[i]), __builtin_bit_cast(i16x2, b)));  // v_pk_min_i16
            if (OP == 13) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, a[i]) - __builtin_bit_cast(i16x2, b));                          // v_pk_sub_i16
            if (OP == 14) a[i] = __sad(a[i], b, a[i]);                                                                                                  // v_sad_u32
            if (OP == 15) a[i] = max(a[i], b);                                                                                                          // v_max_u32
            if (OP == 16) a[i] = __popc(a[i]) + b;                                                                                                      // v_bcnt_u32_b32
            if (OP == 17) a[i] = (a[i] >> 8) & 0xFFu;                                                                                                   // v_bfe_u32
            if (OP == 18) a[i] = (a[i] << 16) | b;                                                                                                      // v_lshl_or_b32
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d, int blocks_per_cu) {
    const int iters = 4096, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 16, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8;  // wave-instructions of the op
    // per SIMD: blocks_per_cu waves; cycles per wave-instr per SIMD at 2.4 GHz
    const double per_simd = winstr / 1024.0;
    printf("%-18s %d waves/SIMD  %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int bpc : {2, 8}) {
        run<0>("v_add_u32", d, bpc);
        run<1>("v_dot4_u32_u8", d, bpc);
        run<2>("v_dot2_u32_u16", d, bpc);
        run<3>("v_mad_u32_u24", d, bpc);
        run<4>("v_alignbyte_b32", d, bpc);
        run<5>("v_mul_lo_u32", d, bpc);
        run<6>("v_min3_u32", d, bpc);
        run<7>("v_perm_b32", d, bpc);
        run<8>("shift+and+min", d, bpc);
        run<9>("add + mov_dpp", d, bpc);
        run<10>("and,sub,max", d, bpc);
        run<11>("v_pk_max_u16", d, bpc);
        run<12>("v_pk_min_i16", d, bpc);
        run<13>("v_pk_sub_i16", d, bpc);
        run<14>("v_sad_u32", d, bpc);
        run<15>("v_max_u32", d, bpc);
        run<16>("v_bcnt_u32_b32", d, bpc);
        run<17>("v_bfe_u32", d, bpc);
        run<18>("v_lshl_or_b32", d, bpc);
    }
    return 0;
}
