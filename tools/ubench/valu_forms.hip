// Micro-benchmark: does the ENCODING of a VALU instruction change its issue rate on gfx950?
// Same operation as VOP2 with register operands, with a 32-bit literal, with an SGPR operand,
// and the VOP3 / VOPC forms the kernels use.  Each variant is written in inline assembly so the
// compiler cannot pick another form.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_forms valu_forms.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(stmt) \
    { { const int i = 0; stmt; } { const int i = 1; stmt; } { const int i = 2; stmt; } { const int i = 3; stmt; } \
      { const int i = 4; stmt; } { const int i = 5; stmt; } { const int i = 6; stmt; } { const int i = 7; stmt; } }

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t b = seed * 3 + threadIdx.x;
    const uint32_t sc = __builtin_amdgcn_readfirstlane(seed * 0x00FF00FFu);
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) REP8(asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
        if (OP == 1) REP8(asm volatile("v_and_b32 %0, 0xff00ff, %0" : "+v"(a[i])))
        if (OP == 2) REP8(asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sc)))
        if (OP == 3) REP8(asm volatile("v_and_b32 %0, 15, %0" : "+v"(a[i])))
        if (OP == 4) REP8(asm volatile("v_cmp_gt_i32 vcc, 0, %0" : : "v"(a[i]) : "vcc"))
        if (OP == 5) REP8(asm volatile("v_cmp_gt_i32_e64 s[20:21], 0, %0" : : "v"(a[i]) : "s20", "s21"))
        if (OP == 6) REP8(asm volatile("v_cmp_gt_i16 vcc, 0, %0" : : "v"(a[i]) : "vcc"))
        if (OP == 7) REP8(asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" : : "v"(a[i]), "s"(sc) : "s20", "s21"))
        if (OP == 8) REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sc)))
        if (OP == 9) REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 10) REP8(asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i])))
        if (OP == 11) REP8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(sc)))
        if (OP == 12) REP8(asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 13) REP8(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 14) REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
        if (OP == 15) REP8(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(sc)))
        if (OP == 16) REP8(asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 17) REP8(asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i])))
        if (OP == 18) REP8(asm volatile("v_or_b32 %0, 0x1000100, %0" : "+v"(a[i])))
        if (OP == 19) REP8(asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 20) REP8(asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 21) REP8(asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b)))
        if (OP == 22) REP8(asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "s"(sc)))
        if (OP == 23) REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 24) REP8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 25) REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<uint64_t*>(&a[i & 6])) : "v"((uint64_t)b)))
        if (OP == 26) REP8(asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i])))
        if (OP == 27) REP8(asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])))
        if (OP == 28) REP8(asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i])))
        if (OP == 29) REP8(asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 30) REP8(asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 31) REP8(asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 32) REP8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 33) REP8(asm volatile("v_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 34) REP8(asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 35) REP8(asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 36) REP8(asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 37) REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"((uint64_t)sc * 0x100000001ull)))
        if (OP == 38) REP8(asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(sc)))
        if (OP == 39) REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
        if (OP == 40) REP8(asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 41) REP8(asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 42) REP8(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i])))
        if (OP == 43) REP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 44) REP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 46) REP8(asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 47) REP8(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 48) REP8(asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 49) REP8(asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 50) REP8(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
        if (OP == 51) REP8(asm volatile("v_ashrrev_i32 %0, 4, %0" : "+v"(a[i])))
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
    const double winstr = (double)blocks * 4 * iters * 8;
    const double per_simd = winstr / 1024.0;
    printf("%-34s %d waves/SIMD  %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int bpc : {8}) {
        run<0>("v_and_b32 vgpr,vgpr", d, bpc);
        run<1>("v_and_b32 literal", d, bpc);
        run<2>("v_and_b32 sgpr", d, bpc);
        run<3>("v_and_b32 inline const", d, bpc);
        run<4>("v_cmp_gt_i32 vcc (VOPC)", d, bpc);
        run<5>("v_cmp_gt_i32_e64 sgpr pair", d, bpc);
        run<6>("v_cmp_gt_i16 vcc", d, bpc);
        run<7>("v_cmp_gt_u32_e64 v,s", d, bpc);
        run<8>("v_mbcnt_lo_u32_b32", d, bpc);
        run<9>("v_lshl_add_u32", d, bpc);
        run<10>("v_bfe_u32", d, bpc);
        run<11>("v_perm_b32", d, bpc);
        run<12>("v_pk_sub_i16", d, bpc);
        run<13>("v_sub_u32", d, bpc);
        run<14>("v_cndmask_b32 vcc", d, bpc);
        run<15>("v_and_or_b32", d, bpc);
        run<16>("v_add3_u32", d, bpc);
        run<17>("v_lshrrev_b32 inline", d, bpc);
        run<18>("v_or_b32 literal", d, bpc);
        run<19>("v_min_i32", d, bpc);
        run<20>("v_pk_max_u16", d, bpc);
        run<21>("v_sub_u32_sdwa bytes", d, bpc);
        run<22>("v_mov_b32 sgpr", d, bpc);
        run<23>("v_fma_f32", d, bpc);
        run<24>("v_mul_f32", d, bpc);
        run<25>("v_pk_fma_f32", d, bpc);
        run<26>("v_cvt_f32_ubyte1", d, bpc);
        run<27>("v_rndne_f32", d, bpc);
        run<28>("v_cvt_i32_f32", d, bpc);
        run<29>("v_mad_i32_i24", d, bpc);
        run<30>("v_max_i16", d, bpc);
        run<31>("v_sub_u16", d, bpc);
        run<32>("v_max_i32", d, bpc);
        run<33>("v_min_i16", d, bpc);
        run<34>("v_min_u32", d, bpc);
        run<35>("v_max_u32", d, bpc);
        run<36>("v_min_u16", d, bpc);
        run<37>("v_cndmask_b32_e64 sgpr pair", d, bpc);
        run<38>("v_add_u32 sgpr", d, bpc);
        run<39>("v_addc_co_u32 vcc", d, bpc);
        run<40>("v_add_u16", d, bpc);
        run<41>("v_max3_i32", d, bpc);
        run<42>("v_lshlrev_b32", d, bpc);
        run<43>("v_xor_b32", d, bpc);
        run<44>("v_add_f32", d, bpc);
        run<46>("v_fmac_f32", d, bpc);
        run<47>("v_mul_u32_u24", d, bpc);
        run<48>("v_bcnt_u32_b32", d, bpc);
        run<49>("v_mov_b32 vgpr", d, bpc);
        run<50>("v_max_f32", d, bpc);
        run<51>("v_ashrrev_i32", d, bpc);
    }
    return 0;
}
