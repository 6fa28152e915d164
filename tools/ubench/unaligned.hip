// Probe: do unaligned dword / dwordx4 global loads and LDS dword reads return the right bytes on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t u32u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) U4 { uint32_t a, b, c, d; };
__global__ void k(const uint8_t* src, uint32_t* out, uint32_t* out4, uint32_t* outl) {
    __shared__ uint8_t s[256];
    s[threadIdx.x] = src[threadIdx.x];
    s[threadIdx.x + 64] = src[threadIdx.x + 64];
    __syncthreads();
    const int off = threadIdx.x;  // 0..63: every alignment
    out[threadIdx.x] = *reinterpret_cast<const u32u*>(src + off);
    U4 v = *reinterpret_cast<const U4*>(src + off);
    out4[threadIdx.x] = v.a ^ v.b ^ v.c ^ v.d;
    outl[threadIdx.x] = *reinterpret_cast<const u32u*>(s + off);
}
int main() {
    uint8_t h[256];
    for (int i = 0; i < 256; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t *o, *o4, *ol;
    hipMalloc(&d, 256); hipMalloc(&o, 256); hipMalloc(&o4, 256); hipMalloc(&ol, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o4, ol);
    uint32_t r[64], r4[64], rl[64];
    hipMemcpy(r, o, 256, hipMemcpyDeviceToHost); hipMemcpy(r4, o4, 256, hipMemcpyDeviceToHost); hipMemcpy(rl, ol, 256, hipMemcpyDeviceToHost);
    int bad = 0, bad4 = 0, badl = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t e, a, b, c, dd; memcpy(&e, h + i, 4);
        memcpy(&a, h + i, 4); memcpy(&b, h + i + 4, 4); memcpy(&c, h + i + 8, 4); memcpy(&dd, h + i + 12, 4);
        bad += r[i] != e; bad4 += r4[i] != (a ^ b ^ c ^ dd); badl += rl[i] != e;
    }
    printf("unaligned global dword: %s, dwordx4: %s, LDS dword: %s\n", bad ? "WRONG" : "ok", bad4 ? "WRONG" : "ok", badl ? "WRONG" : "ok");
    return 0;
}
