// Micro-benchmark: streaming bandwidth a plain kernel reaches on this GPU (read, write, copy),
// 16 bytes per lane, buffers far larger than the caches.  Gives the practical ceiling next to the
// 8 TB/s data-sheet figure.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_bw hbm_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, uint32_t* out, size_t n) {
    uint32_t s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
        s ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ a, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_uint4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
    const size_t bytes = 2ull << 30, n = bytes / 16;
    uint4 *a, *b;
    uint32_t* o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int op = 0; op < 3; ++op) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (op == 0) k_read<<<blocks, 256>>>(a, o, n);
                if (op == 1) k_write<<<blocks, 256>>>(b, n, 7u);
                if (op == 2) k_copy<<<blocks, 256>>>(a, b, n);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = (op == 2 ? 2.0 : 1.0) * (double)bytes;
            printf("%-5s blocks %6d  %.3f ms  %.0f GB/s\n", op == 0 ? "read" : op == 1 ? "write" : "copy", blocks, best, moved / best / 1e6);
        }
    }
    return 0;
}
