// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library
// uses: each kernel reads (or writes) a known byte count once, so counter / bytes gives the factor.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib      (and a second run with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) u128u { uint32_t x, y, z, w; };

__global__ void rd_b128(const uint4* p, size_t n, uint32_t* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_b32(const uint32_t* p, size_t n, uint32_t* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_b128_unaligned(const uint8_t* p, size_t n, uint32_t* sink) {  // 16-byte loads at byte offset 3
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u128u v = *reinterpret_cast<const u128u*>(p + 3 + 16 * i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_b32_unaligned(const uint8_t* p, size_t n, uint32_t* sink) {  // dword loads at byte offset 1
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= *reinterpret_cast<const u32u*>(p + 1 + 4 * i);
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void wr_b128(uint4* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void wr_b32(uint32_t* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 7;
}

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB: well past the 256 MiB last-level cache
    uint8_t* buf;
    uint32_t* sink;
    if (hipMalloc(&buf, bytes + 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes + 64);
    hipDeviceSynchronize();
    const int grid = 256 * 16, block = 256;
    rd_b128<<<grid, block>>>((const uint4*)buf, bytes / 16, sink);
    rd_b32<<<grid, block>>>((const uint32_t*)buf, bytes / 4, sink);
    rd_b128_unaligned<<<grid, block>>>(buf, bytes / 16, sink);
    rd_b32_unaligned<<<grid, block>>>(buf, bytes / 4, sink);
    wr_b128<<<grid, block>>>((uint4*)buf, bytes / 16);
    wr_b32<<<grid, block>>>((uint32_t*)buf, bytes / 4);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
