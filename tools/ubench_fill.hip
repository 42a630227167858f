// HBM write / read rates in isolation (16 B per lane, grid-stride): what a kernel that only writes (the PCPS column
// kernel's 525 MB intermediate) or only reads (the row kernel) can expect.   hipcc --offload-arch=gfx950 -O3 -o ubench_fill ubench_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void fill_kernel(uint4* __restrict__ dst, size_t n16, uint4 v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void sum_kernel(const uint4* __restrict__ src, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
int main() {
    const size_t bytes = 525ull << 20, n16 = bytes / 16;
    uint4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 8192, 32768}) {
        for (int k = 0; k < 60; ++k) fill_kernel<<<blocks, 256>>>(buf, n16, make_uint4(k, 1, 2, 3));
        hipEventRecord(a);
        for (int k = 0; k < 20; ++k) fill_kernel<<<blocks, 256>>>(buf, n16, make_uint4(k, 1, 2, 3));
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("fill  %6d blocks: %.1f us per 525 MiB = %.2f TB/s\n", blocks, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
        hipEventRecord(a);
        for (int k = 0; k < 20; ++k) sum_kernel<<<blocks, 256>>>(buf, n16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("read  %6d blocks: %.1f us per 525 MiB = %.2f TB/s\n", blocks, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
