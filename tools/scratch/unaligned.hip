// Does a global_load_dwordx4 from a 2-byte-aligned address return the right bytes on gfx950 (unaligned access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const unsigned char* src, uint4* out, int* offs) {
    const int t = threadIdx.x;
    const uint4 v = *reinterpret_cast<const uint4*>(src + offs[t]);
    out[t] = v;
}
int main() {
    const int n = 1 << 16;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 131 + 7);
    unsigned char* d; uint4* o; int* doff;
    hipMalloc(&d, n); hipMalloc(&o, 64 * sizeof(uint4)); hipMalloc(&doff, 64 * sizeof(int));
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    int offs[64];
    for (int t = 0; t < 64; ++t) offs[t] = 49 * t + 2 * (t % 7) + (t & 1) * 2 + 126;   // 2-byte aligned, crossing lines
    for (int t = 0; t < 64; ++t) offs[t] &= ~1;
    hipMemcpy(doff, offs, sizeof(offs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, doff);
    uint4 res[64];
    hipError_t e = hipMemcpy(res, o, sizeof(res), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        unsigned char* p = (unsigned char*)&res[t];
        for (int b = 0; b < 16; ++b) bad += p[b] != h[offs[t] + b];
    }
    printf("unaligned dwordx4 loads: %d wrong bytes (offsets mod 4: %d %d %d)\n", bad, offs[1] % 4, offs[2] % 4, offs[3] % 4);
    return bad != 0;
}
