// Microbenchmark: the shader clock a sparse persistent kernel really runs at (clock64 vs the 100 MHz wall_clock64) and
// what dependent instructions cost there: fp64 FMA chain, fp64 division, v_readlane -> VALU, LDS round trip,
// a relaxed agent-scope store -> load round trip through L2.   hipcc --offload-arch=gfx950 -O3 -o ubench_clock ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(double* out, unsigned long long* t, int iters, unsigned long long* flag) {
    __shared__ double lds[256];
    const int tid = threadIdx.x;
    double x = 1.0 + tid * 1e-9, y = 0.999999;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) x = __builtin_fma(x, y, 1e-9);          // dependent FMA chain
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    double d = x;
    for (int i = 0; i < iters / 8; ++i) d = 1.0 / (d + 1.5);                 // dependent divisions
    unsigned long long c2 = clock64();
    double r = d;
    for (int i = 0; i < iters / 8; ++i) {                                    // readlane -> VALU dependent
        int lo = __builtin_amdgcn_readlane(__double2loint(r), 3);
        int hi = __builtin_amdgcn_readlane(__double2hiint(r), 3);
        r = __hiloint2double(hi, lo) + 1e-3;
    }
    unsigned long long c3 = clock64();
    double l = r;
    for (int i = 0; i < iters / 8; ++i) {                                    // LDS write -> read dependent (same wave)
        lds[tid] = l;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        l = lds[tid ^ 1] + 1.0;
    }
    unsigned long long c4 = clock64();
    double a = l;
    for (int i = 0; i < iters / 8; ++i) a = atan(a * 0.5) + 0.25;           // dependent atan
    unsigned long long c5 = clock64();
    // store -> load through L2 (own line)
    unsigned long long v = 0;
    unsigned long long* mine = flag + (blockIdx.x * 4 + (tid >> 6)) * 16;
    for (int i = 0; i < iters / 64; ++i) {
        if ((tid & 63) == 0) __hip_atomic_store(mine, (unsigned long long)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        do { v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (v != (unsigned long long)(i + 1));
    }
    unsigned long long c6 = clock64(), w6 = wall_clock64();
    __syncthreads();
    unsigned long long c7 = clock64();
    for (int i = 0; i < iters / 64; ++i) __syncthreads();
    unsigned long long c8 = clock64();
    out[blockIdx.x * 256 + tid] = x + d + r + l + a + (double)v;
    if (blockIdx.x == 0 && tid == 0) {
        t[0] = c1 - c0; t[1] = w1 - w0; t[2] = c2 - c1; t[3] = c3 - c2; t[4] = c4 - c3; t[5] = c5 - c4; t[6] = c6 - c5; t[7] = w6 - w0; t[8] = c6 - c0; t[9] = c8 - c7;
    }
}

int main() {
    double* out; unsigned long long *t, *flag;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&t, 128); hipMalloc(&flag, 256 * 4 * 16 * 8);
    for (int grid : {1, 256}) for (int rep = 0; rep < 2; ++rep) {
        const int iters = 1 << 16;
        hipMemset(flag, 0, 256 * 4 * 16 * 8);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t, iters, flag);
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, t, 128, hipMemcpyDeviceToHost);
        double mhz = (double)h[8] / ((double)h[7] * 10e-9) / 1e6;
        printf("grid %3d: shader clock %.0f MHz (whole kernel %.2f ms) | cycles per dependent: fma %.1f  div %.1f  readlane+add %.1f  lds wr->rd %.1f  atan %.1f  L2 store->load %.0f  barrier %.0f\n",
               grid, mhz, h[7] * 10e-6, (double)h[0] / iters, (double)h[2] / (iters / 8), (double)h[3] / (iters / 8), (double)h[4] / (iters / 8),
               (double)h[5] / (iters / 8), (double)h[6] / (iters / 64), (double)h[9] / (iters / 64));
    }
    return 0;
}
