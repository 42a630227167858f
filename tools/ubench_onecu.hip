// Microbenchmark for the one-workgroup-per-transform PCPS design: what ONE workgroup
// per CU gets from the chip.
//   (1) issue cost of INDEPENDENT fp64 FMAs from 1 or 2 waves per SIMD (256- or 512-thread workgroups, one per CU);
//   (2) v_accvgpr_write / v_accvgpr_read round trips (parking state in the accumulation half of the register file);
//   (3) per-CU read bandwidth of 16-B loads from operands that live in L2 / the Infinity Cache: every workgroup of an
//       XCD streams the same few 400 KB arrays (the spectra of a PCPS call), K loads in flight per thread.
// hipcc --offload-arch=gfx950 -O3 -o ubench_onecu ubench_onecu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int THREADS>
__global__ __launch_bounds__(THREADS) void fma_kernel(double* out, unsigned long long* t, int iters) {
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = 1.0 + threadIdx.x * 1e-9 + j;
    const double y = 0.999999, z = 1e-9;
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(a[j], y, z);
    }
    const unsigned long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = c1 - c0;
}

// fp64 add + multiply mix with an accvgpr round trip per value
template <int THREADS>
__global__ __launch_bounds__(THREADS) void acc_kernel(double* out, unsigned long long* t, int iters) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + threadIdx.x * 1e-9 + j;
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int lo = __double2loint(a[j]), hi = __double2hiint(a[j]);
            int alo, ahi;
            asm volatile("v_accvgpr_write_b32 a0, %2\n v_accvgpr_write_b32 a1, %3\n s_nop 1\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1"
                         : "=v"(alo), "=v"(ahi) : "v"(lo), "v"(hi) : "a0", "a1");
            a[j] = __hiloint2double(ahi, alo) * 0.999999;
        }
    }
    const unsigned long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = c1 - c0;
}

// Every workgroup streams `nbuf` arrays of `len` double2 (chosen by XCD slot), K loads in flight per thread.
template <int THREADS, int K>
__global__ __launch_bounds__(THREADS) void read_kernel(const double2* __restrict__ src, int len, int nbuf_per_xcd, int reps, double* out) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    double2 acc = make_double2(0, 0);
    for (int rep = 0; rep < reps; ++rep) {
        const int b = (slot + rep) % nbuf_per_xcd;
        const double2* __restrict__ p = src + (size_t)(xcd * nbuf_per_xcd + b) * len;
        for (int i = threadIdx.x; i + (K - 1) * THREADS < len; i += K * THREADS) {
            double2 v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = p[i + k * THREADS];
#pragma unroll
            for (int k = 0; k < K; ++k) { acc.x += v[k].x; acc.y += v[k].y; }
        }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc.x + acc.y;
}

template <int THREADS>
static void run_fma(double* out, unsigned long long* t, int grid) {
    const int iters = 1 << 14;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(fma_kernel<THREADS>, dim3(grid), dim3(THREADS), 0, 0, out, t, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("fp64 FMA, independent x16, %d threads/WG, grid %3d: %.2f cycles per wave-instruction (per wave)\n", THREADS, grid, (double)h / (iters * 16.0));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(acc_kernel<THREADS>, dim3(grid), dim3(THREADS), 0, 0, out, t, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("  accvgpr write x2 + read x2 + mul, x8, %d threads/WG, grid %3d: %.2f cycles per value (5 instructions + s_nop)\n", THREADS, grid, (double)h / (iters * 8.0));
}

template <int THREADS, int K>
static void run_read(const double2* src, int len, int nbuf, double* out) {
    const int reps = 40;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((read_kernel<THREADS, K>), dim3(256), dim3(THREADS), 0, 0, src, len, nbuf, reps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = 256.0 * reps * (double)(len / (K * THREADS) * (K * THREADS)) * 16.0;
    printf("read: %d thr/WG, %2d loads in flight, %2d arrays of %d KB per XCD: %.3f ms, %.1f GB/s per CU, %.2f TB/s chip\n", THREADS, K, nbuf,
           len * 16 / 1024, best, bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
}

int main() {
    double* out; unsigned long long* t;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&t, 128);
    for (int grid : {1, 256}) { run_fma<256>(out, t, grid); run_fma<512>(out, t, grid); run_fma<1024>(out, t, grid); }
    const int len = 25000;
    double2* src; hipMalloc(&src, (size_t)8 * 16 * len * 16);
    hipMemset(src, 0, (size_t)8 * 16 * len * 16);
    for (int nbuf : {2, 8, 12, 16}) {
        run_read<256, 8>(src, len, nbuf, out);
        run_read<256, 16>(src, len, nbuf, out);
        run_read<512, 4>(src, len, nbuf, out);
        run_read<512, 8>(src, len, nbuf, out);
        run_read<512, 16>(src, len, nbuf, out);
    }
    return 0;
}
