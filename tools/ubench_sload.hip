// store (vector, relaxed agent scope) -> poll with a SCALAR load (glc): round trip in cycles, vs the vector poll
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned long long* t, int iters, unsigned long long* flag) {
    const int tid = threadIdx.x;
    unsigned long long* mine = flag + (blockIdx.x * 4 + (tid >> 6)) * 16;
    unsigned long long c0 = clock64();
    unsigned long long v = 0;
    for (int i = 0; i < iters; ++i) {
        if ((tid & 63) == 0) __hip_atomic_store(mine, (unsigned long long)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        do { v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (v != (unsigned long long)(i + 1));
    }
    unsigned long long c1 = clock64();
    unsigned long long* mine2v = mine + 8;
    const unsigned long long a_ = (unsigned long long)mine2v;
    const unsigned long long au = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a_ >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)a_);
    unsigned long long* mine2 = (unsigned long long*)au;
    for (int i = 0; i < iters; ++i) {
        if ((tid & 63) == 0) __hip_atomic_store(mine2, (unsigned long long)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long s;
        do {
            asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(mine2) : "memory");
        } while (s != (unsigned long long)(i + 1));
    }
    unsigned long long c2 = clock64();
    // plain vector load latency of a line nobody writes (L2 hit), dependent chain
    unsigned long long acc = 0;
    for (int i = 0; i < iters; ++i) acc += __hip_atomic_load(mine + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1;
    unsigned long long c3 = clock64();
    unsigned long long sacc = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned long long s;
        asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(mine2) : "memory");
        sacc += s & 1;
    }
    unsigned long long c4 = clock64();
    if (blockIdx.x == 0 && tid == 0) { t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c3 - c2; t[3] = c4 - c3; t[4] = v + acc + sacc; }
}
int main() {
    unsigned long long *t, *flag;
    hipMalloc(&t, 128); hipMalloc(&flag, 256 * 4 * 16 * 8);
    for (int grid : {1, 256}) {
        hipMemset(flag, 0, 256 * 4 * 16 * 8);
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, t, iters, flag);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("grid %3d: store->vector poll %.0f cycles, store->scalar poll %.0f, vector load %.0f, scalar load (glc) %.0f\n", grid,
               (double)h[0] / iters, (double)h[1] / iters, (double)h[2] / iters, (double)h[3] / iters);
    }
    return 0;
}
