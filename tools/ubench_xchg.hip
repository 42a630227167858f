// Diagnostics: (1) which XCD a workgroup lands on (normal / cooperative launch);
// (2) ping-pong latency between two workgroups through global memory with different cache scopes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}

template <int MODE>  // 0: agent-scope atomics (sc1)   1: sc0 loads + plain stores   2: plain volatile
__device__ unsigned long long ld(const unsigned long long* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return *(volatile const unsigned long long*)p;
}
template <int MODE>
__device__ void st(unsigned long long* p, unsigned long long v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *(volatile unsigned long long*)p = v;
}

// blocks a and b play ping-pong `iters` times; everyone else exits.  result[0] = ticks (100 MHz), result[1] = failed
template <int MODE>
__global__ void pingpong(unsigned long long* line, int a, int b, int iters, unsigned long long* result) {
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    const bool first = blockIdx.x == a;
    unsigned long long* mine = line + (first ? 0 : 16);
    unsigned long long* theirs = line + (first ? 16 : 0);
    const unsigned long long t0 = wall_clock64();
    int failed = 0;
    for (int i = 1; i <= iters && !failed; ++i) {
        if (first) st<MODE>(mine, i);
        long spins = 0;
        while (ld<MODE>(theirs) < (unsigned long long)i) {
            if (++spins > (1L << 18)) { failed = 1; break; }
        }
        if (!first) st<MODE>(mine, i);
    }
    if (first) {
        result[0] = wall_clock64() - t0;
        result[1] = failed;
    } else if (failed) {
        result[2] = 1;
    }
}

template <int MODE>
int run_pp(const char* name, unsigned long long* line, unsigned long long* res, int a, int b, int grid) {
    CK(hipMemset(line, 0, 512));
    CK(hipMemset(res, 0, 64));
    const int iters = 2000;
    hipLaunchKernelGGL(pingpong<MODE>, dim3(grid), dim3(64), 0, 0, line, a, b, iters, res);
    CK(hipDeviceSynchronize());
    unsigned long long h[3];
    CK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-28s blocks %3d <-> %3d: %s  round trip %.2f us\n", name, a, b, (h[1] || h[2]) ? "FAILED (never saw the peer)" : "ok",
           h[0] * 10.0 / 1e3 / iters);
    return 0;
}

int main() {
    const int grid = 256;
    unsigned* d;
    CK(hipMalloc(&d, grid * 8));
    std::vector<unsigned> h(2 * grid);
    hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(128), 0, 0, d);
    CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
    printf("normal launch      xcc of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    void* args[] = {&d};
    CK(hipLaunchCooperativeKernel((const void*)where_kernel, dim3(grid), dim3(128), args, 0, 0));
    CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
    printf("cooperative launch xcc of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    int bad = 0;
    for (int i = 0; i < grid; ++i) bad += (h[2 * i] & 0xf) != (unsigned)(i % 8);
    printf("cooperative: blocks with xcc != blockIdx %% 8: %d of %d\n", bad, grid);
    unsigned long long *line, *res;
    CK(hipMalloc(&line, 512));
    CK(hipMalloc(&res, 64));
    run_pp<0>("agent-scope atomics (sc1)", line, res, 0, 8, grid);   // same XCD (if round-robin)
    run_pp<0>("agent-scope atomics (sc1)", line, res, 0, 1, grid);   // different XCDs
    run_pp<1>("sc0 loads, wg-scope stores", line, res, 0, 8, grid);
    run_pp<1>("sc0 loads, wg-scope stores", line, res, 0, 1, grid);
    run_pp<2>("plain volatile", line, res, 0, 8, grid);
    return 0;
}
