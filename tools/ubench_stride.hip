// How the lane-to-lane stride of a wave's unaligned 16-byte loads prices them: every lane reads a 52-byte window (three
// 16-byte loads + one dword, as correlator_chip.h's block load does) at base + lane * stride, stride = 36 .. 52 bytes, from a
// buffer that stays in the L2.  Prints nanoseconds per wave-wide window and the ratio to the best stride.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_stride.hip -o tools/ubench_stride && tools/ubench_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void windows(const char* __restrict__ base, int stride, int rounds, size_t span, unsigned* out) {
    const int lane = threadIdx.x;
    unsigned acc = 0;
    size_t at = ((size_t)blockIdx.x * 977 * 64) % span;
    for (int r = 0; r < rounds; ++r) {
        const char* p = base + ((at + (size_t)lane * stride) & ~(size_t)1);
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint4 b = *reinterpret_cast<const uint4*>(p + 16);
        const uint4 c = *reinterpret_cast<const uint4*>(p + 32);
        const unsigned d = *reinterpret_cast<const unsigned*>(p + 48);
        acc ^= a.x ^ a.w ^ b.y ^ c.z ^ d;
        at += (size_t)64 * stride;
        if (at + 64 * 64 >= span) at -= span - 64 * 64;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main() {
    const size_t span = 2u << 20;
    char* buf;
    unsigned* out;
    hipMalloc(&buf, span + 4096);
    hipMalloc(&out, 4096 * 4);
    hipMemset(buf, 1, span + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 12, rounds = 400;
    std::vector<double> t;
    for (int stride = 36; stride <= 52; ++stride) {
        hipLaunchKernelGGL(windows, dim3(blocks), dim3(64), 0, 0, buf, stride, rounds, span, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(windows, dim3(blocks), dim3(64), 0, 0, buf, stride, rounds, span, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms / 5 * 1e6 / ((double)blocks * rounds));
    }
    double best = 1e30;
    for (double v : t) best = v < best ? v : best;
    for (int i = 0; i < (int)t.size(); ++i) printf("stride %2d B: %.4f ns per wave-window (x %.2f)\n", 36 + i, t[i], t[i] / best);
    return 0;
}
