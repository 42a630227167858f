// What a resident "tick server" kernel could save: the round trip host -> device -> host through page-locked memory (a
// workgroup that stays resident, woken by a sequence word the host writes, answering with a word the host polls) against the
// same handshake done with a kernel launch + hipStreamSynchronize per round.  Every wait is bounded (the kernel gives up
// after ~0.2 s without a new sequence number, the host after 1 s): nothing can hang.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pingpong.hip -o tools/ubench_pingpong && tools/ubench_pingpong
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void server(volatile unsigned* request, volatile unsigned* answer, int rounds, unsigned long long give_up_cycles) {
    if (threadIdx.x != 0) return;
    for (int i = 1; i <= rounds; ++i) {
        const unsigned long long t0 = wall_clock64();
        while (*request < (unsigned)i) {
            if (wall_clock64() - t0 > give_up_cycles) return;         // (100 MHz counter)
            __builtin_amdgcn_s_sleep(1);
        }
        *answer = (unsigned)i;
        __threadfence_system();
    }
}

__global__ void one_round(volatile unsigned* answer, unsigned i) {
    if (threadIdx.x == 0) *answer = i;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    unsigned *request, *answer;
    hipHostMalloc(&request, 64, hipHostMallocDefault);
    hipHostMalloc(&answer, 64, hipHostMallocDefault);
    *request = 0, *answer = 0;
    const int rounds = 20000;
    hipStream_t s;
    hipStreamCreate(&s);
    // (a) resident kernel
    hipLaunchKernelGGL(server, dim3(1), dim3(64), 0, s, request, answer, rounds, 20000000ull);
    volatile unsigned* ans = answer;
    double t0 = now_us(), worst = 0.0;
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        const double a = now_us();
        *(volatile unsigned*)request = (unsigned)i;
        while (*ans < (unsigned)i)
            if (now_us() - a > 1e6) goto out;
        const double d = now_us() - a;
        worst = d > worst ? d : worst;
        done = i;
    }
out:
    {
        const double per = (now_us() - t0) / (done ? done : 1);
        hipStreamSynchronize(s);
        printf("resident kernel, request/answer through page-locked words: %d rounds, %.2f us per round trip (worst %.1f)\n", done, per, worst);
    }
    // (b) a launch and a synchronisation per round
    *answer = 0;
    for (int i = 1; i <= 200; ++i) {
        hipLaunchKernelGGL(one_round, dim3(1), dim3(64), 0, s, answer, (unsigned)i);
        hipStreamSynchronize(s);
    }
    t0 = now_us();
    for (int i = 1; i <= 5000; ++i) {
        hipLaunchKernelGGL(one_round, dim3(1), dim3(64), 0, s, answer, (unsigned)i);
        hipStreamSynchronize(s);
    }
    printf("launch + hipStreamSynchronize per round: %.2f us\n", (now_us() - t0) / 5000);
    // (c) two dependent launches + one synchronisation (the two-launch tick's shape)
    t0 = now_us();
    for (int i = 1; i <= 5000; ++i) {
        hipLaunchKernelGGL(one_round, dim3(1), dim3(64), 0, s, answer, (unsigned)i);
        hipLaunchKernelGGL(one_round, dim3(1), dim3(64), 0, s, answer, (unsigned)i);
        hipStreamSynchronize(s);
    }
    printf("two launches + hipStreamSynchronize per round: %.2f us\n", (now_us() - t0) / 5000);
    return 0;
}
