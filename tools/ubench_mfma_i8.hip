// Costing of an exact-integer MFMA formulation of the E/P/L block sums on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8 && tools/ubench_mfma_i8
//
// The formulation: a chip-aligned block's sum P = sum_k x_k * r_k (x: int8 I/Q samples, r: fp64 in-block rotations)
// becomes an int8 x int8 matrix product if every rotation component is split into eight signed 7-bit limbs
// (r ~ sum_l d_l * 2^(-7l-6), |d_l| <= 64): A[block][k] = the block's bytes (K = 64: 32 samples, I/Q interleaved),
// B[k][column] = limb l of the +-cos / +-sin pattern of one captured prefix sum.  V_MFMA_I32_16X16X64_I8 forms 16 blocks x
// 16 columns exactly; the limbs are recombined in fp64.  What has to be known to price it:
//   1. the operand layout of v_mfma_i32_16x16x64_i8 (checked here with exact integer data, asymmetric B);
//   2. its issue cost per SIMD (back-to-back, independent accumulators);
//   3. the cost of the recombination: every int32 the MFMA delivers is one limb of one component of one captured sum,
//      so a block with C captured complex sums hands the VALU 16*C integers to turn into 2*C doubles
//      (pairs of limbs joined by v_lshl_add_u32, v_cvt_f64_i32, a Horner chain of v_fma_f64).
// Result of the costing (measured on MI355X, gpurun_out/ubench_mfma_i8.txt; discussion in DESIGN.md K1): the layout is
// the bf16 form's at twice the K (exact on random data); the MFMA issues every 17.5 cycles with independent
// accumulators (44 on one dependent chain) -- 16 blocks x 16 limb-columns in the time of ~4 VALU instructions; the
// recombination is 13 VALU instructions per double (52 cycles per SIMD, 78 on a lone wave's dependent chain).  Per round
// of 64 blocks with the minimum of two captured sums per block (the halves before and after the outer taps' switch):
// 8 MFMAs (~35 instruction slots) + 4 recombinations per lane (52) + the two single samples of the M / M + 1 and
// 12 / 13 cases by VALU (~16) + handing the sums to the lanes that own the blocks (~24) = ~127 against the ~160 of the
// fp64 FMA loop they replace, in a round of ~250: a projected 1.15x for the kernel, far below the 1.5x that would justify
// a second correlator core.  Not built.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- 1. layout: D = A (16 x 64) * B (64 x 16), one wave
__global__ void layout_kernel(const int8_t* A, const int8_t* B, int* D) {
    const int l = threadIdx.x;
    // expected map (as the bf16 form at twice the K): lane l holds A[row l & 15][k = 16 * (l >> 4) + j], j = 0..15
    // and B[k = 16 * (l >> 4) + j][col l & 15]
    v4i a, b, c = {0, 0, 0, 0};
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) {
        ab[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        bb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    // C/D map (dtype-independent on gfx950): col = lane & 15, row = 4 * (lane >> 4) + reg
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// ---- 2. issue rate: N independent accumulators, back to back
template <int ACC>
__global__ void mfma_rate_kernel(unsigned long long* out, int iters, int seed) {
    v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, seed + 1, seed + 2, (int)threadIdx.x};
    v4i c[ACC];
    for (int i = 0; i < ACC; ++i) c[i] = (v4i){i, i, i, i};
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int sink = 0;
    for (int i = 0; i < ACC; ++i) sink += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0, out[blockIdx.x * 2 + 1] = (unsigned long long)sink;
}

// ---- 3. recombination: 8 limbs (as 4 N-tiles' registers of one lane, two limbs per register pair joined in int32)
//         -> one double; per lane and M-tile: 4 rows x this
__global__ void recombine_rate_kernel(unsigned long long* out, int iters, int seed) {
    int limb[8][4];
    for (int l = 0; l < 8; ++l)
        for (int r = 0; r < 4; ++r) limb[l][r] = seed * (l + 3) + threadIdx.x * (r + 1);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double s14 = 6.103515625e-05;   // 2^-14
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int l = 0; l < 8; ++l) asm volatile("" : "+v"(limb[l][r]));   // (fresh MFMA results every time: nothing is hoisted)
            // pairs of limbs exactly in int32 (|limb sum| < 2^18), four conversions, a Horner chain
            const int p0 = (limb[0][r] << 7) + limb[1][r], p1 = (limb[2][r] << 7) + limb[3][r];
            const int p2 = (limb[4][r] << 7) + limb[5][r], p3 = (limb[6][r] << 7) + limb[7][r];
            double v = (double)p3;
            v = __builtin_fma(v, s14, (double)p2);
            v = __builtin_fma(v, s14, (double)p1);
            v = __builtin_fma(v, s14, (double)p0);
            acc[r] += v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678) out[1] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    // 1. layout
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    srand(7);
    for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B) v = (int8_t)(rand() % 129 - 64);
    std::vector<int> ref(256, 0), got(256, 0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 64; ++k) ref[i * 16 + j] += (int)A[i * 64 + k] * (int)B[k * 16 + j];
    int8_t *dA, *dB;
    int* dD;
    unsigned long long* dT;
    CK(hipMalloc(&dA, A.size()));
    CK(hipMalloc(&dB, B.size()));
    CK(hipMalloc(&dD, 256 * sizeof(int)));
    CK(hipMalloc(&dT, 4096 * sizeof(unsigned long long)));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(got.data(), dD, 256 * sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += got[i] != ref[i];
    printf("layout v_mfma_i32_16x16x64_i8: A[l&15][16*(l>>4)+j], B[16*(l>>4)+j][l&15], D[4*(l>>4)+r][l&15]: %s (%d of 256 wrong)\n",
           bad ? "WRONG" : "exact", bad);
    // 2. MFMA issue rate: one wave per SIMD (256 threads = 4 waves per CU, one workgroup per CU)
    const int iters = 20000;
    unsigned long long t[2 * 256];
    auto cycles = [&](auto launch, int per_iter, const char* what) {
        launch();
        launch();
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, dT, sizeof(t), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int b = 0; b < 256; ++b) mean += (double)t[2 * b];
        mean /= 256;
        printf("%-64s %8.2f cycles per instruction-group (%d per iteration)\n", what, mean / iters / per_iter, per_iter);
    };
    cycles([&] { hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(256), dim3(256), 0, 0, dT, iters, 3); }, 1,
           "v_mfma_i32_16x16x64_i8, one dependent accumulator");
    cycles([&] { hipLaunchKernelGGL(mfma_rate_kernel<4>, dim3(256), dim3(256), 0, 0, dT, iters, 3); }, 4,
           "v_mfma_i32_16x16x64_i8, four independent accumulators");
    cycles([&] { hipLaunchKernelGGL(mfma_rate_kernel<16>, dim3(256), dim3(256), 0, 0, dT, iters, 3); }, 16,
           "v_mfma_i32_16x16x64_i8, sixteen independent accumulators");
    // 3. recombination: per (row register) 8 limbs -> 1 double; 4 rows per iteration
    cycles([&] { hipLaunchKernelGGL(recombine_rate_kernel, dim3(256), dim3(256), 0, 0, dT, iters, 5); }, 4,
           "8 limbs -> fp64 (2 lshl_add x2, 4 cvt, 3 fma, 1 add) per row");
    cycles([&] { hipLaunchKernelGGL(recombine_rate_kernel, dim3(256), dim3(1024), 0, 0, dT, iters, 5); }, 4,
           "the same, four waves per SIMD");
    printf("per round of 64 blocks, C captured complex sums per block: MFMA 4*C*16 cycles; recombination 2*C*4*4 row-groups\n");
    return bad ? 2 : 0;
}
