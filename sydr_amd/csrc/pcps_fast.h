// Register-resident four-step kernels for the inverse transforms of the map-free PCPS search at N = 25 000 = 125 x 200
// (25 MHz, 1 ms: BASELINE configs[1]) -- included by pcps.hip, which keeps the general mixed-radix kernels for every
// other length and for the stages that are not 1312 transforms per call.
//
// The general kernels give every thread ONE radix-R butterfly per pass and go through LDS (read R, write R, a barrier)
// for each of the three passes of a sub-transform: at 125 x 200 they execute ~140 vector instructions and ~11 LDS
// accesses per point, their SIMDs and the LDS are each ~50 % busy and the rest is barrier / latency.  Here a thread
// holds 25 (columns) or 20 (rows) points of ONE sub-transform in registers and does two radix stages on them back to
// back; the sub-transform's third stage runs after a single exchange through LDS:
//
//   columns (length 125 = 25 x 5, five threads per column; thread r owns n1 = r + 5m):
//       A_r[k'] = sum_m x[r + 5m] w25^(m k')           25-point transform in registers (5 x 5, twiddles w25^(m1 kA))
//       B_r[k'] = A_r[k'] w125^(r k')                  per-lane table read
//       Y[k' + 25q] = sum_r B_r[k'] w5^(r q)           after the exchange: thread s takes k' = s + 5 kB
//       Z[k1][n2]   = Y[k1] w_N^(n2 k1)                four-step twiddle, one table read per point
//   rows (length 200 = 20 x 10, ten threads per row; thread r owns n2 = r + 10m):
//       A_r[k'] = sum_m z[r + 10m] w20^(m k')          20-point transform in registers (5 x 4)
//       B_r[k'] = A_r[k'] w200^(r k')
//       X[k1 + 125 (k' + 20q)] = sum_r B_r[k'] w10^(r q)   after the exchange: thread s takes the registers s and s + 10
//   (w = the conjugate table entries: these are inverse transforms, unnormalised; 1/N enters with |.| as before.)
//
// The fused stages are those of the general kernels: spectrum x code spectrum on the way in (XCD-aware workgroup
// mapping: an XCD owns an eighth of the (column tile, bin) pairs and runs every PRN of a pair back to back), and the
// running (|.|/N, first index) maximum with per-wave 16-byte records instead of the map on the way out.
#pragma once

namespace fast25k {

constexpr int N1 = 125, N2 = 200, N = 25000;
constexpr int kColT = 25;            // columns per workgroup: 125 of 128 threads hold 25 points each
constexpr int kColThreads = 128;
#ifndef SDR_PCPS_FAST_ROWT
#define SDR_PCPS_FAST_ROWT 12
#endif
constexpr int kRowT = SDR_PCPS_FAST_ROWT;                 // rows per workgroup: 120 of 128 threads hold 20 points each
constexpr int kRowThreads = (10 * kRowT + 63) / 64 * 64;
constexpr int kRowPitch = 10 * kRowT + 1;                 // 16-byte slots per exchanged register (odd: ten-strided reads hit sixteen different slots)
constexpr int kRowTiles = (N1 + kRowT - 1) / kRowT;
constexpr int kRecordsPerTransform = kRowTiles * (kRowThreads / 64);

// 25-point inverse transform of v[m], m = m1 + 5 m2: result A[kA + 5 kB] in v[5 kA + kB].
__device__ __forceinline__ void idft25(double2* v, const double2* __restrict__ tw) {
#pragma unroll
    for (int m1 = 0; m1 < 5; ++m1) {
        double2 t[5] = {v[m1], v[m1 + 5], v[m1 + 10], v[m1 + 15], v[m1 + 20]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[m1 + 5 * k] = t[k];
    }
#pragma unroll
    for (int m1 = 1; m1 < 5; ++m1)
#pragma unroll
        for (int kA = 1; kA < 5; ++kA) v[m1 + 5 * kA] = cmul_conj(v[m1 + 5 * kA], tw[(N / 25) * m1 * kA]);   // (uniform: scalar loads)
#pragma unroll
    for (int kA = 0; kA < 5; ++kA) {
        double2 t[5] = {v[5 * kA], v[5 * kA + 1], v[5 * kA + 2], v[5 * kA + 3], v[5 * kA + 4]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[5 * kA + k] = t[k];
    }
}

// 20-point inverse transform of v[m], m = m1 + 4 m2: result A[kA + 5 kB] in v[4 kA + kB].
__device__ __forceinline__ void idft20(double2* v, const double2* __restrict__ tw) {
#pragma unroll
    for (int m1 = 0; m1 < 4; ++m1) {
        double2 t[5] = {v[m1], v[m1 + 4], v[m1 + 8], v[m1 + 12], v[m1 + 16]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[m1 + 4 * k] = t[k];
    }
#pragma unroll
    for (int m1 = 1; m1 < 4; ++m1)
#pragma unroll
        for (int kA = 1; kA < 5; ++kA) v[m1 + 4 * kA] = cmul_conj(v[m1 + 4 * kA], tw[(N / 20) * m1 * kA]);
#pragma unroll
    for (int kA = 0; kA < 5; ++kA) {
        double2 t[4] = {v[4 * kA], v[4 * kA + 1], v[4 * kA + 2], v[4 * kA + 3]};
        Butterfly<4, true>::run(t);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * kA + k] = t[k];
    }
}

// 10-point inverse transform of u[r], r = r1 + 2 r2: result X[qA + 5 qB] in u[2 qA + qB].
__device__ __forceinline__ void idft10(double2* u, const double2* __restrict__ tw) {
#pragma unroll
    for (int r1 = 0; r1 < 2; ++r1) {
        double2 t[5] = {u[r1], u[r1 + 2], u[r1 + 4], u[r1 + 6], u[r1 + 8]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) u[r1 + 2 * k] = t[k];
    }
#pragma unroll
    for (int qA = 1; qA < 5; ++qA) u[1 + 2 * qA] = cmul_conj(u[1 + 2 * qA], tw[(N / 10) * qA]);
#pragma unroll
    for (int qA = 0; qA < 5; ++qA) {
        const double2 a = u[2 * qA], b = u[2 * qA + 1];
        u[2 * qA] = cadd(a, b);
        u[2 * qA + 1] = csub(a, b);
    }
}

// Columns: workgroup = kColT adjacent columns n2 of one (PRN, bin) transform.  LDS: N1 * kColT + N1 double2 (52 000 B).
__global__ __launch_bounds__(kColThreads) void cols_kernel(const PassArgs a, double2* __restrict__ Z) {
    extern __shared__ double2 lds4[];
    constexpr int T = kColT;
    constexpr int tiles = N2 / T;
    static_assert(N2 % T == 0 && 5 * T <= kColThreads, "tile geometry");
    // XCD-aware mapping, as in fft4_cols_kernel<LOAD_MUL_CODE>
    const int n_prn = a.n_prn;
    const int pairs = tiles * a.nbins;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p_lo = (int)(((long long)pairs * xcd) >> 3), p_hi = (int)(((long long)pairs * (xcd + 1)) >> 3);
    const int pair = p_lo + slot / n_prn;
    if (pair >= p_hi) return;                           // (the grid is padded to the largest eighth)
    const int prn = slot - (slot / n_prn) * n_prn;
    const int tile = pair / a.nbins;
    const int bin = pair - tile * a.nbins;
    const int batch = prn * a.nbins + bin;
    const int n2_0 = tile * T;

    const int tid = threadIdx.x;
    const bool live = tid < 5 * T;
    const int r = live ? tid / T : 0, c = live ? tid - r * T : 0;
    const double2* __restrict__ tw = a.tw;
    // w125^e, e < 125, for the per-lane twiddles between the two stages (a scattered global read per register otherwise):
    // requested first, stored once the data loads are out -- no barrier between the launch and the data's round trip
    double2* w125 = lds4 + N1 * T;
    const double2 w125_mine = tw[(N / 125) * (tid < N1 ? tid : 0)];
    const int n2 = n2_0 + c;
    const int s = r;
    double2 st[5], base[5];
    double2 v[25];
    if (live) {
        const double2* __restrict__ xs = a.in + (size_t)bin * N + n2_0 + c;
        const double2* __restrict__ cs = a.code_spec + (size_t)prn * N + n2_0 + c;
        double2 xv[25], cv[25];
#pragma unroll
        for (int m = 0; m < 25; ++m) {
            const int off = N2 * (r + 5 * m);
            xv[m] = xs[off];
            cv[m] = cs[off];
        }
        // (behind the data in the queue -- loads return in order -- and in flight while the first stages compute)
        st[1] = tw[25 * n2];                                 // 25 * n2 < N
#pragma unroll
        for (int kB = 0; kB < 5; ++kB) base[kB] = tw[n2 * (s + 5 * kB)];
#pragma unroll
        for (int m = 0; m < 25; ++m) v[m] = cmulf(xv[m], cv[m]);
    }
    if (tid < N1) w125[tid] = w125_mine;
    if (live) idft25(v, tw);
    __syncthreads();                                         // the table
    if (live) {
        // B_r[k'] = A_r[k'] * conj(w125^(r k')); register g = 5 kA + kB holds k' = kA + 5 kB
#pragma unroll
        for (int g = 1; g < 25; ++g) {
            const int kp = g / 5 + 5 * (g % 5);
            v[g] = cmul_conj(v[g], w125[r * kp]);                // r * kp <= 96
        }
#pragma unroll
        for (int g = 0; g < 25; ++g) lds4[(g * 5 + r) * T + c] = v[g];
    }
    __syncthreads();
    if (!live) return;
    double2* __restrict__ zt = Z + (size_t)batch * N + n2_0 + c;
    // four-step twiddle w_N^(n2 k1), k1 = k' + 25 q: w_N^(n2 k') * (w_N^(25 n2))^q -- six scattered table reads per
    // thread instead of 25 (a wave-wide read of 64 different cache lines costs the texture path 64 cycles: with one
    // per point the kernel was bound by exactly that)
    st[2] = cmulf(st[1], st[1]);
    st[3] = cmulf(st[2], st[1]);
    st[4] = cmulf(st[2], st[2]);
#pragma unroll
    for (int kB = 0; kB < 5; ++kB) {
        double2 t[5];
#pragma unroll
        for (int rr = 0; rr < 5; ++rr) t[rr] = lds4[((5 * s + kB) * 5 + rr) * T + c];
        ibf5(t);
        const int kp = s + 5 * kB;
#ifdef SDR_EXP_NOSTORE
        {   // TIMING EXPERIMENT: everything but the stores
            const double2 z0 = cmul_conj(t[0], base[kB]);
            if (z0.x == 1.2345e300) zt[kp * N2] = z0;
#pragma unroll
            for (int q = 1; q < 5; ++q) {
                const double2 zq = cmul_conj(t[q], cmulf(base[kB], st[q]));
                if (zq.x == 1.2345e300) zt[(kp + 25 * q) * N2] = zq;
            }
        }
#else
        zt[kp * N2] = cmul_conj(t[0], base[kB]);
#pragma unroll
        for (int q = 1; q < 5; ++q) zt[(kp + 25 * q) * N2] = cmul_conj(t[q], cmulf(base[kB], st[q]));
#endif
    }
}

// Rows: workgroup = kRowT adjacent rows k1 of one transform; per-wave (maximum, first index) records instead of the map.
// LDS: 10 * kRowPitch + N2 double2 (22 560 B): the exchange takes two rounds.
__global__ __launch_bounds__(kRowThreads) void rows_kernel(const PassArgs a, const double2* __restrict__ Z) {
    extern __shared__ double2 lds4[];
    constexpr int T = kRowT;
#ifdef SDR_PCPS_ROWS_FORWARD
    const int batch = blockIdx.y;
#else
    // the transforms the column kernel wrote LAST are read FIRST: they are still in the 256 MB Infinity Cache
    const int batch = gridDim.y - 1 - blockIdx.y;
#endif
    const int k1_0 = blockIdx.x * T;
    const int tid = threadIdx.x;
    const int i = tid / 10, r = tid - i * 10;
    const bool live = tid < 10 * T && k1_0 + i < N1;
    const double2* __restrict__ tw = a.tw;
    double2* w200 = lds4 + 10 * kRowPitch;
    constexpr int kTabPerThread = (N2 + kRowThreads - 1) / kRowThreads;
    double2 wt[kTabPerThread];
#pragma unroll
    for (int q = 0; q < kTabPerThread; ++q) wt[q] = tw[(N / 200) * (tid + q * kRowThreads < N2 ? tid + q * kRowThreads : 0)];
    double2 v[20];
    if (live) {
        const double2* __restrict__ row = Z + ((size_t)batch * N1 + k1_0 + i) * N2 + r;
#pragma unroll
        for (int m = 0; m < 20; ++m) v[m] = row[10 * m];
    }
#pragma unroll
    for (int q = 0; q < kTabPerThread; ++q)
        if (tid + q * kRowThreads < N2) w200[tid + q * kRowThreads] = wt[q];
    if (live) idft20(v, tw);
    __syncthreads();                                         // the table
    if (live) {
#pragma unroll
        for (int g = 1; g < 20; ++g) {
            const int kp = g / 4 + 5 * (g % 4);
            v[g] = cmul_conj(v[g], w200[r * kp]);                // r * kp <= 171
        }
    }
    int best_i = 0x7fffffff;
    double best_v = -1.0;
    {
        const int s = r;
        const int prn = batch / a.nbins;
        const int bin = batch - prn * a.nbins;
        const int k1 = k1_0 + i;
        double best_sq = -1.0, best_x = 0.0, best_y = 0.0;
        int best_k = -1;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int g = s + 10 * o;
            const int kp = g / 4 + 5 * (g % 4);
            double2 u[10];
            // the exchange in two rounds of ten registers: half the LDS per workgroup, twice the workgroups on a CU (a
            // workgroup's load, arithmetic and exchange phases do not overlap: more of them do) -- 42 -> 35 us per sweep
            if (o) __syncthreads();
            if (live) {
#pragma unroll
                for (int gg = 0; gg < 10; ++gg) lds4[gg * kRowPitch + tid] = v[10 * o + gg];
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int rr = 0; rr < 10; ++rr) u[rr] = lds4[s * kRowPitch + rr + 10 * i];
                idft10(u, tw);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const int q = j / 2 + 5 * (j % 2);
                    const int k = k1 + N1 * (kp + 20 * q);           // position in the transform = code phase
                    const double2 x = u[j];
                    const double sq = __builtin_fma(x.x, x.x, x.y * x.y);
                    bool take = sq > best_sq;
                    // (ordering by the squared magnitude; candidates within 2^-48 of the lane's best go through the scaled
                    // hypot -- the reference's np.abs -- and an exact tie keeps the smaller index: np.argmax's first one.
                    // A wave-uniform branch: left as a lane condition the compiler flattens it and every candidate
                    // pays for two hypots)
                    const bool near = fabs(sq - best_sq) <= best_sq * 0x1p-48;
                    if (__builtin_expect(__any(near), 0)) {
                        if (near) {
                            const double m_new = hypot(x.x * a.scale, x.y * a.scale), m_old = hypot(best_x * a.scale, best_y * a.scale);
                            take = m_new > m_old || (m_new == m_old && k < best_k);
                        }
                    }
                    best_sq = take ? sq : best_sq;
                    best_x = take ? x.x : best_x;
                    best_y = take ? x.y : best_y;
                    best_k = take ? k : best_k;
                }
            }
        }
        if (live) {
            best_i = (bin + a.bin0) * N + best_k;
            best_v = 0.0 + hypot(best_x * a.scale, best_y * a.scale);   // (0.0 + |.|: the map's own rounding)
        }
    }
    wave_best(best_v, best_i);
    if ((tid & 63) == 63) {
        Best rec = {best_v, (long long)best_i};
        a.partials[((size_t)batch * gridDim.x + blockIdx.x) * (kRowThreads / 64) + (tid >> 6)] = rec;
    }
}

// the column kernel alone (the map-writing row kernels are pcps_fastn.h's)
inline void run_cols(PassArgs& a, int batch, double2* Z, hipStream_t stream) {
    const size_t shA = (size_t)(N1 * kColT + N1) * sizeof(double2);
    (void)hipFuncSetAttribute((const void*)cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shA);
    a.n_prn = batch / a.nbins;
    const int pairs = (N2 / kColT) * a.nbins;
    const unsigned gridA = 8u * (unsigned)((pairs + 7) / 8) * (unsigned)a.n_prn;
    hipLaunchKernelGGL(cols_kernel, dim3(gridA), dim3(kColThreads), shA, stream, a, Z);
}

inline void run(sdr_engine* e, PassArgs a, int batch, double2* Z, hipStream_t stream) {
    const size_t shA = (size_t)(N1 * kColT + N1) * sizeof(double2);
    const size_t shB = (size_t)(10 * kRowPitch + N2) * sizeof(double2);
    (void)hipFuncSetAttribute((const void*)cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shA);
    (void)hipFuncSetAttribute((const void*)rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shB);
    a.n_prn = batch / a.nbins;
    const int pairs = (N2 / kColT) * a.nbins;
    const unsigned gridA = 8u * (unsigned)((pairs + 7) / 8) * (unsigned)a.n_prn;
    hipLaunchKernelGGL(cols_kernel, dim3(gridA), dim3(kColThreads), shA, stream, a, Z);
    hipLaunchKernelGGL(rows_kernel, dim3(kRowTiles, batch), dim3(kRowThreads), shB, stream, a, Z);
}

}  // namespace fast25k
