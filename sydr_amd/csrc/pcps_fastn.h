// The register-resident four-step kernels of pcps_fast.h (N = 25 000 = 125 x 200) carried to the other usual code lengths:
// every length that is a multiple of 200 splits as N = N1 x 200 with the SAME row stage (200 = 20 x 10, ten threads per row)
// and a column stage of N1 = P x R2 points, P = 25 points of a column held by each of R2 threads (20 for N1 = 20: one thread
// per column, no exchange):
//      N =  4 000 (4 MHz)  =  20 x 200      columns: one 20-point transform per thread
//      N = 10 000 (10 MHz) =  50 x 200      columns: 25 x 2
//      N = 50 000 (50 MHz) = 250 x 200      columns: 25 x 10
// (N = 25 000 keeps its own, tuned kernels.)  Same fused stages: spectrum x code spectrum on the way in with the XCD-aware
// mapping, running (|.|/N, first index) maximum with per-wave records on the way out.  Included by pcps.hip after pcps_fast.h.
#pragma once

namespace fastn {

using fast25k::cmulf;
using fast25k::cmul_conj;
using fast25k::ibf5;

constexpr int N2 = 200;
constexpr int kThreadsN = 128;
constexpr int kRowT = fast25k::kRowT;
constexpr int kRowThreads = fast25k::kRowThreads;
constexpr int kRowPitch = fast25k::kRowPitch;
constexpr int row_tiles(int n1) { return (n1 + kRowT - 1) / kRowT; }
constexpr int records_per_transform(int n1) { return row_tiles(n1) * (kRowThreads / 64); }

// 25-point inverse transform of v[m], m = m1 + 5 m2: result A[kA + 5 kB] in v[5 kA + kB]  (tw = w_N^k table)
template <int N>
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
        for (int kA = 1; kA < 5; ++kA) v[m1 + 5 * kA] = cmul_conj(v[m1 + 5 * kA], tw[(N / 25) * m1 * kA]);
#pragma unroll
    for (int kA = 0; kA < 5; ++kA) {
        double2 t[5] = {v[5 * kA], v[5 * kA + 1], v[5 * kA + 2], v[5 * kA + 3], v[5 * kA + 4]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[5 * kA + k] = t[k];
    }
}

// 20-point inverse transform of v[m], m = m1 + 4 m2: result A[kA + 5 kB] in v[4 kA + kB]
template <int N>
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

// 10-point inverse transform of u[r], r = r1 + 2 r2: result X[qA + 5 qB] in u[2 qA + qB]
template <int N>
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

// Columns: workgroup = T adjacent columns n2 of one (PRN, bin) transform; R2 threads per column.
// LDS (R2 > 1): N1 * T exchanged points + the N1 twiddles w_N1^e.
template <int N1, int R2, int T>
__global__ __launch_bounds__(kThreadsN) void cols_kernel(const PassArgs a, double2* __restrict__ Z) {
    constexpr int N = N1 * N2, P = N1 / R2;
    static_assert((P == 25 && (R2 == 2 || R2 == 5 || R2 == 10)) || (P == 20 && R2 == 1), "column split");
    static_assert(N2 % T == 0 && R2 * T <= kThreadsN, "tile geometry");
    extern __shared__ double2 lds4[];
    constexpr int tiles = N2 / T;
    // XCD-aware mapping, as in fast25k::cols_kernel
    const int n_prn = a.n_prn;
    const int pairs = tiles * a.nbins;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p_lo = (int)(((long long)pairs * xcd) >> 3), p_hi = (int)(((long long)pairs * (xcd + 1)) >> 3);
    const int pair = p_lo + slot / n_prn;
    if (pair >= p_hi) return;
    const int prn = slot - (slot / n_prn) * n_prn;
    const int tile = pair / a.nbins;
    const int bin = pair - tile * a.nbins;
    const int batch = prn * a.nbins + bin;
    const int n2_0 = tile * T;

    const int tid = threadIdx.x;
    const bool live = tid < R2 * T;
    const int r = live ? tid / T : 0, c = live ? tid - r * T : 0;
    const double2* __restrict__ tw = a.tw;
    const int n2 = n2_0 + c;
    double2 v[P];
    if (live) {
        const double2* __restrict__ xs = a.in + (size_t)bin * N + n2;
        const double2* __restrict__ cs = a.code_spec + (size_t)prn * N + n2;
        double2 xv[P], cv[P];
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int off = N2 * (r + R2 * m);
            xv[m] = xs[off];
            cv[m] = cs[off];
        }
#pragma unroll
        for (int m = 0; m < P; ++m) v[m] = cmulf(xv[m], cv[m]);
    }
    double2* __restrict__ zt = Z + (size_t)batch * N + n2;
    if constexpr (R2 == 1) {
        // the whole column in one thread: Y[kA + 5 kB] in v[4 kA + kB]; four-step twiddle w_N^(n2 k1), store
        if (!live) return;
        idft20<N>(v, tw);
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            const int k1 = g / 4 + 5 * (g % 4);
            zt[k1 * N2] = g == 0 ? v[0] : cmul_conj(v[g], tw[n2 * k1]);          // n2 * k1 < 200 * 20 = N
        }
    } else {
        double2* wn1 = lds4 + N1 * T;                      // w_N1^e = w_N^(200 e), e < N1
        for (int e = tid; e < N1; e += kThreadsN) wn1[e] = tw[N2 * e];
        double2 st1 = make_double2(1.0, 0.0);
        if (live) {
            st1 = tw[25 * n2];                              // 25 * n2 < 5000 <= N
            idft25<N>(v, tw);
        }
        __syncthreads();                                    // the table
        if (live) {
            // B_r[k'] = A_r[k'] * conj(w_N1^(r k')); register g = 5 kA + kB holds k' = kA + 5 kB
#pragma unroll
            for (int g = 1; g < 25; ++g) {
                const int kp = g / 5 + 5 * (g % 5);
                v[g] = cmul_conj(v[g], wn1[r * kp]);        // r * kp <= (R2 - 1) * 24 < N1
            }
#pragma unroll
            for (int g = 0; g < 25; ++g) lds4[(g * R2 + r) * T + c] = v[g];
        }
        __syncthreads();
        if (!live) return;
        // four-step twiddle w_N^(n2 k1), k1 = k' + 25 q:  w_N^(n2 k') * (w_N^(25 n2))^q
        double2 st[R2];
        st[0] = make_double2(1.0, 0.0);
        st[1] = st1;
#pragma unroll
        for (int q = 2; q < R2; ++q) st[q] = cmulf(st[q - 1], st1);
        const int s = r;
#pragma unroll
        for (int j = 0; j < (25 + R2 - 1) / R2; ++j) {
            const int g = s + R2 * j;
            if (g < 25) {
                const int kp = g / 5 + 5 * (g % 5);
                double2 t[R2];
#pragma unroll
                for (int rr = 0; rr < R2; ++rr) t[rr] = lds4[(g * R2 + rr) * T + c];
                const double2 base = tw[n2 * kp];           // n2 * kp < 200 * 25 <= N
                if constexpr (R2 == 5) {
                    ibf5(t);
#pragma unroll
                    for (int q = 0; q < 5; ++q) zt[(kp + 25 * q) * N2] = cmul_conj(t[q], q ? cmulf(base, st[q]) : base);
                } else if constexpr (R2 == 2) {
                    Butterfly<2, true>::run(t);
                    zt[kp * N2] = cmul_conj(t[0], base);
                    zt[(kp + 25) * N2] = cmul_conj(t[1], cmulf(base, st[1]));
                } else {
                    idft10<N>(t, tw);                        // X[qA + 5 qB] in t[2 qA + qB]
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int q = i / 2 + 5 * (i % 2);
                        zt[(kp + 25 * q) * N2] = cmul_conj(t[i], q ? cmulf(base, st[q]) : base);
                    }
                }
            }
        }
    }
}

// Rows: fast25k::rows_kernel with the number of rows (and with it N) as a parameter, and with the map-writing stores:
// STORE_MAG_MAX keeps the running maximum (map-free search); STORE_MAG_ACC writes / adds |.|/N into the map (a search
// that returns its map, non-coherent sums); STORE_CPLX_ACC writes / adds the scaled complex value (coherent sums).
// A thread's 20 outputs sit N1 apart; adjacent rows of the tile are adjacent in memory (runs of kRowT values).
template <int N1, int STORE>
__global__ __launch_bounds__(kRowThreads) void rows_kernel(const PassArgs a, const double2* __restrict__ Z) {
    constexpr int N = N1 * N2;
    extern __shared__ double2 lds4[];
    constexpr int T = kRowT;
    const int batch = gridDim.y - 1 - blockIdx.y;            // (the transforms written last are read first)
    const int k1_0 = blockIdx.x * T;
    const int tid = threadIdx.x;
    const int i = tid / 10, r = tid - i * 10;
    const bool live = tid < 10 * T && k1_0 + i < N1;
    const double2* __restrict__ tw = a.tw;
    double2* w200 = lds4 + 10 * kRowPitch;                   // w200^e, e < 200 (behind the ten exchange slots)
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
    if (live) idft20<N>(v, tw);
    __syncthreads();
    if (live) {
#pragma unroll
        for (int g = 1; g < 20; ++g) {
            const int kp = g / 4 + 5 * (g % 4);
            v[g] = cmul_conj(v[g], w200[r * kp]);
        }
    }
    // The exchange goes through LDS in two rounds of ten registers: half the LDS per workgroup, twice the workgroups on a CU
    // (the load, arithmetic and exchange phases of a workgroup do not overlap: more of them do) -- 42 -> 35 us per sweep of
    // 437 transforms at N = 25 000.  Every thread takes the barriers, live or not.
    int best_i = 0x7fffffff;
    double best_v = -1.0;
    {
        const int s = r;
        const int prn = batch / a.nbins;
        const int bin = batch - prn * a.nbins;
        const int k1 = k1_0 + i;
        const size_t base_o = (size_t)batch * N + k1;
        double best_sq = -1.0, best_x = 0.0, best_y = 0.0;
        int best_k = -1;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int g = s + 10 * o;
            const int kp = g / 4 + 5 * (g % 4);
            if (o) __syncthreads();
            if (live) {
#pragma unroll
                for (int gg = 0; gg < 10; ++gg) lds4[gg * kRowPitch + tid] = v[10 * o + gg];
            }
            __syncthreads();
            if (live) {
                double2 u[10];
#pragma unroll
                for (int rr = 0; rr < 10; ++rr) u[rr] = lds4[s * kRowPitch + rr + 10 * i];
                idft10<N>(u, tw);
                if constexpr (STORE == STORE_MAG_ACC) {
                    double old[10];
                    if (!a.first_block) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) old[j] = a.map[base_o + N1 * (kp + 20 * (j / 2 + 5 * (j % 2)))];
                    }
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const double mag = hypot(u[j].x * a.scale, u[j].y * a.scale);
                        a.map[base_o + N1 * (kp + 20 * (j / 2 + 5 * (j % 2)))] = a.first_block ? 0.0 + mag : old[j] + mag;
                    }
                } else if constexpr (STORE == STORE_CPLX_ACC) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const size_t at = base_o + N1 * (kp + 20 * (j / 2 + 5 * (j % 2)));
                        const double2 sc = make_double2(u[j].x * a.scale, u[j].y * a.scale);
                        a.csum[at] = a.first_block ? sc : cadd(a.csum[at], sc);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const int q = j / 2 + 5 * (j % 2);
                        const int k = k1 + N1 * (kp + 20 * q);
                        const double2 x = u[j];
                        const double sq = __builtin_fma(x.x, x.x, x.y * x.y);
                        bool take = sq > best_sq;
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
        }
        if constexpr (STORE != STORE_MAG_MAX) return;
        if (live) {
            best_i = (bin + a.bin0) * N + best_k;
            best_v = 0.0 + hypot(best_x * a.scale, best_y * a.scale);
        }
    }
    wave_best(best_v, best_i);
    if ((tid & 63) == 63) {
        Best rec = {best_v, (long long)best_i};
        a.partials[((size_t)batch * gridDim.x + blockIdx.x) * (kRowThreads / 64) + (tid >> 6)] = rec;
    }
}

template <int N1, int STORE>
inline void run_rows(const PassArgs& a, int batch, const double2* Z, hipStream_t stream) {
    const size_t shB = (size_t)(10 * kRowPitch + N2) * sizeof(double2);
    (void)hipFuncSetAttribute((const void*)rows_kernel<N1, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shB);
    hipLaunchKernelGGL((rows_kernel<N1, STORE>), dim3(row_tiles(N1), batch), dim3(kRowThreads), shB, stream, a, Z);
}

template <int N1, int R2, int T, int STORE>
inline void run_n(PassArgs a, int batch, double2* Z, hipStream_t stream) {
    const size_t shA = R2 > 1 ? (size_t)(N1 * T + N1) * sizeof(double2) : 0;
    if (shA > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)cols_kernel<N1, R2, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shA);
    a.n_prn = batch / a.nbins;
    const int pairs = (N2 / T) * a.nbins;
    const unsigned gridA = 8u * (unsigned)((pairs + 7) / 8) * (unsigned)a.n_prn;
    hipLaunchKernelGGL((cols_kernel<N1, R2, T>), dim3(gridA), dim3(kThreadsN), shA, stream, a, Z);
    run_rows<N1, STORE>(a, batch, Z, stream);
}

inline bool handles(int N) { return N == 4000 || N == 10000 || N == 50000; }
inline int records(int N) { return records_per_transform(N / N2); }
template <int STORE>
inline void run(int N, PassArgs a, int batch, double2* Z, hipStream_t stream) {
    if (N == 4000) run_n<20, 1, 100, STORE>(a, batch, Z, stream);
    else if (N == 10000) run_n<50, 2, 50, STORE>(a, batch, Z, stream);
    else run_n<250, 10, 10, STORE>(a, batch, Z, stream);
}

}  // namespace fastn
