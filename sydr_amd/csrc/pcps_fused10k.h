// One workgroup per (PRN, Doppler bin) of a PCPS search at N = 10 000 = 50 x 200 (10 MHz, 1 ms blocks: the rate and the
// 1 ms x 10 non-coherent search of the reference's shipped configuration, config/receiver.ini:18-20 and
// config/channels/channel_GPS_L1CA_kaplan.ini:6-10) when the caller wants indices and ratio, not the map.  Included by
// pcps_fused.hip behind pcps_fused.h (its literal twiddles, idft10c, the radix-5 codelet).
//
// The two-kernel path accumulates the map in memory: per (PRN, bin) and millisecond block 160 KB of intermediate out and
// back plus 80 KB of map read and written -- 7.2 GB of counter traffic per 32-PRN x 34-bin x 10-block search.  Here the whole
// 160 KB transform lives in the workgroup's LDS (two buffers of 25 rows x 200: no parking in registers as at N = 25 000),
// the non-coherent sum of |.|/N over the blocks (acquisition.py:57-70: `+= abs(ifft(fft(x) * codeFFT))`) stays in registers
// -- twenty values per lane (the PRN's code spectrum is re-read per block: it sits in the XCD's L2);
// when the last block is in, the unit's map row is complete in the register file and BOTH peaks come out of it:
// the row's first maximum, and the maximum over the columns TwoCorrelationPeakComparison allows around it
// (acquisition.py:98-111).  The PRN's winning row is then simply the unit with the largest first maximum, so there is no
// second sweep: one launch, then one small kernel per PRN.
//
//   column stage (50 = 10 x 5; five threads per column, 100 columns at a time, two items per thread):
//       A_r[k'] = sum_m x[r + 5m] c[r + 5m] w10^(m k'),  B_r[k'] = A_r[k'] w50^(r k')     10-point transform in registers
//     k' = qA + 5 qB goes to buffer qB (round qB): E[(qA, r)][n2]
//   round rho = 0, 1 (25 rows k1 = qA + 5 rho + 10 q, in place in buffer rho): as pcps_fused.h's rounds --
//       Y[k' + 10 q] = sum_r B_r[k'] w5^(r q);  rows of 200 = 10 x 20 with twenty threads per row: four-step twiddle,
//       10-point transform, w200 twiddle, exchange in place, radix-2 decimation in frequency, 10-point transform;
//       acc[rho][g] += hypot(x / N, y / N)                                          (the reference's np.abs, block by block)
#pragma once

namespace fused10k {

using fast25k::cmul_conj;
using fast25k::cmulf;
using fast25k::ibf5;
using fused25k::idft10c;
using fused25k::kW20X;
using fused25k::kW20Y;

constexpr int N1 = 50, N2 = 200, N = 10000;
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kBuf = 25 * N2;                 // double2 per round buffer
constexpr int kTab50 = 40, kTab200 = 180;     // w50^e (e <= 36); w200^(e k''), k'' = 1..9, e < 20, stored [k'' - 1][e] (pcps_fused.h): both resident
constexpr size_t kLdsBytes = (size_t)(2 * kBuf + 240) * sizeof(double2);
static_assert(kLdsBytes == 160 * 1024 && kTab50 + kTab200 <= 240, "the whole LDS of a CU");

struct UnitRecord {        // what a (PRN, bin) unit leaves: its row's first maximum and the second peak of the SAME row
    double top;            // < 0: none
    long long index;       // bin * N + code phase
    double second;         // maximum over the allowed columns around `index` (< 0: none)
    double pad;
};

struct Args {
    const double2* spec;       // [noncoh][nbins][N] forward spectra of the Doppler-mixed blocks
    // shared spectra (pcps.hip): a block is blk_stride elements long, bin b's spectrum starts spec_off[b] elements into it
    // (spec_off == nullptr: b * N, blk_stride = nbins * N)
    const long long* spec_off;
    long long blk_stride;
    const double2* code_spec;  // [n_prn][N]
    const double2* tw;         // exp(-2 pi i m / N)
    int n_prn, nbins, noncoh, spc;
    double scale;              // 1 / N
    UnitRecord* records;       // [prn][bin]
};

// (value, flat index) maximum over the workgroup: larger value, smaller index on ties; every thread gets the result.
__device__ __forceinline__ void block_best(double& v, int& i, double* sh_v, int* sh_i, int tid) {
    wave_best(v, i);
    if ((tid & 63) == 63) sh_v[tid >> 6] = v, sh_i[tid >> 6] = i;
    __syncthreads();
    v = sh_v[0], i = sh_i[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        const double ov = sh_v[w];
        const int oi = sh_i[w];
        const bool take = ov > v || (ov == v && oi < i);
        v = take ? ov : v;
        i = take ? oi : i;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void search_kernel(const Args a) {
    extern __shared__ double2 lds4[];
    double2* const tab50 = lds4 + 2 * kBuf;
    double2* const tab200 = tab50 + kTab50;
    const double2* __restrict__ tw = a.tw;
    const int tid = threadIdx.x;
    if (tid < 37) tab50[tid] = tw[(N / 50) * tid];
    if (tid < 180) tab200[tid] = tw[(N / 200) * ((tid % 20) * (tid / 20 + 1))];

    // units in bin-major order: the 32 workgroups of an XCD work on the PRNs of ONE bin at a time and share its ten
    // spectra through their L2
    const int n_units = a.n_prn * a.nbins;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (n_units + 7) / 8;
    const int u_end = min(n_units, (xcd + 1) * per_xcd);
    for (int unit = xcd * per_xcd + slot; unit < u_end; unit += (int)(gridDim.x >> 3)) {
        const int bin = __builtin_amdgcn_readfirstlane(unit / a.n_prn);
        const int prn = __builtin_amdgcn_readfirstlane(unit - bin * a.n_prn);
        // (roles from an opaque copy of the thread number: pcps_fused.h)
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const bool live = t_ < 500;
        const int r = live ? t_ / 100 : 0, c = live ? t_ - 100 * r : 0;
        const int cb = r * N2 + c;
        // the row stages' lanes laid out for the LDS banks, and the exchange swizzled, exactly as in pcps_fused.h (round 6;
        // tools/lds_conflicts_fused.py): read groups of sixteen lanes take sixteen elements that differ modulo 16, write
        // groups of eight lanes eight that differ modulo 8
        const int l5 = t_ & 31;
        const int gq = 2 * ((t_ >> 5) & 1) + ((l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28 ? 1 : 0);
        const int gi = l5 < 4 ? l5 : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : (l5 < 28 ? l5 - 12 : l5 - 16)));
        auto swz = [](int row) { return 2 * ((row >> 1) & 3) + ((row >> 1) & 1); };
        const int g1 = 4 * (t_ >> 6) + gq;
        const int h = tid >> 8;
        int role1, role2;      // (row, e | k'', swizzle, k1b | first output) packed, < 0: no part in the stage; unpacked per round
        {
            int ri, re;
            bool live1 = true;
            if (g1 <= 24) {
                const int n = g1 >> 1;
                ri = 4 * (n >> 1) + (n & 1) + 2 * (g1 & 1);
                re = gi;
            } else {
                const int u = g1 - 25, q = gi >> 2;
                ri = 8 * (u >> 1) + ((u & 1) ? 0 : 2) + (q & 1) + 4 * (q >> 1);
                re = 16 + (gi & 3);
                if (g1 == 31) ri = 24, live1 = q == 0;
            }
            role1 = live1 ? (ri | re << 5 | swz(ri) << 10 | ((ri / 5) + 10 * (ri % 5)) << 13) : -1;   // (k1b: its row's k1 = k1b + 5 rho)
            const int g2 = 4 * ((t_ >> 6) & 3) + gq;
            int si, sk;
            bool live2 = true;
            if (g2 < 12) si = 2 * g2 + (gi >> 3), sk = gi & 7;
            else if (g2 < 15) si = 8 * (g2 - 12) + (gi >> 1), sk = 8 + (gi & 1);
            else si = 24, sk = gi, live2 = gi < 10;
            role2 = live2 ? (si | sk << 5 | swz(si) << 10 | ((si / 5) + 10 * (si % 5) + N1 * (sk + 10 * h)) << 13) : -1;
        }
        const bool live2 = role2 >= 0;
        const int kf0 = (role2 >> 13) & 4095;

        // (the PRN's code spectrum at this lane's points is re-read with every block: kept in registers for the whole unit --
        // FUSED10K_KEEP_CODE 1, 80 registers beside the 40 of the running sums -- the kernel spilled 33 and a search measured
        // 0.743 against 0.712 ms of kernel time)
#ifndef FUSED10K_KEEP_CODE
#define FUSED10K_KEEP_CODE 0
#endif
        const double2* __restrict__ cs = a.code_spec + (size_t)prn * N + cb;
#if FUSED10K_KEEP_CODE
        // the PRN's code spectrum at this lane's twenty input points: once per unit
        double2 cv[2][10];
        if (live) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < 10; ++m) cv[j][m] = cs[100 * j + N2 * 5 * m];
        }
#endif
        double acc[2][10];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 10; ++g) acc[j][g] = 0.0;

        for (int blk = 0; blk < a.noncoh; ++blk) {
            const double2* __restrict__ xs = a.spec + (size_t)blk * (size_t)a.blk_stride + (a.spec_off ? (size_t)a.spec_off[bin] : (size_t)bin * N) + cb;
            __syncthreads();   // the tables (first block); the previous block's / unit's readers are done with the buffers
            // ---- column stage (the block's spectrum is read as it is needed: its item-0 values requested a block ahead --
            // 40 more registers across the rounds -- made the kernel spill 199 of them: 1.04 instead of 0.75 ms per search, with the code spectrum kept in registers)
            if (live) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double2 v[10];
#pragma unroll
                    for (int m = 0; m < 10; ++m) v[m] = xs[100 * j + N2 * 5 * m];
#if FUSED10K_KEEP_CODE
#pragma unroll
                    for (int m = 0; m < 10; ++m) v[m] = cmulf(v[m], cv[j][m]);
#else
                    {
                        double2 cc[10];
#pragma unroll
                        for (int m = 0; m < 10; ++m) cc[m] = cs[100 * j + N2 * 5 * m];
#pragma unroll
                        for (int m = 0; m < 10; ++m) v[m] = cmulf(v[m], cc[m]);
                    }
#endif
                    idft10c(v);                             // A[k' = qA + 5 qB] in v[2 qA + qB]
#pragma unroll
                    for (int g = 1; g < 10; ++g) {
                        const int kp = g / 2 + 5 * (g % 2);
                        v[g] = cmul_conj(v[g], tab50[r * kp]);
                    }
                    double2* const e0 = lds4 + cb + 100 * j;
#pragma unroll
                    for (int qA = 0; qA < 5; ++qA) {
                        e0[5 * N2 * qA] = v[2 * qA];
                        e0[kBuf + 5 * N2 * qA] = v[2 * qA + 1];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) {
                double2* const X = lds4 + rho * kBuf;
                int r1_ = role1, r2_ = role2;
                asm volatile("" : "+v"(r1_), "+v"(r2_));            // (this round's own unpacking: nothing of it lives across rounds)
                const bool live1 = r1_ >= 0;
                const int ri = r1_ & 31, re = (r1_ >> 5) & 31, sw1 = (r1_ >> 10) & 7, k1b = (r1_ >> 13) & 127;
                const int si = r2_ & 31, sk = (r2_ >> 5) & 15, sw2 = (r2_ >> 10) & 7;
                const int k1 = k1b + 5 * rho;
                const double2 tw_base = tw[k1 * re], tw_step = tw[20 * k1];
                // Y[k' + 10 q], in place
                if (live) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double2* const col = X + 5 * cb - 4 * c + 100 * j;      // (r * 5) * N2 + c
                        double2 t5[5];
#pragma unroll
                        for (int rr = 0; rr < 5; ++rr) t5[rr] = col[rr * N2];
                        ibf5(t5);
#pragma unroll
                        for (int q = 0; q < 5; ++q) col[q * N2] = t5[q];
                    }
                }
                __syncthreads();
                // rows, first stage
                double2 z[10];
                if (live1) {
                    const double2* __restrict__ rowz = X + ri * N2 + re;
#pragma unroll
                    for (int m = 0; m < 10; ++m) z[m] = rowz[20 * m];
                    double2 t = tw_base;
                    z[0] = cmul_conj(z[0], t);
#pragma unroll
                    for (int m = 1; m < 10; ++m) {
                        t = cmulf(t, tw_step);
                        z[m] = cmul_conj(z[m], t);
                    }
                    idft10c(z);
#pragma unroll
                    for (int g = 1; g < 10; ++g) {
                        const int kpp = g / 2 + 5 * (g % 2);
                        z[g] = cmul_conj(z[g], tab200[20 * (kpp - 1) + re]);
                    }
                }
                __syncthreads();
                if (live1) {
                    // element (e, k'') at (10 e + k'') ^ swizzle(row): bits 4..6 of the 32-bit LDS byte address (pcps_fused.h)
                    typedef __attribute__((address_space(3))) char lds_char;
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
                    const unsigned roww = (unsigned)(size_t)((lds_char*)(X + ri * N2)) + 160u * (unsigned)re;
                    const unsigned sw16 = (unsigned)sw1 << 4;
#pragma unroll
                    for (int g = 0; g < 10; ++g) {
                        const int kpp = g / 2 + 5 * (g % 2);
                        *(lds_f64x2*)(size_t)((roww + 16u * kpp) ^ sw16) = f64x2{z[g].x, z[g].y};
                    }
                }
                __syncthreads();
                // rows, second stage; the block's magnitudes into the running sums
                if (live2) {
                    double2 u[10];
                    const double2* __restrict__ const row = X + si * N2;
                    const double2* __restrict__ const rowq[4] = {row + (sk ^ sw2), row + ((sk + 2) ^ sw2), row + ((sk + 4) ^ sw2), row + ((sk + 6) ^ sw2)};
#pragma unroll
                    for (int t0 = 0; t0 < 10; t0 += 5) {
                        double2 lo[5], hi[5];
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            const int e_lo = t0 + t, e_hi = t0 + t + 10;
                            lo[t] = rowq[e_lo & 3][10 * e_lo - 2 * (e_lo & 3)];
                            hi[t] = rowq[e_hi & 3][10 * e_hi - 2 * (e_hi & 3)];
                        }
#pragma unroll
                        for (int t = 0; t < 5; ++t) u[t0 + t] = h ? csub(lo[t], hi[t]) : cadd(lo[t], hi[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (h) {
#pragma unroll
                        for (int t = 1; t < 10; ++t) u[t] = cmul_conj(u[t], make_double2(kW20X[t], kW20Y[t]));
                    }
                    idft10c(u);
#pragma unroll
                    for (int g = 0; g < 10; ++g) {
                        // |.|/N as sqrt(re^2 + im^2) of the scaled parts: within an ulp of the hypot the map-writing kernels
                        // call (as that is of NumPy's), a third of its instructions -- twenty magnitudes per lane and block
                        // were 40 % of this kernel's arithmetic
                        const double xr = u[g].x * a.scale, xi = u[g].y * a.scale;
                        acc[rho][g] += sqrt(__builtin_fma(xr, xr, xi * xi));
                    }
                }
            }
        }
        // ---- the row is complete: first maximum (first index on ties), then the maximum over the allowed columns around it
        double best_v = -1.0;
        int best_k = 0x7fffffff;
        if (live2) {
#pragma unroll
            for (int rho = 0; rho < 2; ++rho)
#pragma unroll
                for (int g = 0; g < 10; ++g) {
                    const int k = kf0 + 5 * rho + 20 * N1 * (g / 2 + 5 * (g % 2));
                    const double v = 0.0 + acc[rho][g];                  // (0.0 + |.|: the map's own first addition)
                    const bool take = v > best_v || (v == best_v && k < best_k);
                    best_v = take ? v : best_v;
                    best_k = take ? k : best_k;
                }
        }
        double* const sh_v = reinterpret_cast<double*>(lds4);
        int* const sh_i = reinterpret_cast<int*>(sh_v + kWaves);
        __syncthreads();                                                 // (the buffers' last readers)
        block_best(best_v, best_k, sh_v, sh_i, tid);
        const double top_v = best_v;
        const int top_k = best_k;
        int a1 = 0, b0 = 0, b1 = 0;                                      // allowed columns [0, a1) U [b0, b1): SURVEY T7
        {
            const int e0 = top_k - a.spc, e1 = top_k + a.spc;
            if (e0 < 1) {
                b0 = e1;
                b1 = N - 1;
            } else if (e1 >= N) {
                a1 = e0;
            } else {
                a1 = e0;
                b0 = e1;
                b1 = N - 1;
            }
        }
        double sec_v = -1.0;
        int sec_k = 0x7fffffff;
        if (live2) {
#pragma unroll
            for (int rho = 0; rho < 2; ++rho)
#pragma unroll
                for (int g = 0; g < 10; ++g) {
                    const int k = kf0 + 5 * rho + 20 * N1 * (g / 2 + 5 * (g % 2));
                    const double v = 0.0 + acc[rho][g];
                    const bool allowed = k < a1 || (k >= b0 && k < b1);
                    const bool take = allowed && (v > sec_v || (v == sec_v && k < sec_k));
                    sec_v = take ? v : sec_v;
                    sec_k = take ? k : sec_k;
                }
        }
        block_best(sec_v, sec_k, sh_v, sh_i, tid);
        if (tid == 0) {
            UnitRecord rec;
            rec.top = top_v;
            rec.index = (long long)bin * N + top_k;
            rec.second = sec_v;
            rec.pad = 0.0;
            a.records[(size_t)prn * a.nbins + bin] = rec;
        }
    }
}

// Per PRN: the unit with the largest first maximum (smallest flat index on ties: np.argmax over the row-major map) is the
// winning row; its own second peak gives the ratio.  The call's last kernel: results may go to page-locked host memory.
// done (nullable): a page-locked word per PRN, raised to done_seq behind the PRN's results (what sdr_pcps waits for).
__global__ __launch_bounds__(64) void peaks_kernel(const UnitRecord* __restrict__ records, int nbins, long long* __restrict__ out_bin,
                                                   long long* __restrict__ out_code, double* __restrict__ out_ratio,
                                                   unsigned* __restrict__ done, unsigned done_seq) {
    const int prn = blockIdx.x;
    double v = -1.0, second = -1.0;
    long long i = 0x7fffffffffffffffLL;
    for (int b = threadIdx.x; b < nbins; b += 64) {
        const UnitRecord r = records[(size_t)prn * nbins + b];
        if (r.top > v || (r.top == v && r.index < i)) v = r.top, i = r.index, second = r.second;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off, 64), os = __shfl_down(second, off, 64);
        const long long oi = __shfl_down(i, off, 64);
        if (ov > v || (ov == v && oi < i)) v = ov, i = oi, second = os;
    }
    if (threadIdx.x == 0) {
        if (v < 0.0) i = 0;
        out_bin[prn] = i / N;
        out_code[prn] = i - (i / N) * N;
        out_ratio[prn] = second >= 0.0 ? v / second : nan("");
        if (done) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(&done[prn], done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace fused10k
