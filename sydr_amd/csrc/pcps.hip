// K2-K6: Parallel Code Phase Search on device (sydr/dsp/acquisition.py:9-115).
//
//   for every Doppler bin b:   F_b = FFT_N( x[n] * exp(-1j*(IF - bin_b)*((2n)*pi/fs)) )      (PRN independent,
//                                                                                             computed ONCE, not per PRN)
//   for every (PRN p, bin b):  map[p][b][:] += | IFFT_N( F_b * conj(FFT_N(upsampled code_p)) ) |
//   per PRN: first global argmax (row-major) + second peak in the winning row.
//
// N = samples per code = 4000 / 10000 / 25000 / 50000 are not powers of two, so
// the transform is a hand-written mixed-radix (2,3,4,5,8 + small primes)
// Stockham autosort FFT in fp64, batched over (PRN, bin); each pass streams the
// batch through HBM/L2 with 16-byte complex loads.  The Doppler mix and the
// int8->fp64 conversion are fused into the first forward pass, the code-spectrum
// multiply into the first inverse pass, and 1/N scale + magnitude + (non-)coherent
// accumulation into the last inverse pass.  fp64 throughout: the peak INDEX must
// match NumPy's bit for bit, and a 1e-7 relative error could reorder near-ties.
#include "engine_internal.h"
#include <chrono>
#include "sincos_reduced.h"
#include "pcps_codelets.h"

#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;

enum LoadMode { LOAD_PLAIN = 0, LOAD_IQ_MIX = 1, LOAD_MUL_CODE = 2, LOAD_CODE_REAL = 3, LOAD_MUL_CODE_SEL = 4 };
// STORE_MAG_MAX: nothing is stored -- the last stage keeps a running (maximum, first index) of |.|/N per wave and
// writes one 16-byte record per wave (the map-free search: the caller only wants indices and ratio).
enum StoreMode { STORE_PLAIN = 0, STORE_CONJ = 1, STORE_MAG_ACC = 2, STORE_CPLX_ACC = 3, STORE_MAG_MAX = 4 };


struct PassArgs {
    const double2* in;    // [batch][N]
    double2* out;         // [batch][N]
    const double2* tw;    // W[m] = exp(-2*pi*i*m/N)
    int N, R, Ns, nbf;    // nbf = N / R butterflies per transform
    int tw_stride;        // N / (Ns * R)
    // LOAD_IQ_MIX
    const void* ring;
    int64_t capacity;
    int64_t first_sample;  // ring index of sample 0 of this 1 ms block
    int64_t carrier_offset;  // idx_coh * N (index into phasePoints)
    double fs, if_hz, bin_start, bin_delta;
    // LOAD_MUL_CODE: in = F[nbins][N]; code spectra [n_prn][N]
    const double2* code_spec;
    int nbins, n_prn;
    // LOAD_CODE_REAL
    const int8_t* code_samples;  // [batch][N]
    // STORE_*_ACC
    double* map;        // [batch][N] magnitudes
    double2* csum;      // [batch][N]
    double scale;
    int first_block;    // 1: overwrite, 0: accumulate
    // LOAD_MUL_CODE_SEL: transform p takes Doppler row sel_bin[p] (the winning row of PRN p)
    const long long* sel_bin;
    // STORE_MAG_MAX: per-wave records, [batch][tiles][waves]
    Best* partials;
    // STORE_MAG_MAX behind LOAD_MUL_CODE_SEL (second peak of the winning row): the first peaks and samples per chip
    const Best* tops;
    int spc;
    // STORE_MAG_MAX of the register-resident kernels: `in` starts at Doppler bin bin0 of the search (a sweep over the last
    // bins only: the fused sweep took the others) -- added to the bin of the records' flat index
    int bin0;
    // LOAD_IQ_MIX over several blocks at once: bins per block (0: the batch is one block's bins)
    int iq_blocks;
    // STORE_PLAIN with a halo (shared spectra): row t of the output is halo + N elements long, element k goes to halo + k and
    // the row's last `halo` elements are written in front of it as well -- out[t][i] = X_t[(i - halo) mod N]
    int halo;
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ double2 mul_mi(double2 a) {
    return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

template <int R, bool INV>
struct Butterfly;

template <bool INV>
struct Butterfly<2, INV> {
    static __device__ __forceinline__ void run(double2* v) {
        double2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV>
struct Butterfly<3, INV> {
    static __device__ __forceinline__ void run(double2* v) {
        const double c = -0.5, s = 0.86602540378443864676;  // cos, sin of 2*pi/3
        double2 t1 = cadd(v[1], v[2]);
        double2 t2 = csub(v[1], v[2]);
        double2 m = make_double2(v[0].x + c * t1.x, v[0].y + c * t1.y);
        double2 r = mul_mi<INV>(make_double2(s * t2.x, s * t2.y));
        v[0] = cadd(v[0], t1);
        v[1] = cadd(m, r);
        v[2] = csub(m, r);
    }
};

template <bool INV>
struct Butterfly<4, INV> {
    static __device__ __forceinline__ void run(double2* v) {
        double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        double2 t2 = cadd(v[1], v[3]), t3 = mul_mi<INV>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[1] = cadd(t1, t3);
        v[2] = csub(t0, t2);
        v[3] = csub(t1, t3);
    }
};

template <bool INV>
struct Butterfly<5, INV> {
    static __device__ __forceinline__ void run(double2* v) {
        const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos(2pi/5), cos(4pi/5)
        const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin(2pi/5), sin(4pi/5)
        double2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
        double2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
        double2 m1 = make_double2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        double2 m2 = make_double2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        double2 r1 = mul_mi<INV>(make_double2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        double2 r2 = mul_mi<INV>(make_double2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = cadd(m1, r1);
        v[4] = csub(m1, r1);
        v[2] = cadd(m2, r2);
        v[3] = csub(m2, r2);
    }
};

template <bool INV>
struct Butterfly<8, INV> {
    static __device__ __forceinline__ void run(double2* v) {
        const double h = 0.70710678118654752440;
        // radix-2 split into evens / odds, two radix-4 butterflies, then combine.
        double2 e[4] = {v[0], v[2], v[4], v[6]};
        double2 o[4] = {v[1], v[3], v[5], v[7]};
        Butterfly<4, INV>::run(e);
        Butterfly<4, INV>::run(o);
        // twiddles W8^k, k = 1..3: (h, -+h), (0, -+1), (-h, -+h)
        double2 w1 = INV ? make_double2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
                         : make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
        double2 w2 = mul_mi<INV>(o[2]);
        double2 w3 = INV ? make_double2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
                         : make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
        v[0] = cadd(e[0], o[0]);
        v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], w1);
        v[5] = csub(e[1], w1);
        v[2] = cadd(e[2], w2);
        v[6] = csub(e[2], w2);
        v[3] = cadd(e[3], w3);
        v[7] = csub(e[3], w3);
    }
};

template <int FMT>
__device__ __forceinline__ double2 ring_sample(const void* ring, int64_t pos) {
    if (FMT == SDR_FMT_CI8) {
        const char2 v = static_cast<const char2*>(ring)[pos];      // (sign-flipped bytes: correlator.h kCi8Flip)
        return make_double2((double)(int8_t)(v.x ^ 0x80), (double)(int8_t)(v.y ^ 0x80));
    } else if (FMT == SDR_FMT_CI16) {
        const short2 v = static_cast<const short2*>(ring)[pos];
        return make_double2((double)v.x, (double)v.y);
    } else if (FMT == SDR_FMT_CF32) {
        const float2 v = static_cast<const float2*>(ring)[pos];
        return make_double2((double)v.x, (double)v.y);
    } else {
        return static_cast<const double2*>(ring)[pos];
    }
}

template <int LOAD, int FMT, bool INV>
__device__ __forceinline__ double2 load_elem(const PassArgs& a, int batch, int idx) {
    if (LOAD == LOAD_PLAIN) {
        return a.in[(size_t)batch * a.N + idx];
    } else if (LOAD == LOAD_IQ_MIX) {
        // acquisition.py:33,42-45,53: carrier = exp(-1j*freq*phasePoints), phasePoints[m] = ((m*2)*pi)/fs
        // (iq_blocks > 0: the batch holds that many Doppler bins of several consecutive blocks of N samples -- transform b is
        // bin b % iq_blocks of block b / iq_blocks: every non-coherent block of a search in one launch)
        const int blk = a.iq_blocks > 0 ? batch / a.iq_blocks : 0;
        batch -= blk * a.iq_blocks;
        int64_t pos = (a.first_sample + (int64_t)blk * a.N + idx) % a.capacity;
        double2 x = ring_sample<FMT>(a.ring, pos);
        double bin = a.bin_start + (double)batch * a.bin_delta;
        double freq = a.if_hz - bin;
        int64_t m = a.carrier_offset + idx;
        double pp = (double)(m * 2) * M_PI;
        pp = pp / a.fs;
        double arg = freq * pp;
        double s, c;
        // (|arg| stays below ~2*pi*f*T: a few hundred radians -- the exact two-constant reduction and the minimax kernels
        // of the correlators, < 1 ulp, instead of libm's general sincos with its large-argument path)
        sdr::sincos_reduced(arg, &s, &c);
        return cmul(make_double2(c, -s), x);
    } else if (LOAD == LOAD_MUL_CODE) {
        int prn = batch / a.nbins, bin = batch - prn * a.nbins;
        double2 f = a.in[(size_t)bin * a.N + idx];
        double2 c = a.code_spec[(size_t)prn * a.N + idx];
        return cmul(f, c);
    } else if (LOAD == LOAD_MUL_CODE_SEL) {
        const int bin = (int)a.sel_bin[batch];
        double2 f = a.in[(size_t)bin * a.N + idx];
        double2 c = a.code_spec[(size_t)batch * a.N + idx];
        return cmul(f, c);
    } else {
        return make_double2((double)a.code_samples[(size_t)batch * a.N + idx], 0.0);
    }
}

template <int STORE>
__device__ __forceinline__ void store_elem(const PassArgs& a, int batch, int idx, double2 v) {
    const size_t o = (size_t)batch * a.N + idx;
    if (STORE == STORE_PLAIN) {
        if (a.halo) {
            const size_t row = (size_t)batch * (size_t)(a.halo + a.N);
            a.out[row + a.halo + idx] = v;
            if (idx >= a.N - a.halo) a.out[row + (idx - (a.N - a.halo))] = v;
        } else {
            a.out[o] = v;
        }
    } else if (STORE == STORE_CONJ) {
        a.out[o] = make_double2(v.x, -v.y);
    } else if (STORE == STORE_MAG_ACC) {
        double mag = hypot(v.x * a.scale, v.y * a.scale);
        a.map[o] = a.first_block ? 0.0 + mag : a.map[o] + mag;
    } else if (STORE == STORE_MAG_MAX) {
        // (only the four-step row kernel implements the running maximum; the launcher never selects it elsewhere)
    } else {
        double2 s = make_double2(v.x * a.scale, v.y * a.scale);
        a.csum[o] = a.first_block ? s : cadd(a.csum[o], s);
    }
}

// One Stockham pass of radix R: thread j owns butterfly j of transform blockIdx.y.
template <int R, bool INV, int LOAD, int STORE, int FMT>
__global__ __launch_bounds__(kThreads) void fft_pass_kernel(const PassArgs a) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= a.nbf) return;
    const int batch = blockIdx.y;
    const int k = j % a.Ns;
    double2 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = load_elem<LOAD, FMT, INV>(a, batch, j + t * a.nbf);
    if (a.Ns > 1) {
#pragma unroll
        for (int t = 1; t < R; ++t) {
            double2 w = a.tw[(size_t)t * k * a.tw_stride];
            if (INV) w.y = -w.y;
            v[t] = cmul(v[t], w);
        }
    }
    Butterfly<R, INV>::run(v);
    const int o = (j - k) * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) store_elem<STORE>(a, batch, o + t * a.Ns, v[t]);
}

// Any other prime radix (7, 11, 13, ...): O(R^2) DFT with roots from the twiddle table.
constexpr int kMaxGenericRadix = 64;
template <bool INV, int LOAD, int STORE, int FMT>
__global__ __launch_bounds__(kThreads) void fft_pass_generic_kernel(const PassArgs a) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= a.nbf) return;
    const int batch = blockIdx.y;
    const int R = a.R;
    const int k = j % a.Ns;
    const int root_stride = a.N / R;
    const int o = (j - k) * R + k;
    for (int q = 0; q < R; ++q) {
        double2 acc = make_double2(0.0, 0.0);
        for (int t = 0; t < R; ++t) {
            double2 x = load_elem<LOAD, FMT, INV>(a, batch, j + t * a.nbf);
            if (a.Ns > 1 && t > 0) {
                double2 w = a.tw[(size_t)t * k * a.tw_stride];
                if (INV) w.y = -w.y;
                x = cmul(x, w);
            }
            double2 r = a.tw[(size_t)((t * q) % R) * root_stride];
            if (INV) r.y = -r.y;
            acc = cadd(acc, cmul(x, r));
        }
        store_elem<STORE>(a, batch, o + q * a.Ns, acc);
    }
}

__global__ __launch_bounds__(kThreads) void twiddle_kernel(double2* tw, int N) {
    int m = blockIdx.x * kThreads + threadIdx.x;
    if (m >= N) return;
    double s, c;
    sincospi(2.0 * (double)m / (double)N, &s, &c);
    tw[m] = make_double2(c, -s);
}

// The two operand images of a PRN's code spectrum for the fused search at N = 2 M = 50 000 (pcps_fused.h): parity 0 is the
// spectrum itself, parity 1 carries the twiddle of the odd half of the radix-2 decimation-in-frequency step,
//     C1[k] = C[k] w^-k,  C1[k + M] = -C[k + M] w^-k,  w = exp(-2 pi i / N)   (tw[k] = w^k: multiplied conjugated).
__global__ __launch_bounds__(kThreads) void code_parity_kernel(const double2* __restrict__ C, const double2* __restrict__ tw, int N,
                                                               double2* __restrict__ out) {
    const int k = blockIdx.x * kThreads + threadIdx.x, prn = blockIdx.y;
    if (k >= N) return;
    const int M = N / 2;
    const double2 c = C[(size_t)prn * N + k];
    const double2 w = tw[k < M ? k : k - M];
    double2 o = make_double2(__builtin_fma(c.y, w.y, c.x * w.x), __builtin_fma(-c.x, w.y, c.y * w.x));    // c * conj(w)
    if (k >= M) o = make_double2(-o.x, -o.y);
    out[((size_t)prn * 2) * N + k] = c;
    out[((size_t)prn * 2 + 1) * N + k] = o;
}

__global__ __launch_bounds__(kThreads) void upsample_batch_kernel(const int8_t* __restrict__ codes,
                                                                  const int32_t* __restrict__ code_len,
                                                                  int code_stride, const int32_t* __restrict__ slots,
                                                                  double ts, double tc, int N,
                                                                  int8_t* __restrict__ out) {
    int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= N) return;
    const int slot = slots[blockIdx.y];
    const int L = code_len[slot];
    double v = (ts * (double)k) / tc;  // gnsssignal.py:53
    int idx = (int)trunc(v);
    if (idx >= L) idx = L - 1;
    out[(size_t)blockIdx.y * N + k] = codes[(size_t)slot * code_stride + idx];
}

// map += |csum| (acquisition.py:68)
__global__ __launch_bounds__(kThreads) void mag_acc_kernel(const double2* __restrict__ csum, double* __restrict__ map,
                                                           size_t count, int first_block) {
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= count) return;
    double mag = hypot(csum[i].x, csum[i].y);
    map[i] = first_block ? 0.0 + mag : map[i] + mag;
}

/* ------------------------------------------------------- peak search (K6) */

__device__ __forceinline__ Best better(Best a, Best b) {
    // larger value wins; on ties the smaller flat index (first occurrence, np.argmax)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ Best block_best(Best mine, Best* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_down(mine.v, off, 64);
        o.i = __shfl_down(mine.i, off, 64);
        mine = better(mine, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = mine;
    __syncthreads();
    Best r = sh[0];
    for (int w = 1; w < kThreads / 64; ++w) r = better(r, sh[w]);
    __syncthreads();
    return r;
}

constexpr int kPeakParts = 64;

__global__ __launch_bounds__(kThreads) void argmax_part_kernel(const double* __restrict__ map, long long per_prn,
                                                               Best* __restrict__ parts) {
    __shared__ Best sh[kThreads / 64];
    const int prn = blockIdx.y, part = blockIdx.x;
    const double* m = map + (size_t)prn * per_prn;
    long long chunk = (per_prn + kPeakParts - 1) / kPeakParts;
    long long lo = (long long)part * chunk, hi = lo + chunk < per_prn ? lo + chunk : per_prn;
    Best mine = {-1.0, 0x7fffffffffffffffLL};
    for (long long i = lo + threadIdx.x; i < hi; i += kThreads) {
        Best c = {m[i], i};
        mine = better(mine, c);
    }
    Best r = block_best(mine, sh);
    if (threadIdx.x == 0) parts[(size_t)prn * kPeakParts + part] = r;
}

// TwoCorrelationPeakComparison (acquisition.py:78-115), including its quirks.
__global__ __launch_bounds__(kThreads) void peak_finish_kernel(const double* __restrict__ map, int nbins, int N,
                                                               int spc, const Best* __restrict__ parts,
                                                               long long* __restrict__ out_bin,
                                                               long long* __restrict__ out_code,
                                                               double* __restrict__ out_ratio) {
    __shared__ Best sh[kThreads / 64];
    const int prn = blockIdx.x;
    Best mine = {-1.0, 0x7fffffffffffffffLL};
    if (threadIdx.x < kPeakParts) mine = parts[(size_t)prn * kPeakParts + threadIdx.x];
    Best top = block_best(mine, sh);
    const long long bin = top.i / N;
    const int code = (int)(top.i - bin * N);
    const double* row = map + ((size_t)prn * nbins + bin) * N;
    const int e0 = code - spc, e1 = code + spc;
    // allowed = [a0,a1) U [b0,b1)
    int a0 = 0, a1 = 0, b0 = 0, b1 = 0;
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
    Best second = {-1.0, 0x7fffffffffffffffLL};
    for (int i = threadIdx.x; i < N; i += kThreads) {
        bool ok = (i >= a0 && i < a1) || (i >= b0 && i < b1);
        if (ok) {
            Best c = {row[i], i};
            second = better(second, c);
        }
    }
    Best p2 = block_best(second, sh);
    if (threadIdx.x == 0) {
        out_bin[prn] = bin;
        out_code[prn] = code;
        out_ratio[prn] = p2.v >= 0.0 ? top.v / p2.v : nan("");
    }
}

// Map-free search, step 2: the first maximum of PRN p's (never materialised) map from the per-wave records of the
// inverse row kernel.  Writes the winning (bin, code) and keeps the record for the ratio.
__global__ __launch_bounds__(kThreads) void argmax_records_kernel(const Best* __restrict__ recs, int per_prn, int N,
                                                                  Best* __restrict__ tops,
                                                                  long long* __restrict__ out_bin,
                                                                  long long* __restrict__ out_code) {
    __shared__ Best sh[kThreads / 64];
    const int prn = blockIdx.x;
    Best mine = {-1.0, 0x7fffffffffffffffLL};
    for (int i = threadIdx.x; i < per_prn; i += kThreads) mine = better(mine, recs[(size_t)prn * per_prn + i]);
    Best top = block_best(mine, sh);
    if (threadIdx.x == 0) {
        if (top.v < 0.0) top.i = 0;      // (no record at all: never an index the second sweep would read a row with)
        tops[prn] = top;
        out_bin[prn] = top.i / N;
        out_code[prn] = top.i - (top.i / N) * N;
    }
}

// Map-free search, step 4: the second peak of PRN p from the per-wave records of the pass over its winning row
// (which ran the maximum over the allowed columns only), and the ratio.
// (dev_bin / dev_code: the first peaks as argmax_records_kernel left them in device memory, where the second sweep
// read its rows from; out_* may be page-locked host memory -- this is the call's last kernel and hands everything over)
__global__ __launch_bounds__(64) void ratio_kernel(const Best* __restrict__ seconds, int per_prn,
                                                   const Best* __restrict__ tops, const long long* __restrict__ dev_bin,
                                                   const long long* __restrict__ dev_code, long long* __restrict__ out_bin,
                                                   long long* __restrict__ out_code, double* __restrict__ out_ratio) {
    const int prn = blockIdx.x;
    Best mine = {-1.0, 0x7fffffffffffffffLL};
    for (int i = threadIdx.x; i < per_prn; i += 64) mine = better(mine, seconds[(size_t)prn * per_prn + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_down(mine.v, off, 64);
        o.i = __shfl_down(mine.i, off, 64);
        mine = better(mine, o);
    }
    if (threadIdx.x == 0) {
        out_ratio[prn] = mine.v >= 0.0 ? tops[prn].v / mine.v : nan("");
        if (out_bin != dev_bin) {
            out_bin[prn] = dev_bin[prn];
            out_code[prn] = dev_code[prn];
        }
    }
}

/* ------------------------------------------------------------ host driver */

std::vector<int> factor_radices(int64_t N) {
    std::vector<int> r;
    int64_t n = N;
    while (n % 5 == 0) { r.push_back(5); n /= 5; }
    while (n % 3 == 0) { r.push_back(3); n /= 3; }
    while (n % 8 == 0) { r.push_back(8); n /= 8; }
    while (n % 4 == 0) { r.push_back(4); n /= 4; }
    while (n % 2 == 0) { r.push_back(2); n /= 2; }
    for (int p = 7; (int64_t)p <= n && p <= kMaxGenericRadix; p += 2)
        while (n % p == 0) { r.push_back(p); n /= p; }
    if (n != 1) r.clear();
    return r;
}

template <bool INV, int LOAD, int STORE, int FMT>
void launch_pass(sdr_engine* e, const PassArgs& a, int batch) {
    dim3 grid((a.nbf + kThreads - 1) / kThreads, batch);
    switch (a.R) {
        case 2: hipLaunchKernelGGL((fft_pass_kernel<2, INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<3, INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
        case 4: hipLaunchKernelGGL((fft_pass_kernel<4, INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
        case 5: hipLaunchKernelGGL((fft_pass_kernel<5, INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
        case 8: hipLaunchKernelGGL((fft_pass_kernel<8, INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
        default: hipLaunchKernelGGL((fft_pass_generic_kernel<INV, LOAD, STORE, FMT>), grid, dim3(kThreads), 0, e->stream, a); break;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Four-step transform: N = N1*N2, both sub-transforms done in LDS, two kernels and ONE round trip
 * through HBM per transform instead of one per radix pass (7 passes at N = 25000):
 *   A  columns:  Y[k1][n2] = sum_n1 x[N2*n1 + n2] W_N1^(n1 k1),  Z[k1][n2] = Y[k1][n2] * W_N^(n2 k1)
 *   B  rows:     X[k1 + N1*k2] = sum_n2 Z[k1][n2] W_N2^(n2 k2)
 * A workgroup owns a tile of T adjacent columns (A) or rows (B) so that every global access is a run
 * of T*16 contiguous bytes; the fused loads (Doppler mix / code spectrum product) sit in A and the
 * fused stores (conj / |.|/N accumulate / running maximum) in B.
 * ------------------------------------------------------------------------------------------------ */
// LDS row pitch (in complex values) of a tile of T columns.  No padding for the 4- and 8-column tiles in use: the
// butterflies walk the tile row by row (consecutive lanes = consecutive columns, then the next row), so an unpadded
// tile is read and written in contiguous 64- / 128-byte runs; a pad of one complex measured 7 % slower (47 % of the
// LDS cycles of the row kernel were bank conflicts), two 16 % slower.
constexpr int lds_pitch(int T) { return (T == 4 || T == 8) ? T : T + 1; }

struct Radices {
    int n;
    int r[10];
};

struct FourStep {
    bool ok = false;
    int N1 = 0, N2 = 0;
    Radices rad1{}, rad2{};
};

FourStep plan_four_step(int N) {
    FourStep f;
    int best = 0;
    for (int d = 2; (int64_t)d * d <= N; ++d)
        if (N % d == 0) best = d;  // largest divisor <= sqrt(N)
    if (best < 4 || N / best > 512) return f;
    std::vector<int> r1 = factor_radices(best), r2 = factor_radices(N / best);
    if (r1.empty() || r2.empty() || r1.size() > 10 || r2.size() > 10) return f;
    f.N1 = best;
    f.N2 = N / best;
    f.rad1.n = (int)r1.size();
    f.rad2.n = (int)r2.size();
    for (size_t i = 0; i < r1.size(); ++i) f.rad1.r[i] = r1[i];
    for (size_t i = 0; i < r2.size(); ++i) f.rad2.r[i] = r2[i];
    f.ok = true;
    return f;
}

// x / d for x*d < 2^32 with magic = 2^32/d + 1: three integer ops instead of a ~25-instruction division.
__device__ __forceinline__ int fast_div(int x, uint32_t magic) { return (int)__umulhi((uint32_t)x, magic); }
__host__ __device__ __forceinline__ uint32_t div_magic(int d) { return (uint32_t)(0x100000000ull / (uint32_t)d) + 1u; }

// One radix-R butterfly of a Stockham pass over data laid out [index][pitch] in LDS, column c.
template <int R, bool INV>
__device__ __forceinline__ void lds_butterfly(const double2* in, double2* out, int j, int c, int pitch, int nbf,
                                              int Ns, int tws, const double2* wsub, uint32_t ns_magic) {
    const int k = Ns == 1 ? 0 : j - fast_div(j, ns_magic) * Ns;
    double2 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = in[(j + t * nbf) * pitch + c];
    if (Ns > 1) {
#pragma unroll
        for (int t = 1; t < R; ++t) {
            double2 w = wsub[t * k * tws];
            if (INV) w.y = -w.y;
            v[t] = cmul(v[t], w);
        }
    }
    Butterfly<R, INV>::run(v);
    const int o = (j - k) * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) out[(o + t * Ns) * pitch + c] = v[t];
}

template <bool INV>
__device__ __forceinline__ void lds_butterfly_generic(const double2* in, double2* out, int R, int j, int c, int pitch,
                                                      int nbf, int Ns, int tws, int Nsub, const double2* wsub,
                                                      uint32_t ns_magic) {
    const int k = Ns == 1 ? 0 : j - fast_div(j, ns_magic) * Ns;
    const int o = (j - k) * R + k;
    const int root = Nsub / R;
    for (int q = 0; q < R; ++q) {
        double2 acc = make_double2(0.0, 0.0);
        for (int t = 0; t < R; ++t) {
            double2 x = in[(j + t * nbf) * pitch + c];
            if (Ns > 1 && t > 0) {
                double2 w = wsub[t * k * tws];
                if (INV) w.y = -w.y;
                x = cmul(x, w);
            }
            double2 r = wsub[((t * q) % R) * root];
            if (INV) r.y = -r.y;
            acc = cadd(acc, cmul(x, r));
        }
        out[(o + q * Ns) * pitch + c] = acc;
    }
}

// In-LDS Stockham transform of `T` columns of length Nsub; returns the buffer holding the result.
template <bool INV, int T>
__device__ __forceinline__ double2* lds_fft(double2* a, double2* b, int Nsub, const Radices& rad, const double2* wsub) {
    constexpr int pitch = lds_pitch(T);
    int Ns = 1;
    double2* in = a;
    double2* out = b;
    for (int p = 0; p < rad.n; ++p) {
        const int R = rad.r[p];
        const int nbf = Nsub / R;
        const int tws = Nsub / (Ns * R);
        const uint32_t ns_magic = div_magic(Ns);
        for (int bf = threadIdx.x; bf < nbf * T; bf += kThreads) {
            const int j = bf / T, c = bf - j * T;
            switch (R) {
                case 2: lds_butterfly<2, INV>(in, out, j, c, pitch, nbf, Ns, tws, wsub, ns_magic); break;
                case 3: lds_butterfly<3, INV>(in, out, j, c, pitch, nbf, Ns, tws, wsub, ns_magic); break;
                case 4: lds_butterfly<4, INV>(in, out, j, c, pitch, nbf, Ns, tws, wsub, ns_magic); break;
                case 5: lds_butterfly<5, INV>(in, out, j, c, pitch, nbf, Ns, tws, wsub, ns_magic); break;
                case 8: lds_butterfly<8, INV>(in, out, j, c, pitch, nbf, Ns, tws, wsub, ns_magic); break;
                default: lds_butterfly_generic<INV>(in, out, R, j, c, pitch, nbf, Ns, tws, Nsub, wsub, ns_magic); break;
            }
        }
        __syncthreads();
        double2* tmp = in;
        in = out;
        out = tmp;
        Ns *= R;
    }
    return in;
}

// The same transform with the radix sequence known at compile time (every index constant folds: no
// runtime division, no radix switch, twiddle strides are immediates).
template <bool INV, int T, int NSUB, int NS, int R0, int... Rs>
__device__ __forceinline__ double2* lds_fft_static(double2* in, double2* out, const double2* wsub) {
    constexpr int pitch = lds_pitch(T);
    constexpr int nbf = NSUB / R0;
    constexpr int tws = NSUB / (NS * R0);
    for (int bf = threadIdx.x; bf < nbf * T; bf += kThreads) {
        const int j = bf / T, c = bf - j * T;
        lds_butterfly<R0, INV>(in, out, j, c, pitch, nbf, NS, tws, wsub, div_magic(NS));
    }
    __syncthreads();
    if constexpr (sizeof...(Rs) == 0)
        return out;
    else
        return lds_fft_static<INV, T, NSUB, NS * R0, Rs...>(out, in, wsub);
}

// Sub-transform lengths of the usual sampling rates (N = 4000: 50 x 80, 10000: 100 x 100, 25000: 125 x 200,
// 50000: 200 x 250) run the compile-time version -- 13 % off a whole acquisition at N = 25000; the sequences are
// those factor_radices() produces, so both versions do the same arithmetic in the same order.
template <bool INV, int T>
__device__ __forceinline__ double2* lds_fft_auto(double2* a, double2* b, int Nsub, const Radices& rad, const double2* wsub) {
    switch (Nsub) {
        case 50: return lds_fft_static<INV, T, 50, 1, 5, 5, 2>(a, b, wsub);
        case 80: return lds_fft_static<INV, T, 80, 1, 5, 8, 2>(a, b, wsub);
        case 100: return lds_fft_static<INV, T, 100, 1, 5, 5, 4>(a, b, wsub);
        case 125: return lds_fft_static<INV, T, 125, 1, 5, 5, 5>(a, b, wsub);
        case 200: return lds_fft_static<INV, T, 200, 1, 5, 5, 8>(a, b, wsub);
        case 250: return lds_fft_static<INV, T, 250, 1, 5, 5, 5, 2>(a, b, wsub);
        default: return lds_fft<INV, T>(a, b, Nsub, rad, wsub);
    }
}

// The raw operands of one element of the fused first stage, fetched without being combined: the global loads of
// several elements are issued back to back and waited for once (a loop of load -> use pays one memory round trip per
// iteration, ~1 us each under load, which is what these kernels' time was made of).
struct Fetched {
    double2 p, q;
};
template <int LOAD, int FMT, bool INV>
__device__ __forceinline__ Fetched fetch_elem(const PassArgs& a, int batch, int bin, int prn, int idx) {
    Fetched f;
    if constexpr (LOAD == LOAD_MUL_CODE || LOAD == LOAD_MUL_CODE_SEL) {
        f.p = a.in[(size_t)bin * a.N + idx];
        f.q = a.code_spec[(size_t)prn * a.N + idx];
    } else if constexpr (LOAD == LOAD_PLAIN) {
        f.p = a.in[(size_t)batch * a.N + idx];
        f.q = make_double2(0.0, 0.0);
    } else {
        f.p = load_elem<LOAD, FMT, INV>(a, batch, idx);
        f.q = make_double2(0.0, 0.0);
    }
    return f;
}
template <int LOAD>
__device__ __forceinline__ double2 combine_elem(const Fetched& f) {
    if constexpr (LOAD == LOAD_MUL_CODE || LOAD == LOAD_MUL_CODE_SEL) return cmul(f.p, f.q);
    else return f.p;
}

constexpr int kBatchLoads = 4;  // elements per lane whose loads are in flight together

// Kernel A: T adjacent columns n2 of one transform.  Dynamic LDS: 2 * N1*(T+1) + N1 double2.
template <bool INV, int LOAD, int FMT, int T>
__global__ __launch_bounds__(kThreads) void fft4_cols_kernel(const PassArgs a, int N1, int N2, const Radices rad,
                                                             double2* __restrict__ Z) {
    extern __shared__ double2 lds4[];
    constexpr int pitch = lds_pitch(T);
    double2* bufA = lds4;
    double2* bufB = bufA + N1 * pitch;
    double2* wsub = bufB + N1 * pitch;
    int batch = blockIdx.y;
    int n2_0 = blockIdx.x * T;
    int prn = batch, bin = batch;
    if constexpr (LOAD == LOAD_MUL_CODE) {
        // XCD-aware mapping (1-D grid).  Every (PRN, bin) transform multiplies the bin's spectrum by the PRN's code
        // spectrum: 29 MB of unique operands at 32 PRN x 41 bins x 25000, read 1312 times over.  Workgroups are dealt
        // to the 8 XCDs round-robin by linear id, each XCD with its own 4 MB L2 -- in (tile, batch) order every XCD
        // ends up streaming all of it (measured 629 MB per call, 8 x the code spectra + every spectrum tile per PRN).
        // Here XCD x = id % 8 owns a contiguous eighth of the (column tile, bin) pairs -- 2 MB of spectrum tiles and
        // the code tiles of ~3 column tiles, both L2-resident -- and runs all PRNs of a pair back to back.
        const int tiles = (N2 + T - 1) / T;
        const int n_prn = a.n_prn;
        const int pairs = tiles * a.nbins;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int p_lo = (int)(((long long)pairs * xcd) >> 3), p_hi = (int)(((long long)pairs * (xcd + 1)) >> 3);
        const int pair = p_lo + slot / n_prn;
        if (pair >= p_hi) return;                       // (the grid is padded to the largest eighth)
        prn = slot - (slot / n_prn) * n_prn;
        const int tile = pair / a.nbins;
        bin = pair - tile * a.nbins;
        batch = prn * a.nbins + bin;
        n2_0 = tile * T;
    } else if constexpr (LOAD == LOAD_MUL_CODE_SEL) {
        bin = (int)a.sel_bin[batch];
    }
    const int total = N1 * T;
    const uint32_t t_magic = div_magic(T);
    // W_N1^m = W_N^(m*N2) for the sub-transform (read together with the first elements)
    for (int base = 0; base < total; base += kBatchLoads * kThreads) {
        Fetched f[kBatchLoads];
        int slot[kBatchLoads];
        double2 wv = make_double2(0.0, 0.0);
        // (each chunk of kBatchLoads*256 elements also brings 256 entries of the sub-transform's twiddle table:
        // T >= kBatchLoads chunks' worth cover all N1 of them)
        static_assert(T >= kBatchLoads, "the twiddle table rides along with the element chunks");
        const int m = base / kBatchLoads + threadIdx.x;
        const bool wm = m < N1;
        if (wm) wv = a.tw[(size_t)m * N2];
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u) {
            const int e = base + u * kThreads + threadIdx.x;
            const int n1 = fast_div(e, t_magic), c = e - n1 * T;
            const int n2 = n2_0 + c;
            const bool ok = e < total && n2 < N2;
            slot[u] = e < total ? n1 * pitch + c : -1;
            f[u].p = f[u].q = make_double2(0.0, 0.0);
            if (ok) f[u] = fetch_elem<LOAD, FMT, INV>(a, batch, bin, prn, N2 * n1 + n2);
        }
        if (wm) wsub[m] = wv;
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u)
            if (slot[u] >= 0) bufA[slot[u]] = combine_elem<LOAD>(f[u]);
    }
    __syncthreads();
    double2* res = lds_fft_auto<INV, T>(bufA, bufB, N1, rad, wsub);
    // Twiddle W_N^(n2*k1) = W_N^(n2_0*k1) * W_N^(c*k1): the first factor is one scattered table read per ROW
    // (kept in the now free wsub), the second comes from the first T*N1 entries of the table -- cache resident --
    // instead of one scattered read per element.
    for (int m = threadIdx.x; m < N1; m += kThreads) wsub[m] = a.tw[(size_t)n2_0 * m];  // n2_0*m < N
    __syncthreads();
    double2* __restrict__ zt = Z + (size_t)batch * N1 * N2 + n2_0;   // (k1, c) of the tile sits at zt[k1*N2 + c]: 32-bit offsets
    for (int base = 0; base < total; base += kBatchLoads * kThreads) {
        double2 w2[kBatchLoads];
        int k1s[kBatchLoads], cs[kBatchLoads];
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u) {
            const int e = base + u * kThreads + threadIdx.x;
            const int k1 = fast_div(e, t_magic), c = e - k1 * T;
            const bool ok = e < total && n2_0 + c < N2;
            k1s[u] = ok ? k1 : -1;
            cs[u] = c;
            w2[u] = make_double2(1.0, 0.0);
            if (ok) w2[u] = a.tw[c * k1];
        }
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u) {
            if (k1s[u] >= 0) {
                double2 w = cmul(wsub[k1s[u]], w2[u]);
                if (INV) w.y = -w.y;
                zt[k1s[u] * N2 + cs[u]] = cmul(res[k1s[u] * pitch + cs[u]], w);
            }
        }
    }
}

// Kernel B: T adjacent rows k1 of one transform.  Dynamic LDS: 2 * N2*(T+1) + N2 double2.
template <bool INV, int STORE, int T, bool LOADSEL = false>
__global__ __launch_bounds__(kThreads) void fft4_rows_kernel(const PassArgs a, int N1, int N2, const Radices rad,
                                                             const double2* __restrict__ Z) {
    extern __shared__ double2 lds4[];
    constexpr int pitch = lds_pitch(T);
    double2* bufA = lds4;
    double2* bufB = bufA + N2 * pitch;
    double2* wsub = bufB + N2 * pitch;
    const int batch = blockIdx.y;
    const int k1_0 = blockIdx.x * T;
    const uint32_t n2_magic = div_magic(N2);
    const int total = N2 * T;
    // the T rows of a tile are ONE contiguous run of Z (row k1 of transform `batch` starts at (batch*N1 + k1)*N2):
    // element e of the tile sits at tile + e, no per-element 64-bit index arithmetic
    const double2* __restrict__ tile = Z + ((size_t)batch * N1 + k1_0) * N2;
    const int exist = (N1 - k1_0 < T ? N1 - k1_0 : T) * N2;    // elements of the tile that belong to the transform
    for (int base = 0; base < total; base += kBatchLoads * kThreads) {
        double2 z[kBatchLoads];
        int slot[kBatchLoads];
        double2 wv = make_double2(0.0, 0.0);
        static_assert(T >= kBatchLoads, "the twiddle table rides along with the element chunks");
        const int m = base / kBatchLoads + threadIdx.x;
        const bool wm = m < N2;
        if (wm) wv = a.tw[(size_t)m * N1];     // W_N2^m = W_N^(m*N1)
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u) {
            const int e = base + u * kThreads + threadIdx.x;
            const int c = fast_div(e, n2_magic), n2 = e - c * N2;  // n2 fastest: each row of Z is read as one contiguous run
            slot[u] = e < total ? n2 * pitch + c : -1;
            z[u] = make_double2(0.0, 0.0);
            if (e < exist) z[u] = tile[e];
        }
        if (wm) wsub[m] = wv;
#pragma unroll
        for (int u = 0; u < kBatchLoads; ++u)
            if (slot[u] >= 0) bufA[slot[u]] = z[u];
    }
    __syncthreads();
    const double2* res = lds_fft_auto<INV, T>(bufA, bufB, N2, rad, wsub);
    if constexpr (STORE == STORE_MAG_MAX) {
        // map[p][b][k] = |v|/N is never written: each lane keeps the largest value it produced (first index on
        // ties, as np.argmax over the row-major map), the wave reduces with DPP moves and lane 63 writes 16 bytes.
        // Inside a lane the candidates are ordered by the squared magnitude (3 instructions instead of a ~40
        // instruction hypot per point); hypot -- the reference's np.abs -- is monotone in it up to its rounding, so
        // only candidates within 2^-48 (relative) of the lane's best are compared through hypot itself, and the
        // lane's winner gets its exact hypot once, before lanes are compared.  With SECOND (the pass over the
        // winning row alone) the same maximum runs over the columns TwoCorrelationPeakComparison allows.
        const int prn = LOADSEL ? batch : batch / a.nbins;
        const int bin = LOADSEL ? 0 : batch - prn * a.nbins;
        int a1 = 0, b0 = 0, b1 = 0;                    // allowed columns [0,a1) U [b0,b1) (acquisition.py:98-111)
        if constexpr (LOADSEL) {
            const long long ti = a.tops[batch].i;
            const int code = (int)(ti - (ti / a.N) * a.N);
            const int e0 = code - a.spc, e1 = code + a.spc;
            if (e0 < 1) {
                b0 = e1;
                b1 = a.N - 1;
            } else if (e1 >= a.N) {
                a1 = e0;
            } else {
                a1 = e0;
                b0 = e1;
                b1 = a.N - 1;
            }
        }
        // A lane walks its elements e = k2*T + c in increasing order, and the map index bin*N + (k1_0 + c) + N1*k2 grows
        // with e: the first occurrence of a lane's maximum is simply the earliest e, so only the winner's index is ever
        // formed.  The coarse ordering uses the UNSCALED squared magnitude (the 1/N of the inverse transform is a
        // positive constant: it moves the ordering by rounding only, far inside the 2^-48 band that goes through the
        // exact comparison of the scaled hypot, the reference's np.abs).
        double best_sq = -1.0, best_x = 0.0, best_y = 0.0;
        int best_e = -1;
        for (int e = threadIdx.x; e < N2 * T; e += kThreads) {
            const int k2 = e / T, c = e - k2 * T;
            bool live = k1_0 + c < N1;
            if (LOADSEL) {
                const int col = k1_0 + c + N1 * k2;
                live = live && (col < a1 || (col >= b0 && col < b1));
            }
            if (live) {
                const double2 v = res[k2 * pitch + c];
                const double sq = v.x * v.x + v.y * v.y;
                bool take = sq > best_sq;
                if (__builtin_expect(fabs(sq - best_sq) <= best_sq * 0x1p-48, 0)) {
                    const double m_new = hypot(v.x * a.scale, v.y * a.scale), m_old = hypot(best_x * a.scale, best_y * a.scale);
                    take = m_new > m_old;      // (equal: the earlier element, i.e. the smaller index, stays)
                }
                best_sq = take ? sq : best_sq;
                best_x = take ? v.x : best_x;
                best_y = take ? v.y : best_y;
                best_e = take ? e : best_e;
            }
        }
        int best_i = 0x7fffffff;
        double best_v = -1.0;
        if (best_e >= 0) {
            const int k2 = best_e / T, c = best_e - k2 * T;
            best_i = bin * a.N + k1_0 + c + N1 * k2;
            best_v = 0.0 + hypot(best_x * a.scale, best_y * a.scale);   // (0.0 + |.|: the map's own rounding)
        }
        wave_best(best_v, best_i);
        if ((threadIdx.x & 63) == 63) {
            Best r = {best_v, (long long)best_i};
            a.partials[((size_t)batch * gridDim.x + blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6)] = r;
        }
        return;
    }
    for (int e = threadIdx.x; e < N2 * T; e += kThreads) {
        const int k2 = e / T, c = e - k2 * T;
        const int k1 = k1_0 + c;
        if (k1 < N1) store_elem<STORE>(a, batch, k1 + N1 * k2, res[k2 * pitch + c]);
    }
}

#ifndef SDR_PCPS_OVERLAP_MB
#define SDR_PCPS_OVERLAP_MB 100ll
#endif
#include "pcps_fast.h"   // register-resident kernels for N = 125 x 200 (needs Butterfly, PassArgs, Best, wave_best)
#include "pcps_fastn.h"  // ... and for N = 20 / 50 / 250 x 200

template <bool INV, int LOAD0, int STORE_LAST, int FMT, int TA, int TB>
void run_four_step_t(sdr_engine* e, const FourStep& f, PassArgs a, int batch, double2* Z, double2* final_out) {
    a.out = final_out;
    const size_t shA = (size_t)(2 * f.N1 * lds_pitch(TA) + f.N1) * sizeof(double2);
    const size_t shB = (size_t)(2 * f.N2 * lds_pitch(TB) + f.N2) * sizeof(double2);
    // more than 64 KiB of dynamic LDS has to be requested explicitly (160 KiB per CU on MI355X)
    (void)hipFuncSetAttribute((const void*)fft4_cols_kernel<INV, LOAD0, FMT, TA>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shA);
    constexpr bool SEL = LOAD0 == LOAD_MUL_CODE_SEL;
    (void)hipFuncSetAttribute((const void*)fft4_rows_kernel<INV, STORE_LAST, TB, SEL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shB);
    dim3 gridA((f.N2 + TA - 1) / TA, batch);
    if constexpr (LOAD0 == LOAD_MUL_CODE) {             // 1-D, XCD-aware: 8 x the largest eighth of the (tile, bin) pairs x PRNs
        const int pairs = (int)gridA.x * a.nbins;
        a.n_prn = batch / a.nbins;
        gridA = dim3(8u * (unsigned)((pairs + 7) / 8) * (unsigned)a.n_prn, 1);
    }
    hipLaunchKernelGGL((fft4_cols_kernel<INV, LOAD0, FMT, TA>), gridA, dim3(kThreads), shA, e->stream, a, f.N1, f.N2, f.rad1, Z);
    hipLaunchKernelGGL((fft4_rows_kernel<INV, STORE_LAST, TB, SEL>), dim3((f.N1 + TB - 1) / TB, batch), dim3(kThreads), shB,
                       e->stream, a, f.N1, f.N2, f.rad2, Z);
}

// Tile widths: 8 columns for kernel A (128-byte runs), 4 rows for kernel B -- its LDS footprint is the larger
// one (N2 >= N1) and halving it (5 instead of 2 workgroups per CU at N = 25000) measured 9 % faster than 8.
#ifndef SDR_PCPS_ROW_TILE
#define SDR_PCPS_ROW_TILE 4
#endif
#ifndef SDR_PCPS_COL_TILE
#define SDR_PCPS_COL_TILE 8
#endif
constexpr int kRowTile = SDR_PCPS_ROW_TILE;
inline bool fast25k_applies(const sdr_engine* e, const FourStep& f) {
    return f.ok && f.N1 == fast25k::N1 && f.N2 == fast25k::N2 && !e->pcps_no_fast;
}
// the register-resident kernels of either header serve this length (they split it as N1 x 200 themselves)
inline bool fast_applies(const sdr_engine* e, const FourStep& f) {
    return fast25k_applies(e, f) || (f.ok && fastn::handles(f.N1 * f.N2) && !e->pcps_no_fast);
}
template <int STORE = STORE_MAG_MAX>
inline void fast_run(sdr_engine* e, const FourStep& f, PassArgs a, int batch, double2* Z, hipStream_t stream) {
    if (fast25k_applies(e, f)) {
        if constexpr (STORE == STORE_MAG_MAX) {
            fast25k::run(e, a, batch, Z, stream);
        } else {
            fast25k::run_cols(a, batch, Z, stream);
            fastn::run_rows<fast25k::N1, STORE>(a, batch, Z, stream);
        }
    } else {
        fastn::run<STORE>(f.N1 * f.N2, a, batch, Z, stream);
    }
}
template <bool INV, int LOAD0, int STORE_LAST, int FMT>
void run_four_step(sdr_engine* e, const FourStep& f, PassArgs a, int batch, double2* Z, double2* final_out) {
    if constexpr (INV && LOAD0 == LOAD_MUL_CODE &&
                  (STORE_LAST == STORE_MAG_MAX || STORE_LAST == STORE_MAG_ACC || STORE_LAST == STORE_CPLX_ACC)) {
        if (fast_applies(e, f)) {         // the inverse transforms of a search at 4 / 10 / 25 / 50 MHz
            fast_run<STORE_LAST>(e, f, a, batch, Z, e->stream);
            return;
        }
    }
    if constexpr (!INV && LOAD0 == LOAD_IQ_MIX && STORE_LAST == STORE_PLAIN) {
        // a handful of forward transforms (shared spectra: four for a 250 Hz grid) are a chain of latencies, not work: narrower
        // column tiles put twice as many workgroups on its first half
        if (batch <= 8) {           // (measured: 0.2179 -> 0.2159 ms per 32-PRN call at 25 MHz, 0.4863 -> 0.4833 at 50 MHz)
            run_four_step_t<INV, LOAD0, STORE_LAST, FMT, 4, 4>(e, f, a, batch, Z, final_out);
            return;
        }
    }
    run_four_step_t<INV, LOAD0, STORE_LAST, FMT, SDR_PCPS_COL_TILE, kRowTile>(e, f, a, batch, Z, final_out);
}
// per-wave records one map-free inverse sweep leaves per transform
inline int records_per_transform(const FourStep& f) { return ((f.N1 + kRowTile - 1) / kRowTile) * (kThreads / 64); }
// ... and of the main sweep, which may run the register-resident kernels
// operand terms per point of the one-workgroup-per-transform sweep: 1 at N = 25 000, 2 at N = 50 000 (a radix-2 step in
// front of two 25 000-point transforms, pcps_fused.h), 0 = the sweep does not serve this length
inline int fused_terms(const sdr_engine* e, const FourStep& f) {
    if (!e->pcps_fused || !f.ok || e->pcps_no_fast) return 0;
    if (f.N1 == fast25k::N1 && f.N2 == fast25k::N2) return 1;
    return f.N1 * f.N2 == 2 * fast25k::N ? 2 : 0;
}
// The fused sweep takes a map-free search whole when it fills at least one round of its 256 persistent workgroups (a
// smaller search is quicker through the two-kernel path: ~8 us + 0.19 us per transform against ~35 us per round).
inline bool fused_takes(const sdr_engine* e, const FourStep& f, int n_prn, int nbins) {
    return fused_terms(e, f) * n_prn * nbins >= 256;
}
inline int records_main_sweep(const sdr_engine* e, const FourStep& f) {
    if (fast25k_applies(e, f)) return fast25k::kRecordsPerTransform;
    return fast_applies(e, f) ? fastn::records(f.N1 * f.N2) : records_per_transform(f);
}

/* ------------------------------------------------------------------------------------------------
 * Chirp-z (Bluestein) transform for a length N the planner cannot factor (a prime factor above 64):
 * what NumPy's pocketfft does for such sizes.  With c[m] = exp(+i*pi*m^2/N) (m^2 taken mod 2N in
 * integers, so the phase is exact to an ulp):
 *   forward  X[k] = conj(c[k]) * sum_n (x[n] conj(c[n])) * c[k-n]
 *   inverse  Y[n] =      c[n]  * sum_k (X[k]      c[k] ) * conj(c[n-k])          (unnormalised)
 * The sums are circular convolutions of length M >= 2N-1 (M = 2^a 3^b 5^c), done with the ordinary
 * transforms above.  The fused loads / stores of the PCPS stages sit in the pre / post kernels.
 * ------------------------------------------------------------------------------------------------ */
struct BluPlan {
    int N = 0, M = 0;
    const double2* chirp = nullptr;   // [N]
    const double2* spec_fwd = nullptr;  // FFT_M of the forward kernel, [M]
    const double2* spec_inv = nullptr;  // FFT_M of the inverse kernel, [M]
    const double2* twM = nullptr;     // twiddles of the length-M transform
    double2 *x = nullptr, *a = nullptr, *b = nullptr;  // [max batch][M] work buffers
    std::vector<int> radM;
};

__global__ __launch_bounds__(kThreads) void blu_chirp_kernel(double2* chirp, double2* kern_fwd, double2* kern_inv, int N, int M) {
    const int m = blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    double2 c = make_double2(0.0, 0.0);
    const int d = m < N ? m : (M - m < N ? M - m : -1);  // kernel index: b[m] = c[|m|] for |m| < N (wrapped), else 0
    if (d >= 0) {
        const long long r = ((long long)d * d) % (2LL * N);
        double sn, cs;
        sincospi((double)r / (double)N, &sn, &cs);
        c = make_double2(cs, sn);
    }
    if (m < N) chirp[m] = c;
    kern_fwd[m] = c;
    kern_inv[m] = make_double2(c.x, -c.y);
}

template <int LOAD, int FMT, bool INV>
__global__ __launch_bounds__(kThreads) void blu_pre_kernel(const PassArgs a, const double2* __restrict__ chirp, int M,
                                                          double2* __restrict__ x) {
    const int m = blockIdx.x * kThreads + threadIdx.x;
    const int batch = blockIdx.y;
    if (m >= M) return;
    double2 v = make_double2(0.0, 0.0);
    if (m < a.N) {
        double2 c = chirp[m];
        if (!INV) c.y = -c.y;
        v = cmul(load_elem<LOAD, FMT, INV>(a, batch, m), c);
    }
    x[(size_t)batch * M + m] = v;
}

__global__ __launch_bounds__(kThreads) void blu_mul_kernel(double2* __restrict__ x, const double2* __restrict__ spec, int M) {
    const int m = blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    const size_t o = (size_t)blockIdx.y * M + m;
    x[o] = cmul(x[o], spec[m]);
}

template <int STORE, bool INV>
__global__ __launch_bounds__(kThreads) void blu_post_kernel(const PassArgs a, const double2* __restrict__ chirp, int M,
                                                           const double2* __restrict__ x) {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    const int batch = blockIdx.y;
    if (k >= a.N) return;
    double2 c = chirp[k];
    if (!INV) c.y = -c.y;
    const double inv_m = 1.0 / (double)M;
    double2 v = cmul(x[(size_t)batch * M + k], c);
    v.x *= inv_m;
    v.y *= inv_m;
    store_elem<STORE>(a, batch, k, v);
}

// Runs all passes of one batched transform.  `first` carries the fused load of
// pass 0, `last_store` the fused store of the final pass.  Ping-pongs bufA/bufB.
template <bool INV, int LOAD0, int STORE_LAST, int FMT>
void run_fft(sdr_engine* e, const std::vector<int>& radices, PassArgs a, int batch, double2* bufA, double2* bufB,
             double2* final_out, const char* prof_name, const BluPlan* blu = nullptr);

template <bool INV, int LOAD0, int STORE_LAST, int FMT>
void run_bluestein(sdr_engine* e, const BluPlan& blu, PassArgs a, int batch, double2* final_out) {
    a.out = final_out;
    const int M = blu.M;
    const dim3 gm((M + kThreads - 1) / kThreads, batch), gn((a.N + kThreads - 1) / kThreads, batch);
    hipLaunchKernelGGL((blu_pre_kernel<LOAD0, FMT, INV>), gm, dim3(kThreads), 0, e->stream, a, blu.chirp, M, blu.x);
    PassArgs p = {};
    p.N = M;
    p.tw = blu.twM;
    p.in = blu.x;
    run_fft<false, LOAD_PLAIN, STORE_PLAIN, FMT>(e, blu.radM, p, batch, blu.a, blu.b, blu.x, "pcps_bluestein_fft");
    hipLaunchKernelGGL(blu_mul_kernel, gm, dim3(kThreads), 0, e->stream, blu.x, INV ? blu.spec_inv : blu.spec_fwd, M);
    run_fft<true, LOAD_PLAIN, STORE_PLAIN, FMT>(e, blu.radM, p, batch, blu.a, blu.b, blu.x, "pcps_bluestein_fft");
    hipLaunchKernelGGL((blu_post_kernel<STORE_LAST, INV>), gn, dim3(kThreads), 0, e->stream, a, blu.chirp, M, blu.x);
}

template <bool INV, int LOAD0, int STORE_LAST, int FMT>
void run_fft(sdr_engine* e, const std::vector<int>& radices, PassArgs a, int batch, double2* bufA, double2* bufB,
             double2* final_out, const char* prof_name, const BluPlan* blu) {
    ProfScope ps(e, prof_name);
    if (blu) {
        run_bluestein<INV, LOAD0, STORE_LAST, FMT>(e, *blu, a, batch, final_out);
        return;
    }
    const FourStep four = plan_four_step(a.N);
    if (four.ok && !e->pcps_force_passes) {
        run_four_step<INV, LOAD0, STORE_LAST, FMT>(e, four, a, batch, bufA, final_out);
        return;
    }
    const int np = (int)radices.size();
    int Ns = 1;
    const double2* src = a.in;
    for (int p = 0; p < np; ++p) {
        a.R = radices[p];
        a.Ns = Ns;
        a.nbf = a.N / a.R;
        a.tw_stride = a.N / (Ns * a.R);
        const bool first = p == 0, last = p == np - 1;
        a.in = first ? a.in : src;
        double2* dst = last ? final_out : ((p & 1) ? bufB : bufA);
        a.out = dst;
        if (first && last) launch_pass<INV, LOAD0, STORE_LAST, FMT>(e, a, batch);
        else if (first) launch_pass<INV, LOAD0, STORE_PLAIN, FMT>(e, a, batch);
        else if (last) launch_pass<INV, LOAD_PLAIN, STORE_LAST, FMT>(e, a, batch);
        else launch_pass<INV, LOAD_PLAIN, STORE_PLAIN, FMT>(e, a, batch);
        src = dst;
        Ns *= a.R;
    }
}

// ---- shared spectra (round 5).  x mixed with exp(-2 pi i (f - q / T) t) is x mixed with f, times exp(+2 pi i q n / N): its
// spectrum is the other's, circularly shifted by q elements.  Bins P apart whose distance P * step is a whole number q of
// transform bins (1 / T = fs / N: 250 Hz steps over 1 ms -> P = 4, q = 1; 300 Hz -> P = 10, q = 3) therefore share ONE
// forward transform: a search transforms its first P bins and reads bin c + P j as the spectrum of bin c shifted by j q
// (acquisition.py:42-59 transforms every bin; the values agree to rounding, ~1e-15 relative, like any two orders of the same
// additions).  4 forward transforms instead of 41 for the headline grid, and the spectra side of the inverse sweep's operands
// is 4 arrays.  The class spectra are written with the row's last H elements in front of it (PassArgs::halo), so that a
// shift is a contiguous read H - j q elements into the row.
struct SharedSpectra {
    int P = 0, q = 0, H = 0;            // P == 0: every bin has its own transform
    const long long* off = nullptr;     // device: element offset of every bin's spectrum inside ITS BLOCK of P rows of H + N
};
inline int plan_shared_spectra(sdr_engine* e, int nbins, int N, double fs, double bin_delta, SharedSpectra* out) {
    *out = SharedSpectra{};
    if (e->pcps_no_shared_spectra || !(bin_delta > 0.0)) return SDR_OK;
    for (int P = 1; P <= 64 && 2 * P <= nbins; ++P) {
        const double v = (double)P * bin_delta * (double)N / fs, r = std::nearbyint(v);
        if (r >= 1.0 && std::fabs(v - r) <= 1e-9 * r && r * (double)((nbins - 1) / P) <= 4096.0) {
            out->P = P, out->q = (int)r;
            break;
        }
    }
    if (!out->P) return SDR_OK;
    out->H = out->q * ((nbins - 1) / out->P);
    std::vector<int64_t> key = {(int64_t)N, (int64_t)nbins, (int64_t)out->P, (int64_t)out->q};
    if (key != e->pcps_spec_off_key) {
        std::vector<long long> off((size_t)nbins);
        for (int b = 0; b < nbins; ++b)
            off[(size_t)b] = (long long)(b % out->P) * (out->H + N) + out->H - (long long)(b / out->P) * out->q;
        e->pcps_spec_off_key.clear();
        if (int rc = sdr_devbuf_reserve(e, &e->pcps_spec_off, off.size() * sizeof(long long))) return rc;
        SDR_HIP(hipMemcpyAsync(e->pcps_spec_off.ptr, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice, e->stream));
        SDR_HIP(hipStreamSynchronize(e->stream));      // (pageable source: complete before `off` goes)
        e->pcps_spec_off_key = key;
    }
    out->off = (const long long*)e->pcps_spec_off.ptr;
    return SDR_OK;
}

template <int FMT>
int pcps_run(sdr_engine* e, const int32_t* d_slots, const int32_t* h_slots, int n_prn, int64_t start, double fs, double if_hz,
             double bin_start, double bin_delta, int nbins, int N, int spc, int coh, int noncoh,
             const std::vector<int>& radices, int prn_chunk, bool have_spectra, const BluPlan* blu, bool map_free, bool fused10k) {
    double2* F = (double2*)e->pcps_fwd.ptr;
    double2* A = (double2*)e->pcps_a.ptr;
    double2* B = (double2*)e->pcps_b.ptr;
    double2* C = (double2*)e->pcps_code.ptr;
    double* map = (double*)e->pcps_map.ptr;
    double2* csum = (double2*)e->pcps_csum.ptr;
    const double2* tw = (const double2*)e->pcps_tw.ptr;

    // K7/a3: upsample every code and take conj(fft(code)) (channel_l1ca_kaplan.py:184-185), unless the caller
    // handed in its own codeFFT arrays (function-level drop-in of PCPS()) -- or the spectra of exactly these
    // staged codes at this rate are still in the buffer from the previous search (the reference recomputes them
    // on every acquisition, kaplan:184-185; they only change when a slot is re-staged).
    std::vector<int64_t> key;
    if (!have_spectra) {
        int64_t fs_bits;
        memcpy(&fs_bits, &fs, sizeof(fs_bits));
        key = {(int64_t)N, fs_bits, e->code_generation};
        for (int i = 0; i < n_prn; ++i) {
            key.push_back(h_slots[i]);
            key.push_back(e->code_stamp[h_slots[i]]);
        }
    }
    const bool spectra_cached = !have_spectra && key == e->pcps_spec_key && !e->pcps_no_spec_cache;
    if (have_spectra) e->pcps_spec_key.clear();
    if (!spectra_cached) e->pcps_code2_ok = false;
    if (!have_spectra && !spectra_cached) {
        e->pcps_spec_key.clear();   // (valid again only once the new spectra are queued without error, below)
        int8_t* up = (int8_t*)B;  // scratch: n_prn*N bytes fits easily in a work buffer
        // (the slot numbers reach the device only now: a search whose spectra are still there needs no copy command in
        // front of its first kernel -- 2.7 us of copy and two boundaries on the stream)
        SDR_HIP(hipMemcpyAsync(const_cast<int32_t*>(d_slots), e->pcps_slots_pinned, (size_t)n_prn * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        {
            ProfScope ps(e, "pcps_upsample");
            hipLaunchKernelGGL(upsample_batch_kernel, dim3((N + kThreads - 1) / kThreads, n_prn), dim3(kThreads), 0,
                               e->stream, e->codes, e->code_len, e->code_stride, d_slots, 1.0 / fs, 1.0 / 1.023e6, N, up);
        }
        PassArgs a = {};
        a.tw = tw;
        a.N = N;
        a.code_samples = up;
        // ping-pong inside A (two halves are not needed: code batch is small) -> use A and F as scratch
        run_fft<false, LOAD_CODE_REAL, STORE_CONJ, FMT>(e, radices, a, n_prn, A, F, C, "pcps_code_fft", blu);
        SDR_HIP(hipGetLastError());
        e->pcps_spec_key = key;
    }

    if (fused10k) {
        // N = 10 000, indices and ratio only, one coherent millisecond per block: the forward transforms of every block
        // first, then ONE launch that keeps each (PRN, bin)'s transform in LDS and its non-coherent sum in registers and
        // finds both peaks (pcps_fused10k.h) -- no map, no intermediate, no second sweep
        SharedSpectra sh10;
        {
            PassArgs f = {};
            f.tw = tw;
            f.N = N;
            f.ring = e->iq;
            f.capacity = e->iq_capacity;
            f.first_sample = start;
            f.carrier_offset = 0;            // (one coherent millisecond per block: the carrier restarts with every block)
            f.fs = fs;
            f.if_hz = if_hz;
            f.bin_start = bin_start;
            f.bin_delta = bin_delta;
            // (shared spectra: the shipped 300 Hz grid has ten classes -- 10 transforms per block instead of 34)
            if (int rcs = plan_shared_spectra(e, nbins, N, fs, bin_delta, &sh10)) return rcs;
            const int per_block = sh10.P ? sh10.P : nbins;
            f.iq_blocks = per_block;         // all blocks in one batch: [block][bin]
            f.halo = sh10.H;
            run_fft<false, LOAD_IQ_MIX, STORE_PLAIN, FMT>(e, radices, f, per_block * noncoh, A, B, F, "pcps_fwd_fft", blu);
        }
        long long* out_bin = e->pcps_res_direct ? (long long*)e->pcps_res_direct : (long long*)e->pcps_res.ptr;
        return sdr_pcps_fused10k_search(e, F, sh10.off, sh10.P ? (long long)sh10.P * (sh10.H + N) : (long long)nbins * N, C, tw, n_prn, nbins,
                                        noncoh, N, spc, e->pcps_part.ptr, out_bin, out_bin + n_prn, (double*)(out_bin + 2 * n_prn));
    }
    for (int inc = 0; inc < noncoh; ++inc) {
        for (int ic = 0; ic < coh; ++ic) {
            // Forward transforms of the Doppler-mixed millisecond, all bins at once.
            PassArgs f = {};
            f.tw = tw;
            f.N = N;
            f.ring = e->iq;
            f.capacity = e->iq_capacity;
            f.first_sample = start + ((int64_t)inc * coh + ic) * N;
            f.carrier_offset = (int64_t)ic * N;
            f.fs = fs;
            f.if_hz = if_hz;
            f.bin_start = bin_start;
            f.bin_delta = bin_delta;
            SharedSpectra sh;
            if (map_free && fused_takes(e, plan_four_step(N), n_prn, nbins))
                if (int rcs = plan_shared_spectra(e, nbins, N, fs, bin_delta, &sh)) return rcs;
            const int share_P = sh.P;
            const long long* spec_off = sh.off;
            if (share_P) {
                // (the class spectra with their halos, written so by the transform's last pass: F holds share_P rows of H + N)
                f.halo = sh.H;
                run_fft<false, LOAD_IQ_MIX, STORE_PLAIN, FMT>(e, radices, f, share_P, A, B, F, "pcps_fwd_fft", blu);
            } else {
                run_fft<false, LOAD_IQ_MIX, STORE_PLAIN, FMT>(e, radices, f, nbins, A, B, F, "pcps_fwd_fft", blu);
            }
            e->pcps_shared = share_P != 0;
            e->pcps_shared_off = spec_off;

            if (map_free && fused_takes(e, plan_four_step(N), n_prn, nbins)) {
                // every (PRN, bin) transform of the search in ONE launch of persistent workgroups: no intermediate, no sweeps
                if (fused_terms(e, plan_four_step(N)) == 2) {
                    if (!e->pcps_code2_ok) {       // (made with the spectra, kept as long as they are)
                        ProfScope ps(e, "pcps_code_fft");
                        hipLaunchKernelGGL(code_parity_kernel, dim3((N + kThreads - 1) / kThreads, n_prn), dim3(kThreads), 0, e->stream, C,
                                           tw, N, (double2*)e->pcps_code2.ptr);
                        e->pcps_code2_ok = true;
                    }
                    C = (double2*)e->pcps_code2.ptr;
                }
                if (int rcf = sdr_pcps_fused_sweep(e, F, spec_off, C, tw, n_prn, nbins, N, e->pcps_part.ptr)) return rcf;
                continue;
            }
            // Register-resident kernels, several sweeps, nobody timing the stages: the sweeps alternate between two
            // streams and two intermediates, so that one sweep's partial last rounds of workgroups (and its launch
            // ramps) are filled by the other's -- ordered behind the forward transforms and in front of the peak
            // kernels by two events.
            const bool overlap = map_free && fast_applies(e, plan_four_step(N)) && n_prn > prn_chunk && !e->prof &&
                                 !e->pcps_no_overlap;
            if (overlap) {
                if (!e->pcps_aux) {
                    SDR_HIP(hipStreamCreateWithFlags(&e->pcps_aux, hipStreamNonBlocking));
                    SDR_HIP(hipEventCreateWithFlags(&e->pcps_ev[0], hipEventDisableTiming));
                    SDR_HIP(hipEventCreateWithFlags(&e->pcps_ev[1], hipEventDisableTiming));
                }
                SDR_HIP(hipEventRecord(e->pcps_ev[0], e->stream));
                SDR_HIP(hipStreamWaitEvent(e->pcps_aux, e->pcps_ev[0], 0));
            }
            int sweep = 0;
            for (int p0 = 0; p0 < n_prn; p0 += prn_chunk, ++sweep) {
                const int pc = n_prn - p0 < prn_chunk ? n_prn - p0 : prn_chunk;
                if (overlap) {
                    PassArgs g = {};
                    g.tw = tw;
                    g.N = N;
                    g.in = F;
                    g.code_spec = C + (size_t)p0 * N;
                    g.nbins = nbins;
                    g.scale = 1.0 / (double)N;
                    g.partials = (Best*)e->pcps_part.ptr + (size_t)p0 * nbins * records_main_sweep(e, plan_four_step(N));
                    fast_run(e, plan_four_step(N), g, pc * nbins, (sweep & 1) ? B : A, (sweep & 1) ? e->pcps_aux : e->stream);
                    continue;
                }
                PassArgs g = {};
                g.tw = tw;
                g.N = N;
                g.in = F;
                g.code_spec = C + (size_t)p0 * N;
                g.nbins = nbins;
                g.scale = 1.0 / (double)N;
                g.map = map + (size_t)p0 * nbins * N;
                g.csum = csum ? csum + (size_t)p0 * nbins * N : nullptr;
                if (map_free) {
                    g.partials = (Best*)e->pcps_part.ptr + (size_t)p0 * nbins * records_main_sweep(e, plan_four_step(N));
                    run_fft<true, LOAD_MUL_CODE, STORE_MAG_MAX, FMT>(e, radices, g, pc * nbins, A, B, nullptr, "pcps_inv_fft", blu);
                } else if (coh == 1) {
                    g.first_block = inc == 0;
                    run_fft<true, LOAD_MUL_CODE, STORE_MAG_ACC, FMT>(e, radices, g, pc * nbins, A, B, nullptr, "pcps_inv_fft", blu);
                } else {
                    g.first_block = ic == 0;
                    run_fft<true, LOAD_MUL_CODE, STORE_CPLX_ACC, FMT>(e, radices, g, pc * nbins, A, B, nullptr, "pcps_inv_fft", blu);
                }
            }
            if (overlap) {
                SDR_HIP(hipEventRecord(e->pcps_ev[1], e->pcps_aux));
                SDR_HIP(hipStreamWaitEvent(e->stream, e->pcps_ev[1], 0));
            }
        }
        if (coh > 1) {
            ProfScope ps(e, "pcps_mag_acc");
            size_t count = (size_t)n_prn * nbins * N;
            hipLaunchKernelGGL(mag_acc_kernel, dim3((unsigned)((count + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                               e->stream, csum, map, count, inc == 0 ? 1 : 0);
        }
    }

    // K6
    Best* parts = (Best*)e->pcps_part.ptr;
    // (the three result arrays are a few hundred bytes: the peak kernels write them straight into page-locked host
    // memory when the caller set it up -- one copy command and its stream latency less per acquisition)
    long long* res_bin = e->pcps_res_direct ? (long long*)e->pcps_res_direct : (long long*)e->pcps_res.ptr;
    long long* res_code = res_bin + n_prn;
    double* res_ratio = (double*)(res_code + n_prn);
    long long* dev_bin = (long long*)e->pcps_res.ptr;      // device copies of the first peaks (read by the second sweep)
    long long* dev_code = dev_bin + n_prn;
    if (map_free) {
        // the map was never written: maximum from the per-wave records, then the winning row of every PRN alone
        // (1/nbins of one inverse sweep) for the second peak
        const FourStep four = plan_four_step(N);
        const int per_prn = fused_takes(e, four, n_prn, nbins) ? sdr_pcps_fused_records_per_prn(n_prn, nbins, fused_terms(e, four))
                                                               : nbins * records_main_sweep(e, four);
        Best* tops = parts + (size_t)n_prn * per_prn;
        Best* seconds = tops + n_prn;      // [n_prn][records of the second sweep]
        int per_second = records_per_transform(four);
        if (fused_takes(e, four, n_prn, nbins) && !e->pcps_slow_second) {
            // first peaks + second sweep in one launch of n_prn x 5 workgroups (pcps_fused.h ifft_second_kernel)
            // ... which also divides the two peaks (its last workgroup per PRN): the whole K6 in one launch
            // (C: at N = 50 000 the parity images the first sweep used; the spectra: shared ones when that sweep read them so)
            const bool two = fused_terms(e, four) == 2;
            return sdr_pcps_fused_second(e, F, e->pcps_shared_off, two ? (double2*)e->pcps_code2.ptr : C, tw, n_prn, N, spc,
                                         parts, per_prn, tops, dev_bin, dev_code, seconds, res_bin, res_code, res_ratio);
        } else {
            {
                ProfScope ps(e, "pcps_peak");
                hipLaunchKernelGGL(argmax_records_kernel, dim3(n_prn), dim3(kThreads), 0, e->stream, parts, per_prn, N, tops,
                                   dev_bin, dev_code);
            }
            PassArgs g = {};
            g.tw = tw;
            g.N = N;
            g.in = F;
            g.code_spec = C;
            g.nbins = nbins;
            g.scale = 1.0 / (double)N;
            g.sel_bin = dev_bin;
            g.tops = tops;
            g.spc = spc;
            g.partials = seconds;
            run_fft<true, LOAD_MUL_CODE_SEL, STORE_MAG_MAX, FMT>(e, radices, g, n_prn, A, B, nullptr, "pcps_inv_fft", blu);
        }
        {
            ProfScope ps(e, "pcps_peak");
            hipLaunchKernelGGL(ratio_kernel, dim3(n_prn), dim3(64), 0, e->stream, seconds, per_second,
                               tops, dev_bin, dev_code, res_bin, res_code, res_ratio);
        }
        SDR_HIP(hipGetLastError());
        return SDR_OK;
    }
    {
        ProfScope ps(e, "pcps_peak");
        hipLaunchKernelGGL(argmax_part_kernel, dim3(kPeakParts, n_prn), dim3(kThreads), 0, e->stream, map,
                           (long long)nbins * N, parts);
        hipLaunchKernelGGL(peak_finish_kernel, dim3(n_prn), dim3(kThreads), 0, e->stream, map, nbins, N, spc, parts,
                           res_bin, res_code, res_ratio);
    }
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

}  // namespace

extern "C" {

int sdr_pcps_bins(double doppler_range, double doppler_step) {
    if (!(doppler_step > 0.0) || !(doppler_range >= 0.0)) return 0;
    // len(np.arange(-R, R+1, S)) = ceil((stop - start)/step)
    double len = std::ceil(((doppler_range + 1.0) - (-doppler_range)) / doppler_step);
    return len > 0 ? (int)len : 0;
}

int sdr_two_peak_compare(sdr_engine* e, const double* corr_map, int n_bins, int n_code, int samples_per_chip,
                         int64_t* peak_bin, int64_t* peak_code, double* peak_ratio) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!corr_map || !peak_bin || !peak_code || !peak_ratio) return sdr_fail(SDR_ERR_INVALID, "NULL argument");
    if (n_bins < 1 || n_code < 2 || samples_per_chip < 0) return sdr_fail(SDR_ERR_INVALID, "bad map geometry");
    const size_t count = (size_t)n_bins * n_code;
    int rc = sdr_devbuf_reserve(e, &e->pcps_map, count * sizeof(double));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_part, kPeakParts * sizeof(Best));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_res, 3 * sizeof(double) + sizeof(int32_t));
    if (rc) return rc;
    double* map = (double*)e->pcps_map.ptr;
    long long* res = (long long*)e->pcps_res.ptr;
    SDR_HIP(hipMemcpyAsync(map, corr_map, count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    {
        ProfScope ps(e, "pcps_peak");
        hipLaunchKernelGGL(argmax_part_kernel, dim3(kPeakParts, 1), dim3(kThreads), 0, e->stream, map,
                           (long long)count, (Best*)e->pcps_part.ptr);
        hipLaunchKernelGGL(peak_finish_kernel, dim3(1), dim3(kThreads), 0, e->stream, map, n_bins, n_code,
                           samples_per_chip, (const Best*)e->pcps_part.ptr, res, res + 1, (double*)(res + 2));
    }
    SDR_HIP(hipGetLastError());
    long long host[3];
    SDR_HIP(hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    *peak_bin = host[0];
    *peak_code = host[1];
    memcpy(peak_ratio, &host[2], sizeof(double));
    return SDR_OK;
}

static int pcps_impl(sdr_engine* e, const int32_t* code_slots, const double* code_spectra, int n_code_in, int n_prn,
                     int64_t start_sample, double fs, double if_hz, double doppler_range, double doppler_step,
                     int coh, int noncoh, int64_t* peak_bin, int64_t* peak_code, double* peak_ratio,
                     double* corr_map, int* n_bins_out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!code_spectra && !e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if ((!code_slots && !code_spectra) || n_prn <= 0) return sdr_fail(SDR_ERR_INVALID, "no PRN to search");
    if (!peak_bin || !peak_code || !peak_ratio) return sdr_fail(SDR_ERR_INVALID, "NULL peak outputs");
    if (!(fs > 0.0) || coh < 1 || noncoh < 1) return sdr_fail(SDR_ERR_INVALID, "bad fs / integration counts");
    const int nbins = sdr_pcps_bins(doppler_range, doppler_step);
    if (nbins <= 0) return sdr_fail(SDR_ERR_INVALID, "empty Doppler grid");
    if (n_bins_out) *n_bins_out = nbins;
    // samplesPerCode / samplesPerCodeChip (channel_l1ca_kaplan.py:186,205-206), Python round = half-even
    const int64_t N64 = code_spectra ? (int64_t)n_code_in : (int64_t)std::nearbyint(fs * 1023.0 / 1.023e6);
    const int spc = (int)std::nearbyint(fs / 1.023e6);
    if (N64 < 2 || N64 > (1 << 24)) return sdr_fail(SDR_ERR_UNSUPPORTED, "samples per code %lld unsupported", (long long)N64);
    const int N = (int)N64;
    const std::vector<int> radices = factor_radices(N);
    // A code length the mixed-radix planner cannot factor (prime factor above 64) goes through the chirp-z
    // transform with M = the next 2^a 3^b 5^c >= 2N-1.
    const bool use_blu = radices.empty();
    int M = 0;
    if (use_blu) {
        for (int64_t m = 2 * (int64_t)N - 1;; ++m) {
            int64_t q = m;
            for (int f : {2, 3, 5})
                while (q % f == 0) q /= f;
            if (q == 1) {
                M = (int)m;
                break;
            }
        }
    }
    const int64_t need = (int64_t)N * coh * noncoh;
    if (start_sample < 0 || need > e->iq_capacity)
        return sdr_fail(SDR_ERR_RANGE, "acquisition needs %lld samples, ring holds %lld", (long long)need,
                        (long long)e->iq_capacity);
    for (int i = 0; i < n_prn && !code_spectra; ++i)
        if (code_slots[i] < 0 || code_slots[i] >= e->n_slots || e->code_len_host[code_slots[i]] <= 0)
            return sdr_fail(SDR_ERR_INVALID, "PRN entry %d: code slot %d is not staged", i, code_slots[i]);

    // Work-buffer sizing: transforms in flight per inverse sweep are capped at 8 GiB per buffer.
    const size_t tbytes = (size_t)N * sizeof(double2);
    const size_t mbytes = (size_t)M * sizeof(double2);  // (0 without the chirp-z path)
    int prn_chunk = (int)std::min<int64_t>(n_prn, std::max<int64_t>(1, (int64_t)((8ull << 30) / (std::max(tbytes, mbytes) * nbins))));
    if ((int64_t)prn_chunk * nbins > 65535) prn_chunk = std::max(1, 65535 / nbins);
    // Indices and ratio only, one block, four-step transform available: the map is never materialised.
    const FourStep four = plan_four_step(N);
    const bool map_free = !corr_map && coh == 1 && noncoh == 1 && !use_blu && four.ok && !e->pcps_force_passes &&
                          !e->pcps_force_map;
    // The column kernel of an inverse sweep writes 16 N bytes per (PRN, bin) and the row kernel reads them back: as
    // many PRNs per sweep as keep that intermediate inside the 256 MB Infinity Cache (a 200 MB budget), in sweeps of
    // equal size -- 32 PRNs x 41 bins x 25 000: three sweeps of 11 / 11 / 10 PRNs instead of one of 525 MB, measured
    // 0.40 -> 0.34 ms per acquisition (tools/pcps_breakdown.py <chunk>: 12: 0.341, 11: 0.339, 10: 0.348, 8: 0.357,
    // 16: 0.379, 6: 0.392 -- the smaller the sweep, the larger the share of its partial last round of workgroups).
    if (map_free && e->pcps_prn_chunk == 0) {
        const int64_t per_prn = (int64_t)tbytes * nbins;
        // (two sweeps alive at a time where they alternate between two streams: pcps_run)
        const bool two_alive = fast_applies(e, four) && !e->prof && !e->pcps_no_overlap;
        const int fit = (int)std::max<int64_t>(1, ((two_alive ? SDR_PCPS_OVERLAP_MB : 200ll) << 20) / per_prn);
        if (fit < prn_chunk) {
            const int sweeps = (n_prn + fit - 1) / fit;
            prn_chunk = (n_prn + sweeps - 1) / sweeps;
        }
    }
    if (e->pcps_prn_chunk > 0) prn_chunk = std::min(prn_chunk, e->pcps_prn_chunk);
    // (32 units and more: below that the two-kernel path's many small workgroups finish sooner than one unit per CU)
    const bool fused10k = !corr_map && coh == 1 && N == 10000 && !use_blu && four.ok && !e->pcps_force_passes && !e->pcps_force_map &&
                          e->pcps_fused && !e->pcps_no_fast && (int64_t)n_prn * nbins >= 32;
    if (nbins > 65535 || n_prn > 65535) return sdr_fail(SDR_ERR_UNSUPPORTED, "grid too large");
    const size_t work = tbytes * (size_t)std::max(fused10k ? nbins * noncoh : prn_chunk * nbins, std::max(n_prn, nbins));
    int rc = sdr_devbuf_reserve(e, &e->pcps_fwd, tbytes * (size_t)std::max(fused10k ? nbins * noncoh : nbins, n_prn));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_a, work);
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_b, work);
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_code, tbytes * n_prn);
    if (!rc && map_free && fused_terms(e, four) == 2 && fused_takes(e, four, n_prn, nbins)) {
        void* before = e->pcps_code2.ptr;
        rc = sdr_devbuf_reserve(e, &e->pcps_code2, 2 * tbytes * n_prn);
        if (e->pcps_code2.ptr != before) e->pcps_code2_ok = false;
    }
    // (the fused sweep leaves at most 5 x SDR_PCPS_FUSED_RECORDS records per transform, twice that at N = 50 000)
    const size_t n_records = map_free ? (size_t)n_prn * (nbins + 1) * std::max(std::max(records_per_transform(four), records_main_sweep(e, four)), 10 * SDR_PCPS_FUSED_RECORDS) + n_prn : 0;
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_map, (size_t)n_prn * ((map_free || fused10k) ? 1 : nbins) * N * sizeof(double));
    if (!rc && coh > 1) rc = sdr_devbuf_reserve(e, &e->pcps_csum, (size_t)n_prn * nbins * tbytes);
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_part, std::max(std::max((size_t)n_prn * kPeakParts, n_records) * sizeof(Best),
                                                                 fused10k ? (size_t)n_prn * nbins * SDR_PCPS_FUSED10K_RECORD_BYTES : 0));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_res, (size_t)n_prn * 3 * sizeof(double) + n_prn * sizeof(int32_t));
    if (rc) return rc;
    if (e->pcps_tw_n != N) {
        if ((rc = sdr_devbuf_reserve(e, &e->pcps_tw, tbytes))) return rc;
        hipLaunchKernelGGL(twiddle_kernel, dim3((N + kThreads - 1) / kThreads), dim3(kThreads), 0, e->stream,
                           (double2*)e->pcps_tw.ptr, N);
        SDR_HIP(hipGetLastError());
        e->pcps_tw_n = N;
    }
    BluPlan blu;
    if (use_blu) {
        const size_t max_batch = (size_t)std::max(2, std::max(prn_chunk * nbins, std::max(n_prn, nbins)));
        if ((rc = sdr_devbuf_reserve(e, &e->pcps_blu, tbytes + 3 * mbytes))) return rc;
        if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_blu_x, mbytes * max_batch);
        if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_blu_a, mbytes * max_batch);
        if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_blu_b, mbytes * max_batch);
        if (rc) return rc;
        double2* chirp = (double2*)e->pcps_blu.ptr;
        double2* spec_fwd = chirp + N;
        double2* spec_inv = spec_fwd + M;
        double2* twM = spec_inv + M;
        blu.N = N;
        blu.M = M;
        blu.chirp = chirp;
        blu.spec_fwd = spec_fwd;
        blu.spec_inv = spec_inv;
        blu.twM = twM;
        blu.x = (double2*)e->pcps_blu_x.ptr;
        blu.a = (double2*)e->pcps_blu_a.ptr;
        blu.b = (double2*)e->pcps_blu_b.ptr;
        blu.radM = factor_radices(M);
        if (e->pcps_blu_n != N) {  // plan cached per code length: chirp, length-M twiddles, spectra of both kernels
            hipLaunchKernelGGL(twiddle_kernel, dim3((M + kThreads - 1) / kThreads), dim3(kThreads), 0, e->stream, twM, M);
            hipLaunchKernelGGL(blu_chirp_kernel, dim3((M + kThreads - 1) / kThreads), dim3(kThreads), 0, e->stream, chirp,
                               blu.x, blu.x + M, N, M);
            PassArgs p = {};
            p.N = M;
            p.tw = twM;
            p.in = blu.x;  // two transforms at once: [kernel_fwd, kernel_inv] -> [spec_fwd, spec_inv] (adjacent)
            run_fft<false, LOAD_PLAIN, STORE_PLAIN, SDR_FMT_CF64>(e, blu.radM, p, 2, blu.a, blu.b, spec_fwd, "pcps_bluestein_plan");
            SDR_HIP(hipGetLastError());
            e->pcps_blu_n = N;
        }
    }
    int32_t* d_slots = (int32_t*)((char*)e->pcps_res.ptr + (size_t)n_prn * 3 * sizeof(double));
    // small transfers go through page-locked staging: one copy each way, no hidden synchronisation
    const size_t res_bytes = (size_t)n_prn * 3 * sizeof(double);
    // page-locked: [results: bin, code, ratio per PRN][slot numbers][done words]
    if ((rc = sdr_pinned_reserve(e, &e->ctx0, res_bytes + 2 * (size_t)n_prn * sizeof(int32_t)))) return rc;
    char* pin = (char*)e->ctx0.pinned;
    if (code_spectra) {
        SDR_HIP(hipMemcpyAsync(e->pcps_code.ptr, code_spectra, tbytes * n_prn, hipMemcpyHostToDevice, e->stream));
    } else {
        memcpy(pin + res_bytes, code_slots, (size_t)n_prn * sizeof(int32_t));
        e->pcps_slots_pinned = pin + res_bytes;      // (copied to the device by pcps_run when it has to make spectra)
    }
    e->pcps_res_direct = pin;
    // the searches that end in the fused second sweep raise a word per PRN behind its results (sdr_pcps_fused_second): those
    // are waited for instead of the stream's signal, which follows the last store by ~9 us (the receiver tick's finding)
    e->pcps_done = (unsigned*)(pin + res_bytes + (size_t)n_prn * sizeof(int32_t));
    memset(e->pcps_done, 0, (size_t)n_prn * sizeof(unsigned));
    e->pcps_done_seq = ++e->pcps_done_counter ? e->pcps_done_counter : ++e->pcps_done_counter;
    e->pcps_done_used = false;
    const bool hs = code_spectra != nullptr;

    // np.arange(-R, R+1, S): element k = start + k*delta with delta = (start+step) - start
    const double bin_start = -doppler_range;
    const double bin_delta = (bin_start + doppler_step) - bin_start;

    {
    // (one event pair around every kernel of the search: its in-stream time; while it records, the search stays on one stream)
    ProfScope whole(e, "call_pcps");
    switch (e->iq_fmt) {
        case SDR_FMT_CI8: rc = pcps_run<SDR_FMT_CI8>(e, d_slots, code_slots, n_prn, start_sample, fs, if_hz, bin_start, bin_delta, nbins, N, spc, coh, noncoh, radices, prn_chunk, hs, use_blu ? &blu : nullptr, map_free, fused10k); break;
        case SDR_FMT_CI16: rc = pcps_run<SDR_FMT_CI16>(e, d_slots, code_slots, n_prn, start_sample, fs, if_hz, bin_start, bin_delta, nbins, N, spc, coh, noncoh, radices, prn_chunk, hs, use_blu ? &blu : nullptr, map_free, fused10k); break;
        case SDR_FMT_CF32: rc = pcps_run<SDR_FMT_CF32>(e, d_slots, code_slots, n_prn, start_sample, fs, if_hz, bin_start, bin_delta, nbins, N, spc, coh, noncoh, radices, prn_chunk, hs, use_blu ? &blu : nullptr, map_free, fused10k); break;
        default: rc = pcps_run<SDR_FMT_CF64>(e, d_slots, code_slots, n_prn, start_sample, fs, if_hz, bin_start, bin_delta, nbins, N, spc, coh, noncoh, radices, prn_chunk, hs, use_blu ? &blu : nullptr, map_free, fused10k); break;
    }
    }
    e->pcps_res_direct = nullptr;
    unsigned* const done_words = e->pcps_done_used ? e->pcps_done : nullptr;
    e->pcps_done = nullptr;
    if (rc) {
        // a search that stopped half way: what the code-spectra buffer holds is unknown, and a sweep may still be
        // running on the second stream -- join it before anybody re-uses the work buffers
        e->pcps_spec_key.clear();
        if (e->pcps_aux) (void)hipStreamSynchronize(e->pcps_aux);
        return rc;
    }

    if (corr_map)
        SDR_HIP(hipMemcpyAsync(corr_map, e->pcps_map.ptr, (size_t)n_prn * nbins * N * sizeof(double),
                               hipMemcpyDeviceToHost, e->stream));
    bool seen = false;
    if (done_words && !corr_map) {       // (bounded: a launch that died never raises the words -- the stream is asked then)
        const auto t0 = std::chrono::steady_clock::now();
        int c = 0;
        for (long spins = 0;; ++spins) {
            while (c < n_prn && __atomic_load_n(&done_words[c], __ATOMIC_ACQUIRE) == e->pcps_done_seq) ++c;
            if (c == n_prn) {
                seen = true;
                break;
            }
            if ((spins & 4095) == 4095 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) break;
        }
    }
    if (!seen) SDR_HIP(hipStreamSynchronize(e->stream));
    const long long* hb = (const long long*)pin;
    for (int i = 0; i < n_prn; ++i) {
        peak_bin[i] = hb[i];
        peak_code[i] = hb[n_prn + i];
    }
    memcpy(peak_ratio, pin + 2 * (size_t)n_prn * sizeof(long long), (size_t)n_prn * sizeof(double));
    return SDR_OK;
}

int sdr_pcps(sdr_engine* e, const int32_t* code_slots, int n_prn, int64_t start_sample, double fs, double if_hz,
             double doppler_range, double doppler_step, int coh, int noncoh, int64_t* peak_bin,
             int64_t* peak_code, double* peak_ratio, double* corr_map, int* n_bins_out) {
    if (!code_slots) return sdr_fail(SDR_ERR_INVALID, "code_slots is NULL");
    return pcps_impl(e, code_slots, nullptr, 0, n_prn, start_sample, fs, if_hz, doppler_range, doppler_step, coh,
                     noncoh, peak_bin, peak_code, peak_ratio, corr_map, n_bins_out);
}

int sdr_pcps_spectra(sdr_engine* e, const double* code_spectra, int n_prn, int n_code, int64_t start_sample,
                     double fs, double if_hz, double doppler_range, double doppler_step, int coh, int noncoh,
                     int64_t* peak_bin, int64_t* peak_code, double* peak_ratio, double* corr_map,
                     int* n_bins_out) {
    if (!code_spectra || n_code < 2) return sdr_fail(SDR_ERR_INVALID, "code_spectra is NULL or too short");
    return pcps_impl(e, nullptr, code_spectra, n_code, n_prn, start_sample, fs, if_hz, doppler_range, doppler_step,
                     coh, noncoh, peak_bin, peak_code, peak_ratio, corr_map, n_bins_out);
}

}  // extern "C"

/* =====================================================================================================
 * SerialSearch acquisition (sydr/dsp/acquisition.py:119-193; plugin sydr/channel/channel_l1ca_kaplan_ss.py).
 *
 *   map[b][k] = | sum_n x[n]*exp(+1j*bin_b*((2n)*pi/fs)) * code[(u(n) - k) mod L] |^2 ,  u(n) = trunc((ts*n)/tc)
 *
 * The reference evaluates 41 x 1023 dot products of length N.  Every sample enters only through its
 * chip number u(n), so the device first folds the Doppler-mixed millisecond into L per-chip sums
 * (PRN independent, once per bin) and then correlates those L sums circularly with each PRN's chips.
 * ===================================================================================================== */
namespace {

constexpr int kSsMaxChips = 4096;

// A[b][m] = sum over the samples of chip m of x[n] * exp(+1j*bin_b*((2n)*pi/fs))   (acquisition.py:125-131)
template <int FMT>
__global__ __launch_bounds__(kThreads) void ss_chipsum_kernel(const void* __restrict__ ring, int64_t capacity,
                                                              int64_t first_sample, int N, int L, double fs,
                                                              double bin_start, double bin_delta,
                                                              double2* __restrict__ A) {
    const int b = blockIdx.x;
    const double freq = bin_start + (double)b * bin_delta;
    const double ts = 1.0 / fs, tc = 1.0 / 1.023e6;
    const double per_chip = tc / ts;
    for (int m = threadIdx.x; m < L; m += kThreads) {
        int n = (int)floor((double)m * per_chip) - 1;
        if (n < 0) n = 0;
        double sr = 0.0, si = 0.0;
        for (; n < N; ++n) {
            const double v = (ts * (double)n) / tc;  // UpsampleCode index (gnsssignal.py:53)
            const int u = (int)trunc(v);
            if (u < m) continue;
            if (u > m) break;
            const double2 x = ring_sample<FMT>(ring, (first_sample + n) % capacity);
            double pp = (double)((int64_t)n * 2) * M_PI;
            pp = pp / fs;
            double s, c;
            sincos(freq * pp, &s, &c);
            sr += x.x * c - x.y * s;
            si += x.x * s + x.y * c;
        }
        A[(size_t)b * L + m] = make_double2(sr, si);
    }
}

// map[p][b][k] (+)= |sum_m A[b][m] * chip_p[(m - k) mod L]|^2
__global__ __launch_bounds__(kThreads) void ss_correlate_kernel(const double2* __restrict__ A,
                                                                const int8_t* __restrict__ codes, int code_stride,
                                                                const int32_t* __restrict__ slots, int L, int nbins,
                                                                double* __restrict__ map, int accumulate) {
    __shared__ double2 a_s[kSsMaxChips];
    __shared__ float c_s[2 * kSsMaxChips];
    const int b = blockIdx.x, p = blockIdx.y;
    const int8_t* chips = codes + (size_t)slots[p] * code_stride;
    for (int m = threadIdx.x; m < L; m += kThreads) {
        a_s[m] = A[(size_t)b * L + m];
        const float c = chips[m] > 0 ? 1.f : -1.f;
        c_s[m] = c;
        c_s[m + L] = c;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < L; k += kThreads) {
        double sr = 0.0, si = 0.0;
        const float* c = c_s + (L - k);  // c[m] = chip[(m - k) mod L]
        for (int m = 0; m < L; ++m) {
            const double w = (double)c[m];
            sr += w * a_s[m].x;
            si += w * a_s[m].y;
        }
        const size_t o = ((size_t)p * nbins + b) * L + k;
        const double v = sr * sr + si * si;
        map[o] = accumulate ? map[o] + v : 0.0 + v;
    }
}

// TwoCorrelationPeakComparison_SS (acquisition.py:159-193): second peak = max outside the 3x3 block around
// the first, with Python's slice semantics (a block that starts at index -1 selects nothing).
__global__ __launch_bounds__(kThreads) void peak_finish_ss_kernel(const double* __restrict__ map, int nrows, int ncols,
                                                                  const Best* __restrict__ parts,
                                                                  long long* __restrict__ out_bin,
                                                                  long long* __restrict__ out_code,
                                                                  double* __restrict__ out_ratio) {
    __shared__ Best sh[kThreads / 64];
    const int prn = blockIdx.x;
    Best mine = {-1.0, 0x7fffffffffffffffLL};
    if (threadIdx.x < kPeakParts) mine = parts[(size_t)prn * kPeakParts + threadIdx.x];
    Best top = block_best(mine, sh);
    const int i0 = (int)(top.i / ncols), i1 = (int)(top.i - (long long)i0 * ncols);
    int r0 = i0 - 1, r1 = i0 + 2 < nrows ? i0 + 2 : nrows;
    int c0 = i1 - 1, c1 = i1 + 2 < ncols ? i1 + 2 : ncols;
    if (r0 < 0) r0 += nrows;
    if (c0 < 0) c0 += ncols;
    const bool block_empty = r0 >= r1 || c0 >= c1;
    const double* m = map + (size_t)prn * nrows * ncols;
    Best second = {-1.0, 0x7fffffffffffffffLL};
    for (long long i = threadIdx.x; i < (long long)nrows * ncols; i += kThreads) {
        const int r = (int)(i / ncols), c = (int)(i - (long long)r * ncols);
        const bool excluded = !block_empty && r >= r0 && r < r1 && c >= c0 && c < c1;
        if (!excluded) {
            Best cand = {m[i], i};
            second = better(second, cand);
        }
    }
    Best p2 = block_best(second, sh);
    if (threadIdx.x == 0) {
        out_bin[prn] = i0;
        out_code[prn] = i1;
        out_ratio[prn] = top.v / p2.v;
    }
}

template <int FMT>
void ss_launch_chipsum(sdr_engine* e, int64_t first, int N, int L, double fs, double bin_start, double bin_delta,
                       int nbins, double2* A) {
    hipLaunchKernelGGL(ss_chipsum_kernel<FMT>, dim3(nbins), dim3(kThreads), 0, e->stream, e->iq, e->iq_capacity, first,
                       N, L, fs, bin_start, bin_delta, A);
}

int ss_finish(sdr_engine* e, const double* d_map, int n_prn, int nbins, int L, int64_t* peak_bin, int64_t* peak_code,
              double* peak_ratio, double* corr_map) {
    Best* parts = (Best*)e->pcps_part.ptr;
    long long* res_bin = (long long*)e->pcps_res.ptr;
    long long* res_code = res_bin + n_prn;
    double* res_ratio = (double*)(res_code + n_prn);
    {
        ProfScope ps(e, "ss_peak");
        hipLaunchKernelGGL(argmax_part_kernel, dim3(kPeakParts, n_prn), dim3(kThreads), 0, e->stream, d_map,
                           (long long)nbins * L, parts);
        hipLaunchKernelGGL(peak_finish_ss_kernel, dim3(n_prn), dim3(kThreads), 0, e->stream, d_map, nbins, L, parts,
                           res_bin, res_code, res_ratio);
    }
    SDR_HIP(hipGetLastError());
    std::vector<long long> hb(2 * (size_t)n_prn);
    SDR_HIP(hipMemcpyAsync(hb.data(), res_bin, 2 * (size_t)n_prn * sizeof(long long), hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipMemcpyAsync(peak_ratio, res_ratio, (size_t)n_prn * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (corr_map)
        SDR_HIP(hipMemcpyAsync(corr_map, d_map, (size_t)n_prn * nbins * L * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n_prn; ++i) {
        peak_bin[i] = hb[i];
        peak_code[i] = hb[n_prn + i];
    }
    return SDR_OK;
}

}  // namespace

extern "C" {

int sdr_serial_search(sdr_engine* e, const int32_t* code_slots, int n_prn, int64_t start_sample, double fs,
                      double doppler_range, double doppler_step, int noncoh, int64_t* peak_bin, int64_t* peak_code,
                      double* peak_ratio, double* corr_map, int* n_bins_out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if (!code_slots || n_prn <= 0 || n_prn > 65535) return sdr_fail(SDR_ERR_INVALID, "no PRN to search");
    if (!peak_bin || !peak_code || !peak_ratio) return sdr_fail(SDR_ERR_INVALID, "NULL peak outputs");
    if (!(fs > 0.0) || noncoh < 1) return sdr_fail(SDR_ERR_INVALID, "bad fs / integration count");
    const int nbins = sdr_pcps_bins(doppler_range, doppler_step);
    if (nbins <= 0 || nbins > 65535) return sdr_fail(SDR_ERR_INVALID, "bad Doppler grid");
    if (n_bins_out) *n_bins_out = nbins;
    const int N = (int)std::nearbyint(fs * 1023.0 / 1.023e6);
    if (N < 2 || (int64_t)N * noncoh > e->iq_capacity || start_sample < 0)
        return sdr_fail(SDR_ERR_RANGE, "serial search needs %lld samples, ring holds %lld", (long long)N * noncoh,
                        (long long)e->iq_capacity);
    int L = 0;
    for (int i = 0; i < n_prn; ++i) {
        const int s = code_slots[i];
        if (s < 0 || s >= e->n_slots || e->code_len_host[s] <= 0)
            return sdr_fail(SDR_ERR_INVALID, "PRN entry %d: code slot %d is not staged", i, s);
        if (L && e->code_len_host[s] != L) return sdr_fail(SDR_ERR_INVALID, "serial search needs codes of one length");
        L = e->code_len_host[s];
    }
    if (L > kSsMaxChips) return sdr_fail(SDR_ERR_UNSUPPORTED, "codes longer than %d chips", kSsMaxChips);
    int rc = sdr_devbuf_reserve(e, &e->pcps_a, (size_t)nbins * L * sizeof(double2));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_map, (size_t)n_prn * nbins * L * sizeof(double));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_part, (size_t)n_prn * kPeakParts * sizeof(Best));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_res, (size_t)n_prn * 3 * sizeof(double) + n_prn * sizeof(int32_t));
    if (rc) return rc;
    int32_t* d_slots = (int32_t*)((char*)e->pcps_res.ptr + (size_t)n_prn * 3 * sizeof(double));
    SDR_HIP(hipMemcpyAsync(d_slots, code_slots, n_prn * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    const double bin_start = -doppler_range;
    const double bin_delta = (bin_start + doppler_step) - bin_start;
    double2* A = (double2*)e->pcps_a.ptr;
    double* map = (double*)e->pcps_map.ptr;
    for (int k = 0; k < noncoh; ++k) {  // channel_l1ca_kaplan_ss.py:14-21: maps of successive milliseconds are added
        const int64_t first = start_sample + (int64_t)k * N;
        {
            ProfScope ps(e, "ss_chipsum");
            switch (e->iq_fmt) {
                case SDR_FMT_CI8: ss_launch_chipsum<SDR_FMT_CI8>(e, first, N, L, fs, bin_start, bin_delta, nbins, A); break;
                case SDR_FMT_CI16: ss_launch_chipsum<SDR_FMT_CI16>(e, first, N, L, fs, bin_start, bin_delta, nbins, A); break;
                case SDR_FMT_CF32: ss_launch_chipsum<SDR_FMT_CF32>(e, first, N, L, fs, bin_start, bin_delta, nbins, A); break;
                default: ss_launch_chipsum<SDR_FMT_CF64>(e, first, N, L, fs, bin_start, bin_delta, nbins, A); break;
            }
        }
        {
            ProfScope ps(e, "ss_correlate");
            hipLaunchKernelGGL(ss_correlate_kernel, dim3(nbins, n_prn), dim3(kThreads), 0, e->stream, A, e->codes,
                               e->code_stride, d_slots, L, nbins, map, k > 0 ? 1 : 0);
        }
        SDR_HIP(hipGetLastError());
    }
    return ss_finish(e, map, n_prn, nbins, L, peak_bin, peak_code, peak_ratio, corr_map);
}

int sdr_two_peak_compare_ss(sdr_engine* e, const double* corr_map, int n_rows, int n_cols, int64_t* peak_bin,
                            int64_t* peak_code, double* peak_ratio) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!corr_map || !peak_bin || !peak_code || !peak_ratio || n_rows < 1 || n_cols < 1)
        return sdr_fail(SDR_ERR_INVALID, "bad map");
    const size_t count = (size_t)n_rows * n_cols;
    int rc = sdr_devbuf_reserve(e, &e->pcps_map, count * sizeof(double));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_part, kPeakParts * sizeof(Best));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->pcps_res, 3 * sizeof(double) + sizeof(int32_t));
    if (rc) return rc;
    SDR_HIP(hipMemcpyAsync(e->pcps_map.ptr, corr_map, count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    return ss_finish(e, (const double*)e->pcps_map.ptr, 1, n_rows, n_cols, peak_bin, peak_code, peak_ratio, nullptr);
}

}  // extern "C"
