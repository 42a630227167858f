// Device-side core of the E/P/L correlator, shared by the open-loop batch kernel (epl.hip)
// and the persistent closed-loop tracking kernel (track.hip).
//
// Follows EPL of the reference (sydr/dsp/tracking.py:92-116):
//   replica_i = exp(1j*(-(f*2.0*pi*(i/fs)) + rem_carrier))
//   idx_i     = ceil(linspace(shift, code_step*n + shift, n, endpoint=False)),  shift = rem_code + spacing
//   I,Q       = sum code[idx_i] * Re/Im(replica_i * x_i)
//
// What is reference arithmetic and what is not:
//   * the chip index is the reference's, operation for operation, in IEEE fp64 with no FMA
//     (np.linspace + np.ceil) -- it must be bit-exact;
//   * the carrier replica is evaluated once per 8-sample group in fp64 (exact Cody-Waite
//     reduction + minimax sin/cos) and advanced inside the group by 8 precomputed fp64
//     rotations; mix and accumulation use FMAs.  They agree with NumPy to ~1e-13 relative,
//     the bar being 1e-6.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "engine_internal.h"

#pragma clang fp contract(off)

namespace sdr {

constexpr int kGroup = 8;  // samples per lane per iteration (one 16-byte load of ci8)

constexpr double kHalfPiHi = 1.57079632679489655800e+00;  // fl(pi/2)
constexpr double kHalfPiLo = 6.12323399573676603587e-17;  // pi/2 - fl(pi/2)
constexpr double kTwoOverPi = 6.36619772367581382433e-01;

// fp64 sin/cos of a phase of up to ~1e6 rad: Cody-Waite reduction to |t| <= pi/4 by whole
// quarter turns (the k*hi product is exact inside the FMA), then the classic degree-13 / 14
// minimax kernels (fdlibm k_sin / k_cos coefficients, < 1 ulp on the reduced range).  This
// replaces the general libm sincos, whose Payne-Hanek path is dead weight here.
__device__ __forceinline__ void sincos_reduced(double ph, double* s, double* c) {
    const double k = rint(ph * kTwoOverPi);
    double t = __builtin_fma(-k, kHalfPiHi, ph);
    t = __builtin_fma(-k, kHalfPiLo, t);
    const double z = t * t;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(t * z, ps, t);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k;
    const double a = (q & 1) ? cs : sn;
    const double b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}

// Load 8 consecutive ring samples starting at aligned position `pos` and widen to fp64.
template <int FMT>
struct Loader;

template <>
struct Loader<SDR_FMT_CI8> {
    static __device__ __forceinline__ void load(const void* ring, int64_t pos, double* xr, double* xi) {
        const int4 v = *reinterpret_cast<const int4*>(static_cast<const char*>(ring) + pos * 2);
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            xr[2 * d] = (double)(int)(int8_t)(w[d]);
            xi[2 * d] = (double)(int)(int8_t)(w[d] >> 8);
            xr[2 * d + 1] = (double)(int)(int8_t)(w[d] >> 16);
            xi[2 * d + 1] = (double)(w[d] >> 24);
        }
    }
};

template <>
struct Loader<SDR_FMT_CI16> {
    static __device__ __forceinline__ void load(const void* ring, int64_t pos, double* xr, double* xi) {
        const int4* p = reinterpret_cast<const int4*>(static_cast<const char*>(ring) + pos * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int4 v = p[h];
            const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                xr[4 * h + d] = (double)(int)(int16_t)(w[d]);
                xi[4 * h + d] = (double)(w[d] >> 16);
            }
        }
    }
};

template <>
struct Loader<SDR_FMT_CF32> {
    static __device__ __forceinline__ void load(const void* ring, int64_t pos, double* xr, double* xi) {
        const float4* p = reinterpret_cast<const float4*>(static_cast<const char*>(ring) + pos * 8);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float4 v = p[h];
            xr[2 * h] = v.x;
            xi[2 * h] = v.y;
            xr[2 * h + 1] = v.z;
            xi[2 * h + 1] = v.w;
        }
    }
};

template <>
struct Loader<SDR_FMT_CF64> {
    static __device__ __forceinline__ void load(const void* ring, int64_t pos, double* xr, double* xi) {
        const double2* p = reinterpret_cast<const double2*>(static_cast<const char*>(ring) + pos * 16);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const double2 v = p[h];
            xr[h] = v.x;
            xi[h] = v.y;
        }
    }
};

// Stage a PRN replica into LDS: lut[q] = chip[(q - PAD - 1) mod L] as the HIGH WORD of +-1.0
// (so a gathered entry becomes an fp64 multiplier with no conversion).
template <int THREADS>
__device__ __forceinline__ void stage_lut(uint32_t* lut, const int8_t* __restrict__ chips, int L, int tid) {
    const int words = L + 2 * SDR_LUT_PAD + 2;
    for (int q = tid; q < words; q += THREADS) {
        int c = q - SDR_LUT_PAD - 1;
        c = c < 0 ? c + L : (c >= L ? c - L : c);
        c = c < 0 ? c + L : (c >= L ? c - L : c);  // PAD + 1 < L is checked on the host
        lut[q] = chips[c] > 0 ? 0x3FF00000u : 0xBFF00000u;
    }
}

// Per-epoch NCO inputs of one channel (what the reference passes to EPL).
struct EpochParams {
    int64_t start_sample;
    int n;
    double carrier_hz, rem_carrier, rem_code, code_step;
};

// exp(-1j*j*dphi), j = 0..7, computed by lanes 0..7 into rot[16]; caller barriers afterwards.
__device__ __forceinline__ double carrier_step(double carrier_hz, double fs) {
    const double w = (carrier_hz * 2.0) * M_PI;  // tracking.py:102 uses np.pi
    return w / fs;
}
__device__ __forceinline__ void stage_rotations(double* rot, double dphi, int tid) {
    if (tid < kGroup) {
        double s, c;
        sincos_reduced(-(double)tid * dphi, &s, &c);
        rot[2 * tid] = c;
        rot[2 * tid + 1] = s;
    }
}

// Correlate this thread's share (groups tid, tid+THREADS, ...) of one epoch.
// accr/acci[NT] receive the thread-partial fp64 accumulators.
template <int FMT, int NT, int THREADS>
__device__ __forceinline__ void correlate_epoch(const void* __restrict__ ring, int64_t capacity,
                                                const EpochParams& ep, const double* spacing, double dphi,
                                                const double* rot, const uint32_t* lut, int tid, double* accr,
                                                double* acci) {
    const int n = ep.n;
    double rc[kGroup], rs[kGroup];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        rc[j] = rot[2 * j];
        rs[j] = rot[2 * j + 1];
    }
    // np.linspace(shift, code_step*n + shift, n, endpoint=False) per tap (tracking.py:111-112).
    const double nd = (double)n;
    double shift[NT], step[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        shift[t] = ep.rem_code + spacing[t];
        double stop = ep.code_step * nd;
        stop = stop + shift[t];
        double delta = stop - shift[t];
        step[t] = delta / nd;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;

    const int64_t aligned = ep.start_sample & ~(int64_t)(kGroup - 1);
    const int head = (int)(ep.start_sample - aligned);
    const int n_groups = (head + n + kGroup - 1) / kGroup;
    const int64_t base = aligned % capacity;

    // One 8-sample group.  EDGE = the group straddles the start or the end of the epoch:
    // samples outside [0,n) are zeroed and their (unused) chip index is clamped into range.
    auto group = [&](int g, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        int64_t pos = base + (int64_t)g * kGroup;
        if (pos >= capacity) pos -= capacity;
        double xr[kGroup], xi[kGroup];
        Loader<FMT>::load(ring, pos, xr, xi);

        const int i0 = g * kGroup - head;
        double sb, cb;
        sincos_reduced(__builtin_fma(-(double)i0, dphi, ep.rem_carrier), &sb, &cb);

        double gr[NT], gi[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) gr[t] = gi[t] = 0.0;

#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            int i = i0 + j;
            double ar = xr[j], ai = xi[j];
            if (EDGE) {
                const bool valid = (unsigned)i < (unsigned)n;
                ar = valid ? ar : 0.0;
                ai = valid ? ai : 0.0;
                i = i < 0 ? 0 : (i >= n ? n - 1 : i);
            }
            const double zr = __builtin_fma(-ai, rs[j], ar * rc[j]);
            const double zi = __builtin_fma(ai, rc[j], ar * rs[j]);
            const double di = (double)i;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double y = di * step[t];  // reference arithmetic: separate mul, add, ceil
                y = y + shift[t];
                const int p = (int)ceil(y);
                const double c = __hiloint2double((int)lut[p + SDR_LUT_PAD], 0);
                gr[t] = __builtin_fma(c, zr, gr[t]);
                gi[t] = __builtin_fma(c, zi, gi[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            accr[t] += __builtin_fma(cb, gr[t], -sb * gi[t]);
            acci[t] += __builtin_fma(cb, gi[t], sb * gr[t]);
        }
    };

    for (int g = tid; g < n_groups; g += THREADS) {
        const int i0 = g * kGroup - head;
        if (i0 >= 0 && i0 + kGroup <= n)
            group(g, std::false_type{});
        else
            group(g, std::true_type{});
    }
}

// Workgroup reduction of 2*NT fp64 accumulators: wavefront shuffles (64 lanes), then the waves
// through LDS in a fixed order -- deterministic, so results do not depend on GPU sharding.
// red needs (THREADS/64)*2*NT doubles.  After the call threads 0..2*NT-1 hold the totals
// (thread 2t: I_t, thread 2t+1: Q_t) in the return value.
template <int NT, int THREADS>
__device__ __forceinline__ double reduce_taps(const double* accr, const double* acci, double* red, int tid) {
    constexpr int kWaves = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double a = accr[t], b = acci[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off, 64);
            b += __shfl_down(b, off, 64);
        }
        if (lane == 0) {
            red[wave * 2 * NT + 2 * t] = a;
            red[wave * 2 * NT + 2 * t + 1] = b;
        }
    }
    __syncthreads();
    double s = 0.0;
    if (tid < 2 * NT) {
        s = red[tid];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) s += red[wv * 2 * NT + tid];
    }
    return s;
}

}  // namespace sdr
