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
//   * the carrier replica is evaluated once per lane in fp64 (exact Cody-Waite reduction +
//     minimax sin/cos), advanced inside a group by 8 / 16 precomputed fp64 rotations and from
//     group to group by one more; mix and accumulation use FMAs.  They agree with NumPy to
//     ~1e-15 relative on the benchmark stream (1.9e-12 worst case in the stress runs), the bar being 1e-6.
//
// Contents: the per-sample core (correlate_epoch), the boundary variant (wide_group /
// correlate_epoch_wide: a lane owns 16 or 8 consecutive samples), its single-round form for the
// closed-loop clusters (correlate_epoch_single), the wave / workgroup reductions.  The chip-aligned
// variant of the open-loop kernel lives in correlator_chip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "engine_internal.h"
#include "sincos_reduced.h"

#pragma clang fp contract(off)

namespace sdr {

constexpr int kGroup = 8;  // samples per lane per iteration (one 16-byte load of ci8)
constexpr int kWide = 16;  // samples per lane per iteration of the boundary variant

// Values that are the same in every lane (per-epoch constants) are pinned to SGPRs: it frees ~60
// VGPRs per lane, which is what decides between 2 and 3-4 resident waves per SIMD here.
__device__ __forceinline__ double uniform(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// A ci8 ring holds its samples with the sign bit of every byte flipped (u = x + 128 as an unsigned byte: "offset binary") --
// the form the straight-line correlators build their doubles from with one v_perm_b32 per component (correlator_chip.h
// biased_sample).  Round 6: that IS the ring, not an image beside it -- the engine flips where samples enter (uploads, the
// ingest kernels, the synthesiser) and flips back where they leave (sdr_iq_download); every other reader takes the bits back
// with one exclusive or per dword of two samples, here.
constexpr uint32_t kCi8Flip = 0x80808080u;
__device__ __forceinline__ int ci8_native(int w) { return w ^ (int)kCi8Flip; }

// Load 8 consecutive ring samples starting at aligned position `pos` and widen to fp64.
template <int FMT>
struct Loader;

template <>
struct Loader<SDR_FMT_CI8> {
    static __device__ __forceinline__ void load(const void* ring, int64_t pos, double* xr, double* xi) {
        const int4 v = *reinterpret_cast<const int4*>(static_cast<const char*>(ring) + pos * 2);
        const int w[4] = {ci8_native(v.x), ci8_native(v.y), ci8_native(v.z), ci8_native(v.w)};
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

// Stage a PRN replica into LDS: the per-slot table prepared by expand_lut_kernel (high words of +-1.0,
// lut[q] = chip[(q - PAD - 1) mod L]) is copied with 16-byte loads -- two per lane for a C/A code.
template <int THREADS>
__device__ __forceinline__ void stage_lut(uint32_t* lut, const uint32_t* __restrict__ glut, int words, int tid) {
    const int quads = (words + 3) >> 2;
    const uint4* src = reinterpret_cast<const uint4*>(glut);
    uint4* dst = reinterpret_cast<uint4*>(lut);
    for (int q = tid; q < quads; q += THREADS) dst[q] = src[q];
}

// Per-epoch NCO inputs of one channel (what the reference passes to EPL).
struct EpochParams {
    int64_t start_sample;
    int n;
    double carrier_hz, rem_carrier, rem_code, code_step;
};

__host__ __device__ __forceinline__ double carrier_step(double carrier_hz, double fs) {
    const double w = (carrier_hz * 2.0) * M_PI;  // tracking.py:102 uses np.pi
    return w / fs;
}

// One ring sample, widened to fp64 (edge samples only).
template <int FMT>
__device__ __forceinline__ void load_one(const void* ring, int64_t pos, double& xr, double& xi) {
    if (FMT == SDR_FMT_CI8) {
        const char2 v = static_cast<const char2*>(ring)[pos];
        xr = (double)(int8_t)(v.x ^ 0x80);
        xi = (double)(int8_t)(v.y ^ 0x80);
    } else if (FMT == SDR_FMT_CI16) {
        const short2 v = static_cast<const short2*>(ring)[pos];
        xr = (double)v.x;
        xi = (double)v.y;
    } else if (FMT == SDR_FMT_CF32) {
        const float2 v = static_cast<const float2*>(ring)[pos];
        xr = (double)v.x;
        xi = (double)v.y;
    } else {
        const double2 v = static_cast<const double2*>(ring)[pos];
        xr = v.x;
        xi = v.y;
    }
}

// The few samples of an epoch that do not fill a whole aligned group -- [0, head_end) and
// [tail_start, n), at most 2*(group-1) of them -- are correlated one per lane with the plain
// per-sample form of the reference arithmetic, in a single pass of the first wave.  (Routing them
// through a masked copy of the group body cost two extra ~1000-instruction passes per epoch.)
template <int FMT, int NT>
__device__ __forceinline__ void edge_samples(const void* __restrict__ ring, int64_t capacity,
                                             const EpochParams& ep, double dphi, const double* shift,
                                             const double* step, const uint32_t* lut, int lane, int head_end,
                                             int tail_start, double* accr, double* acci, int64_t base = -1) {
    const int count = head_end + (ep.n - tail_start);
    if (lane >= count) return;
    const int i = lane < head_end ? lane : tail_start + (lane - head_end);
    // base >= 0: the ring position of the epoch's first sample, from a caller whose epoch does not wrap (a 64-bit
    // modulo is ~60 instructions)
    int64_t pos = base >= 0 ? base + i : (ep.start_sample + i) % capacity;
    double xr, xi;
    load_one<FMT>(ring, pos, xr, xi);
    double sn, cs;
    sincos_reduced(__builtin_fma(-(double)i, dphi, ep.rem_carrier), &sn, &cs);
    const double zr = __builtin_fma(-xi, sn, xr * cs);
    const double zi = __builtin_fma(xi, cs, xr * sn);
    const double di = (double)i;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double y = di * step[t];  // reference arithmetic: separate mul, add, ceil
        y = y + shift[t];
        const int p = (int)ceil(y);
        const double c = __hiloint2double((int)lut[p + SDR_LUT_PAD], 0);
        accr[t] = __builtin_fma(c, zr, accr[t]);
        acci[t] = __builtin_fma(c, zi, acci[t]);
    }
}

// Per-epoch constants.  Every wave computes them for itself in ONE pass -- lane l evaluates a different
// quantity (lanes 0..10: sin/cos of -m_l*dphi; lanes 16..16+NT-1: the np.linspace constants of tap
// l-16) -- and v_readlane hands each result to the whole wave as an SGPR pair.  No LDS, no barrier,
// and ~60 fewer VGPRs per lane than keeping them in vector registers.
template <int NT>
struct EpochConsts {
    double rc[kWide], rs[kWide];  // cos,sin(-j*dphi): per-sample rotations inside a group (the 8-sample loop uses 0..7)
    double cN, sN;                // -8*THREADS*dphi:  lane stride of the 8-sample loop
    double cW, sW;                // -16*THREADS*dphi: lane stride of the 16-sample loop
    double shift[NT], step[NT];   // np.linspace(shift, code_step*n+shift, n, endpoint=False) (tracking.py:111-112)
    double inv_step[NT];          // 1/step: only ever used to PREDICT a chip-switch position
};

__device__ __forceinline__ double lane_value(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

constexpr int kConstTapLane = 24;  // lanes 0..15: in-group rotations, 16/17: lane strides, 24..: taps

// `stride` = number of lanes that share the epoch (the workgroup's threads, or all threads of the
// workgroups that cooperate on one channel).
template <int NT>
__device__ __forceinline__ void compute_constants(EpochConsts<NT>& k, const EpochParams& ep,
                                                  const double* __restrict__ spacing, double dphi, int stride) {
    const int lane = threadIdx.x & 63;
    const double mult = lane < kWide ? (double)lane
                                     : (lane == kWide ? (double)(kGroup * stride) : (double)(kWide * stride));
    double sn, cs;
    sincos_reduced(-mult * dphi, &sn, &cs);
    int t = lane - kConstTapLane;
    t = t < 0 ? 0 : (t > NT - 1 ? NT - 1 : t);
    const double nd = (double)ep.n;
    const double shift = ep.rem_code + spacing[t];  // reference arithmetic, operation for operation
    double stop = ep.code_step * nd;
    stop = stop + shift;
    const double delta = stop - shift;
    const double step = delta / nd;
    // 1/step only ever PREDICTS a chip-switch position (the prediction is re-checked exactly whenever it falls within
    // kNearInteger of a sample): hardware reciprocal estimate + one Newton step (~2^-50) instead of a division chain
    const double r0 = __builtin_amdgcn_rcp(step);
    const double inv = __builtin_fma(__builtin_fma(-step, r0, 1.0), r0, r0);
#pragma unroll
    for (int j = 0; j < kWide; ++j) {
        k.rc[j] = lane_value(cs, j);
        k.rs[j] = lane_value(sn, j);
    }
    k.cN = lane_value(cs, kWide);
    k.sN = lane_value(sn, kWide);
    k.cW = lane_value(cs, kWide + 1);
    k.sW = lane_value(sn, kWide + 1);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        k.shift[q] = lane_value(shift, kConstTapLane + q);
        k.step[q] = lane_value(step, kConstTapLane + q);
        k.inv_step[q] = lane_value(inv, kConstTapLane + q);
    }
}

// Only the np.linspace constants of the taps (what the chip-aligned core reads of EpochConsts): no rotations, no
// sincos -- the per-sample rotations are evaluated where an epoch actually falls back to the per-sample core.
template <int NT>
__device__ __forceinline__ void compute_tap_constants(EpochConsts<NT>& k, const EpochParams& ep,
                                                      const double* __restrict__ spacing) {
    const int lane = threadIdx.x & 63;
    const int t = lane < NT ? lane : NT - 1;
    const double nd = (double)ep.n;
    const double shift = ep.rem_code + spacing[t];  // reference arithmetic, operation for operation
    double stop = ep.code_step * nd;
    stop = stop + shift;
    const double delta = stop - shift;
    const double step = delta / nd;
    const double r0 = __builtin_amdgcn_rcp(step);
    const double inv = __builtin_fma(__builtin_fma(-step, r0, 1.0), r0, r0);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        k.shift[q] = lane_value(shift, q);
        k.step[q] = lane_value(step, q);
        k.inv_step[q] = lane_value(inv, q);
    }
}

// Correlate this lane's share (groups lane, lane+stride, ...) of one epoch; `lane` is the index among
// the `stride` lanes that share the epoch.  edge_lane = 0..63 in the ONE wave that also takes the
// epoch's edge samples, -1 elsewhere.  accr/acci[NT] receive the lane-partial fp64 accumulators.
template <int FMT, int NT>
__device__ __forceinline__ void correlate_epoch(const void* __restrict__ ring, int64_t capacity,
                                                const EpochParams& ep, double dphi, const EpochConsts<NT>& K,
                                                const uint32_t* lut, int lane, int stride, int edge_lane,
                                                double* accr, double* acci) {
    const int n = ep.n;
    const double* rc = K.rc;
    const double* rs = K.rs;
    const double c_it = K.cN, s_it = K.sN;
    const double* shift = K.shift;
    const double* step = K.step;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;

    const int64_t aligned = ep.start_sample & ~(int64_t)(kGroup - 1);
    const int head = (int)(ep.start_sample - aligned);
    const int64_t base = aligned % capacity;
    // whole groups g in [g_lo, g_hi) lie inside the epoch; the rest are edge samples
    const int g_lo = head ? 1 : 0;
    const int g_hi = (head + n) / kGroup;
    const int head_end = g_hi > g_lo ? g_lo * kGroup - head : n;   // no whole group: every sample is "edge"
    const int tail_start = g_hi > g_lo ? g_hi * kGroup - head : n;

    // One whole 8-sample group.
    auto group = [&](int g, double sb, double cb) {
        int64_t pos = base + (int64_t)g * kGroup;
        if (pos >= capacity) pos -= capacity;
        double xr[kGroup], xi[kGroup];
        Loader<FMT>::load(ring, pos, xr, xi);

        const int i0 = g * kGroup - head;

        double gr[NT], gi[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) gr[t] = gi[t] = 0.0;

#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const int i = i0 + j;
            const double ar = xr[j], ai = xi[j];
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

    // carrier phase at the lane's first sample: one exact evaluation, then a fixed rotation per iteration
    double sb, cb;
    sincos_reduced(__builtin_fma(-(double)((g_lo + lane) * kGroup - head), dphi, ep.rem_carrier), &sb, &cb);
    for (int g = g_lo + lane; g < g_hi; g += stride) {
        group(g, sb, cb);
        const double cbn = __builtin_fma(cb, c_it, -sb * s_it);
        sb = __builtin_fma(sb, c_it, cb * s_it);
        cb = cbn;
    }
    if (edge_lane >= 0 && head_end + (n - tail_start) > 64) {
        // (only when n < 8 + 64: a tiny epoch; walk the edge list in wave-sized pieces)
        for (int off = 0; off < head_end + (n - tail_start); off += 64)
            edge_samples<FMT, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane + off, head_end, tail_start, accr, acci);
    } else if (edge_lane >= 0) {
        edge_samples<FMT, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane, head_end, tail_start, accr, acci);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Boundary variant (used when 16*code_step < 1, i.e. fs above ~17 MHz for C/A code).
 *
 * A lane owns 16 consecutive samples (two 16-byte loads), i.e. < 1 chip, so for every tap the
 * reference chip index takes at most two values p0, p0+1 inside the group and the sequence is
 * monotone: the first `nlead` samples sit on chip p0, the rest on p0+1.  The lane mixes its 16
 * samples with the carrier once, keeps the running sums P_1..P_16 in a private LDS strip
 * (P_0 = 0; 8 slots, used twice), and a tap's share of the group is then
 *        c(p0+1) * P_16 + (c(p0) - c(p0+1)) * P_nlead
 * -- two indexed 16-byte LDS reads and a handful of FMAs per tap per 16 samples, instead of a chip
 * index, a gather and two FMAs per tap per sample.
 *
 * nlead has to be EXACTLY what NumPy's linspace/ceil gives.  y0 = fl(fl(i0*step)+shift) and
 * p0 = ceil(y0) are the reference expression itself.  The crossing is predicted as
 * e = (p0 - y0)/step, nlead = floor(e)+1.  The reference's y_k differ from the real line
 * y0 + k*step by at most ~4 ulp(y) (< 2^-33 for y < 2^18), i.e. by < 2^-33/step samples; the
 * launcher admits this variant only for step >= 1e-4, so the prediction can be wrong only when e
 * lies within 2^-19 of an integer.  Whenever any tap of any lane of the wave is within 2^-16
 * (kNearInteger) of one, the whole wave recomputes nlead from exact evaluations of the reference
 * expression on both sides of the predicted crossing (a branch taken by < 1 % of the groups).
 * ------------------------------------------------------------------------------------------------ */
constexpr int kPrefixSlots = kGroup + 1;  // double2 slots of LDS per lane used by the boundary variant
constexpr double kNearInteger = 1.0 / 65536.0;  // predicted crossings this close to a sample are re-checked exactly
constexpr double kFastMaxCodeStep = 0.06;  // 16-sample groups: 15 * step <= 0.9 chip
constexpr double kFastMaxCodeStep8 = 0.125;  // 8-sample groups:  7 * step <= 0.875 chip (fs above ~8.2 MHz for C/A code)
constexpr double kFastMinCodeStep = 1e-4;  // keeps the switch prediction's error far below kNearInteger (see above)
constexpr int kFastMaxLutWords = 1 << 18;   // chip coordinates below 2^18: ulp(y) <= 2^-35

template <int FMT>
struct Raw8;  // 8 consecutive samples kept in their storage format until they are needed

template <>
struct Raw8<SDR_FMT_CI8> {
    int4 v;
    __device__ __forceinline__ void load(const void* ring, int64_t pos) {
        v = *reinterpret_cast<const int4*>(static_cast<const char*>(ring) + pos * 2);
    }
    __device__ __forceinline__ void get(int j, double& xr, double& xi) const {
        // (the ring's bytes are sign-flipped: one exclusive or per dword gives the two's-complement samples back)
        const int w = ci8_native((j >> 1) == 0 ? v.x : ((j >> 1) == 1 ? v.y : ((j >> 1) == 2 ? v.z : v.w)));
        if (j & 1) {
            xr = (double)(int)(int8_t)(w >> 16);
            xi = (double)(w >> 24);
        } else {
            xr = (double)(int)(int8_t)(w);
            xi = (double)(int)(int8_t)(w >> 8);
        }
    }
};

template <>
struct Raw8<SDR_FMT_CI16> {
    int4 v[2];
    __device__ __forceinline__ void load(const void* ring, int64_t pos) {
        const int4* p = reinterpret_cast<const int4*>(static_cast<const char*>(ring) + pos * 4);
        v[0] = p[0];
        v[1] = p[1];
    }
    __device__ __forceinline__ void get(int j, double& xr, double& xi) const {
        const int4 q = v[j >> 2];
        const int w = (j & 3) == 0 ? q.x : ((j & 3) == 1 ? q.y : ((j & 3) == 2 ? q.z : q.w));
        xr = (double)(int)(int16_t)(w);
        xi = (double)(w >> 16);
    }
};

template <>
struct Raw8<SDR_FMT_CF32> {
    float4 v[4];
    __device__ __forceinline__ void load(const void* ring, int64_t pos) {
        const float4* p = reinterpret_cast<const float4*>(static_cast<const char*>(ring) + pos * 8);
#pragma unroll
        for (int h = 0; h < 4; ++h) v[h] = p[h];
    }
    __device__ __forceinline__ void get(int j, double& xr, double& xi) const {
        const float4 q = v[j >> 1];
        xr = (j & 1) ? q.z : q.x;
        xi = (j & 1) ? q.w : q.y;
    }
};

template <>
struct Raw8<SDR_FMT_CF64> {
    double2 v[8];
    __device__ __forceinline__ void load(const void* ring, int64_t pos) {
        const double2* p = reinterpret_cast<const double2*>(static_cast<const char*>(ring) + pos * 16);
#pragma unroll
        for (int h = 0; h < 8; ++h) v[h] = p[h];
    }
    __device__ __forceinline__ void get(int j, double& xr, double& xi) const {
        xr = v[j].x;
        xi = v[j].y;
    }
};

// True when the epoch's samples cross the end of the ring (once per ring revolution per channel): such
// an epoch goes through the per-sample variant, whose group loop re-wraps every position.
__device__ __forceinline__ bool epoch_wraps(const EpochParams& ep, int64_t capacity) {
    const int64_t aligned = ep.start_sample & ~(int64_t)(kGroup - 1);
    return aligned % capacity + (ep.start_sample - aligned) + ep.n + kWide > capacity;
}

// One group of W (8 or 16) consecutive samples of one lane, boundary variant: samples i0 .. i0+W-1 of the epoch
// (raw[] holds them in storage format), carrier phasor (cb, sb) at sample i0.  TAIL: only the first v (1..W) samples
// belong to the epoch -- its last group in the single-round form of the closed-loop kernel, which has no separate
// edge-sample pass: the running sums are there anyway, so the group's share is c(p0+1)*P_v + (c(p0)-c(p0+1))*P_min(nlead,v)
// and what lies behind sample v is never read.  strip: this lane's kPrefixSlots double2 slots of LDS, slot 0 holding zero.
// ROT8 (16-sample groups only): rc / rs hold the rotations of samples 0..7 alone; the second half is mixed with the
// same eight and its running sums are turned by (c8, s8) = exp(-1j*8*dphi) where they are read -- 16 FMAs per group
// instead of eight more rotations per epoch, which cost the closed-loop kernel 28 v_readlane and 32 v_mov.
template <int FMT, int NT, int W, bool TAIL, bool ROT8 = false>
__device__ __forceinline__ void wide_group(const Raw8<FMT>* raw, int i0, int v, const double* rc, const double* rs,
                                           const double* shift, const double* step, const double* inv_step,
                                           const uint32_t* lut, double2* strip, double sb, double cb, double* accr,
                                           double* acci, double c8 = 1.0, double s8 = 0.0) {
    constexpr int kHalves = W / kGroup;
    const double di0 = (double)i0;

    int nlead[NT];                    // leading samples on chip p0, 1..16 (16: the whole group)
    double sign_b[NT], sign_diff[NT];  // c(p0+1) and c(p0) - c(p0+1)
    double y0[NT];
    int p0[NT];
    bool near = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double y = di0 * step[t];  // the reference's chip index, exactly: separate multiply, add, ceil
        y = y + shift[t];
        const double cy = ceil(y);
        y0[t] = y;
        p0[t] = (int)cy;
        const double e = (cy - y) * inv_step[t];  // >= 0
        const double fr = __builtin_amdgcn_fract(e);  // v_fract_f64: e - floor(e)
        near |= fabs(fr - 0.5) > 0.5 - kNearInteger;
        const double ec = fmin(e, (double)(W - 1));
        nlead[t] = (int)ec + 1;
    }
    if (__builtin_expect(__any(near), 0)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            auto chip = [&](int i) {
                double y = (double)i * step[t];
                y = y + shift[t];
                return (int)ceil(y);
            };
            const int b = nlead[t] > W - 1 ? W - 1 : nlead[t];  // compare samples b-1 | b, both in the group
            const int pa = chip(i0 + b - 1);
            const int pb = chip(i0 + b);
            nlead[t] = (pa != p0[t]) ? b - 1 : ((pb == p0[t]) ? b + 1 : b);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double ca = __hiloint2double((int)lut[p0[t] + SDR_LUT_PAD], 0);
        sign_b[t] = __hiloint2double((int)lut[p0[t] + 1 + SDR_LUT_PAD], 0);
        sign_diff[t] = ca - sign_b[t];
        if (TAIL) nlead[t] = nlead[t] < v ? nlead[t] : v;
    }

    // First half: P_1..P_8 into slots 1..8 (slot 0 stays 0); every tap reads P_min(nlead,8).
    // Second half: the sums restart at sample 8 (Q_1..Q_8) and reuse the same slots -- LDS operations
    // of a wave execute in order, so the reads above are served first; every tap reads Q_max(nlead-8,0).
    // P_nlead = P_min(nlead,8) + Q_max(nlead-8,0), and P_16 = P_8 + Q_8.
    double pr = 0.0, pi = 0.0;
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        double ar, ai;
        raw[0].get(j, ar, ai);
        pr = __builtin_fma(-ai, rs[j], __builtin_fma(ar, rc[j], pr));
        pi = __builtin_fma(ai, rc[j], __builtin_fma(ar, rs[j], pi));
        strip[1 + j] = make_double2(pr, pi);
    }
    if (kHalves == 1) {  // 8-sample groups: the strip holds everything, P_nlead is one read
        if (TAIL) {
            const double2 pv = strip[v];
            pr = pv.x, pi = pv.y;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double2 pm = strip[nlead[t]];
            const double gr = __builtin_fma(sign_diff[t], pm.x, sign_b[t] * pr);
            const double gi = __builtin_fma(sign_diff[t], pm.y, sign_b[t] * pi);
            accr[t] = __builtin_fma(-sb, gi, __builtin_fma(cb, gr, accr[t]));
            acci[t] = __builtin_fma(sb, gr, __builtin_fma(cb, gi, acci[t]));
        }
        return;
    }
    double2 pa[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pa[t] = strip[nlead[t] < kGroup ? nlead[t] : kGroup];
    double2 pva = make_double2(0.0, 0.0);
    if (TAIL) pva = strip[v < kGroup ? v : kGroup];
    constexpr int kOff = ROT8 ? 0 : (kHalves - 1) * kGroup;   // which rotations the second half is mixed with
    double qr = 0.0, qi = 0.0;
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        double ar, ai;
        raw[kHalves - 1].get(j, ar, ai);
        qr = __builtin_fma(-ai, rs[kOff + j], __builtin_fma(ar, rc[kOff + j], qr));
        qi = __builtin_fma(ai, rc[kOff + j], __builtin_fma(ar, rs[kOff + j], qi));
        strip[1 + j] = make_double2(qr, qi);
    }
    auto turn = [&](double2 q) {   // a running sum of the second half as mixed with the full rotations
        if (!ROT8) return q;
        return make_double2(__builtin_fma(-q.y, s8, q.x * c8), __builtin_fma(q.y, c8, q.x * s8));
    };
    if (TAIL) {
        const double2 pvb = turn(strip[(v > kGroup ? v : kGroup) - kGroup]);
        pr = pva.x + pvb.x;
        pi = pva.y + pvb.y;
    } else {
        const double2 qt = turn(make_double2(qr, qi));
        pr += qt.x;
        pi += qt.y;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double2 qb = turn(strip[(nlead[t] > kGroup ? nlead[t] : kGroup) - kGroup]);
        const double gr = __builtin_fma(sign_diff[t], pa[t].x + qb.x, sign_b[t] * pr);
        const double gi = __builtin_fma(sign_diff[t], pa[t].y + qb.y, sign_b[t] * pi);
        accr[t] = __builtin_fma(-sb, gi, __builtin_fma(cb, gr, accr[t]));
        acci[t] = __builtin_fma(sb, gr, __builtin_fma(cb, gi, acci[t]));
    }
}

// tid = index of the thread in its workgroup (selects the LDS strip); lane/stride/edge_lane as in
// correlate_epoch.  SINGLE_WAVE: the epoch belongs to one wave alone (the batched kernel).
template <int FMT, int NT, bool SINGLE_WAVE, int W = kWide>
__device__ __forceinline__ void correlate_epoch_wide(const void* __restrict__ ring, int64_t capacity,
                                                     const EpochParams& ep, double dphi, const EpochConsts<NT>& K,
                                                     const uint32_t* lut, double2* prefix_lds, int tid, int lane,
                                                     int stride, int edge_lane, double* accr, double* acci) {
    const int n = ep.n;
    double2* strip = prefix_lds + tid * kPrefixSlots;  // this lane's 9 x 16 B (odd multiple of 16 B: conflict-free)
    strip[0] = make_double2(0.0, 0.0);
    // 16 rotations = 64 scalar registers would overflow the SGPR file (and the spills come back as
    // v_readlane in the loop): the second half lives in vector registers instead, which are plentiful here.
    static_assert(W == kGroup || W == 2 * kGroup, "a lane owns one or two 16-byte loads of ci8");
    constexpr int kHalves = W / kGroup;
    double rc[W], rs[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        rc[j] = K.rc[j];
        rs[j] = K.rs[j];
        if (j >= kGroup) {
            asm volatile("" : "+v"(rc[j]));
            asm volatile("" : "+v"(rs[j]));
        }
    }
    const double c_it = W == kWide ? K.cW : K.cN, s_it = W == kWide ? K.sW : K.sN;  // lane stride of this loop
    const double dphi_u = uniform(dphi), rem_carrier_u = uniform(ep.rem_carrier);
    const double* shift = K.shift;
    const double* step = K.step;
    const double* inv_step = K.inv_step;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;

    const int64_t aligned = ep.start_sample & ~(int64_t)(kGroup - 1);
    const int head = (int)(ep.start_sample - aligned);
    const int64_t base = aligned % capacity;
    // whole 16-sample groups g in [g_lo, g_hi); everything else is an edge sample (<= 30 of them)
    const int g_lo = head ? 1 : 0;
    const int g_hi = (head + n) / W;
    const int head_end = g_hi > g_lo ? g_lo * W - head : n;
    const int tail_start = g_hi > g_lo ? g_hi * W - head : n;

    // (the caller guarantees that the whole groups do not wrap around the ring: see epoch_wraps())
    auto load_group = [&](int64_t pos, Raw8<FMT>* raw) {
        raw[0].load(ring, pos);
        if (kHalves == 2) raw[kHalves - 1].load(ring, pos + kGroup);
    };

    auto group = [&](int g, const Raw8<FMT>* raw, double sb, double cb) {
        wide_group<FMT, NT, W, false>(raw, g * W - head, W, rc, rs, shift, step, inv_step, lut, strip, sb, cb, accr, acci);
    };

    // Software prefetch: the next group's 16-byte loads are in flight while this one is computed.  The
    // loop is unrolled by two over a ping-pong pair of buffers and is free of divergent control flow:
    // the trip count is wave-uniform, every load is unconditional (a lane that has run out of groups
    // stays on its last one) and such a lane's carrier phasor is zeroed, so what it computes adds 0.
    // (Handing prefetched registers over by copy, or loading under a lane mask, makes the compiler
    // wait for a load in the iteration that issued it.)
    const int n_groups = g_hi - g_lo;
    if (n_groups > 0) {
        Raw8<FMT> buf_a[kHalves], buf_b[kHalves];
        int g = g_lo + lane;
        bool alive = g < g_hi;
        g = alive ? g : g_hi - 1;
        int64_t pos = base + (int64_t)g * W;
        load_group(pos, buf_a);
        // Carrier phase at the lane's first sample: one exact evaluation, then a fixed rotation per
        // iteration (the lane's groups are kWide*THREADS samples apart; <= a few dozen steps, so the
        // recurrence stays within ~1e-15 of a fresh evaluation).
        double sb, cb;
        sincos_reduced(__builtin_fma(-(double)(g * W - head), dphi_u, rem_carrier_u), &sb, &cb);
        sb = alive ? sb : 0.0;
        cb = alive ? cb : 0.0;
        auto advance = [&]() {  // to the lane's next group, or stay (with a zero phasor) when there is none
            const bool more = g + stride < g_hi;
            const double cbn = __builtin_fma(cb, c_it, -sb * s_it);
            const double sbn = __builtin_fma(sb, c_it, cb * s_it);
            sb = more ? sbn : 0.0;
            cb = more ? cbn : 0.0;
            alive = more;
            g += more ? stride : 0;
            pos += more ? W * stride : 0;
        };
        // A wave all of whose lanes have run out (the last, partial round of a multi-wave workgroup)
        // skips the group body; the single-wave kernel never pays for the test.
        auto wave_has_work = [&](bool lane_alive) { return SINGLE_WAVE || __any(lane_alive) != 0; };
        const int rounds = (n_groups + stride - 1) / stride;
        for (int it = 0; it < rounds / 2; ++it) {
            const double sb0 = sb, cb0 = cb;
            const int g0 = g;
            const bool alive0 = alive;
            advance();
            load_group(pos, buf_b);
            if (wave_has_work(alive0)) group(g0, buf_a, sb0, cb0);
            const double sb1 = sb, cb1 = cb;
            const int g1 = g;
            const bool alive1 = alive;
            advance();
            load_group(pos, buf_a);
            if (wave_has_work(alive1)) group(g1, buf_b, sb1, cb1);
        }
        if ((rounds & 1) && wave_has_work(alive)) group(g, buf_a, sb, cb);
    }
    // (callers give the edge samples to the LAST wave: the partial last round lands on the first ones)
    if (edge_lane >= 0) {
        if (head_end + (n - tail_start) > 64) {   // epoch shorter than a group + a wave: walk the list
            for (int off = 0; off < head_end + (n - tail_start); off += 64)
                edge_samples<FMT, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane + off, head_end, tail_start, accr, acci);
        } else {
            edge_samples<FMT, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane, head_end, tail_start, accr, acci);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Single-round form (closed-loop kernel, a channel on a cluster of workgroups whose lanes cover the
 * whole epoch in one pass): lane g of the cluster owns samples 16g .. 16g+15 of the epoch, loaded from
 * wherever the epoch starts (gfx950 serves 16-byte loads from any 2-byte aligned address), so there is
 * no head; the epoch's last group is cut with the running sums it keeps anyway (wide_group<TAIL>) instead
 * of handing its samples to a separate edge pass.  The caller loads the groups itself -- one epoch ahead:
 * where epoch k+1 starts is known when epoch k starts (start + n), only its length waits for the code loop.
 * ------------------------------------------------------------------------------------------------ */
struct SingleGeometry {
    int64_t pos0;  // ring position of the epoch's first sample
    int groups;    // 16-sample groups that hold a sample of the epoch
    bool fits;     // every group lies inside the ring (an epoch that wraps goes through the per-sample variant)
};

// pos0 = start_sample % capacity, kept by the caller from epoch to epoch (+n, minus the capacity when it passes it):
// a 64-bit modulo is ~60 instructions, and a lone wave pays for every one of them.
__device__ __forceinline__ SingleGeometry single_geometry(int64_t pos0, int n, int64_t capacity) {
    SingleGeometry s;
    s.pos0 = pos0;
    s.groups = (n + kWide - 1) / kWide;
    s.fits = s.pos0 + (int64_t)s.groups * kWide <= capacity;
    return s;
}

// Ring position lane g loads from: its group, kept inside the ring for lanes beyond the epoch (and for an epoch
// that wraps the ring, which ignores what was loaded here).
__device__ __forceinline__ int64_t single_load_pos(const SingleGeometry& s, int g, int64_t capacity) {
    const int64_t pos = s.pos0 + (int64_t)g * kWide;
    return pos > capacity - kWide ? capacity - kWide : pos;
}

template <int FMT>
__device__ __forceinline__ void single_load(const void* __restrict__ ring, int64_t pos, Raw8<FMT>* raw) {
    raw[0].load(ring, pos);
    raw[1].load(ring, pos + kGroup);
}

template <int FMT, int NT>
__device__ __forceinline__ void correlate_epoch_single(const Raw8<FMT>* raw, const EpochParams& ep, double dphi,
                                                       const EpochConsts<NT>& K, const uint32_t* lut,
                                                       double2* prefix_lds, int tid, int g, const SingleGeometry& geo,
                                                       double* accr, double* acci) {
    double2* strip = prefix_lds + tid * kPrefixSlots;
    strip[0] = make_double2(0.0, 0.0);
    const double* rc = K.rc;       // rotations 0..7 (scalar registers) and exp(-1j*8*dphi) for the second half
    const double* rs = K.rs;
    const double c8 = K.rc[kGroup], s8 = K.rs[kGroup];
    constexpr bool kRot8 = true;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;
    const bool alive = g < geo.groups;
    if (!__any(alive)) return;                                   // a wave wholly beyond the epoch
    const int ge = alive ? g : geo.groups - 1;                   // (a lane beyond the epoch computes on the last group's indices)
    const int i0 = ge * kWide;
    double sb, cb;
    sincos_reduced(__builtin_fma(-(double)i0, uniform(dphi), uniform(ep.rem_carrier)), &sb, &cb);
    const int v = ep.n - i0 < kWide ? ep.n - i0 : kWide;         // samples of the group that belong to the epoch
    if (__any(v < kWide))
        wide_group<FMT, NT, kWide, true, kRot8>(raw, i0, v, rc, rs, K.shift, K.step, K.inv_step, lut, strip, sb, cb, accr, acci, c8, s8);
    else
        wide_group<FMT, NT, kWide, false, kRot8>(raw, i0, v, rc, rs, K.shift, K.step, K.inv_step, lut, strip, sb, cb, accr, acci, c8, s8);
#pragma unroll
    for (int t = 0; t < NT; ++t) {                               // (not a zero phasor: float rings may hold NaNs out there)
        accr[t] = alive ? accr[t] : 0.0;
        acci[t] = alive ? acci[t] : 0.0;
    }
}

// Sum of x over the 64 lanes of the wave, returned to every lane: xor-butterfly inside each row of
// 16 lanes with DPP moves (no LDS traffic, no address arithmetic), then the four row sums in a
// fixed order through v_readlane.  The order never depends on the data: results are deterministic.
__device__ __forceinline__ double wave_sum(double x) {
    auto dpp_add = [](double v, auto ctrl) {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), decltype(ctrl)::value, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), decltype(ctrl)::value, 0xf, 0xf, true);
        return v + __hiloint2double(hi, lo);
    };
    x = dpp_add(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]: lane ^ 1
    x = dpp_add(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]: lane ^ 2
    x = dpp_add(x, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad of the 8
    x = dpp_add(x, std::integral_constant<int, 0x140>{});  // row_mirror: the other half of the 16
    return ((lane_value(x, 0) + lane_value(x, 16)) + lane_value(x, 32)) + lane_value(x, 48);
}

// The same for a workgroup whose totals are wanted by ONE wave only (the closed-loop clusters): the four DPP
// stages leave every lane of a row of 16 with the row's sum, the row leaders park the 2*NT row sums in LDS, and after
// the barrier the collector wave adds the 4*waves rows -- G lanes per value, each over its share of the rows, then a
// DPP butterfly over the G lanes.  No v_readlane at all (the cross-row step of wave_sum is 8 of them per value, and a
// lone wave pays ~8 cycles per instruction whatever it is).  Fixed order: deterministic.  The sum of value v comes
// back in lanes v*G .. v*G+G-1 of the collector (collector_group_lanes(NT) = G); red needs 4*waves*2*NT doubles.
constexpr int collector_group_lanes(int nt) { return 64 / (2 * nt) >= 8 ? 8 : (64 / (2 * nt) >= 4 ? 4 : 2); }

__device__ __forceinline__ double dpp_add_f64(double v, int which) {
    // which: 0 = lane ^ 1, 1 = lane ^ 2, 2 = the other quad of the 8, 3 = the other half of the 16
    int lo, hi;
    switch (which) {
        case 0:
            lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true);
            hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
            break;
        case 1:
            lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xf, 0xf, true);
            hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xf, 0xf, true);
            break;
        case 2:
            lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x141, 0xf, 0xf, true);
            hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x141, 0xf, 0xf, true);
            break;
        default:
            lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x140, 0xf, 0xf, true);
            hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x140, 0xf, 0xf, true);
            break;
    }
    return v + __hiloint2double(hi, lo);
}

template <int NT, int THREADS, int COLLECTOR>
__device__ __forceinline__ double reduce_taps_rows(const double* accr, const double* acci, double* red, int tid) {
    constexpr int kWaves = THREADS / 64, kRows = 4 * kWaves, kVals = 2 * NT, G = collector_group_lanes(NT);
    static_assert(kRows % G == 0, "every collector lane adds the same number of rows");
    const int lane = tid & 63, wave = tid >> 6;
    double rsum[kVals];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double a = accr[t], b = acci[t];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            a = dpp_add_f64(a, st);
            b = dpp_add_f64(b, st);
        }
        rsum[2 * t] = a;
        rsum[2 * t + 1] = b;
    }
    if ((lane & 15) == 0) {
        double* mine = red + (wave * 4 + (lane >> 4)) * kVals;
#pragma unroll
        for (int k = 0; k < kVals; ++k) mine[k] = rsum[k];
    }
    __syncthreads();
    double s = 0.0;
    if (wave == COLLECTOR && lane < kVals * G) {
        const int v = lane / G, g = lane - v * G;
#pragma unroll
        for (int r = 0; r < kRows / G; ++r) s += red[(g + r * G) * kVals + v];
        s = dpp_add_f64(s, 0);
        if (G >= 4) s = dpp_add_f64(s, 1);
        if (G >= 8) s = dpp_add_f64(s, 2);
    }
    return s;
}

// Workgroup reduction of 2*NT fp64 accumulators: wave sums, then the waves through LDS in a fixed
// order.  red needs (THREADS/64)*2*NT doubles.  After the call threads 0..2*NT-1 hold the totals
// (thread 2t: I_t, thread 2t+1: Q_t) in the return value.
template <int NT, int THREADS, int COLLECTOR = 0>
__device__ __forceinline__ double reduce_taps(const double* accr, const double* acci, double* red, int tid) {
    constexpr int kWaves = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
    double mine = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double ta = wave_sum(accr[t]), tb = wave_sum(acci[t]);
        mine = lane == 2 * t ? ta : (lane == 2 * t + 1 ? tb : mine);
    }
    if (kWaves == 1) return mine;  // one wave: lanes 0..2*NT-1 already hold the totals, no LDS, no barrier
    if (lane < 2 * NT) red[wave * 2 * NT + lane] = mine;
    __syncthreads();
    double s = 0.0;
    if (wave == COLLECTOR && lane < 2 * NT) {
        s = red[lane];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) s += red[wv * 2 * NT + lane];
    }
    return s;
}

// ---- one wave, reduce-scatter form ------------------------------------------------------------------------------------
// reduce_taps() sums each of the 2*NT accumulators over the 64 lanes separately (4 DPP stages + 8 v_readlane per value:
// ~27 instructions each).  Here the lanes SHARE the work: at the step that pairs lane l with lane l ^ d, a lane keeps
// one value of every pair of values and hands the other to its partner, so the list halves with every step (6 -> 3 ->
// 2 -> 1 values for three taps) and only the last value is carried through the remaining steps: ~55 instructions in
// all.  Afterwards every lane holds the wave total of ONE value -- which one depends on its low lane bits and is
// returned in `slot` (2t: I of tap t, 2t + 1: Q); lanes 0..7 between them hold all of them.  The order of the additions
// is fixed by the lane numbers: deterministic, as reduce_taps() is (the two differ in the last bits: another tree).
template <int D>
__device__ __forceinline__ double lane_xor_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (D == 1) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
    } else if constexpr (D == 2) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true);
    } else if constexpr (D == 4) {
        // lanes with bit 2 set read four lanes down (row_shr:4 into banks 1 and 3), the others four lanes up
        const int l1 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xA, false), h1 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xA, false);
        lo = __builtin_amdgcn_update_dpp(l1, lo, 0x104, 0xf, 0x5, false), hi = __builtin_amdgcn_update_dpp(h1, hi, 0x104, 0xf, 0x5, false);
    } else if constexpr (D == 8) {
        const int l1 = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xC, false), h1 = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xC, false);
        lo = __builtin_amdgcn_update_dpp(l1, lo, 0x108, 0xf, 0x3, false), hi = __builtin_amdgcn_update_dpp(h1, hi, 0x108, 0xf, 0x3, false);
    } else if constexpr (D == 16) {
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F), hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);   // bit mode: xor 0x10
    } else {
        const int addr = ((int)(threadIdx.x & 63) ^ 32) << 2;
        lo = __builtin_amdgcn_ds_bpermute(addr, lo), hi = __builtin_amdgcn_ds_bpermute(addr, hi);
    }
    return __hiloint2double(hi, lo);
}

template <int N, int D>
__device__ __forceinline__ void scatter_steps(double (&v)[N > 0 ? N : 1], int (&slot)[N > 0 ? N : 1], int lane, double& total, int& which) {
    if constexpr (D > 32) {
        static_assert(N == 1, "six halvings leave one value");
        total = v[0], which = slot[0];
    } else {
        constexpr int M = (N + 1) / 2;
        double nv[M];
        int ns[M];
        const bool upper = (lane & D) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const double keep = upper ? v[2 * i + 1] : v[2 * i];
            const double give = upper ? v[2 * i] : v[2 * i + 1];
            nv[i] = keep + lane_xor_f64<D>(give);
            ns[i] = upper ? slot[2 * i + 1] : slot[2 * i];
        }
        if constexpr (N & 1) {
            nv[M - 1] = v[N - 1] + lane_xor_f64<D>(v[N - 1]);
            ns[M - 1] = slot[N - 1];
        }
        scatter_steps<M, D * 2>(nv, ns, lane, total, which);
    }
}

template <int NT>
__device__ __forceinline__ double reduce_taps_scatter(const double* accr, const double* acci, int lane, int& slot) {
    double v[2 * NT];
    int s[2 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        v[2 * t] = accr[t], v[2 * t + 1] = acci[t];
        s[2 * t] = 2 * t, s[2 * t + 1] = 2 * t + 1;
    }
    double total;
    scatter_steps<2 * NT, 1>(v, s, lane, total, slot);
    return total;
}

}  // namespace sdr
