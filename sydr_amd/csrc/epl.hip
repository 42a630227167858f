// K1: batched carrier wipe-off + multi-tap PRN correlate-accumulate.
//
// One workgroup = one channel-epoch = one call of the reference's EPL
// (sydr/dsp/tracking.py:92-116).  IQ is streamed from the HBM ring with 16-byte
// coalesced loads (8 ci8 samples per lane per load), the PRN replica sits in LDS
// as the high words of +-1.0, accumulators are fp64 and are reduced with
// wavefront shuffles, then across the 4 waves through LDS in a fixed order (so
// results do not depend on how channels are sharded over GPUs).
//
// Arithmetic that selects a chip is the reference's, operation for operation
// (np.linspace + np.ceil), in IEEE fp64 with contraction off:
//     shift = rem_code + spacing            stop = code_step*n + shift
//     step  = (stop - shift) / n            idx_i = ceil(i*step + shift)
// The carrier replica exp(1j*(-(f*2.0*pi*(i/fs)) + rem)) is evaluated once per
// 8-sample group in fp64 (exact range reduction + libm-grade sincos) and
// advanced inside the group by 8 precomputed fp64 rotations, which agrees with
// the reference to ~1e-12 rad (far inside the 1e-6 relative bar on accumulators).
#include "engine_internal.h"

#include <cmath>

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kGroup = 8;  // samples per lane per iteration

constexpr double kTwoPiHi = 6.283185307179586232e+00;   // fl(2*pi)
constexpr double kTwoPiLo = 2.449293598294706414e-16;   // 2*pi - fl(2*pi)
constexpr double kInvTwoPi = 1.591549430918953456e-01;

__device__ __forceinline__ void sincos_reduced(double ph, double* s, double* c) {
    // ph may be thousands of radians (non-zero IF): remove whole turns exactly first.
    double k = rint(ph * kInvTwoPi);
    double r = fma(-k, kTwoPiHi, ph);
    r = fma(-k, kTwoPiLo, r);
    sincos(r, s, c);
}

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

// Dynamic LDS: [rot: 16 doubles][red: kWaves*2*NT doubles][lut: lut_words uint32]
template <int FMT, int NT>
__global__ __launch_bounds__(kThreads) void epl_kernel(const void* __restrict__ ring, int64_t capacity,
                                                       const sdr_epl_item* __restrict__ items,
                                                       const int8_t* __restrict__ codes,
                                                       const int32_t* __restrict__ code_len, int code_stride,
                                                       const double* __restrict__ spacing, double fs,
                                                       int tap0, int n_taps_total,
                                                       double* __restrict__ out) {
    extern __shared__ double smem[];
    double* rot = smem;
    double* red = smem + 2 * kGroup;
    uint32_t* lut = reinterpret_cast<uint32_t*>(red + kWaves * 2 * NT);

    const int tid = threadIdx.x;
    const sdr_epl_item it = items[blockIdx.x];
    const int n = it.n_samples;
    const int L = code_len[it.code_slot];

    // Stage the PRN replica: lut[q] = chip[(q - PAD - 1) mod L] as the high word of +-1.0.
    {
        const int8_t* chips = codes + (size_t)it.code_slot * code_stride;
        const int words = L + 2 * SDR_LUT_PAD + 2;
        for (int q = tid; q < words; q += kThreads) {
            int c = q - SDR_LUT_PAD - 1;
            c %= L;
            if (c < 0) c += L;
            lut[q] = chips[c] > 0 ? 0x3FF00000u : 0xBFF00000u;
        }
    }

    // Per-sample carrier advance inside a group: rot_j = exp(-1j*j*dphi).
    const double w = (it.carrier_hz * 2.0) * M_PI;
    const double dphi = w / fs;
    if (tid < kGroup) {
        double s, c;
        sincos_reduced(-(double)tid * dphi, &s, &c);
        rot[2 * tid] = c;
        rot[2 * tid + 1] = s;
    }
    __syncthreads();
    double rc[kGroup], rs[kGroup];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        rc[j] = rot[2 * j];
        rs[j] = rot[2 * j + 1];
    }

    // np.linspace(shift, code_step*n + shift, n, endpoint=False) per tap.
    const double nd = (double)n;
    double shift[NT], step[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        shift[t] = it.rem_code + spacing[tap0 + t];
        double stop = it.code_step * nd;
        stop = stop + shift[t];
        double delta = stop - shift[t];
        step[t] = delta / nd;
    }

    double accr[NT], acci[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;

    const int64_t aligned = it.start_sample & ~(int64_t)(kGroup - 1);
    const int head = (int)(it.start_sample - aligned);
    const int n_groups = (head + n + kGroup - 1) / kGroup;
    const int64_t base = aligned % capacity;

    for (int g = tid; g < n_groups; g += kThreads) {
        int64_t pos = base + (int64_t)g * kGroup;
        if (pos >= capacity) pos -= capacity;
        double xr[kGroup], xi[kGroup];
        Loader<FMT>::load(ring, pos, xr, xi);

        const int i0 = g * kGroup - head;
        double sb, cb;
        sincos_reduced(fma(-(double)i0, dphi, it.rem_carrier), &sb, &cb);

        double gr[NT], gi[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) gr[t] = gi[t] = 0.0;

#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const int i = i0 + j;
            const bool valid = (unsigned)i < (unsigned)n;
            const double ar = valid ? xr[j] : 0.0;
            const double ai = valid ? xi[j] : 0.0;
            const double zr = ar * rc[j] - ai * rs[j];
            const double zi = ar * rs[j] + ai * rc[j];
            const int ic = i < 0 ? 0 : (i >= n ? n - 1 : i);
            const double di = (double)ic;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double y = di * step[t];
                y = y + shift[t];
                const int p = (int)ceil(y);
                const double c = __hiloint2double((int)lut[p + SDR_LUT_PAD], 0);
                gr[t] += c * zr;
                gi[t] += c * zi;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            accr[t] += cb * gr[t] - sb * gi[t];
            acci[t] += cb * gi[t] + sb * gr[t];
        }
    }

    // Wavefront shuffle reduction (64 lanes), then the 4 waves through LDS.
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
    if (tid < 2 * NT) {
        double s = red[tid];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) s += red[wv * 2 * NT + tid];
        out[(size_t)blockIdx.x * 2 * n_taps_total + 2 * tap0 + tid] = s;
    }
}

template <int FMT, int NT>
void launch_one(sdr_engine* e, const sdr_epl_item* d_items, int n_items, const double* d_spacing, double fs,
                int tap0, int n_taps_total, int lut_words, double* d_out) {
    size_t shmem = (2 * kGroup + kWaves * 2 * NT) * sizeof(double) + (size_t)lut_words * sizeof(uint32_t);
    hipLaunchKernelGGL((epl_kernel<FMT, NT>), dim3(n_items), dim3(kThreads), shmem, e->stream, e->iq,
                       e->iq_capacity, d_items, e->codes, e->code_len, e->code_stride, d_spacing, fs, tap0,
                       n_taps_total, d_out);
}

template <int FMT>
void launch_fmt(sdr_engine* e, const sdr_epl_item* d_items, int n_items, const double* d_spacing, double fs,
                int n_taps, int lut_words, double* d_out) {
    // Taps are served in register-resident chunks of 5/3/2/1.
    int t0 = 0;
    while (t0 < n_taps) {
        int left = n_taps - t0;
        if (left >= 5) {
            launch_one<FMT, 5>(e, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, d_out);
            t0 += 5;
        } else if (left >= 3) {
            launch_one<FMT, 3>(e, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, d_out);
            t0 += 3;
        } else if (left == 2) {
            launch_one<FMT, 2>(e, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, d_out);
            t0 += 2;
        } else {
            launch_one<FMT, 1>(e, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, d_out);
            t0 += 1;
        }
    }
}

}  // namespace

struct sdr_epl_plan {
    sdr_epl_item* d_items = nullptr;
    double* d_out = nullptr;
    double* d_spacing = nullptr;
    int n_items = 0;
    int n_taps = 0;
    int lut_words = 0;
    double fs = 0.0;
};

// Host-side check that no item can index outside the ring or the staged LUT.
static int validate_items(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing,
                          int n_taps, int* lut_words) {
    double smin = spacing[0], smax = spacing[0];
    for (int t = 1; t < n_taps; ++t) {
        smin = spacing[t] < smin ? spacing[t] : smin;
        smax = spacing[t] > smax ? spacing[t] : smax;
    }
    int maxlen = 0;
    for (int i = 0; i < n_items; ++i) {
        const sdr_epl_item& it = items[i];
        if (it.code_slot < 0 || it.code_slot >= e->n_slots || e->code_len_host[it.code_slot] <= 0)
            return sdr_fail(SDR_ERR_INVALID, "item %d: code slot %d is not staged", i, it.code_slot);
        if (it.n_samples <= 0 || it.n_samples > e->iq_capacity)
            return sdr_fail(SDR_ERR_RANGE, "item %d: n_samples %d outside (0, ring capacity]", i, it.n_samples);
        if (it.start_sample < 0) return sdr_fail(SDR_ERR_RANGE, "item %d: negative start_sample", i);
        if (!(it.code_step > 0.0) || !std::isfinite(it.rem_code) || !std::isfinite(it.rem_carrier) ||
            !std::isfinite(it.carrier_hz))
            return sdr_fail(SDR_ERR_INVALID, "item %d: non-finite or non-positive NCO parameter", i);
        const int L = e->code_len_host[it.code_slot];
        const double lo = std::ceil(it.rem_code + smin);
        const double hi = std::ceil(it.code_step * (double)it.n_samples + it.rem_code + smax);
        if (lo < -(double)SDR_LUT_PAD || hi > (double)(L + SDR_LUT_PAD))
            return sdr_fail(SDR_ERR_RANGE,
                            "item %d: code phase range [%g, %g] leaves the staged replica [-%d, %d]", i, lo, hi,
                            SDR_LUT_PAD, L + SDR_LUT_PAD);
        if (L > maxlen) maxlen = L;
    }
    *lut_words = maxlen + 2 * SDR_LUT_PAD + 2;
    return SDR_OK;
}

extern "C" {

int sdr_epl_plan_create(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing,
                        int n_taps, double fs, sdr_epl_plan** out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if (!items || n_items <= 0) return sdr_fail(SDR_ERR_INVALID, "no items");
    if (!spacing || n_taps < 1 || n_taps > SDR_MAX_TAPS)
        return sdr_fail(SDR_ERR_INVALID, "n_taps %d outside 1..%d", n_taps, SDR_MAX_TAPS);
    if (!(fs > 0.0)) return sdr_fail(SDR_ERR_INVALID, "fs must be positive");
    int lut_words = 0;
    if (int rc = validate_items(e, items, n_items, spacing, n_taps, &lut_words)) return rc;

    sdr_epl_plan* p = new (std::nothrow) sdr_epl_plan();
    if (!p) return sdr_fail(SDR_ERR_NOMEM, "host allocation failed");
    p->n_items = n_items;
    p->n_taps = n_taps;
    p->fs = fs;
    p->lut_words = lut_words;
    hipError_t err = hipMalloc(&p->d_items, (size_t)n_items * sizeof(sdr_epl_item));
    if (err == hipSuccess) err = hipMalloc(&p->d_out, (size_t)n_items * 2 * n_taps * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&p->d_spacing, SDR_MAX_TAPS * sizeof(double));
    if (err == hipSuccess)
        err = hipMemcpyAsync(p->d_items, items, (size_t)n_items * sizeof(sdr_epl_item), hipMemcpyHostToDevice,
                             e->stream);
    if (err == hipSuccess)
        err = hipMemcpyAsync(p->d_spacing, spacing, n_taps * sizeof(double), hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
        sdr_epl_plan_destroy(e, p);
        return sdr_fail(err == hipErrorOutOfMemory ? SDR_ERR_NOMEM : SDR_ERR_HIP, "plan setup failed: %s",
                        hipGetErrorString(err));
    }
    *out = p;
    return SDR_OK;
}

int sdr_epl_plan_run_range(sdr_engine* e, sdr_epl_plan* p, int64_t first, int64_t count) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!p) return sdr_fail(SDR_ERR_INVALID, "plan is NULL");
    if (first < 0 || count <= 0 || first + count > p->n_items)
        return sdr_fail(SDR_ERR_RANGE, "item range [%lld, %lld) outside the plan's %d items", (long long)first,
                        (long long)(first + count), p->n_items);
    const sdr_epl_item* items = p->d_items + first;
    double* out = p->d_out + (size_t)first * 2 * p->n_taps;
    const int n = (int)count;
    {
        ProfScope ps(e, "epl_kernel");
        switch (e->iq_fmt) {
            case SDR_FMT_CI8: launch_fmt<SDR_FMT_CI8>(e, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, out); break;
            case SDR_FMT_CI16: launch_fmt<SDR_FMT_CI16>(e, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, out); break;
            case SDR_FMT_CF32: launch_fmt<SDR_FMT_CF32>(e, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, out); break;
            default: launch_fmt<SDR_FMT_CF64>(e, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, out); break;
        }
    }
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int sdr_epl_plan_run(sdr_engine* e, sdr_epl_plan* p) {
    if (!p) return sdr_fail(SDR_ERR_INVALID, "plan is NULL");
    return sdr_epl_plan_run_range(e, p, 0, p->n_items);
}

int sdr_epl_plan_fetch(sdr_engine* e, sdr_epl_plan* p, double* out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!p || !out) return sdr_fail(SDR_ERR_INVALID, "NULL plan or output");
    SDR_HIP(hipMemcpyAsync(out, p->d_out, (size_t)p->n_items * 2 * p->n_taps * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    return SDR_OK;
}

void sdr_epl_plan_destroy(sdr_engine* e, sdr_epl_plan* p) {
    if (!p) return;
    if (e) {
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->stream);
    }
    if (p->d_items) (void)hipFree(p->d_items);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->d_spacing) (void)hipFree(p->d_spacing);
    delete p;
}

int sdr_epl_batch(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing, int n_taps,
                  double fs, double* out) {
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    sdr_epl_plan* p = nullptr;
    if (int rc = sdr_epl_plan_create(e, items, n_items, spacing, n_taps, fs, &p)) return rc;
    int rc = sdr_epl_plan_run(e, p);
    if (!rc) rc = sdr_epl_plan_fetch(e, p, out);
    sdr_epl_plan_destroy(e, p);
    return rc;
}

}  // extern "C"
