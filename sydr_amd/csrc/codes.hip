// PRN replica generation on device: GPS L1 C/A Gold codes, nearest-chip
// upsampling, and the seeded synthetic multi-satellite IQ generator.
//
// Follows (restated, not translated):
//   Gold code   sydr/signal/ca.py:70-112  (G1 x^10+x^3+1, G2 x^10+x^9+x^8+x^6+x^3+x^2+1,
//               all-ones start, output = stage 10, code = G1 xor G2 delayed by g2_delay[prn],
//               bit 1 -> +1, bit 0 -> -1: ca.py:112)
//   Upsample    sydr/signal/gnsssignal.py:35-58 (idx = trunc((ts*k)/tc), ts = 1/fs, tc = 1/1.023e6)
#include "engine_internal.h"

// G2 delays in chips for PRN 1..210 (IS-GPS-200 Table 3-Ia/3-Ib; same data as ca.py:13-68).
__constant__ int16_t k_g2_delay[211] = {
       0,    5,    6,    7,    8,   17,   18,  139,  140,  141,  251,  252,
     254,  255,  256,  257,  258,  469,  470,  471,  472,  473,  474,  509,
     512,  513,  514,  515,  516,  859,  860,  861,  862,  863,  950,  947,
     948,  950,   67,  103,   91,   19,  679,  225,  625,  946,  638,  161,
    1001,  554,  280,  710,  709,  775,  864,  558,  220,  397,   55,  898,
     759,  367,  299, 1018,  729,  695,  780,  801,  788,  732,   34,  320,
     327,  389,  407,  525,  405,  221,  761,  260,  326,  955,  653,  699,
     422,  188,  438,  959,  539,  879,  677,  586,  153,  792,  814,  446,
     264, 1015,  278,  536,  819,  156,  957,  159,  712,  885,  461,  248,
     713,  126,  807,  279,  122,  197,  693,  632,  771,  467,  647,  203,
     145,  175,   52,   21,  237,  235,  886,  657,  634,  762,  355, 1012,
     176,  603,  130,  359,  595,   68,  386,  797,  456,  499,  883,  307,
     127,  211,  121,  118,  163,  628,  853,  484,  289,  811,  202, 1021,
     463,  568,  904,  670,  230,  911,  684,  309,  644,  932,   12,  314,
     891,  212,  185,  675,  503,  150,  395,  345,  846,  798,  992,  357,
     995,  877,  112,  144,  476,  193,  109,  445,  291,   87,  399,  292,
     901,  339,  208,  711,  189,  263,  537,  663,  942,  173,  900,   30,
     500,  935,  556,  373,   85,  652,  310,
};

// One workgroup per PRN.  Lane 0 clocks both 10-stage registers (kept as bit
// masks: bit s = stage s+1) and leaves the two m-sequences in LDS; every lane
// then combines G1[i] with G2[(i - delay) mod 1023].
__global__ __launch_bounds__(256) void gold_code_kernel(const int32_t* __restrict__ prns,
                                                        int8_t* __restrict__ out, int out_stride) {
    __shared__ uint8_t g1[SDR_GPS_L1CA_CHIPS];
    __shared__ uint8_t g2[SDR_GPS_L1CA_CHIPS];
    const int prn = prns[blockIdx.x];
    if (threadIdx.x == 0) {
        uint32_t r1 = 0x3FF, r2 = 0x3FF;
        for (int i = 0; i < SDR_GPS_L1CA_CHIPS; ++i) {
            g1[i] = (r1 >> 9) & 1u;
            g2[i] = (r2 >> 9) & 1u;
            uint32_t f1 = ((r1 >> 9) ^ (r1 >> 2)) & 1u;                                    // taps 10,3
            uint32_t f2 = ((r2 >> 9) ^ (r2 >> 8) ^ (r2 >> 7) ^ (r2 >> 5) ^ (r2 >> 2) ^ (r2 >> 1)) & 1u;  // 10,9,8,6,3,2
            r1 = ((r1 << 1) | f1) & 0x3FF;
            r2 = ((r2 << 1) | f2) & 0x3FF;
        }
    }
    __syncthreads();
    const int delay = k_g2_delay[prn];
    for (int i = threadIdx.x; i < SDR_GPS_L1CA_CHIPS; i += blockDim.x) {
        int j = i - delay;
        if (j < 0) j += SDR_GPS_L1CA_CHIPS;
        uint8_t bit = g1[i] ^ g2[j];
        out[(size_t)blockIdx.x * out_stride + i] = bit ? 1 : -1;
    }
}

// out[k] = code[trunc((ts*k)/tc)] with ts = fl(1/fs), tc = fl(1/1.023e6) (gnsssignal.py:46-56).
__global__ __launch_bounds__(256) void upsample_kernel(const int8_t* __restrict__ code, int n_chips,
                                                       double ts, double tc, int64_t n,
                                                       int8_t* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double v = (ts * (double)k) / tc;
    int idx = (int)trunc(v);
    if (idx >= n_chips) idx = n_chips - 1;
    out[k] = code[idx];
}

// Replica in LDS form: out[q] = high word of (double)chip[(q - PAD - 1) mod L], periodic over the whole
// row, so that multi-period epochs (4 ms of C/A code) and far-out taps index it without a modulo.
__global__ __launch_bounds__(256) void expand_lut_kernel(const int8_t* __restrict__ chips, int L,
                                                         uint32_t* __restrict__ out, int words) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= words) return;
    int c = (q - SDR_LUT_PAD - 1) % L;
    if (c < 0) c += L;
    out[q] = chips[c] > 0 ? 0x3FF00000u : 0xBFF00000u;
}

/* ------------------------------------------------ synthetic IQ generator */

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
__device__ __forceinline__ T quantise(float v);
template <>
__device__ __forceinline__ int8_t quantise<int8_t>(float v) {
    return (int8_t)fminf(fmaxf(rintf(v), -127.f), 127.f);
}
template <>
__device__ __forceinline__ int16_t quantise<int16_t>(float v) {
    return (int16_t)fminf(fmaxf(rintf(v), -32767.f), 32767.f);
}
template <>
__device__ __forceinline__ float quantise<float>(float v) { return v; }
template <>
__device__ __forceinline__ double quantise<double>(float v) { return (double)v; }

struct SynthSatDev {
    double cstep;     // chips per sample
    double code0;     // chips at ring sample 0
    double fcyc;      // carrier cycles per sample
    double phase0;    // cycles at sample 0
    float amp;
    int32_t id;          // seeds the data-bit stream
    int32_t code_off;    // first chip of this satellite's code in the packed chip buffer
    int32_t code_len;    // chips per code period
    int32_t boc;         // 1: multiply by the BOC(1,1) square sub-carrier (sign flips every half chip)
    int32_t bit_periods; // code periods per data bit
};

template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(T* __restrict__ ring, int64_t capacity,
                                                    const SynthSatDev* __restrict__ sats, int n_sats,
                                                    const int8_t* __restrict__ codes, float sigma,
                                                    uint64_t seed, int64_t first, int64_t count) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t n = first + k;
    float re = 0.f, im = 0.f;
    for (int s = 0; s < n_sats; ++s) {
        const SynthSatDev sat = sats[s];
        double chips = sat.code0 + (double)n * sat.cstep;
        double period = floor(chips / (double)sat.code_len);
        double in_period = chips - period * (double)sat.code_len;
        int chip = (int)floor(in_period);
        chip = chip < 0 ? 0 : (chip >= sat.code_len ? sat.code_len - 1 : chip);
        int64_t bit_index = (int64_t)floor(period / (double)sat.bit_periods);
        uint64_t h = mix64(seed ^ mix64((uint64_t)sat.id * 0x100000001B3ull + (uint64_t)bit_index));
        float sgn = (float)codes[sat.code_off + chip] * ((h & 1ull) ? 1.f : -1.f);
        if (sat.boc && (in_period - floor(in_period)) >= 0.5) sgn = -sgn;
        double cyc = sat.phase0 + (double)n * sat.fcyc;
        cyc -= rint(cyc);
        float sn, cs;
        sincospif(2.0f * (float)cyc, &sn, &cs);
        re += sat.amp * sgn * cs;
        im += sat.amp * sgn * sn;
    }
    if (sigma > 0.f) {
        uint64_t h = mix64(seed * 0xD6E8FEB86659FD93ull + (uint64_t)n);
        float u1 = ((h >> 40) + 1) * (1.0f / 16777217.0f);          // (0,1)
        float u2 = ((h >> 8) & 0xFFFFFFull) * (1.0f / 16777216.0f);  // [0,1)
        float r = sigma * sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        re += r * cs;
        im += r * sn;
    }
    int64_t slot = n % capacity;
    if constexpr (sizeof(T) == 1) {     // (a ci8 ring holds its bytes with the sign bit flipped: correlator.h kCi8Flip)
        ring[2 * slot] = (T)(quantise<T>(re) ^ (T)0x80);
        ring[2 * slot + 1] = (T)(quantise<T>(im) ^ (T)0x80);
    } else {
        ring[2 * slot] = quantise<T>(re);
        ring[2 * slot + 1] = quantise<T>(im);
    }
}

extern "C" {

int sdr_code_slots(sdr_engine* e, int n_slots, int max_chips) { return sdr_code_slots_ex(e, n_slots, max_chips, 1); }

int sdr_code_slots_ex(sdr_engine* e, int n_slots, int max_chips, int max_periods) {
    if (int rc = sdr_set_device(e)) return rc;
    if (n_slots <= 0 || max_chips < 1 || max_chips > 65536 || max_periods < 1 ||
        (int64_t)max_chips * max_periods > 32768)
        return sdr_fail(SDR_ERR_INVALID, "bad code slot geometry (%d slots, %d chips, %d periods; chips*periods <= 32768)",
                        n_slots, max_chips, max_periods);
    SDR_HIP(hipDeviceSynchronize());   // launches on any of the engine's streams may still read the old tables
    e->code_generation += 1;             // plans and bank entries made against the old tables are stale from here on
    if (e->codes) SDR_HIP(hipFree(e->codes));
    if (e->luts) SDR_HIP(hipFree(e->luts));
    e->luts = nullptr;
    if (e->luts2) SDR_HIP(hipFree(e->luts2));
    e->luts2 = nullptr;
    e->luts2_generation = -1;
    if (e->code_len) SDR_HIP(hipFree(e->code_len));
    e->codes = nullptr;
    e->code_len = nullptr;
    e->n_slots = 0;
    int stride = (max_chips + 15) & ~15;
    if (hipMalloc(&e->codes, (size_t)n_slots * stride) != hipSuccess)
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc for code slots failed");
    if (hipMalloc(&e->code_len, (size_t)n_slots * sizeof(int32_t)) != hipSuccess)
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc for code lengths failed");
    const int lut_stride = (stride * max_periods + 2 * SDR_LUT_PAD + 2 + 3) & ~3;
    if (hipMalloc(&e->luts, (size_t)n_slots * lut_stride * sizeof(uint32_t)) != hipSuccess)
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc for replica LUTs failed");
    e->lut_stride = lut_stride;
    SDR_HIP(hipMemsetAsync(e->codes, 0, (size_t)n_slots * stride, e->stream));
    SDR_HIP(hipMemsetAsync(e->code_len, 0, (size_t)n_slots * sizeof(int32_t), e->stream));
    e->code_len_host.assign(n_slots, 0);
    e->code_stamp.assign(n_slots, 0);
    e->n_slots = n_slots;
    e->code_stride = stride;
    return SDR_OK;
}

static int check_slot(sdr_engine* e, int slot) {
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated (call sdr_code_slots)");
    if (slot < 0 || slot >= e->n_slots)
        return sdr_fail(SDR_ERR_INVALID, "code slot %d outside [0,%d)", slot, e->n_slots);
    return SDR_OK;
}

int sdr_code_gps_l1ca(sdr_engine* e, int slot, int prn) {
    if (int rc = sdr_set_device(e)) return rc;
    if (int rc = check_slot(e, slot)) return rc;
    if (prn < 1 || prn > 210) return sdr_fail(SDR_ERR_INVALID, "PRN %d outside 1..210", prn);
    if (e->code_stride < SDR_GPS_L1CA_CHIPS)
        return sdr_fail(SDR_ERR_INVALID, "code slots hold %d chips < 1023", e->code_stride);
    // The PRN number rides in the (not yet valid) length word of the slot.
    int32_t len = SDR_GPS_L1CA_CHIPS;
    SDR_HIP(hipMemcpyAsync(e->code_len + slot, &prn, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    {
        ProfScope ps(e, "gold_code_kernel");
        hipLaunchKernelGGL(gold_code_kernel, dim3(1), dim3(256), 0, e->stream, e->code_len + slot,
                           e->codes + (size_t)slot * e->code_stride, e->code_stride);
    }
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipMemcpyAsync(e->code_len + slot, &len, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(expand_lut_kernel, dim3((e->lut_stride + 255) / 256), dim3(256), 0, e->stream,
                       e->codes + (size_t)slot * e->code_stride, len, e->luts + (size_t)slot * e->lut_stride,
                       e->lut_stride);
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipStreamSynchronize(e->stream));
    e->code_len_host[slot] = len;
    e->code_stamp[slot] = ++e->code_stamp_counter;   // (cached code spectra of this slot are stale now)
    return SDR_OK;
}

int sdr_code_custom(sdr_engine* e, int slot, const int8_t* chips, int n_chips) {
    if (int rc = sdr_set_device(e)) return rc;
    if (int rc = check_slot(e, slot)) return rc;
    if (!chips || n_chips < 16 || n_chips > e->code_stride)
        return sdr_fail(SDR_ERR_INVALID, "custom code of %d chips does not fit a %d-chip slot", n_chips,
                        e->code_stride);
    for (int i = 0; i < n_chips; ++i)
        if (chips[i] != 1 && chips[i] != -1)
            return sdr_fail(SDR_ERR_INVALID, "chip %d is %d, expected +-1", i, (int)chips[i]);
    int32_t len = n_chips;
    SDR_HIP(hipMemcpyAsync(e->codes + (size_t)slot * e->code_stride, chips, n_chips, hipMemcpyHostToDevice,
                           e->stream));
    SDR_HIP(hipMemcpyAsync(e->code_len + slot, &len, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(expand_lut_kernel, dim3((e->lut_stride + 255) / 256), dim3(256), 0, e->stream,
                       e->codes + (size_t)slot * e->code_stride, len, e->luts + (size_t)slot * e->lut_stride,
                       e->lut_stride);
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipStreamSynchronize(e->stream));
    e->code_len_host[slot] = len;
    e->code_stamp[slot] = ++e->code_stamp_counter;   // (cached code spectra of this slot are stale now)
    return SDR_OK;
}

int sdr_code_read(sdr_engine* e, int slot, int8_t* out_chips, int max_chips, int* n_chips) {
    if (int rc = sdr_set_device(e)) return rc;
    if (int rc = check_slot(e, slot)) return rc;
    int len = e->code_len_host[slot];
    if (len <= 0) return sdr_fail(SDR_ERR_STATE, "code slot %d is empty", slot);
    if (!out_chips || max_chips < len) return sdr_fail(SDR_ERR_INVALID, "output holds %d < %d chips", max_chips, len);
    SDR_HIP(hipMemcpyAsync(out_chips, e->codes + (size_t)slot * e->code_stride, len, hipMemcpyDeviceToHost,
                           e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    if (n_chips) *n_chips = len;
    return SDR_OK;
}

int sdr_code_upsample(sdr_engine* e, int slot, double fs, int64_t n_samples, int8_t* out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (int rc = check_slot(e, slot)) return rc;
    int len = e->code_len_host[slot];
    if (len <= 0) return sdr_fail(SDR_ERR_STATE, "code slot %d is empty", slot);
    if (!out || n_samples <= 0 || !(fs > 0.0)) return sdr_fail(SDR_ERR_INVALID, "bad upsample request");
    DevBuf tmp;
    if (int rc = sdr_devbuf_reserve(e, &tmp, (size_t)n_samples)) return rc;
    const double ts = 1.0 / fs, tc = 1.0 / 1.023e6;
    {
        ProfScope ps(e, "upsample_kernel");
        hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, e->stream,
                           e->codes + (size_t)slot * e->code_stride, len, ts, tc, n_samples, (int8_t*)tmp.ptr);
    }
    hipError_t err = hipGetLastError();
    if (err == hipSuccess)
        err = hipMemcpyAsync(out, tmp.ptr, (size_t)n_samples, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp.ptr);
    if (err != hipSuccess) return sdr_fail(SDR_ERR_HIP, "upsample failed: %s", hipGetErrorString(err));
    return SDR_OK;
}

int sdr_iq_synth(sdr_engine* e, const sdr_synth_sat* sats, int n_sats, double fs, double noise_sigma,
                 uint64_t seed, int64_t first_sample, int64_t n_samples) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (n_sats < 0 || (n_sats > 0 && !sats) || !(fs > 0.0) || n_samples < 0 || first_sample < 0)
        return sdr_fail(SDR_ERR_INVALID, "bad synth request");
    if (n_samples > e->iq_capacity)
        return sdr_fail(SDR_ERR_RANGE, "synth of %lld samples exceeds ring capacity %lld", (long long)n_samples,
                        (long long)e->iq_capacity);
    if (n_samples == 0) return SDR_OK;
    std::vector<SynthSatDev> host(n_sats > 0 ? n_sats : 1);
    size_t total_chips = 0;
    for (int s = 0; s < n_sats; ++s) {
        const bool from_slot = (sats[s].flags & SDR_SYNTH_CODE_SLOT) != 0;
        int len = SDR_GPS_L1CA_CHIPS;
        if (from_slot) {
            if (!e->codes || sats[s].prn < 0 || sats[s].prn >= e->n_slots || e->code_len_host[sats[s].prn] <= 0)
                return sdr_fail(SDR_ERR_INVALID, "synth satellite %d: code slot %d is not staged", s, sats[s].prn);
            len = e->code_len_host[sats[s].prn];
        } else if (sats[s].prn < 1 || sats[s].prn > 210) {
            return sdr_fail(SDR_ERR_INVALID, "synth PRN %d outside 1..210", sats[s].prn);
        }
        host[s].cstep = 1.023e6 * (1.0 + sats[s].doppler_hz / 1575.42e6) / fs;
        host[s].code0 = sats[s].code_phase;
        host[s].fcyc = sats[s].doppler_hz / fs;
        host[s].phase0 = sats[s].carrier_phase;
        host[s].amp = (float)sats[s].amplitude;
        host[s].id = sats[s].prn + (from_slot ? 1000 : 0);
        host[s].code_off = (int32_t)total_chips;
        host[s].code_len = len;
        host[s].boc = (sats[s].flags & SDR_SYNTH_BOC11) ? 1 : 0;
        host[s].bit_periods = len == SDR_GPS_L1CA_CHIPS ? 20 : 1;
        total_chips += (size_t)len;
    }
    DevBuf dsat, dprn, dcode;
    int rc = sdr_devbuf_reserve(e, &dsat, host.size() * sizeof(SynthSatDev));
    if (!rc) rc = sdr_devbuf_reserve(e, &dprn, sizeof(int32_t) * (n_sats > 0 ? n_sats : 1));
    if (!rc) rc = sdr_devbuf_reserve(e, &dcode, total_chips ? total_chips : 16);
    hipError_t err = hipSuccess;
    if (!rc) {
        err = hipMemcpyAsync(dsat.ptr, host.data(), host.size() * sizeof(SynthSatDev), hipMemcpyHostToDevice, e->stream);
        for (int s = 0; s < n_sats && err == hipSuccess; ++s) {
            int8_t* dst = (int8_t*)dcode.ptr + host[s].code_off;
            if (sats[s].flags & SDR_SYNTH_CODE_SLOT) {
                err = hipMemcpyAsync(dst, e->codes + (size_t)sats[s].prn * e->code_stride, host[s].code_len,
                                     hipMemcpyDeviceToDevice, e->stream);
            } else {
                int32_t* dp = (int32_t*)dprn.ptr + s;
                err = hipMemcpyAsync(dp, &sats[s].prn, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
                if (err == hipSuccess) {
                    hipLaunchKernelGGL(gold_code_kernel, dim3(1), dim3(256), 0, e->stream, (const int32_t*)dp, dst,
                                       SDR_GPS_L1CA_CHIPS);
                    err = hipGetLastError();
                }
            }
        }
        if (err == hipSuccess) {
            ProfScope ps(e, "synth_kernel");
            dim3 grid((unsigned)((n_samples + 255) / 256));
            const SynthSatDev* ds = (const SynthSatDev*)dsat.ptr;
            const int8_t* dc = (const int8_t*)dcode.ptr;
            float sg = (float)noise_sigma;
            sdr_iq_mark_written(e, first_sample, n_samples);
            switch (e->iq_fmt) {
                case SDR_FMT_CI8:
                    hipLaunchKernelGGL(synth_kernel<int8_t>, grid, dim3(256), 0, e->stream, (int8_t*)e->iq,
                                       e->iq_capacity, ds, n_sats, dc, sg, seed, first_sample, n_samples);
                    break;
                case SDR_FMT_CI16:
                    hipLaunchKernelGGL(synth_kernel<int16_t>, grid, dim3(256), 0, e->stream, (int16_t*)e->iq,
                                       e->iq_capacity, ds, n_sats, dc, sg, seed, first_sample, n_samples);
                    break;
                case SDR_FMT_CF32:
                    hipLaunchKernelGGL(synth_kernel<float>, grid, dim3(256), 0, e->stream, (float*)e->iq,
                                       e->iq_capacity, ds, n_sats, dc, sg, seed, first_sample, n_samples);
                    break;
                default:
                    hipLaunchKernelGGL(synth_kernel<double>, grid, dim3(256), 0, e->stream, (double*)e->iq,
                                       e->iq_capacity, ds, n_sats, dc, sg, seed, first_sample, n_samples);
                    break;
            }
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    }
    if (dsat.ptr) (void)hipFree(dsat.ptr);
    if (dprn.ptr) (void)hipFree(dprn.ptr);
    if (dcode.ptr) (void)hipFree(dcode.ptr);
    if (rc) return rc;
    if (err != hipSuccess) return sdr_fail(SDR_ERR_HIP, "synth failed: %s", hipGetErrorString(err));
    return SDR_OK;
}

}  // extern "C"
