// K1: batched carrier wipe-off + multi-tap PRN correlate-accumulate.
//
// One 64-lane WAVE = one channel-epoch = one call of the reference's EPL (sydr/dsp/tracking.py:92-116); a workgroup
// is one wave (or four waves -- four epochs of one channel -- when the staged replica is 16 KB and more).  IQ is
// streamed from the HBM ring with 16-byte loads, the PRN replica sits in LDS as the high words of +-1.0, accumulators
// are fp64 and are reduced inside the wave with DPP moves + v_readlane in a fixed order (so results do not depend on
// how channels are sharded over GPUs).  Three correlator cores, chosen per launch from the items' code steps:
//   chip-aligned (correlator_chip.h) : ci8, 16-26 samples per chip -- a lane owns a whole chip of the prompt tap; with the
//                                      block length (24/25 samples) and the outer taps' switch position (12/13) compiled
//                                      in when every epoch of the launch has them; 32-52 samples per chip through the
//                                      half-chip view of the replicas (every chip twice, doubled NCO parameters)
//   boundary     (correlator.h)      : a lane owns 16 (8) consecutive samples, < 1 chip; running sums in an LDS strip
//   per-sample   (correlator.h)      : exact chip index per sample and tap; low rates, epochs that wrap the ring
//
// Arithmetic that selects a chip is the reference's, operation for operation (np.linspace + np.ceil),
// in IEEE fp64 with contraction off:
//     shift = rem_code + spacing            stop = code_step*n + shift
//     step  = (stop - shift) / n            idx_i = ceil(i*step + shift)
// The carrier replica exp(1j*(-(f*2.0*pi*(i/fs)) + rem)) is evaluated once per lane in fp64 (exact range reduction +
// minimax sincos) and advanced by precomputed fp64 rotations inside a block and from block to block, which agrees
// with the reference to ~1e-15 relative on the accumulators (the bar is 1e-6).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "correlator.h"
#include "correlator_chip.h"
#include "correlator_chip2.h"
#include "epl_items.h"

#ifdef SDR_TRACE_WG
// Debug build only (tools/wg_trace.py): per-workgroup start/end clock and hardware id.
__device__ unsigned long long g_wg_trace[3 * 65536];
extern "C" __attribute__((visibility("default"))) int sdr_debug_read_trace(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_trace), (size_t)n * 3 * sizeof(unsigned long long));
}
#endif

#include "epl_kernel.h"

namespace {

using namespace sdr;

// Half-chip view of the staged replicas: every chip twice.  A code of 32-52 samples per chip (GPS L1 C/A at 50 MHz)
// presented as a code of twice the chips at twice the rate has 16-26 samples per (half-)chip and runs on the
// chip-aligned correlator.  Exact: the kernel is handed 2*rem_code, 2*code_step and 2*spacing -- scaling by two
// commutes with every fp64 rounding of the reference's index expression, so its half-chip index is ceil(2y) for the
// reference's own y, and ceil(ceil(2y) / 2) = ceil(y) is the chip:  lut2[h + PAD] = lut[((h + 1) >> 1) + PAD].
__global__ __launch_bounds__(256) void double_lut_kernel(const uint32_t* __restrict__ luts, int stride, uint32_t* __restrict__ luts2,
                                                         int stride2) {
    const int slot = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= stride2) return;
    int src = ((j - SDR_LUT_PAD + 1) >> 1) + SDR_LUT_PAD;
    src = src < stride - 1 ? src : stride - 1;
    luts2[(size_t)slot * stride2 + j] = luts[(size_t)slot * stride + src];
}

int ensure_doubled_luts(sdr_engine* e, hipStream_t stream) {
    if (e->luts2 && e->luts2_generation == e->code_generation && e->luts2_stamp == e->code_stamp_counter) return SDR_OK;
    if (!e->luts2 || e->luts2_generation != e->code_generation) {
        if (e->luts2) SDR_HIP(hipFree(e->luts2));
        e->luts2 = nullptr;
        e->lut2_stride = (2 * e->lut_stride + 3) & ~3;
        if (hipMalloc(&e->luts2, (size_t)e->n_slots * e->lut2_stride * sizeof(uint32_t)) != hipSuccess)
            return sdr_fail(SDR_ERR_NOMEM, "hipMalloc for the half-chip replica tables failed");
    }
    // (on the stream the staging kernels use -- they are ordered before this -- and complete before any batch stream reads it)
    hipLaunchKernelGGL(double_lut_kernel, dim3((e->lut2_stride + 255) / 256, e->n_slots), dim3(256), 0, e->stream, e->luts,
                       e->lut_stride, e->luts2, e->lut2_stride);
    SDR_HIP(hipGetLastError());
    if (stream != e->stream) SDR_HIP(hipStreamSynchronize(e->stream));
    e->luts2_generation = e->code_generation;
    e->luts2_stamp = e->code_stamp_counter;
    return SDR_OK;
}

#ifndef SDR_EPL2_WAVES
#define SDR_EPL2_WAVES 3   // (a cap of 128 registers for four waves per SIMD spills into scratch: 0.296 instead of 0.192 ms per 32 000 epochs)
#endif
// Several chips per lane (correlator_chip2.h): one wave per item, three taps, ci8 ring; the item's setup comes from the plan.
// Dynamic LDS: [8 zero words][lut: lut_words uint32].
template <int... P>
__global__ __launch_bounds__(kWaveThreads, (sizeof...(P) > 4 ? 2 : SDR_EPL2_WAVES)) void epl2_kernel(const void* __restrict__ ring, const void* __restrict__ ring_flipped,
                                                            int64_t capacity, const sdr_epl_item* __restrict__ items, int n_items,
                                                            const uint32_t* __restrict__ luts, int lut_words, int lut_stride,
                                                            const double* __restrict__ spacing, double fs, double* __restrict__ out,
                                                            const ChipNSetup<P...>* __restrict__ setups) {
    extern __shared__ double smem[];
    uint32_t* zero_words = reinterpret_cast<uint32_t*>(smem);
    uint32_t* lut = zero_words + 8;
    const int lane = threadIdx.x;
    const int item = blockIdx.x;
    const sdr_epl_item it = items[item];
    EpochParams ep;
    ep.start_sample = it.start_sample;
    ep.n = it.n_samples;
    ep.carrier_hz = it.carrier_hz;
    ep.rem_carrier = it.rem_carrier;
    ep.rem_code = it.rem_code;
    ep.code_step = it.code_step;
    const ChipNSetup<P...>& S = setups[item];
    stage_lut<kWaveThreads>(lut, luts + (size_t)it.code_slot * lut_stride, lut_words, lane);
    if (lane < 8) zero_words[lane] = 0u;
    __syncthreads();  // replica staged
    double accr[3], acci[3];
    const bool done = S.base >= 0 && correlate_epoch_chipn<P...>(ring, ring_flipped, ep, S, lut, zero_words, lane, accr, acci);
    if (!done) {     // an epoch the scheme does not cover: per sample
        const double dphi = carrier_step(it.carrier_hz, fs);
        EpochConsts<3> K2;
        compute_constants<3>(K2, ep, spacing, dphi, kWaveThreads);
        correlate_epoch<SDR_FMT_CI8, 3>(ring, capacity, ep, dphi, K2, lut, lane, kWaveThreads, lane, accr, acci);
    }
    int slot;
    const double total = reduce_taps_scatter<3>(accr, acci, lane, slot);
    if (lane < 8) out[(size_t)item * 6 + slot] = total;
}

template <int FMT, int NT>
void launch_one(sdr_engine* e, hipStream_t stream, const sdr_epl_item* d_items, int n_items, const double* d_spacing, double fs,
                int tap0, int n_taps_total, int lut_words, int wide, int group_stride, bool doubled, double* d_out,
                const void* d_setups) {
    // long replicas: four waves (four epochs of one channel) per workgroup around one staged table
    const int wpw = (lut_words >= kLongLutWords && group_stride > 0) ? 4 : 1;
    const int threads = kWaveThreads * wpw;
    const size_t scratch = wide >= kChipMax ? (size_t)threads * chip_strip_slots<NT>() + wpw * kChipRotSlots
                                            : (wide ? (size_t)threads * kPrefixSlots : 0);
    size_t shmem = (size_t)(wpw * 2 * NT) * sizeof(double) + (size_t)((lut_words + 3) & ~3) * sizeof(uint32_t) +
                   scratch * sizeof(double2);
    const int stride = wpw > 1 ? group_stride : 1;
    const int grid = wpw > 1 ? (int)(((int64_t)n_items + (int64_t)wpw * stride - 1) / ((int64_t)wpw * stride)) * stride : n_items;
    auto launch = [&](auto kernel) {
        if (shmem > 64u * 1024u)  // beyond the default dynamic-LDS grant (long multi-period replicas)
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), shmem, stream, e->iq, (const void*)e->iq, e->iq_capacity, d_items, n_items, stride,
                           doubled ? e->luts2 : e->luts, lut_words, doubled ? e->lut2_stride : e->lut_stride, d_spacing, fs, tap0,
                           n_taps_total, d_out, d_setups);
    };
    constexpr bool kOdd = NT == 3 || NT == 5;               // (the compile-time tap geometries exist for these)
    // (the straight-line kernels read the plan's per-item setups: without them the run-time-position kernel serves the list)
    const bool ki = (wide & kVariantKI) != 0 && FMT == SDR_FMT_CI8 && kOdd && d_setups != nullptr;
    const int ks_of = FMT == SDR_FMT_CI8 && NT == 3 && d_setups != nullptr ? (wide & kVariantKSMask) / 256 : 0;
    const bool ks = ks_of != 0;
    wide &= 255;
    if (wpw == 4) {
        if (ki && wide == kChipMax + 24)                     // taps whole (half-)chips apart: configs 4-5
            launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 24, 4, 0, (kOdd ? 1 : 0)>);
        else if (wide == kChipMax + 24 && FMT == SDR_FMT_CI8)     // (every epoch with 24 or 25 samples per chip: BOC(1,1) half-chips at 50 MHz)
            launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 24, 4>);
        else if (wide >= kChipMax && FMT == SDR_FMT_CI8)
            launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 0, 4>);
        else if (wide == 16)
            launch(epl_kernel<FMT, NT, 16, 0, 4>);
        else if (wide == 8)
            launch(epl_kernel<FMT, NT, 8, 0, 4>);
        else
            launch(epl_kernel<FMT, NT, 0, 0, 4>);
        return;
    }
    // the straight-line kernels of the other block lengths (ci8, three taps) live in epl_straight.hip: launched through their address
    auto launch_by_address = [&](const void* kernel) {
        const void* ring = e->iq;
        const void* flipped = (const void*)e->iq;
        int64_t cap = e->iq_capacity;
        int n = n_items, gs = stride, lw = lut_words, ls = doubled ? e->lut2_stride : e->lut_stride, t0 = tap0, nt = n_taps_total;
        const uint32_t* luts = doubled ? e->luts2 : e->luts;
        void* args[] = {&ring, &flipped, &cap, &d_items, &n, &gs, &luts, &lw, &ls, &d_spacing, &fs, &t0, &nt, &d_out, &d_setups};
        if (shmem > 64u * 1024u) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        (void)hipLaunchKernel(kernel, dim3(grid), dim3(threads), args, shmem, stream);
    };
    const int km = wide - kChipMax;
    const void* elsewhere = (ks && km != 24 && km != 19)                    ? sdr_epl_ks_kernel(km)
                            : (ki && NT == 3 && km != 24 && km != 15)       ? sdr_epl_ki_kernel(km)
                            : (ki && NT == 5 && km != 24)                   ? sdr_epl_ki5_kernel(km)
                            : (!ks && !ki && FMT == SDR_FMT_CI8 && NT == 3 && km >= 17 && km != 24) ? sdr_epl_km_kernel(km)
                                                                            : nullptr;
    if (elsewhere)
        launch_by_address(elsewhere);
    else if (ks && ks_of == 9)                               // 19 / 20 samples per chip, the outer taps switching at sample 9 or 10
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), (NT == 3 ? 19 : 0), 1, (NT == 3 ? 9 : 0)>);
    else if (ks)                                             // 24 / 25 samples per chip, both outer taps switching at sample 12 or 13
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 24, 1, (NT == 3 ? 12 : 0)>);
    else if (ki && wide == kChipMax + 15)                    // 15 / 16 samples per half chip (31-32.7 MHz on the half-chip view)
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), (NT == 3 ? 15 : 24), 1, 0, (kOdd ? 1 : 0)>);
    else if (ki)
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 24, 1, 0, (kOdd ? 1 : 0)>);
    else if (wide == kChipMax + 24 && FMT == SDR_FMT_CI8)   // chip-aligned, every epoch with 24 or 25 samples per chip
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16), 24>);
    else if (wide == kChipMax + kVariantKM1516 && FMT == SDR_FMT_CI8 && NT == 3) {   // 15.x or 16.x samples per chip, epoch by epoch
        if constexpr (FMT == SDR_FMT_CI8 && NT == 3) launch(epl_kernel<FMT, NT, kChipMax, 15, 1, 0, 0, 16>);
    }
    else if (wide >= kChipMax && FMT == SDR_FMT_CI8)
        launch(epl_kernel<FMT, NT, (FMT == SDR_FMT_CI8 ? kChipMax : 16)>);
    else if (wide == 16)
        launch(epl_kernel<FMT, NT, 16>);
    else if (wide == 8)
        launch(epl_kernel<FMT, NT, 8>);
    else
        launch(epl_kernel<FMT, NT, 0>);
}

template <int FMT>
void launch_fmt(sdr_engine* e, hipStream_t stream, const sdr_epl_item* d_items, int n_items, const double* d_spacing, double fs,
                int n_taps, int lut_words, int wide, int group_stride, bool doubled, double* d_out, const void* d_setups) {
    // Taps are served in register-resident chunks of 5/3/2/1.
    int t0 = 0;
    while (t0 < n_taps) {
        int left = n_taps - t0;
        if (left >= 5) {
            launch_one<FMT, 5>(e, stream, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, wide, group_stride, doubled, d_out, d_setups);
            t0 += 5;
        } else if (left >= 3) {
            launch_one<FMT, 3>(e, stream, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, wide, group_stride, doubled, d_out, d_setups);
            t0 += 3;
        } else if (left == 2) {
            launch_one<FMT, 2>(e, stream, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, wide, group_stride, doubled, d_out, d_setups);
            t0 += 2;
        } else {
            launch_one<FMT, 1>(e, stream, d_items, n_items, d_spacing, fs, t0, n_taps, lut_words, wide, group_stride, doubled, d_out, d_setups);
            t0 += 1;
        }
    }
}

}  // namespace

struct sdr_epl_plan {
    sdr_epl_item* d_items = nullptr;
    double* d_out = nullptr;
    double* d_spacing = nullptr;
    char* d_setups = nullptr;   // straight-line kernels: one ChipSetup<n_taps> / ChipNSetup per item (correlator_chip.h, correlator_chip2.h)
    size_t setup_bytes = 0;     // sizeof(ChipSetup<n_taps>), 0: none
    size_t bytes_items = 0, bytes_out = 0, bytes_spacing = 0, bytes_setups = 0;   // what the four allocations hold (plan_give)
    int n_items = 0;
    int n_taps = 0;
    int lut_words = 0;
    int wide = 0;  // 16 / 8: every item has 16 (8) * code_step < 1, the boundary variant with that group width applies
    int group_stride = 0;  // C > 0: items[i] and items[i + C] use the same code slot for every i (epoch-major lists of C channels)
    bool borrowed = false;  // device buffers are the engine's workspaces (sdr_epl_batch): not freed with the plan
    bool doubled = false;  // the plan runs on the half-chip view: its device items / spacings carry 2*rem_code, 2*code_step, 2*spacing
    double fs = 0.0;
    int64_t code_generation = 0;  // of the engine's code tables the plan was validated against
    int64_t ring_capacity = 0;    // and of the ring
    // one event per stream a range of the plan was launched on, re-recorded behind every launch: fetch waits for
    // exactly these instead of the whole device
    std::vector<std::pair<hipStream_t, hipEvent_t>> ran_on;
};

// The per-epoch setups of the straight-line kernels (ChipSetup / ChipNSetup: tap constants, epoch geometry, ring position,
// carrier rotations), made when a plan is created: one THREAD per item -- what every wave of an epoch would otherwise
// repeat in all of its 64 lanes -- with the same functions the run-time-position kernels call per wave.  1.92 M items
// (60 s x 32 channels) take ~0.1 ms of one launch.
template <int NT, int KM, int KS, int KI>
__global__ __launch_bounds__(64) void chip_setup_kernel(const sdr_epl_item* __restrict__ items, int n_items,
                                                        const double* __restrict__ spacing, double fs, int64_t capacity,
                                                        sdr::ChipSetup<NT>* __restrict__ out) {
    // One wave per 64 items.  A thread's setup is 480 / 560 bytes: stored where it belongs by the thread itself, a wave's store
    // instruction touches 64 places that far apart (1.4 ms for the 0.92 GB of a 60 s x 32 channel plan); through LDS the
    // wave writes its 64 setups -- contiguous in memory -- sixteen bytes per lane, a kilobyte per instruction (round 6).
    constexpr int kWords = (int)(sizeof(sdr::ChipSetup<NT>) / 16);
    static_assert(sizeof(sdr::ChipSetup<NT>) % 16 == 0, "setups are copied out in 16-byte granules");
    __shared__ uint4 stage[64 * kWords];
    const int i0 = blockIdx.x * 64, i = i0 + threadIdx.x;
    if (i < n_items) {
        const sdr_epl_item it = items[i];
        double sp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) sp[t] = spacing[t];
        sdr::ChipSetup<NT> S;
        sdr::chip_setup<NT, KM, KS, KI>(it.n_samples, it.start_sample, capacity, it.carrier_hz, it.rem_code, it.code_step, sp, fs,
                                    kWaveThreads, S);
        // (granule w of every lane side by side: lane-major slots would put 64 lanes 30 granules apart on the same banks)
        const uint4* const src = reinterpret_cast<const uint4*>(&S);
#pragma unroll
        for (int w = 0; w < kWords; ++w) stage[w * 64 + threadIdx.x] = src[w];
    }
    __syncthreads();
    const int n_here = n_items - i0 < 64 ? n_items - i0 : 64;
    uint4* const dst = reinterpret_cast<uint4*>(out + i0);
    for (int g = threadIdx.x; g < n_here * kWords; g += 64) {
        const int item = g / kWords, w = g - item * kWords;
        dst[g] = stage[w * 64 + item];
    }
}
// ... *missed counts the items the several-chips-per-lane scheme does not cover (their setups say so, and the kernel
// redoes them per sample).
template <int... P>
__global__ __launch_bounds__(256) void chipn_setup_kernel(const sdr_epl_item* __restrict__ items, int n_items,
                                                          const double* __restrict__ spacing, double fs, int64_t capacity,
                                                          sdr::ChipNSetup<P...>* __restrict__ out, int* __restrict__ missed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    const sdr_epl_item it = items[i];
    const double sp[3] = {spacing[0], spacing[1], spacing[2]};
    sdr::ChipNSetup<P...> S;
    if (!sdr::chipn_setup<P...>(it.n_samples, it.start_sample, capacity, it.carrier_hz, it.rem_code, it.code_step, sp, fs, S))
        atomicAdd(missed, 1);
    out[i] = S;
}

// ---- validation of a list: no item may index outside the ring or the staged replica, and what the list as a whole
// allows (the kernel variant) follows from every item.  One function per item for both sides: the host walks short lists
// (one call per EPL(): latency), one THREAD per item checks long ones behind their upload (1.92 M items of a 60 s x 32
// channel plan took the host 20-30 ms of the 34 ms a plan cost; the launch takes ~0.1 ms).
__device__ __forceinline__ unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }

// Two rule sets at once (the list as it is and its half-chip view): one upload, one launch, one read-back.  A thread walks
// the list with the grid's stride and keeps its own statistics; a wave reduces them and adds them to the list's with ONE set
// of atomics -- at 1.92 M items a set per 64 items was 420 k atomics on seven addresses, 2.2 ms of a 4.5 ms plan (round 6).
__global__ __launch_bounds__(256) void validate_items_kernel(const sdr_epl_item* __restrict__ items, int n_items, ItemRules r1,
                                                             ItemRules r2, const int32_t* __restrict__ code_len,
                                                             ItemStats* __restrict__ out) {
    const int stride = (int)(gridDim.x * 256);
    // (a lane without an item, or with a bad one, carries the neutral element of every reduction: all 64 lanes take part)
    int ml[2] = {0, 0}, mlo[2] = {0x7fffffff, 0x7fffffff}, mhi[2] = {0, 0}, a12[2] = {1, 1}, fb[2] = {0x7fffffff, 0x7fffffff};
    unsigned long long mx[2] = {0ull, 0ull}, mn[2] = {~0ull, ~0ull};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_items; i += stride) {
        const sdr_epl_item it = items[i];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const ItemRules& r = v ? r2 : r1;
            int maxlen = 0, m_chip = 0;
            double step = 0.0, lo, hi;
            bool split = true;
            const int bad = check_item(it, r, code_len, maxlen, step, m_chip, split, lo, hi);
            if (bad) {
                fb[v] = min(fb[v], i);
            } else {
                ml[v] = max(ml[v], maxlen);
                const unsigned long long sb = dbits(step);
                mx[v] = sb > mx[v] ? sb : mx[v];
                mn[v] = sb < mn[v] ? sb : mn[v];
                mlo[v] = min(mlo[v], m_chip);
                mhi[v] = max(mhi[v], m_chip);
                a12[v] &= split ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ml[v] = max(ml[v], __shfl_xor(ml[v], off, 64));
            const unsigned long long ox = __shfl_xor(mx[v], off, 64), on = __shfl_xor(mn[v], off, 64);
            mx[v] = ox > mx[v] ? ox : mx[v];
            mn[v] = on < mn[v] ? on : mn[v];
            mlo[v] = min(mlo[v], __shfl_xor(mlo[v], off, 64));
            mhi[v] = max(mhi[v], __shfl_xor(mhi[v], off, 64));
            a12[v] &= __shfl_xor(a12[v], off, 64);
            fb[v] = min(fb[v], __shfl_xor(fb[v], off, 64));
        }
    }
    // the block's four waves through LDS, then one set of atomics per block and rule set
    __shared__ int sh_i[4][2][5];
    __shared__ unsigned long long sh_u[4][2][2];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            sh_i[wave][v][0] = ml[v], sh_i[wave][v][1] = mlo[v], sh_i[wave][v][2] = mhi[v], sh_i[wave][v][3] = a12[v], sh_i[wave][v][4] = fb[v];
            sh_u[wave][v][0] = mx[v], sh_u[wave][v][1] = mn[v];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int v = threadIdx.x;
        int b_ml = 0, b_lo = 0x7fffffff, b_hi = 0, b_a12 = 1, b_fb = 0x7fffffff;
        unsigned long long b_mx = 0ull, b_mn = ~0ull;
        for (int w = 0; w < 4; ++w) {
            b_ml = max(b_ml, sh_i[w][v][0]);
            b_lo = min(b_lo, sh_i[w][v][1]);
            b_hi = max(b_hi, sh_i[w][v][2]);
            b_a12 &= sh_i[w][v][3];
            b_fb = min(b_fb, sh_i[w][v][4]);
            b_mx = sh_u[w][v][0] > b_mx ? sh_u[w][v][0] : b_mx;
            b_mn = sh_u[w][v][1] < b_mn ? sh_u[w][v][1] : b_mn;
        }
        ItemStats* o = out + v;
        atomicMax(&o->maxlen, b_ml);
        atomicMax(&o->max_step_bits, b_mx);
        atomicMin(&o->min_step_bits, b_mn);
        atomicMin(&o->m_lo, b_lo);
        atomicMax(&o->m_hi, b_hi);
        if (!b_a12) atomicAnd(&o->all_split, 0);
        if (b_fb != 0x7fffffff) atomicMin(&o->first_bad, b_fb);
    }
}

static ItemRules item_rules(const sdr_engine* e, const double* spacing, int n_taps, double scale) {
    ItemRules r = {};
    r.n_slots = e->n_slots;
    r.lut_stride = e->lut_stride;
    r.n_taps = n_taps;
    r.iq_capacity = e->iq_capacity;
    r.scale = scale;
    r.smin = r.smax = spacing[0];
    for (int t = 1; t < n_taps; ++t) {
        r.smin = spacing[t] < r.smin ? spacing[t] : r.smin;
        r.smax = spacing[t] > r.smax ? spacing[t] : r.smax;
    }
    r.s_anchor = scale * spacing[n_taps < 5 ? n_taps / 2 : 2];   // centre tap of the first chunk of taps the kernels serve together
    r.sp0 = spacing[0];
    r.sp2 = n_taps >= 3 ? spacing[2] : 0.0;
    r.want_s12 = n_taps == 3;   // both outer taps switch chips 12.x samples into the anchor's block (KS = 12 kernel)
    return r;
}

static int item_error(const sdr_engine* e, const sdr_epl_item& it, int index, int code, double lo, double hi) {
    switch (code) {
        case ITEM_SLOT: return sdr_fail(SDR_ERR_INVALID, "item %d: code slot %d is not staged", index, it.code_slot);
        case ITEM_SAMPLES: return sdr_fail(SDR_ERR_RANGE, "item %d: n_samples %d outside (0, ring capacity]", index, it.n_samples);
        case ITEM_START: return sdr_fail(SDR_ERR_RANGE, "item %d: negative start_sample", index);
        case ITEM_NCO: return sdr_fail(SDR_ERR_INVALID, "item %d: non-finite or non-positive NCO parameter", index);
        default:
            return sdr_fail(SDR_ERR_RANGE,
                            "item %d: code phase range [%g, %g] leaves the staged replica [-%d, %d] "
                            "(stage more code periods with sdr_code_slots_ex)", index, lo, hi, SDR_LUT_PAD,
                            e->lut_stride - SDR_LUT_PAD - 2);
    }
}

// The kernel variant a list of these statistics gets (0: per sample; 8 / 16: boundary variants; kChipMax + ...: chip-aligned).
// m_lo / m_hi: the list's range of whole samples per chip; all_split: ItemStats::all_split.
static int variant_of(const sdr_engine* e, const ItemRules& r, const double* spacing, double max_step, double min_step, int m_lo, int m_hi,
                      bool all_split) {
    bool all_ki = r.n_taps == 3 || r.n_taps == 5;   // tap t exactly (t - A) chips from the anchor (KI kernel)
    for (int t = 0; all_ki && t < r.n_taps; ++t) all_ki = r.scale * spacing[t] - r.s_anchor == (double)(t - r.n_taps / 2);
    const bool all_m24 = m_lo == 24 && m_hi == 24;
    const bool all_m15 = m_lo == 15 && m_hi == 15;                      // (31-32.7 MHz on the half-chip view: KM = 15, whole-chip taps)
    const bool all_m1516 = m_lo >= 15 && m_hi <= 16 && r.n_taps == 3 && r.scale == 1.0;   // (16.368 MHz: 16.0 -- 15.x or 16.x by the Doppler's sign)
    // one block length KM throughout, three taps: the straight-line kernels exist for KM = 16 .. 25 (epl.hip's own: 24 / 12,
    // 19 / 9, the whole-chip-tap forms of 24 and 15; the rest: epl_straight.hip) -- with the outer taps' switch position
    // floor(KM / 2) compiled in (+-0.5 chip, `all_split`) or with taps whole (half-)chips apart (`all_ki`)
    const int km_one = m_lo == m_hi && m_lo >= 16 && m_lo <= 25 && r.n_taps == 3 ? m_lo : 0;
    const bool boundary_ok = min_step >= sdr::kFastMinCodeStep && r.scale * e->lut_stride < sdr::kFastMaxLutWords;
    int wide = !boundary_ok ? 0 : (max_step <= sdr::kFastMaxCodeStep ? 16 : (max_step <= sdr::kFastMaxCodeStep8 ? 8 : 0));
    // every item inside the chip-aligned variant's range (ci8 ring): lanes own whole chips instead of 16 samples
    // (the half-chip view only where a half chip holds 16 samples or more: at 31-32 MHz -- 15.6 per half chip -- the
    // 16-sample boundary groups of the plain list were measured faster, 0.47 against 0.37 of the roof)
    // ... unless every half chip holds 15.x samples and the taps sit whole half-chips apart: the KM = 15 whole-chip-tap kernel
    const double chip_max_step = r.scale == 2.0 && !(all_m15 && all_ki && r.n_taps == 3 && !e->epl_no_split) ? 1.0 / 16.0 : sdr::kChipMaxCodeStep;
    if (boundary_ok && e->iq_fmt == SDR_FMT_CI8 && min_step >= sdr::kChipMinCodeStep && max_step <= chip_max_step &&
        !e->epl_no_chip) {
        wide = sdr::kChipMax;
        const int km5 = m_lo == m_hi && m_lo >= 16 && m_lo <= 25 && r.n_taps == 5 ? m_lo : 0;
        if (km5 && all_ki && !e->epl_no_split)              // five taps whole (half-)chips apart
            wide += km5 + kVariantKI;
        else if (all_m24 && r.n_taps != 3)                  // (other tap counts: the 24 / 25 form with run-time positions alone)
            wide += 24;
        else if (km_one && all_split && !e->epl_no_split)
            wide += km_one + 256 * (km_one / 2);
        else if (km_one && all_ki && !e->epl_no_split)
            wide += km_one + kVariantKI;
        else if (all_m24 || (km_one >= 17 && !e->epl_no_split))   // block length compiled in, tap positions at run time
            wide += all_m24 ? 24 : km_one;
        else if (all_m15 && all_ki && r.n_taps == 3 && r.scale == 2.0 && !e->epl_no_split)
            wide += 15 + kVariantKI;
        else if (all_m1516 && !e->epl_no_split)
            wide += kVariantKM1516;
    }
    return wide;
}

// Host walk (short lists; and the one bad item of a long list, for its message: index0 = its place in the list).
static int validate_items(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing,
                          int n_taps, double scale, int* lut_words, int* wide, int index0 = 0) {
    const ItemRules r = item_rules(e, spacing, n_taps, scale);
    int maxlen = 0;
    double max_step = 0.0, min_step = 1e300;
    int m_lo = 0x7fffffff, m_hi = 0;
    bool all_split = r.want_s12;
    for (int i = 0; i < n_items; ++i) {
        int ml = 0;
        double step = 0.0, lo = 0.0, hi = 0.0;
        int m_chip = 0;
        bool split = true;
        if (const int bad = check_item(items[i], r, e->code_len_host.data(), ml, step, m_chip, split, lo, hi))
            return item_error(e, items[i], index0 + i, bad, lo, hi);
        m_lo = m_chip < m_lo ? m_chip : m_lo;
        m_hi = m_chip > m_hi ? m_chip : m_hi;
        all_split = all_split && split;
        if (step > max_step) max_step = step;
        if (step < min_step) min_step = step;
        if (ml > maxlen) maxlen = ml;
    }
    *lut_words = maxlen + SDR_LUT_PAD + 2;
    *wide = variant_of(e, r, spacing, max_step, min_step, m_lo, m_hi, all_split);
    return SDR_OK;
}

// The same for a list that is already on the device (long lists: behind their upload).  ok2: the half-chip view's rules
// were met as well (lw2 / wide2 then hold its answers).
static int validate_items_dev(sdr_engine* e, const sdr_epl_item* d_items, const sdr_epl_item* h_items, int n_items,
                              const double* spacing, int n_taps, int* lut_words, int* wide, bool* ok2, int* lw2, int* wide2) {
    const ItemRules r1 = item_rules(e, spacing, n_taps, 1.0), r2 = item_rules(e, spacing, n_taps, 2.0);
    if (int rc = sdr_devbuf_reserve(e, &e->ws_stats, 2 * sizeof(ItemStats))) return rc;
    if (int rc = sdr_pinned_reserve(e, &e->ctx0, 2 * sizeof(ItemStats))) return rc;
    ItemStats* host = static_cast<ItemStats*>(e->ctx0.pinned);
    for (int v = 0; v < 2; ++v) {
        host[v] = ItemStats{0x7fffffff, 0, 0, 0ull, ~0ull, 0x7fffffff, 0, (v ? r2 : r1).want_s12 ? 1 : 0};
    }
    SDR_HIP(hipMemcpyAsync(e->ws_stats.ptr, host, 2 * sizeof(ItemStats), hipMemcpyHostToDevice, e->stream));
    const unsigned check_blocks = (unsigned)((n_items + 255) / 256) < 2048u ? (unsigned)((n_items + 255) / 256) : 2048u;      // (eight per CU)
    hipLaunchKernelGGL(validate_items_kernel, dim3(check_blocks), dim3(256), 0, e->stream, d_items, n_items, r1, r2,
                       (const int32_t*)e->code_len, static_cast<ItemStats*>(e->ws_stats.ptr));
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipMemcpyAsync(host, e->ws_stats.ptr, 2 * sizeof(ItemStats), hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    auto as_double = [](unsigned long long b) { double d; std::memcpy(&d, &b, sizeof(d)); return d; };
    if (host[0].first_bad != 0x7fffffff) {
        const int i = host[0].first_bad;
        sdr_epl_item it;
        if (h_items) it = h_items[i];
        else SDR_HIP(hipMemcpy(&it, d_items + i, sizeof(it), hipMemcpyDeviceToHost));
        int lw, wd;
        const int rc = validate_items(e, &it, 1, spacing, n_taps, 1.0, &lw, &wd, i);   // (the message, from the host's own walk of that item)
        return rc ? rc : sdr_fail(SDR_ERR_INVALID, "item %d: rejected by the device check", i);
    }
    *lut_words = host[0].maxlen + SDR_LUT_PAD + 2;
    *wide = variant_of(e, r1, spacing, as_double(host[0].max_step_bits), as_double(host[0].min_step_bits), host[0].m_lo, host[0].m_hi,
                       host[0].all_split != 0);
    *ok2 = host[1].first_bad == 0x7fffffff;
    if (*ok2) {
        *lw2 = host[1].maxlen + SDR_LUT_PAD + 2;
        *wide2 = variant_of(e, r2, spacing, as_double(host[1].max_step_bits), as_double(host[1].min_step_bits), host[1].m_lo, host[1].m_hi,
                            host[1].all_split != 0);
    }
    return SDR_OK;
}

// (half-chip view of a list on the device: 2 * rem_code, 2 * code_step)
__global__ __launch_bounds__(256) void double_items_kernel(sdr_epl_item* __restrict__ items, int n_items) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    items[i].rem_code *= 2.0;
    items[i].code_step *= 2.0;
}

// Device memory for a plan: a buffer a destroyed plan left in the engine's pool if one fits (not more than twice the size:
// a 60 s plan's 0.9 GB of setups never serve a one-second plan), else a fresh allocation.  plan_give: back into the pool --
// eight buffers at most, the smallest makes room -- instead of hipFree, which waits for every stream of the device.
static hipError_t plan_take(sdr_engine* e, void** out, size_t bytes, size_t* got) {
    int best = -1;
    for (int i = 0; i < (int)e->plan_pool.size(); ++i) {
        const size_t b = e->plan_pool[(size_t)i].bytes;
        if (b >= bytes && b <= 2 * bytes + 4096 && (best < 0 || b < e->plan_pool[(size_t)best].bytes)) best = i;
    }
    if (best >= 0) {
        *out = e->plan_pool[(size_t)best].ptr;
        *got = e->plan_pool[(size_t)best].bytes;
        e->plan_pool.erase(e->plan_pool.begin() + best);
        return hipSuccess;
    }
    *got = bytes;
    return hipMalloc(out, bytes);
}

static void plan_give(sdr_engine* e, void* ptr, size_t bytes) {
    if (!ptr) return;
    if (!e || bytes == 0) {
        (void)hipFree(ptr);
        return;
    }
    e->plan_pool.push_back(DevBuf{ptr, bytes});
    if (e->plan_pool.size() > 8) {
        size_t small = 0;
        for (size_t i = 1; i < e->plan_pool.size(); ++i)
            if (e->plan_pool[i].bytes < e->plan_pool[small].bytes) small = i;
        (void)hipFree(e->plan_pool[small].ptr);
        e->plan_pool.erase(e->plan_pool.begin() + (long)small);
    }
}

extern "C" {

static int plan_create_impl(sdr_engine* e, const sdr_epl_item* items, const sdr_epl_item* d_src, int n_items, const double* spacing,
                            int n_taps, double fs, bool use_workspaces, sdr_epl_plan** out);

int sdr_epl_plan_create(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing,
                        int n_taps, double fs, sdr_epl_plan** out) {
    if (!items) return sdr_fail(SDR_ERR_INVALID, "no items");
    return plan_create_impl(e, items, nullptr, n_items, spacing, n_taps, fs, false, out);
}

int sdr_epl_plan_create_dev(sdr_engine* e, const sdr_epl_item* items_dev, int n_items, const double* spacing,
                            int n_taps, double fs, sdr_epl_plan** out) {
    if (!items_dev) return sdr_fail(SDR_ERR_INVALID, "no items");
    return plan_create_impl(e, nullptr, items_dev, n_items, spacing, n_taps, fs, false, out);
}

// items: the list in host memory, or d_src: the list in device memory (copied: the plan owns its items either way).
// use_workspaces: the one-shot sdr_epl_batch (the function-level drop-in calls it once per EPL()) keeps its three device
// buffers in the engine between calls instead of allocating and freeing them every time.
static int plan_create_impl(sdr_engine* e, const sdr_epl_item* items, const sdr_epl_item* d_src, int n_items, const double* spacing,
                            int n_taps, double fs, bool use_workspaces, sdr_epl_plan** out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if ((!items && !d_src) || n_items <= 0) return sdr_fail(SDR_ERR_INVALID, "no items");
    if (!spacing || n_taps < 1 || n_taps > SDR_MAX_TAPS)
        return sdr_fail(SDR_ERR_INVALID, "n_taps %d outside 1..%d", n_taps, SDR_MAX_TAPS);
    if (!(fs > 0.0)) return sdr_fail(SDR_ERR_INVALID, "fs must be positive");
    int lut_words = 0;
    int wide = 0;
    const bool timing = getenv("SDR_PLAN_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // a short list is checked by the host before anything is allocated (one call per EPL(): latency); a long one by one
    // thread per item behind its upload
    const bool long_list = d_src != nullptr || n_items >= 4096;
    // a list that misses the chip-aligned correlator only because a chip holds too many samples (32-52: GPS L1 C/A at
    // 50 MHz) runs on the half-chip view of its replicas
    bool doubled = false;
    auto consider_half_chip_view = [&](bool ok2, int lw2, int wide2) -> int {
        if (wide < sdr::kChipMax && e->iq_fmt == SDR_FMT_CI8 && !e->epl_no_chip && !e->epl_no_double && ok2 && wide2 >= sdr::kChipMax) {
            if (int rc = ensure_doubled_luts(e, e->stream)) return rc;
            doubled = true;
            wide = wide2;
            lut_words = 2 * lut_words < e->lut2_stride ? 2 * lut_words : e->lut2_stride;
            (void)lw2;
        }
        return SDR_OK;
    };
    if (!long_list) {
        if (int rc = validate_items(e, items, n_items, spacing, n_taps, 1.0, &lut_words, &wide)) return rc;
        int lw2 = 0, wide2 = 0;
        bool ok2 = false;
        if (wide < sdr::kChipMax && e->iq_fmt == SDR_FMT_CI8 && !e->epl_no_chip && !e->epl_no_double)
            ok2 = validate_items(e, items, n_items, spacing, n_taps, 2.0, &lw2, &wide2) == SDR_OK;
        if (int rc = consider_half_chip_view(ok2, lw2, wide2)) return rc;
    }
    const double t_valid = now();

    sdr_epl_plan* p = new (std::nothrow) sdr_epl_plan();
    if (!p) return sdr_fail(SDR_ERR_NOMEM, "host allocation failed");
    p->n_items = n_items;
    p->n_taps = n_taps;
    p->fs = fs;
    p->code_generation = e->code_generation;
    p->ring_capacity = e->iq_capacity;
    hipError_t err = hipSuccess;
    if (use_workspaces) {
        int rc = sdr_devbuf_reserve(e, &e->ws_items, (size_t)n_items * sizeof(sdr_epl_item));
        if (!rc) rc = sdr_devbuf_reserve(e, &e->ws_out, (size_t)n_items * 2 * n_taps * sizeof(double));
        if (!rc) rc = sdr_devbuf_reserve(e, &e->ws_spacing, SDR_MAX_TAPS * sizeof(double));
        if (rc) {
            delete p;
            return rc;
        }
        p->borrowed = true;
        p->d_items = (sdr_epl_item*)e->ws_items.ptr;
        p->d_out = (double*)e->ws_out.ptr;
        p->d_spacing = (double*)e->ws_spacing.ptr;
    } else {
        err = plan_take(e, (void**)&p->d_items, (size_t)n_items * sizeof(sdr_epl_item), &p->bytes_items);
        if (err == hipSuccess) err = plan_take(e, (void**)&p->d_out, (size_t)n_items * 2 * n_taps * sizeof(double), &p->bytes_out);
        if (err == hipSuccess) err = plan_take(e, (void**)&p->d_spacing, SDR_MAX_TAPS * sizeof(double), &p->bytes_spacing);
    }
    const double t_alloc = now();
    if (err == hipSuccess)
        err = hipMemcpyAsync(p->d_items, d_src ? d_src : items, (size_t)n_items * sizeof(sdr_epl_item),
                             d_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess && long_list) {
        int lw2 = 0, wide2 = 0;
        bool ok2 = false;
        int rc = validate_items_dev(e, p->d_items, items, n_items, spacing, n_taps, &lut_words, &wide, &ok2, &lw2, &wide2);
        if (!rc) rc = consider_half_chip_view(ok2, lw2, wide2);
        if (rc) {
            sdr_epl_plan_destroy(e, p);
            return rc;
        }
    }
    p->lut_words = lut_words;
    p->wide = wide;
    p->doubled = doubled;
    // the period of the code-slot pattern, if the list has one (what lets four epochs of a channel share a staged table)
    std::vector<sdr_epl_item> fetched;          // (a device list of long replicas: its slots are read back for this)
    if (err == hipSuccess && lut_words >= kLongLutWords) {
        const sdr_epl_item* h = items;
        if (!h) {
            fetched.resize(n_items);
            err = hipMemcpy(fetched.data(), p->d_items, (size_t)n_items * sizeof(sdr_epl_item), hipMemcpyDeviceToHost);
            h = fetched.data();
        }
        if (err == hipSuccess) {
            int c = 1;
            while (c < n_items && h[c].code_slot != h[0].code_slot) ++c;
            bool periodic = c < n_items || n_items == 1;
            for (int i = 0; periodic && i + c < n_items; ++i) periodic = h[i].code_slot == h[i + c].code_slot;
            p->group_stride = periodic ? c : 0;
        }
    }
    std::vector<sdr_epl_item> items2;   // short lists on the half-chip view: the host's copy with 2 * rem_code, 2 * code_step (for its setups)
    std::vector<char> host_setups;      // short lists: the setups are made on the host (kept until the final synchronisation)
    double spacing2[SDR_MAX_TAPS];
    if (doubled) {
        if (items && !long_list) {
            items2.assign(items, items + n_items);
            for (sdr_epl_item& it : items2) it.rem_code *= 2.0, it.code_step *= 2.0;
        }
        for (int t = 0; t < n_taps; ++t) spacing2[t] = 2.0 * spacing[t];
        if (err == hipSuccess) {
            hipLaunchKernelGGL(double_items_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, e->stream, p->d_items, n_items);
            err = hipGetLastError();
        }
    }
    if (err == hipSuccess)
        err = hipMemcpyAsync(p->d_spacing, doubled ? spacing2 : spacing, n_taps * sizeof(double), hipMemcpyHostToDevice, e->stream);
    // ---- the per-item setups of the straight-line kernels: one launch, one thread per item (items and spacings are on the device)
    auto reserve_setups = [&](size_t bytes_per_item) {
        p->setup_bytes = bytes_per_item;
        const size_t bytes = bytes_per_item * (size_t)n_items;
        if (use_workspaces) {
            if (sdr_devbuf_reserve(e, &e->ws_setups, bytes + 64) != SDR_OK) err = hipErrorOutOfMemory;
            else p->d_setups = (char*)e->ws_setups.ptr;
        } else {
            err = plan_take(e, (void**)&p->d_setups, bytes + 64, &p->bytes_setups);     // (+ the counter of items a scheme does not cover)
        }
    };
    const unsigned setup_grid = (unsigned)((n_items + 255) / 256);       // (the two-chips-per-lane setups: 256 items per block)
    const unsigned setup_grid64 = (unsigned)((n_items + 63) / 64);        // (the one-chip-per-lane setups: a wave per 64 items)
    if (err == hipSuccess && e->iq_fmt == SDR_FMT_CI8 && (wide & 255) >= kChipMax && (wide & (kVariantKSMask | kVariantKI)) &&
        (n_taps == 3 || n_taps == 5)) {
        reserve_setups(n_taps == 3 ? sizeof(sdr::ChipSetup<3>) : sizeof(sdr::ChipSetup<5>));
        if (err == hipSuccess && !long_list) {
            // a short list (sdr_epl_batch: one call per EPL()): the same function on the host, ~2 us per item, instead of a launch
            const sdr_epl_item* src = doubled ? items2.data() : items;
            const double* spc = doubled ? spacing2 : spacing;
            host_setups.resize(p->setup_bytes * (size_t)n_items);
            const int km = (wide & 255) - kChipMax;
            const bool ks = (wide & kVariantKSMask) != 0;
            for (int i = 0; i < n_items; ++i) {
                const sdr_epl_item& it = src[i];
                if (n_taps == 5) {
                    sdr::ChipSetup<5>& S5 = reinterpret_cast<sdr::ChipSetup<5>*>(host_setups.data())[i];
                    switch (km) {
#define SDR_SETUP5_CASE(K) \
    case K: sdr::chip_setup<5, K, 0, 1>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step, spc, fs, kWaveThreads, S5); break;
                        SDR_SETUP5_CASE(16) SDR_SETUP5_CASE(17) SDR_SETUP5_CASE(18) SDR_SETUP5_CASE(19) SDR_SETUP5_CASE(20) SDR_SETUP5_CASE(21)
                        SDR_SETUP5_CASE(22) SDR_SETUP5_CASE(23) SDR_SETUP5_CASE(24) SDR_SETUP5_CASE(25)
#undef SDR_SETUP5_CASE
                        default: err = hipErrorInvalidValue; break;
                    }
                    continue;
                }
                sdr::ChipSetup<3>& S = reinterpret_cast<sdr::ChipSetup<3>*>(host_setups.data())[i];
                switch (km * 2 + (ks ? 1 : 0)) {
#define SDR_SETUP_CASE(K)                                                                                                                      \
    case 2 * K + 1: sdr::chip_setup<3, K, K / 2, 0>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step, spc, fs, kWaveThreads, S); break; \
    case 2 * K: sdr::chip_setup<3, K, 0, 1>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step, spc, fs, kWaveThreads, S); break;
                    SDR_SETUP_CASE(15) SDR_SETUP_CASE(16) SDR_SETUP_CASE(17) SDR_SETUP_CASE(18) SDR_SETUP_CASE(19) SDR_SETUP_CASE(20)
                    SDR_SETUP_CASE(21) SDR_SETUP_CASE(22) SDR_SETUP_CASE(23) SDR_SETUP_CASE(24) SDR_SETUP_CASE(25)
#undef SDR_SETUP_CASE
                    default: err = hipErrorInvalidValue; break;
                }
            }
            if (err == hipSuccess) err = hipMemcpyAsync(p->d_setups, host_setups.data(), host_setups.size(), hipMemcpyHostToDevice, e->stream);
        } else if (err == hipSuccess) {
            const int km = (wide & 255) - kChipMax;
            const bool ks = (wide & kVariantKSMask) != 0;
            if (n_taps == 5) {
                sdr::ChipSetup<5>* d5 = reinterpret_cast<sdr::ChipSetup<5>*>(p->d_setups);
                switch (km) {
#define SDR_SETUP5_CASE(K) \
    case K: hipLaunchKernelGGL((chip_setup_kernel<5, K, 0, 1>), dim3(setup_grid64), dim3(64), 0, e->stream, p->d_items, n_items, p->d_spacing, fs, e->iq_capacity, d5); break;
                    SDR_SETUP5_CASE(16) SDR_SETUP5_CASE(17) SDR_SETUP5_CASE(18) SDR_SETUP5_CASE(19) SDR_SETUP5_CASE(20) SDR_SETUP5_CASE(21)
                    SDR_SETUP5_CASE(22) SDR_SETUP5_CASE(23) SDR_SETUP5_CASE(24) SDR_SETUP5_CASE(25)
#undef SDR_SETUP5_CASE
                    default: err = hipErrorInvalidValue; break;
                }
            } else {
                sdr::ChipSetup<3>* d3 = reinterpret_cast<sdr::ChipSetup<3>*>(p->d_setups);
                switch (km * 2 + (ks ? 1 : 0)) {
#define SDR_SETUP_CASE(K)                                                                                                              \
    case 2 * K + 1: hipLaunchKernelGGL((chip_setup_kernel<3, K, K / 2, 0>), dim3(setup_grid64), dim3(64), 0, e->stream, p->d_items, n_items, p->d_spacing, fs, e->iq_capacity, d3); break; \
    case 2 * K: hipLaunchKernelGGL((chip_setup_kernel<3, K, 0, 1>), dim3(setup_grid64), dim3(64), 0, e->stream, p->d_items, n_items, p->d_spacing, fs, e->iq_capacity, d3); break;
                    SDR_SETUP_CASE(15) SDR_SETUP_CASE(16) SDR_SETUP_CASE(17) SDR_SETUP_CASE(18) SDR_SETUP_CASE(19) SDR_SETUP_CASE(20)
                    SDR_SETUP_CASE(21) SDR_SETUP_CASE(22) SDR_SETUP_CASE(23) SDR_SETUP_CASE(24) SDR_SETUP_CASE(25)
#undef SDR_SETUP_CASE
                    default: err = hipErrorInvalidValue; break;
                }
            }
            if (err == hipSuccess) err = hipGetLastError();
        }
    }
    // three taps half a chip apart and chips of 9.5 .. 10 or 11.5 .. 12 samples (the reference's shipped 10 MHz; 12 MHz): two
    // chips per lane, if (nearly) every item fits the scheme.  (At 24 .. 24.5 samples per chip -- boundaries <12, 24, 36, 48> --
    // the same kernel was measured against the one-chip form: 5 % fewer instructions per sample, 60 % more per epoch, at the
    // register cap: 0.335 instead of 0.316 ms per 32 000 epochs; three chips per lane at 10 MHz -- <4, 9, 14, 19, 24, 29> --
    // spill 27 registers at three waves per SIMD: 0.44 instead of 0.57 of the roof.  Neither is instantiated.)
    // ... and chips of 3.75 .. 4 samples (the reference's 4 MHz, BASELINE configs[0]): FOUR chips per lane, where the list
    // would otherwise go per sample (round 6)
    if (err == hipSuccess && e->iq_fmt == SDR_FMT_CI8 && n_taps == 3 && !doubled && !e->epl_no_chip2 && ((wide & 255) == 8 || (wide & 255) == 0) &&
        lut_words < kLongLutWords) {          // (long multi-period replicas keep the four-epochs-per-workgroup kernels)
        sdr_epl_item first = {};
        if (items) first = items[0];
        else err = hipMemcpy(&first, p->d_items, sizeof(first), hipMemcpyDeviceToHost);
        const double two_chips = err == hipSuccess ? std::floor(2.0 / first.code_step) : 0.0;   // samples in two chips (any positive step got here)
        const double four_chips = err == hipSuccess ? std::floor(4.0 / first.code_step) : 0.0;
        const bool half_apart = spacing[0] == -0.5 && spacing[1] == 0.0 && spacing[2] == 0.5;
        const int shape = (wide & 255) == 8 ? (two_chips == 19.0 ? 1 : (two_chips == 23.0 ? 2 : 0)) : (four_chips == 15.0 && half_apart ? 3 : 0);
        if (shape) {
            reserve_setups(shape == 1 ? sizeof(sdr::ChipNSetup<4, 9, 14, 19>) : shape == 2 ? sizeof(sdr::ChipNSetup<5, 11, 17, 23>)
                                                                                           : sizeof(sdr::ChipNSetup<1, 3, 5, 7, 9, 11, 13, 15>));
            int* d_missed = p->d_setups ? reinterpret_cast<int*>(p->d_setups + p->setup_bytes * (size_t)n_items) : nullptr;
            int missed = n_items;
            if (err == hipSuccess && !long_list) {
                // a short list (sdr_epl_batch: one call per EPL()): the same function on the host, ~2 us per item, instead of a
                // launch and a round trip for the count -- the call is latency, not throughput
                host_setups.resize(p->setup_bytes * (size_t)n_items);
                missed = 0;
                for (int i = 0; i < n_items; ++i) {
                    const sdr_epl_item& it = items[i];
                    const bool ok = shape == 1
                        ? sdr::chipn_setup<4, 9, 14, 19>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step,
                                                         spacing, fs, reinterpret_cast<sdr::ChipNSetup<4, 9, 14, 19>*>(host_setups.data())[i])
                        : shape == 2
                        ? sdr::chipn_setup<5, 11, 17, 23>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step,
                                                          spacing, fs, reinterpret_cast<sdr::ChipNSetup<5, 11, 17, 23>*>(host_setups.data())[i])
                        : sdr::chipn_setup<1, 3, 5, 7, 9, 11, 13, 15>(it.n_samples, it.start_sample, e->iq_capacity, it.carrier_hz, it.rem_code, it.code_step,
                                                                      spacing, fs, reinterpret_cast<sdr::ChipNSetup<1, 3, 5, 7, 9, 11, 13, 15>*>(host_setups.data())[i]);
                    missed += ok ? 0 : 1;
                }
                if (missed <= n_items / 64)      // (host_setups lives until the synchronisation that ends the plan's creation)
                    err = hipMemcpyAsync(p->d_setups, host_setups.data(), host_setups.size(), hipMemcpyHostToDevice, e->stream);
            } else if (err == hipSuccess) {
              err = hipMemsetAsync(d_missed, 0, sizeof(int), e->stream);
              if (err == hipSuccess) {
                if (shape == 1)
                    hipLaunchKernelGGL((chipn_setup_kernel<4, 9, 14, 19>), dim3(setup_grid), dim3(256), 0, e->stream, p->d_items,
                                       n_items, p->d_spacing, fs, e->iq_capacity,
                                       reinterpret_cast<sdr::ChipNSetup<4, 9, 14, 19>*>(p->d_setups), d_missed);
                else if (shape == 2)
                    hipLaunchKernelGGL((chipn_setup_kernel<5, 11, 17, 23>), dim3(setup_grid), dim3(256), 0, e->stream, p->d_items,
                                       n_items, p->d_spacing, fs, e->iq_capacity,
                                       reinterpret_cast<sdr::ChipNSetup<5, 11, 17, 23>*>(p->d_setups), d_missed);
                else
                    hipLaunchKernelGGL((chipn_setup_kernel<1, 3, 5, 7, 9, 11, 13, 15>), dim3(setup_grid), dim3(256), 0, e->stream, p->d_items,
                                       n_items, p->d_spacing, fs, e->iq_capacity,
                                       reinterpret_cast<sdr::ChipNSetup<1, 3, 5, 7, 9, 11, 13, 15>*>(p->d_setups), d_missed);
                err = hipGetLastError();
              }
              if (err == hipSuccess) err = hipMemcpyAsync(&missed, d_missed, sizeof(int), hipMemcpyDeviceToHost, e->stream);
              if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
            }
            if (err == hipSuccess && missed <= n_items / 64) {
                p->wide = wide = (wide & 255) + kVariantC2 * shape;
            } else if (err == hipSuccess) {                 // too many strays: the boundary variant serves the list
                if (!use_workspaces && p->d_setups) plan_give(e, p->d_setups, p->bytes_setups);
                p->d_setups = nullptr;
                p->setup_bytes = 0;
            }
        }
    }
    const double t_queued = now();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (timing)
        fprintf(stderr, "plan_create %d items: validate %.2f ms, alloc %.2f, queue copies+setups %.2f, sync %.2f\n", n_items,
                t_valid - t_begin, t_alloc - t_valid, t_queued - t_alloc, now() - t_queued);
    if (err != hipSuccess) {
        sdr_epl_plan_destroy(e, p);
        return sdr_fail(err == hipErrorOutOfMemory ? SDR_ERR_NOMEM : SDR_ERR_HIP, "plan setup failed: %s",
                        hipGetErrorString(err));
    }
    *out = p;
    return SDR_OK;
}

int sdr_epl_plan_run_range(sdr_engine* e, sdr_epl_plan* p, int64_t first, int64_t count) {
    return sdr_epl_plan_run_range_on(e, p, first, count, 0);
}

int sdr_epl_plan_run_range_on(sdr_engine* e, sdr_epl_plan* p, int64_t first, int64_t count, int stream_id) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!p) return sdr_fail(SDR_ERR_INVALID, "plan is NULL");
    StreamCtx* ctx = sdr_stream_ctx(e, stream_id);
    if (!ctx) return sdr_fail(SDR_ERR_INVALID, "stream id %d does not exist", stream_id);
    // A plan is validated against the code tables and the ring as they were at sdr_epl_plan_create; after
    // sdr_code_slots(_ex) / sdr_iq_alloc its slot indices, LUT length and ring bounds mean something else.
    if (p->code_generation != e->code_generation || p->ring_capacity != e->iq_capacity ||
        p->lut_words > (p->doubled ? 2 * e->lut_stride + 3 : e->lut_stride))
        return sdr_fail(SDR_ERR_STATE, "plan is stale: the code slots or the IQ ring were re-allocated after it was created");
    if (first < 0 || count <= 0 || first + count > p->n_items)
        return sdr_fail(SDR_ERR_RANGE, "item range [%lld, %lld) outside the plan's %d items", (long long)first,
                        (long long)(first + count), p->n_items);
    if (p->doubled)
        if (int rc = ensure_doubled_luts(e, ctx->stream)) return rc;   // (a slot may have been re-staged since the plan was made)
    // (the straight-line kernels build their doubles from sign-flipped bytes: a ci8 ring holds them that way -- correlator.h kCi8Flip)
    if (int rc = sdr_iq_order_reader(e, ctx)) return rc;      // (behind the uploads queued on the engine's stream so far)
    const sdr_epl_item* items = p->d_items + first;
    double* out = p->d_out + (size_t)first * 2 * p->n_taps;
    const int n = (int)count;
    const void* setups = p->d_setups ? p->d_setups + (size_t)first * p->setup_bytes : nullptr;
    if (e->iq_fmt == SDR_FMT_CI8 && (p->wide & (3 * kVariantC2)) && setups) {      // two chips per lane: its own kernel
        hipStream_t st = ctx->stream;
        ProfScope ps(e, "epl_kernel", st);
        const size_t shmem = 8 * sizeof(uint32_t) + (size_t)((p->lut_words + 3) & ~3) * sizeof(uint32_t);
        const int shape = (p->wide / kVariantC2) & 3;
        auto launch2 = [&](auto kernel, auto* typed) {
            hipLaunchKernelGGL(kernel, dim3(n), dim3(kWaveThreads), shmem, st, e->iq, (const void*)e->iq, e->iq_capacity,
                               items, n, e->luts, p->lut_words, e->lut_stride, p->d_spacing, p->fs, out, typed);
        };
        if (shape == 1) launch2(epl2_kernel<4, 9, 14, 19>, reinterpret_cast<const sdr::ChipNSetup<4, 9, 14, 19>*>(setups));
        else if (shape == 2) launch2(epl2_kernel<5, 11, 17, 23>, reinterpret_cast<const sdr::ChipNSetup<5, 11, 17, 23>*>(setups));
        else launch2(epl2_kernel<1, 3, 5, 7, 9, 11, 13, 15>, reinterpret_cast<const sdr::ChipNSetup<1, 3, 5, 7, 9, 11, 13, 15>*>(setups));
    } else {
        hipStream_t st = ctx->stream;
        ProfScope ps(e, "epl_kernel", st);
        switch (e->iq_fmt) {
            case SDR_FMT_CI8: launch_fmt<SDR_FMT_CI8>(e, st, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, p->wide, p->group_stride, p->doubled, out, setups); break;
            case SDR_FMT_CI16: launch_fmt<SDR_FMT_CI16>(e, st, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, p->wide, p->group_stride, p->doubled, out, setups); break;
            case SDR_FMT_CF32: launch_fmt<SDR_FMT_CF32>(e, st, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, p->wide, p->group_stride, p->doubled, out, setups); break;
            default: launch_fmt<SDR_FMT_CF64>(e, st, items, n, p->d_spacing, p->fs, p->n_taps, p->lut_words, p->wide, p->group_stride, p->doubled, out, setups); break;
        }
    }
    SDR_HIP(hipGetLastError());
    if (ctx->stream != e->stream) {   // (fetch copies on e->stream, which is ordered behind its own launches anyway)
        hipEvent_t ev = nullptr;
        for (auto& se : p->ran_on)
            if (se.first == ctx->stream) ev = se.second;
        if (!ev) {
            SDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            p->ran_on.emplace_back(ctx->stream, ev);
        }
        SDR_HIP(hipEventRecord(ev, ctx->stream));
    }
    return SDR_OK;
}

int sdr_epl_plan_variant(const sdr_epl_plan* p) { return p ? p->wide + (p->doubled ? 65536 : 0) : -1; }

int sdr_epl_plan_run(sdr_engine* e, sdr_epl_plan* p) {
    if (!p) return sdr_fail(SDR_ERR_INVALID, "plan is NULL");
    return sdr_epl_plan_run_range(e, p, 0, p->n_items);
}

int sdr_epl_plan_fetch(sdr_engine* e, sdr_epl_plan* p, double* out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!p || !out) return sdr_fail(SDR_ERR_INVALID, "NULL plan or output");
    for (auto& se : p->ran_on) SDR_HIP(hipStreamWaitEvent(e->stream, se.second, 0));   // the streams its ranges ran on, no others
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
    for (auto& se : p->ran_on) {
        (void)hipEventSynchronize(se.second);
        (void)hipEventDestroy(se.second);
    }
    if (!p->borrowed) {     // (every stream the plan ran on has been waited for: the buffers are free to serve the next plan)
        plan_give(e, p->d_items, p->bytes_items);
        plan_give(e, p->d_out, p->bytes_out);
        plan_give(e, p->d_spacing, p->bytes_spacing);
        plan_give(e, p->d_setups, p->bytes_setups);
    }
    delete p;
}

int sdr_epl_batch(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing, int n_taps,
                  double fs, double* out) {
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    sdr_epl_plan* p = nullptr;
    if (int rc = plan_create_impl(e, items, nullptr, n_items, spacing, n_taps, fs, true, &p)) return rc;
    int rc = sdr_epl_plan_run(e, p);
    if (!rc) rc = sdr_epl_plan_fetch(e, p, out);
    sdr_epl_plan_destroy(e, p);
    return rc;
}

}  // extern "C"
