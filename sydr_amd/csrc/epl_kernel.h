// The E/P/L kernel template (K1) and the words of a plan's variant code: included by epl.hip, which instantiates the general
// forms and the headline geometries, and by epl_straight.hip, which instantiates the straight-line forms of the other block
// lengths (one translation unit each: they compile side by side).  Everything here has internal linkage.
#pragma once
#include "correlator.h"
#include "correlator_chip.h"

namespace {

using namespace sdr;

constexpr int kWaveThreads = 64;  // one wave per channel-epoch: the ~5 us fixed latency of a workgroup is amortised over 4x more work
// plan variant word (sdr_epl_plan_variant): low byte = samples a lane owns (0 / 8 / 16 / 26, + 24 when the block length is
// compiled in), then the compile-time tap geometry
constexpr int kVariantKS12 = 256 * 12;   // three taps, the outer ones switching 12.x samples into the anchor's block (24 / 25 samples per chip)
constexpr int kVariantKSMask = 256 * 15; // ... in general: KS in bits 8-11, with the block length KM in the low byte (kChipMax + KM)
constexpr int kVariantKS9 = 256 * 9;     // 19 / 20 samples per chip (20 MHz): KM = 19, KS = 9
constexpr int kVariantKM1516 = 16;       // (added to kChipMax) 15.x or 16.x samples per chip by the epoch: both block lengths compiled in
constexpr int kVariantKI = 4096;         // taps whole (half-)chips apart: no switch inside a block
constexpr int kVariantC2 = 8192;         // several chips per lane (correlator_chip2.h), x 1 / 2: boundaries <4,9,14,19> / <5,11,17,23>
constexpr int kLongLutWords = 4096;  // replicas of 16 KB and more (multi-period / BOC half-chip codes): four epochs share a staged copy

// Dynamic LDS: [red: WPW*2*NT doubles][scratch: strips / rotations][lut: lut_words uint32]
//
// One WAVE per item; WPW = 1: one workgroup per item.  (A persistent grid-stride variant that keeps the replica in
// LDS across items was measured slower: holding two items' parameters pushed the kernel from 4 to 2 resident waves
// per SIMD -- 1.15 ms vs 0.87 ms per 32 000-item launch -- so hardware workgroup dispatch does the scheduling.)
// WPW = 4 (long replicas): a 33 KB table per single-wave workgroup leaves 3 waves on a CU; four waves of a workgroup
// correlate four epochs of the SAME channel -- items i, i+C, i+2C, i+3C of a list whose code slots repeat with
// period C (the plan checks that) -- against one staged copy, 8 waves per CU.
// W: samples a lane owns per iteration of the boundary variant (16 or 8), 0 = the per-sample variant.
#ifndef SDR_EPL_WAVES
#define SDR_EPL_WAVES 1
#endif
// KM2 != 0 (with KS = KI = 0): the list's epochs have KM or KM2 = KM + 1 whole samples per chip by the sign of their code
// Doppler (16.368 MHz: exactly 16.0) -- both block lengths are compiled in and an epoch takes the body of its own.
template <int FMT, int NT, int W, int KM = 0, int WPW = 1, int KS = 0, int KI = 0, int KM2 = 0>
#ifndef SDR_EPL_KS_WAVES
#define SDR_EPL_KS_WAVES 3   // (the KS kernel sits at the 168-register cap of three waves per SIMD; two waves: 0.306 instead of 0.286 ms per stream-second)
#endif
#ifndef SDR_EPL_KI_WAVES
#define SDR_EPL_KI_WAVES 1
#endif
#ifndef SDR_EPL_KM2_WAVES
#define SDR_EPL_KM2_WAVES 1
#endif
__global__ __launch_bounds__(kWaveThreads * WPW, (KS != 0 ? SDR_EPL_KS_WAVES : (KI != 0 ? SDR_EPL_KI_WAVES : (KM2 != 0 ? SDR_EPL_KM2_WAVES : SDR_EPL_WAVES)))) void epl_kernel(const void* __restrict__ ring, const void* __restrict__ ring_flipped, int64_t capacity,
                                                       const sdr_epl_item* __restrict__ items, int n_items, int group_stride,
                                                       const uint32_t* __restrict__ luts,
                                                       int lut_words, int lut_stride,
                                                       const double* __restrict__ spacing, double fs,
                                                       int tap0, int n_taps_total,
                                                       double* __restrict__ out, const void* __restrict__ setups) {
    constexpr int kThreads = kWaveThreads * WPW;
    constexpr bool kPre = W == kChipMax && FMT == SDR_FMT_CI8 && KM != 0 && (KS != 0 || KI != 0);   // the plan holds a ChipSetup per item
    extern __shared__ double smem[];
    double* red = smem;
    double2* prefix = reinterpret_cast<double2*>(red + WPW * 2 * NT);          // boundary variants: kThreads*9 slots; chip variant: strips + rotations
    constexpr int kScratchSlots = W == kChipMax ? kThreads * chip_strip_slots<NT>() + WPW * kChipRotSlots : (W ? kThreads * kPrefixSlots : 0);
    uint32_t* lut = reinterpret_cast<uint32_t*>(prefix + kScratchSlots);

    const int tid = threadIdx.x;
    // (the wave number as a scalar: what is indexed with it -- the item, its setup in the plan -- is then read with scalar loads)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef SDR_TRACE_WG
    const unsigned long long t_start = wall_clock64();
#endif
    int item = blockIdx.x;
    if (WPW > 1) {
        const int g = blockIdx.x / group_stride, c = blockIdx.x - g * group_stride;
        item = (g * WPW + wave) * group_stride + c;
    }
    const bool have = item < n_items;
    // (a wave without an item still stages its share of the table: the slot of its group's column c = item c of the
    // range -- clamped, because a range shorter than the stride launches columns that hold no item at all)
    const int column = (int)(blockIdx.x % group_stride);
    const sdr_epl_item it = items[have ? item : (column < n_items ? column : n_items - 1)];
    stage_lut<kThreads>(lut, luts + (size_t)it.code_slot * lut_stride, lut_words, tid);
    double dphi;
    if constexpr (!kPre) dphi = carrier_step(it.carrier_hz, fs);
    EpochParams ep;
    ep.start_sample = it.start_sample;
    ep.n = it.n_samples;
    ep.carrier_hz = it.carrier_hz;
    ep.rem_carrier = it.rem_carrier;
    ep.rem_code = it.rem_code;
    ep.code_step = it.code_step;
    EpochConsts<NT> K;
    ChipGeom<NT> G;
    const ChipRot* rot_plan = nullptr;
    int64_t base = -1;
    if constexpr (kPre) {
        // everything wave-uniform that is not a sincos was worked out by the host when the plan was made: scalar loads
        const ChipSetup<NT>& S = static_cast<const ChipSetup<NT>*>(setups)[have ? item : 0];
        dphi = S.dphi;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            K.shift[q] = S.shift[q], K.step[q] = S.step[q], K.inv_step[q] = S.inv_step[q];
            // (only the exact re-evaluations near a sample and the edge samples read these: vector registers, of which
            // there are enough -- the scalar ones hold the sample loop's rotations)
            asm volatile("" : "+v"(K.shift[q]), "+v"(K.step[q]), "+v"(K.inv_step[q]));
        }
        G = S.g;
        rot_plan = &S.r;
        base = S.base;
    } else if constexpr (W == kChipMax && FMT == SDR_FMT_CI8) {
        compute_tap_constants<NT>(K, ep, spacing + tap0);   // (the chip-aligned core evaluates its own rotations)
        if (chip_variant_applies(ep, capacity)) {
            base = ep.start_sample % capacity;
            chip_geometry<NT, (KM2 != 0 ? 0 : KM), KS, KI>(ep.n, K.shift, K.step, K.inv_step, G);   // (two lengths: the body is chosen below)
        }
    } else {
        compute_constants<NT>(K, ep, spacing + tap0, dphi, kWaveThreads);
    }
    __syncthreads();  // replica staged
    if (WPW > 1 && !have) return;

    double accr[NT], acci[NT];
    if constexpr (W == kChipMax && FMT == SDR_FMT_CI8) {
        // chip-aligned blocks (correlator_chip.h); an epoch it does not cover is redone per sample
        bool done = false;
        if constexpr (KM2 != 0) {
            static_assert(KS == 0 && KI == 0 && !kPre, "two block lengths: tap positions at run time, no plan setups");
            const int M = __builtin_amdgcn_readfirstlane((int)(G.Tfx >> 32));
            double2* const rot = prefix + kThreads * chip_strip_slots<NT>() + wave * kChipRotSlots;
            if (base >= 0 && M == KM)
                done = correlate_epoch_chip<NT, true, KM, 0, 0>(ring, ring_flipped, capacity, ep, dphi, K, G, base, rot_plan, lut, prefix, rot,
                                                                 tid, lane, kWaveThreads, lane, accr, acci);
            else if (base >= 0 && M == KM2)
                done = correlate_epoch_chip<NT, true, KM2, 0, 0>(ring, ring_flipped, capacity, ep, dphi, K, G, base, rot_plan, lut, prefix, rot,
                                                                  tid, lane, kWaveThreads, lane, accr, acci);
        } else {
            done = base >= 0 &&
                   correlate_epoch_chip<NT, true, KM, KS, KI>(ring, ring_flipped, capacity, ep, dphi, K, G, base, rot_plan, lut, prefix,
                                                              prefix + kThreads * chip_strip_slots<NT>() + wave * kChipRotSlots,
                                                              tid, lane, kWaveThreads, lane, accr, acci);
        }
        if (!done) {
            // (its own copy of the per-epoch constants: the in-group rotations the per-sample routine wants would
            // otherwise sit in 64 scalar registers across the whole chip-aligned path)
            EpochConsts<NT> K2;
            compute_constants<NT>(K2, ep, spacing + tap0, dphi, kWaveThreads);
            correlate_epoch<FMT, NT>(ring, capacity, ep, dphi, K2, lut, lane, kWaveThreads, lane, accr, acci);
        }
    } else if (W != 0 && !epoch_wraps(ep, capacity))
        correlate_epoch_wide<FMT, NT, true, (W ? W : kWide)>(ring, capacity, ep, dphi, K, lut, prefix, tid, lane, kWaveThreads, lane, accr, acci);
    else
        correlate_epoch<FMT, NT>(ring, capacity, ep, dphi, K, lut, lane, kWaveThreads, lane, accr, acci);
    if constexpr (NT == 3 || NT == 5) {
        // one wave per item (also with four items per workgroup): the lanes share the reduction of the 2*NT sums
        // (correlator.h: reduce_taps_scatter); the first eight (sixteen) lanes between them hold every total, each under
        // the slot it ended up with
        int slot;
        const double total = reduce_taps_scatter<NT>(accr, acci, lane, slot);
        if (lane < (NT == 3 ? 8 : 16)) out[(size_t)item * 2 * n_taps_total + 2 * tap0 + slot] = total;
    } else {
        const double total = reduce_taps<NT, kWaveThreads>(accr, acci, red, lane);
        if (lane < 2 * NT) out[(size_t)item * 2 * n_taps_total + 2 * tap0 + lane] = total;
    }
#ifdef SDR_TRACE_WG
    if (tid == 0 && blockIdx.x < 65536) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wg_trace[3 * blockIdx.x] = t_start;
        g_wg_trace[3 * blockIdx.x + 1] = wall_clock64();
        g_wg_trace[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

}  // namespace
