// What sdr_epl_plan_create checks of ONE item of a list (no item may index outside the ring or the staged replica) and what
// it contributes to the choice of the kernel variant -- one function for the host's walk of a short list, for the thread
// per item that checks a long one on the device (epl.hip: validate_items_kernel) and for the sanitizer build that feeds it
// hostile items on the CPU (tests/csrc/fuzz_items.hip, `make check-sanitize`).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "engine_internal.h"

// scale = 2: the variant is chosen for the half-chip view (2*rem_code, 2*code_step, 2*spacing against tables of twice the length).
struct ItemRules {
    int n_slots, lut_stride, n_taps;
    int64_t iq_capacity;
    double scale, smin, smax, s_anchor, sp0, sp2;
    bool want_s12;
};
struct ItemStats {                 // of the items seen so far
    int first_bad, bad_code;       // lowest index of an invalid item (INT_MAX: none) and what is wrong with it
    int maxlen;
    unsigned long long max_step_bits, min_step_bits;   // (positive doubles order as integers)
    int m_lo, m_hi;                // whole samples per chip of the anchor tap's line: lowest and highest of the list
    int all_split;                 // every item's outer taps switch chips floor(M / 2) samples into the anchor's block (with margin)
};
enum ItemError { ITEM_OK = 0, ITEM_SLOT, ITEM_SAMPLES, ITEM_START, ITEM_NCO, ITEM_REPLICA };

// What is wrong with one item (ITEM_OK: nothing) and its share of the list's statistics.  lo / hi: its code phase range.
__host__ __device__ inline int check_item(const sdr_epl_item& it, const ItemRules& r, const int32_t* code_len, int& maxlen,
                                          double& step_scaled, int& m_chip, bool& split, double& lo, double& hi) {
    // (before anything is derived from them: a zero n_samples or a NaN code_step would be cast to an integer below)
    if (it.code_slot < 0 || it.code_slot >= r.n_slots || code_len[it.code_slot] <= 0) return ITEM_SLOT;
    if (it.n_samples <= 0 || it.n_samples > r.iq_capacity) return ITEM_SAMPLES;
    if (it.start_sample < 0) return ITEM_START;
    // (finite: x - x is 0 for every finite x and NaN otherwise -- one spelling for both sides)
    auto finite = [](double x) { return x - x == 0.0; };
    if (!(it.code_step > 0.0) || !finite(it.code_step) || !finite(it.rem_code) || !finite(it.rem_carrier) || !finite(it.carrier_hz))
        return ITEM_NCO;
    // the code phase range against the staged replica -- before anything is converted to an integer: a finite but absurd
    // rem_code (1e9 chips) would overflow the fixed-point conversions below (undefined on the host: `make check-sanitize`)
    step_scaled = r.scale * it.code_step;
    lo = ceil(it.rem_code + r.smin);
    hi = ceil(it.code_step * (double)it.n_samples + it.rem_code + r.smax);
    const int reach = r.lut_stride - SDR_LUT_PAD - 2;  // largest padded index the staged row serves
    if (!(lo >= -(double)SDR_LUT_PAD) || !(hi <= (double)reach)) return ITEM_REPLICA;
    maxlen = (int)hi;
    {   // samples per chip of the anchor tap's np.linspace step, exactly as the kernel derives it (correlator_chip.h)
        const double two32 = 4294967296.0;
        const double nd = (double)it.n_samples;
        auto line = [&](double spc, double& sh, double& inv) {
            sh = r.scale * it.rem_code + spc;
            double stop = (r.scale * it.code_step) * nd;
            stop = stop + sh;
            inv = 1.0 / ((stop - sh) / nd);
        };
        double sh, inv;
        line(r.s_anchor, sh, inv);
        const bool in_range = inv >= 1.0 && inv < 1024.0;   // (samples per chip; false for NaN / Inf as well)
        const int64_t tfx = in_range ? (int64_t)rint(inv * two32) : 0;
        m_chip = (int)(tfx >> 32);
        // (the straight-line kernels: the outer taps switch KS = floor(M / 2) samples into the anchor's block -- 12.x at 24.x
        // samples per chip, 9.x at 19.x)
        const int ks = m_chip >> 1;
        split = r.want_s12 && m_chip >= 2;
        for (int t = 0; split && t < 3; t += 2) {
            // the kernel's own derivation of the switch offset (a Q32.32 sample count); its reciprocal is a Newton
            // step, not a division, so a margin of 2^-10 sample keeps the two from disagreeing about the integer part
            double sht, invt;
            line(r.scale * (t ? r.sp2 : r.sp0), sht, invt);
            const int64_t ufx = (int64_t)floor(-sh * inv * two32), ut = (int64_t)floor(-sht * invt * two32);
            const int j = (int)ceil(sht - sh) - 1;
            int64_t d = (ut - ufx) + (int64_t)(1 + j) * tfx;
            if (d < 0) d += tfx; else if (d >= tfx) d -= tfx;
            const int64_t margin = (int64_t)1 << 22;
            split = d >= ((int64_t)ks << 32) + margin && d < ((int64_t)(ks + 1) << 32) - margin;
        }
    }
    return ITEM_OK;
}

