// Chip-aligned variant of the E/P/L correlator core (ci8 rings, code steps of 1/26 .. 1/16 chip per sample:
// GPS L1 C/A between ~16.4 and ~26.5 MHz, i.e. the 25 MHz of the headline configuration; BOC(1,1) half-chip codes
// at 50 MHz).
//
// The boundary variant of correlator.h gives a lane 16 consecutive SAMPLES: the chip switch of every tap falls
// somewhere inside, so all 16 running sums go through LDS (one 16-byte store per sample) and every tap pays its
// switch arithmetic per 16 samples.  Here a lane owns one whole CHIP of the centre ("anchor") tap -- the samples
// i with ceil(y_a(i)) == q, 24 or 25 of them at 25 MHz -- so
//   * the anchor tap sees no switch at all inside a lane's block;
//   * every other tap switches exactly once per block, and because all taps advance by the same step the switch
//     position is the same in every lane up to +1: split_t in {m_t, m_t + 1}, block length n in {M, M + 1}
//     with wave-uniform m_t, M;
//   * so only the running sums at those 2*NT positions are kept (a uniform branch around one LDS store at ~6 of
//     the 25 sample positions), and a tap's share of the block is
//          c(p+1) * P_n + (c(p) - c(p+1)) * P_split       p = the tap's chip at the block's first sample
//     with P_split read back from the lane's strip at a per-lane slot.
// Per sample that leaves the unavoidable part -- int8 -> fp64 (4 instructions) and the complex mix into the running
// sum (4 FMAs) -- plus ~5 instructions of amortised block overhead, against ~17.8 for the boundary variant.
//
// Exactness: a block boundary is "the first sample whose reference chip index exceeds q", with the reference's
// y(i) = fl(fl(i*step) + shift) (np.linspace).  It is PREDICTED as floor((q - shift)/step) + 1; the
// prediction can only be off when the crossing lies within ~1e-11 of a sample, and whenever any lane of the wave
// is within 2^-16 of one the wave re-derives the boundary from exact evaluations of y on both sides (rare branch).
// A lane that ever finds a split or a length outside its {m, m+1} pair (taps exactly aligned with the anchor whose
// rounding jitters, degenerate steps) raises a flag: the caller then redoes the whole epoch with the per-sample
// routine.  First and last (partial) chips of the epoch are edge samples (one per lane, per-sample arithmetic).
#pragma once

#include "correlator.h"

#pragma clang fp contract(off)

namespace sdr {

constexpr int kChipMax = 26;                      // samples a lane's block may hold (k = 0..25)
constexpr double kChipMinCodeStep = 1.0 / 25.9;   // blocks of at most 26 samples
constexpr double kChipMaxCodeStep = 1.0 / 15.5;   // (16.368 MHz is 16.0 samples per chip: Doppler must not decide the kernel; 31-32 MHz through the half-chip view)
constexpr int kChipRawDwords = 13;                // 52 bytes: 26 samples
// per WAVE, double2 slots of LDS behind the lanes' strips: the in-block rotations exp(-1j*k*dphi), k = 0..25, and (round 6, the
// forms with run-time positions) what the biased conversion's offset puts into a running sum of k samples, k = 0..26
constexpr int kChipRotSlots = 2 * kChipMax + 2;
template <int NT>
constexpr int chip_strip_slots() { return 2 * NT + 1; }   // double2 slots per lane (odd multiple of 16 B: conflict-free)

// Predicted first sample i with y(i) > thr on the line y = i*step + shift.
__host__ __device__ __forceinline__ int chip_first_above(double thr, double shift, double inv_step, bool& near) {
    const double u = (thr - shift) * inv_step;
    const double fl = floor(u);
    const double fr = u - fl;
    near = near || fr < kNearInteger || fr > 1.0 - kNearInteger;
    return (int)fl + 1;
}
// The same from exact evaluations of the reference expression around a prediction c.
__host__ __device__ __forceinline__ int chip_first_above_exact(int c, double thr, double step, double shift) {
    auto above = [&](int i) {
        double y = (double)i * step;   // reference arithmetic: separate multiply and add
        y = y + shift;
        return y > thr;
    };
    if (c >= 1 && above(c - 1)) return (c >= 2 && above(c - 2)) ? c - 2 : c - 1;
    if (c < 0) c = 0;
    if (above(c)) return c;
    return above(c + 1) ? c + 1 : c + 2;
}

// Compile-time loop: f(integral_constant<int, I>) for I in [B, E).
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ int wave_min_i32(int x) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_xor(x, off, 64);
        x = o < x ? o : x;
    }
    return __builtin_amdgcn_readfirstlane(x);
}

// int8 -> fp64 in ONE instruction, with a known offset.  v_perm_b32 drops a sample byte (sign bit flipped: u = x + 128;
// a ci8 ring holds its bytes that way, correlator.h kCi8Flip)
// into bits 8..15 of the high word 0x40B0_0000 of a double whose low word is zero: that double is 4096 + u = 4224 + x,
// exactly.  The straight-line kernels mix THESE into their running sums (against 2 instructions for
// v_bfe_i32 + v_cvt_f64_i32: the conversion was half of the 8-instruction floor per sample) and take the offset's share
// 4224 * (1 + 1j) * sum_k r_k out where a sum is read: the in-block rotations r_k are per-epoch constants, so that share
// is three complex scalars per epoch.  The running sums carry ~2^16 instead of ~2^9 while a half block is summed, i.e. their
// roundings are ~2^7 larger: ~1e-13 relative on an epoch's accumulators (measured; the bar is 1e-6).
#ifndef SDR_BIASED_CVT
#define SDR_BIASED_CVT 1
#endif
constexpr double kCvtBias = 4224.0;
// The double lives in a register PAIR whose low word stays zero for the whole epoch: only the high word is rewritten
// (built from a fresh zero each time, the compiler spends a v_mov on every low word and nothing is gained).
typedef uint32_t sdr_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double biased_sample(sdr_u32x2& pair, uint32_t w_flipped, uint32_t selector, uint32_t hi_const) {
    pair.y = __builtin_amdgcn_perm(w_flipped, hi_const, selector);
    return __builtin_bit_cast(double, pair);
}
// selector of v_perm_b32 for byte `byte` of the sample dword: result = [0x40][0xB0][that byte][0x00]
constexpr uint32_t cvt_selector(int byte) { return 0x03020000u | ((uint32_t)(4 + byte) << 8) | 0x0Cu; }

// What the chip-aligned routine needs of an epoch besides its samples and its rotations: wave-uniform integers that
// follow from the taps' np.linspace constants alone.  The kernels with run-time positions derive them per epoch; for
// the straight-line ones the HOST does, when a plan is made (epl.hip: sdr_epl_plan_create) -- the same function, the same
// IEEE operations -- and a wave reads them with scalar loads: ~300 vector and ~200 scalar instructions less per epoch
// (of ~5000: int64 conversions, the exact boundary evaluations of the first and last chip, a 64-bit modulo).
template <int NT>
struct ChipGeom {
    int64_t Tfx, Ufx;           // samples per chip and the anchor line's offset U = -shift/step, Q32.32
    uint64_t delta[NT];         // tap t's switch, in samples after the block start (Q32.32)
    int q0, F;                  // chips q0 + 1 .. q0 + F of the anchor tap are whole
    int head_end, tail_start;   // samples [0, head_end) and [tail_start, n) belong to the partial first and last chip
    int m[NT], J[NT];           // integer part of delta_t; the chip q + J_t tap t sits on at the start of anchor chip q
    int bad;                    // the uniform-position scheme does not cover this epoch
};

template <int NT, int KM, int KS, int KI>
__host__ __device__ __forceinline__ void chip_geometry(int n, const double* shift, const double* step, const double* inv_step,
                                                       ChipGeom<NT>& g) {
    constexpr int A = NT / 2;
    const double two32 = 4294967296.0;
    g.q0 = (int)ceil(shift[A]);
    double y_last = (double)(n - 1) * step[A];
    y_last = y_last + shift[A];
    const int q_last = (int)ceil(y_last);
    g.F = q_last - g.q0 - 1;
    g.head_end = n, g.tail_start = n;
    if (g.F > 0) {
        bool nr = false;
        g.head_end = chip_first_above_exact(chip_first_above((double)g.q0, shift[A], inv_step[A], nr), (double)g.q0, step[A], shift[A]);
        g.tail_start = chip_first_above_exact(chip_first_above((double)(q_last - 1), shift[A], inv_step[A], nr),
                                              (double)(q_last - 1), step[A], shift[A]);
    }
    g.Tfx = (int64_t)rint(inv_step[A] * two32);
    g.Ufx = (int64_t)floor(-shift[A] * inv_step[A] * two32);
    const int M = (int)(g.Tfx >> 32);                                   // block length M or M + 1
    bool bad = M < 1 || M + 1 > kChipMax || g.F > 16384 || (KM != 0 && M != KM);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        g.m[t] = M, g.J[t] = 0, g.delta[t] = 0;
        if (t == A) continue;
        // tap t sits on chip q + J_t at the block start of anchor chip q; its switch to q + J_t + 1 comes
        // delta_t >= 0 samples later (delta_t < T)
        const int64_t Ut = (int64_t)floor(-shift[t] * inv_step[t] * two32);
        int j = (int)ceil(shift[t] - shift[A]) - 1;
        int64_t d = (Ut - g.Ufx) + (int64_t)(1 + j) * g.Tfx;
        if (d < 0) {
            d += g.Tfx;
            ++j;
        } else if (d >= g.Tfx) {
            d -= g.Tfx;
            --j;
        }
        bad = bad || d < 0 || d >= g.Tfx;
        g.J[t] = j;
        g.delta[t] = (uint64_t)d;
        g.m[t] = (int)(d >> 32);
        bad = bad || (KS != 0 && (g.m[t] != KS || j != (t < A ? -1 : 0)));   // (KS: the tap sits on chip q - 1 / q at the block start)
        // (KI: on chip q + (t - A) - 1 and switching at the block's first sample, or on q + (t - A) until its end)
        bad = bad || (KI != 0 && !((j == (t - A) * KI - 1 && g.m[t] == 0) || (j == (t - A) * KI && g.m[t] >= M)));
    }
    g.bad = bad ? 1 : 0;
}

// The carrier rotations of a straight-line kernel's epoch: exp(-1j*k*dphi) inside a half block (k = 1 .. kStaticHalf),
// over the D / D + 1 samples between a lane's blocks, and the biased conversion's share of a sum of kStaticHalf - 2 / - 1 /
// - 0 samples (correlate_epoch_chip).  Evaluated by the host with the kernels' own sincos_reduced -- fused multiply-adds
// and exact roundings on both sides, i.e. the same bits a wave would get -- and the same order of additions as the
// wave's DPP prefix sum.
constexpr int kStaticHalf = 13;                 // the longest half block a straight-line kernel sums: KS + 1 = (KM + 2) / 2 for KM = 24, KS = 12
// The half block of a straight-line geometry -- samples 0 .. KS (taps switching inside the block) or (KM + 2) / 2 (whole-chip
// taps) -- and the shortest sum whose share of the conversion's offset is kept: KM - half samples (the second half up to
// the M-th sample), then + 1, + 2.
constexpr int chip_half(int KM, int KS, int KI) { return KI != 0 ? (KM + 2) / 2 : KS + 1; }
struct ChipRot {
    double urc[kStaticHalf + 1], urs[kStaticHalf + 1];
    double rd0c, rd0s, rd1c, rd1s;
    double biasc[3], biass[3];
};
// half: chip_half() of the kernel's geometry; cmin = KM - half.
__host__ __device__ inline void chip_rotations(double dphi, int Dmin, ChipRot& r, int half = kStaticHalf, int cmin = kStaticHalf - 2) {
    double cs[16], sn[16];
    for (int k = 0; k < 16; ++k) sincos_reduced(-(double)k * dphi, &sn[k], &cs[k]);
    for (int k = 1; k <= half; ++k) r.urc[k] = cs[k], r.urs[k] = sn[k];
    r.urc[0] = 1.0, r.urs[0] = 0.0;
    sincos_reduced(-(double)Dmin * dphi, &r.rd0s, &r.rd0c);
    sincos_reduced(-(double)(Dmin + 1) * dphi, &r.rd1s, &r.rd1c);
    for (int st = 1; st < 16; st <<= 1)          // inclusive prefix sums as a row of 16 lanes forms them (row_shr:1, 2, 4, 8)
        for (int l = 15; l >= st; --l) cs[l] += cs[l - st], sn[l] += sn[l - st];
    for (int i = 0; i < 3; ++i) {
        const int l = cmin + i - 1;                 // (a sum of cmin + i samples: rotations 0 .. l)
        const double re = cs[l] - sn[l], im = cs[l] + sn[l];
        r.biasc[i] = re * 4224.0, r.biass[i] = im * 4224.0;
    }
}

// One epoch of a plan as the straight-line kernels read it (device memory, one per item).
template <int NT>
struct ChipSetup {
    double dphi;                                // carrier_step(carrier_hz, fs)
    double shift[NT], step[NT], inv_step[NT];   // the taps' np.linspace constants (compute_tap_constants)
    int64_t base;                               // start_sample % capacity; < 0: the chip-aligned routine does not apply (chip_variant_applies)
    ChipGeom<NT> g;
    ChipRot r;
};

// ... and how one is made (one THREAD per item of a plan: epl.hip's chip_setup_kernel; the host builds of the tests).
// stride: the lanes that share an epoch (a lane's blocks are that many chips apart).
template <int NT, int KM, int KS, int KI>
__host__ __device__ inline void chip_setup(int n, int64_t start_sample, int64_t capacity, double carrier_hz, double rem_code,
                                           double code_step, const double* spacing, double fs, int stride, ChipSetup<NT>& S) {
    S = ChipSetup<NT>{};
    S.dphi = carrier_step(carrier_hz, fs);
    const double nd = (double)n;
    for (int t = 0; t < NT; ++t) {             // compute_tap_constants(): np.linspace(shift, code_step*n + shift, n, endpoint=False)
        const double shift = rem_code + spacing[t];
        double stop = code_step * nd;
        stop = stop + shift;
        const double delta = stop - shift;
        S.shift[t] = shift;
        S.step[t] = delta / nd;
        S.inv_step[t] = 1.0 / S.step[t];       // (only ever predicts positions that are re-checked exactly near a sample)
    }
    const int64_t base = start_sample % capacity;
    const bool applies = code_step >= kChipMinCodeStep && code_step <= kChipMaxCodeStep && base + n + 32 <= capacity;   // chip_variant_applies()
    S.base = applies ? base : -1;
    if (applies) {
        chip_geometry<NT, KM, KS, KI>(n, S.shift, S.step, S.inv_step, S.g);
        chip_rotations(S.dphi, (int)(((int64_t)stride * S.g.Tfx) >> 32), S.r, chip_half(KM, KS, KI), KM - chip_half(KM, KS, KI));
    }
}

// One lane's block, prepared one round ahead of its use (its loads are in flight while the previous block computes).
template <int NT>
struct ChipBlock {
    uint32_t raw[kChipRawDwords];
    int S;             // first sample of the block (epoch-relative)
    int dn;            // block length - M            (0 or 1)
    int ds[NT];        // per tap: switch position - m_t (0 or 1; anchor: unused)
};

// Returns false when a lane met a configuration the uniform-position scheme does not cover: the caller redoes the
// epoch with correlate_epoch().  strip: this wave-group's [stride][chip_strip_slots<NT>()] double2; rot: kChipRotSlots double2
// private to the WAVE.
//
// Block boundaries come from a fixed-point line (Q32.32 samples): with T = 1/step samples per chip and
// U = -shift/step, the first sample whose reference y exceeds thr is floor(U + thr*T) + 1.  One 64-bit add per
// boundary per lane; a boundary whose fraction lies within 2^-16 of an integer (where fixed point, the real line
// and the reference's rounded y might disagree) sends the wave through exact evaluations of the reference
// expression instead.  Because every tap advances by the same T, the switch of tap t sits delta_t = const samples
// after the block start on the real line, so its sample position is floor(delta_t) or floor(delta_t) + 1 in every
// lane (floor(a + b) - floor(a) for b >= 0): the two LDS slots per tap are known for the whole epoch.
// KM > 0: the block length M is known when the kernel is compiled (the host checked that every epoch of the launch
// has floor(samples per chip) == KM, e.g. 24 at 25 MHz): the sample loop then has a fixed trip count and the stores
// of P_M and P_(M+1) are unconditional -- two scalar instructions per sample (bit test + branch for a tap position)
// instead of six with two branches (measured: 0.436 -> 0.40 ms per launch; testing two samples per branch with
// both bodies duplicated was slower again: 0.415).
// KS > 0 (with KM > 0): every other tap's switch position m_t is known too (KS = 12: taps half a chip either side of
// the anchor at 24.4 samples per chip -- the reference's default correlator spacing at 25 MHz; the host checked it for
// every epoch of the launch, and an epoch that disagrees is flagged and redone per sample).  All four positions
// somebody reads -- P_KS, P_(KS+1), P_KM, P_(KM+1) -- are then compile-time: they stay in registers, the sample loop
// is one straight line (no strip stores, no bit tests, no branches) and a lane picks its pair member with selects.
// KI = 1 (with KM > 0): the taps sit whole chips apart -- tap t on chip q + (t - A) for ALL of the anchor's block of
// chip q (the five taps VE/E/P/L/VL at -1, -0.5, 0, +0.5, +1 chip of BASELINE configs 4-5 on the half-chip view of
// their replicas: -2 .. +2 half-chips at 24.4 samples per half-chip).  No tap switches inside a block, so the block
// needs ONE sum (P_M or P_(M+1)), turned by the block-start phasor once and added into every tap with that tap's
// replica value: two FMAs per tap and block, no strip, no per-tap fixed-point arithmetic.  The host checks the spacing;
// a block in which the reference's rounding lets a tap switch a sample early or late (only possible within 2^-16 of
// a sample, where the wave evaluates the reference expression exactly anyway) flags the epoch, which is redone per
// sample.
// G: chip_geometry() of the epoch, computed by the caller or read from the plan's ChipSetup; base = start_sample % capacity;
// R (straight-line forms only): the plan's chip_rotations() of the epoch.
template <int NT, bool SINGLE_WAVE, int KM = 0, int KS = 0, int KI = 0>
__device__ __forceinline__ bool correlate_epoch_chip(const void* __restrict__ ring, const void* __restrict__ ring_flipped, int64_t capacity,
                                                     const EpochParams& ep, double dphi, const EpochConsts<NT>& K,
                                                     const ChipGeom<NT>& G, int64_t base, const ChipRot* R,
                                                     const uint32_t* lut, double2* strip_lds, double2* rot, int tid,
                                                     int lane, int stride, int edge_lane, double* accr, double* acci) {
    constexpr int A = NT / 2;                       // anchor tap: the centre one (prompt)
    constexpr int kSlots = chip_strip_slots<NT>();
    const int n = ep.n;
    const double* shift = K.shift;
    const double* step = K.step;
    double2* strip = strip_lds + tid * kSlots;
    const int wlane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;

    // ---- epoch geometry (wave-uniform): chips q0 .. q_last of the anchor tap; q0+1 .. q_last-1 are whole
    const double dphi_u = uniform(dphi), rem_carrier_u = uniform(ep.rem_carrier);
    const int q0 = __builtin_amdgcn_readfirstlane(G.q0);
    const int F = __builtin_amdgcn_readfirstlane(G.F);                  // whole chips
    const int head_end = __builtin_amdgcn_readfirstlane(G.head_end);
    const int tail_start = __builtin_amdgcn_readfirstlane(G.tail_start);
    // (the caller guarantees base + n + 32 <= capacity)
    // (the straight-line forms build their samples from the sign-flipped image of the ring: see biased_sample())
    // (round 6: the ring itself holds the sign-flipped bytes -- `ring_flipped` is the same pointer, kept in the signature)
    (void)ring_flipped;
    const char* ring_base = static_cast<const char*>(ring) + base * 2;

    // in-block rotations exp(-1j*k*dphi), k = 0..25: one per lane, parked in LDS, read back as broadcasts
    // (KS: the block is summed in two halves of KS + 1 and KM - KS samples that both start at rotation 0, so only
    // k = 1 .. KS + 1 are needed and they fit in scalar registers -- no LDS reads in the sample loop at all)
    static_assert(!(KS != 0 && KI != 0), "either the taps switch inside the block (KS) or with it (KI)");
    constexpr bool kStatic = KM != 0 && (KS != 0 || KI != 0);
    constexpr int kHalf = chip_half(KM, KS, KI);
    double urc[kStatic ? kHalf + 1 : 1], urs[kStatic ? kHalf + 1 : 1];
    double biasc[3] = {0.0, 0.0, 0.0}, biass[3] = {0.0, 0.0, 0.0};   // (biased conversion) the offset's share of a sum of kHalf - 2 / - 1 / - 0 samples
    // samples per chip as Q32.32, and the distance to a lane's next block: D or D + 1 samples
    const double two32 = 4294967296.0;
    int64_t Tfx = G.Tfx;
    int64_t stride_fx = (int64_t)stride * Tfx;
    const int Dmin = (int)(stride_fx >> 32);
    // (read from the plan these arrive in scalar registers, which the rotations of the sample loop need: the per-lane
    // 64-bit arithmetic of the block boundaries takes them from vector registers, as when a wave derives them itself)
    if constexpr (kStatic) asm volatile("" : "+v"(Tfx), "+v"(stride_fx));
    double rd0c = 1.0, rd0s = 0.0, rd1c = 1.0, rd1s = 0.0;  // the carrier rotations over D and D + 1 samples
    {
        double sn = 0.0, cs = 0.0;
        if constexpr (kStatic) {
            // (worked out by the host with the plan: chip_rotations())
            static_assert(kHalf <= kStaticHalf, "the plan's rotations are laid out for half blocks of up to 13 samples");
#pragma unroll
            for (int k = 1; k <= kHalf; ++k) urc[k] = R->urc[k], urs[k] = R->urs[k];
            rd0c = R->rd0c, rd0s = R->rd0s, rd1c = R->rd1c, rd1s = R->rd1s;
            if constexpr (SDR_BIASED_CVT) {
#pragma unroll
                for (int i = 0; i < 3; ++i) biasc[i] = R->biasc[i], biass[i] = R->biass[i];
            }
        } else {
            if (wlane < kChipMax) {
                sincos_reduced(-(double)wlane * dphi_u, &sn, &cs);
                rot[wlane] = make_double2(cs, sn);
            }
            if constexpr (SDR_BIASED_CVT) {
                // Round 6: the ring holds sign-flipped bytes, so these forms take the one-instruction conversion too (a
                // sample arrives as 4224 + x: two v_perm_b32 instead of an exclusive or, two bit-field extracts and two
                // converts per sample).  A running sum P_k then carries 4224 (1 + 1j) sum_{i<k} r_i: btab[k], an inclusive
                // scan of the lanes' rotations, taken out where a parked sum is read.
                double br = (cs - sn) * kCvtBias, bi = (cs + sn) * kCvtBias;       // (lanes >= 26 hold zeros)
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    const double o_r = __shfl_up(br, off, 64), o_i = __shfl_up(bi, off, 64);
                    br = wlane >= off ? br + o_r : br;
                    bi = wlane >= off ? bi + o_i : bi;
                }
                double2* const btab = rot + kChipMax;
                if (wlane < kChipMax) btab[wlane + 1] = make_double2(br, bi);
                if (wlane == 0) btab[0] = make_double2(0.0, 0.0);
            }
        }
    }

    bool bad = false;
    const int rounds = F > 0 ? (F + stride - 1) / stride : 0;
    if (rounds > 0) {
        // ---- the fixed-point line of the anchor tap, and each tap's constant offset on it
        int64_t Ufx = G.Ufx;
        const int M = (int)(G.Tfx >> 32);                                 // block length M or M + 1
        if constexpr (kStatic) asm volatile("" : "+v"(Ufx));
        bad = bad || G.bad != 0;
        int m[NT], J[NT];
        uint64_t delta[NT];
        unsigned evmask = (1u << M) | (2u << M);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            m[t] = G.m[t], J[t] = G.J[t], delta[t] = G.delta[t];
            if constexpr (kStatic) asm volatile("" : "+v"(m[t]), "+v"(J[t]), "+v"(delta[t]));
            if (t != A) evmask |= (1u << m[t]) | (2u << m[t]);
        }
        const int k_last = M + 1;                       // no prefix beyond P_(M+1) is ever read
        int rank[NT];                                   // strip slot of position m_t (m_t + 1 sits in the next one)
#pragma unroll
        for (int t = 0; t < NT; ++t) rank[t] = __builtin_popcount(evmask & ((1u << m[t]) - 1u));
        if constexpr (!kStatic) {
            double sn, cs;
            sincos_reduced(-(double)(Dmin + (wlane & 1)) * dphi_u, &sn, &cs);   // lane 0: Dmin, lane 1: Dmin + 1
            rd0c = lane_value(cs, 0), rd0s = lane_value(sn, 0);
            rd1c = lane_value(cs, 1), rd1s = lane_value(sn, 1);
        }
        asm volatile("" : "+v"(rd0c), "+v"(rd0s), "+v"(rd1c), "+v"(rd1s));   // (selected per lane: keep them in vector registers)
        const uint64_t lane_fx = (uint64_t)((int64_t)lane * Tfx);
        // u of the lane's block start in round 0, "+1 sample" folded in: S = u >> 32
        uint64_t u_cur = (uint64_t)(Ufx + (int64_t)q0 * Tfx + (int64_t)two32) + lane_fx;
        const int last_idx = F - 1;

        auto prepare = [&](int round, uint64_t u0, ChipBlock<NT>& b) {
            // (a lane beyond the last whole chip re-does the last one with a zero carrier phasor)
            const int idx = round * stride + lane;
            const bool inside = idx <= last_idx;
            const uint64_t uS = inside ? u0 : (uint64_t)(Ufx + (int64_t)(q0 + last_idx) * Tfx + (int64_t)two32);
            const uint64_t uE = uS + (uint64_t)Tfx;
            int S = (int)(uS >> 32);
            int E = (int)(uE >> 32);
            bool near = (uint32_t)uS + 0x10000u < 0x20000u || (uint32_t)uE + 0x10000u < 0x20000u;
            int split[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                split[t] = 0;
                if (t == A) continue;
                if constexpr (KI != 0) {
                    // (the tap's switch IS the block's boundary up to a few units of 2^-32 sample: away from `near` it
                    // falls on the same sample, so there is nothing to compute)
                    split[t] = J[t] == (t - A) * KI ? E - S : 0;
                    continue;
                }
                const uint64_t uT = uS + delta[t];
                near = near || (uint32_t)uT + 0x10000u < 0x20000u;
                split[t] = (int)(uT >> 32) - S;
            }
            if (__builtin_expect(__any(near), 0)) {
                const int q = q0 + 1 + (inside ? idx : last_idx);
                S = chip_first_above_exact(S, (double)(q - 1), step[A], shift[A]);
                E = chip_first_above_exact(E, (double)q, step[A], shift[A]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t == A) continue;
                    const int p0 = q + J[t];
                    // the tap must sit on p0 or p0 + 1 at S and switch to p0 + 1 at most once inside the block
                    const int bt = chip_first_above_exact(S + split[t], (double)p0, step[t], shift[t]);
                    double y = (double)S * step[t];
                    y = y + shift[t];
                    const int at_s = (int)ceil(y);
                    bad = bad || (at_s != p0 && at_s != p0 + 1) || (at_s == p0 + 1 && bt > S);
                    split[t] = bt < S ? 0 : bt - S;
                }
            }
            b.S = S;
            const int nn = E - S;
            b.dn = nn - M;
            bad = bad || (unsigned)b.dn > 1u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                b.ds[t] = 0;
                if (t == A) continue;
                const int sp = split[t] > nn ? nn : split[t];
                if constexpr (KI != 0) {
                    bad = bad || !(J[t] == (t - A) * KI ? sp == nn : sp == 0);   // on chip q + (t - A) for the whole block
                    continue;
                }
                b.ds[t] = sp - m[t];
                bad = bad || (unsigned)b.ds[t] > 1u;
            }
            // (KS, three taps: E and L switch half a chip into the block, on the same sample away from a `near` one -- one
            // select serves both; a block where they differ flags the epoch)
            if constexpr (KS != 0 && NT == 3) bad = bad || b.ds[0] != b.ds[2];
            // 13 dwords = 26 samples from the (2-byte aligned) address of sample S: gfx950 serves unaligned dword loads
            const char* src = ring_base + (int64_t)S * 2;
            const uint4 w0 = *reinterpret_cast<const uint4*>(src);
            const uint4 w1 = *reinterpret_cast<const uint4*>(src + 16);
            const uint4 w2 = *reinterpret_cast<const uint4*>(src + 32);
            const uint32_t w3 = *reinterpret_cast<const uint32_t*>(src + 48);
            b.raw[0] = w0.x, b.raw[1] = w0.y, b.raw[2] = w0.z, b.raw[3] = w0.w;
            b.raw[4] = w1.x, b.raw[5] = w1.y, b.raw[6] = w1.z, b.raw[7] = w1.w;
            b.raw[8] = w2.x, b.raw[9] = w2.y, b.raw[10] = w2.z, b.raw[11] = w2.w;
            b.raw[12] = w3;
        };

        ChipBlock<NT> blk_a, blk_b;
        prepare(0, u_cur, blk_a);
        // carrier phase at the lane's first block: one exact evaluation; later blocks by a fixed rotation
        double sb, cb;
        sincos_reduced(__builtin_fma(-(double)blk_a.S, dphi_u, rem_carrier_u), &sb, &cb);
        if constexpr (!kStatic) {
            sb = lane <= last_idx ? sb : 0.0;
            cb = lane <= last_idx ? cb : 0.0;
        }
        // per-lane LDS addresses: the strip slots of each tap's position m_t, and the lane's replica entry
        const int q_lane = q0 + 1 + lane + SDR_LUT_PAD;
        // (KS / KI) the lane's own strip slots are not used for running sums: eight zero words stand in for the replica of a
        // lane that has no block in a round (the last one)
        const uint32_t* zero_lq = reinterpret_cast<const uint32_t*>(strip) + 3;   // (words 1 .. 5 serve entries -2 .. +2)
        if constexpr (kStatic) strip[0] = strip[1] = make_double2(0.0, 0.0);

        sdr_u32x2 zI = {0u, 0u}, zQ = {0u, 0u};     // (biased conversion) the two register pairs the samples are built in
        asm volatile("" : "+v"(zI), "+v"(zQ));

        auto process = [&](const ChipBlock<NT>& b, int round, double sbk, double cbk) {
            double pr = 0.0, pi = 0.0;
            double2* wp = strip;
            // the event positions as ONE scalar register, re-read per block: left to itself the compiler hoists the 27
            // bit tests out of the round loop as 27 lane masks and spills them through v_writelane / v_readlane
            unsigned evm = evmask;
            int klast = k_last;
            if constexpr (!kStatic) asm volatile("" : "+s"(evm), "+s"(klast));
            // the in-block rotations come from LDS as broadcasts, fetched a few samples ahead of their use (the
            // uniform branches below end a basic block: nothing is hoisted across them for us)
            constexpr int kAhead = 4;
            double2 rr[kChipMax + kAhead];
            if constexpr (!kStatic) {
#pragma unroll
                for (int k = 0; k < kAhead; ++k) rr[k] = rot[k];
            }
            auto park = [&]() {                         // P_k -> the lane's next strip slot
                *wp = make_double2(pr, pi);
                ++wp;
                asm volatile("" ::: "memory");
            };
            uint32_t hi_rt = 0x40B00000u;                 // (run-time-position forms: the biased conversion's high-word constant)
            // ... and its four byte selectors as scalar registers for the whole block: left as literals each costs the scalar
            // pipe a move in front of EVERY v_perm_b32 (two per sample: measured, + 41 % scalar instructions in the dense
            // closed-loop kernel and no gain from 6 % fewer vector ones)
            uint32_t sel_rt[4] = {cvt_selector(0), cvt_selector(1), cvt_selector(2), cvt_selector(3)};
            if constexpr (!kStatic && SDR_BIASED_CVT) {
                asm volatile("" : "+v"(hi_rt));
                asm volatile("" : "+s"(sel_rt[0]), "+s"(sel_rt[1]), "+s"(sel_rt[2]), "+s"(sel_rt[3]));
            }
            auto sample = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                double ar, ai;
                if constexpr (SDR_BIASED_CVT) {
                    ar = biased_sample(zI, b.raw[k >> 1], sel_rt[(k & 1) ? 2 : 0], hi_rt);
                    ai = biased_sample(zQ, b.raw[k >> 1], sel_rt[(k & 1) ? 3 : 1], hi_rt);
                } else {
                    const int w = ci8_native((int)b.raw[k >> 1]);
                    ar = (k & 1) ? (double)(int)(int8_t)(w >> 16) : (double)(int)(int8_t)w;
                    ai = (k & 1) ? (double)(w >> 24) : (double)(int)(int8_t)(w >> 8);
                }
                if (k + kAhead < kChipMax) rr[k + kAhead] = rot[k + kAhead];
                const double2 r = rr[k];
                pr = __builtin_fma(-ai, r.y, __builtin_fma(ar, r.x, pr));
                pi = __builtin_fma(ai, r.x, __builtin_fma(ar, r.y, pi));
                if constexpr (SDR_BIASED_CVT) asm volatile("" : "+v"(pr), "+v"(pi), "+v"(zI), "+v"(zQ));   // (see the straight-line loop below)
            };
            double capr[3] = {0.0, 0.0, 0.0}, capi[3] = {0.0, 0.0, 0.0};   // KS: P_KS, second half before its last sample, second half
            if constexpr (kStatic) {
                static_assert(2 * kHalf >= KM + 1 && kHalf < KM, "two halves of at most KS + 1 samples cover the block");
                // first half: samples 0 .. KS (P_KS is its running sum before the last one, P_(KS+1) its total);
                // second half: samples KS+1 .. KM summed from rotation 0 again, turned by exp(-1j*(KS+1)*dphi) where read
                static_assert(!SDR_BIASED_CVT || (2 * kHalf - KM >= 1 && 2 * kHalf - KM <= 2 && (KS == 0 || KS == kHalf - 1)),
                              "the offset's shares are kept for sums of KM - kHalf, + 1 and + 2 samples: kHalf must be one of the last two");
                uint32_t flipped[kChipRawDwords];
                uint32_t hi_const = 0x40B00000u;
                if constexpr (SDR_BIASED_CVT) {
                    asm volatile("" : "+v"(hi_const));             // (v_perm_b32 takes one scalar operand: the selector)
#pragma unroll
                    for (int i = 0; i < kChipRawDwords; ++i) flipped[i] = b.raw[i];   // (the ring holds the flipped bytes)
                }
                static_for<0, KM + 1>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    constexpr int j = k < kHalf ? k : k - kHalf;
                    const int w = ci8_native((int)b.raw[k >> 1]);
                    double ar, ai;
                    if constexpr (SDR_BIASED_CVT) {
                        ar = biased_sample(zI, flipped[k >> 1], cvt_selector((k & 1) ? 2 : 0), hi_const);
                        ai = biased_sample(zQ, flipped[k >> 1], cvt_selector((k & 1) ? 3 : 1), hi_const);
                    } else {
                        ar = (k & 1) ? (double)(int)(int8_t)(w >> 16) : (double)(int)(int8_t)w;
                        ai = (k & 1) ? (double)(w >> 24) : (double)(int)(int8_t)(w >> 8);
                    }
                    if constexpr (KS != 0 && k == KS) capr[0] = pr, capi[0] = pi;
                    if constexpr (k == KM) capr[1] = pr, capi[1] = pi;
                    if constexpr (k == kHalf) capr[2] = pr, capi[2] = pi;          // first half's total
                    if constexpr (j == 0) {
                        pr = ar, pi = ai;
                    } else {
                        pr = __builtin_fma(-ai, urs[j], __builtin_fma(ar, urc[j], pr));
                        pi = __builtin_fma(ai, urc[j], __builtin_fma(ar, urs[j], pi));
                    }
                    // (an opaque point per sample: the sums must have read the pairs before their high words are written
                    // again, and the next sample must be built on THESE registers -- seen through, every sample would be
                    // rebuilt from the original pair, i.e. from a copy of its low word)
                    if constexpr (SDR_BIASED_CVT) asm volatile("" : "+v"(pr), "+v"(pi), "+v"(zI), "+v"(zQ));
                });
            } else if constexpr (KM != 0) {
                // positions KM and KM + 1 are always events; the others (the taps' m_t, m_t + 1 < KM) are looked for
                // two samples at a time
                static_for<0, KM>([&](auto kc) {         // samples below KM: a tap's position may sit in front of any of them
                    constexpr int k = decltype(kc)::value;
                    if (evm & (1u << k)) park();
                    sample(kc);
                });
                park();                                  // P_KM
                sample(std::integral_constant<int, KM>{});
                park();                                  // P_(KM+1)
            } else {
#pragma unroll
                for (int k = 0; k <= kChipMax; ++k) {
                    if (evm & (1u << k)) {              // (wave-uniform) one of the positions somebody reads: park P_k
                        park();
                        if (k == klast) break;
                    }
                    if (k < kChipMax) {
                        double ar, ai;
                        if constexpr (SDR_BIASED_CVT) {
                            ar = biased_sample(zI, b.raw[k >> 1], sel_rt[(k & 1) ? 2 : 0], hi_rt);
                            ai = biased_sample(zQ, b.raw[k >> 1], sel_rt[(k & 1) ? 3 : 1], hi_rt);
                        } else {
                            const int w = ci8_native((int)b.raw[k >> 1]);
                            ar = (k & 1) ? (double)(int)(int8_t)(w >> 16) : (double)(int)(int8_t)w;
                            ai = (k & 1) ? (double)(w >> 24) : (double)(int)(int8_t)(w >> 8);
                        }
                        if (k + kAhead < kChipMax) rr[k + kAhead] = rot[k + kAhead];
                        const double2 r = rr[k];
                        pr = __builtin_fma(-ai, r.y, __builtin_fma(ar, r.x, pr));
                        pi = __builtin_fma(ai, r.x, __builtin_fma(ar, r.y, pi));
                        if constexpr (SDR_BIASED_CVT) asm volatile("" : "+v"(pr), "+v"(pi), "+v"(zI), "+v"(zQ));
                    }
                }
            }
            double2 ptot;
            if constexpr (kStatic && SDR_BIASED_CVT) {
                // what the offset of 4224 per sample put into each sum that is read (its first sample has rotation 1:
                // the sums start from the biased sample itself, and sum_{k<n} r_k includes that r_0 = 1)
                constexpr int iFirst = 2 * kHalf - KM;             // (biasc[i]: a sum of KM - kHalf + i samples)
                pr -= biasc[1], pi -= biass[1];                    // second half, KM + 1 - kHalf samples
                capr[1] -= biasc[0], capi[1] -= biass[0];          // ... before its last sample
                capr[2] -= biasc[iFirst], capi[2] -= biass[iFirst];      // first half: kHalf samples
                if constexpr (KS != 0) capr[0] -= biasc[iFirst - 1], capi[0] -= biass[iFirst - 1];   // P_KS: kHalf - 1 samples
            }
            if constexpr (kStatic) {
                const double qr = b.dn ? pr : capr[1], qi = b.dn ? pi : capi[1];   // second half up to M or M + 1 samples
                ptot.x = __builtin_fma(-qi, urs[kHalf], __builtin_fma(qr, urc[kHalf], capr[2]));
                ptot.y = __builtin_fma(qi, urc[kHalf], __builtin_fma(qr, urs[kHalf], capi[2]));
            } else {
                ptot = strip[rank[A] + b.dn];
                if constexpr (SDR_BIASED_CVT) {             // (P_M or P_(M+1): the offset's share of that many samples out)
                    const double2 bb = (rot + kChipMax)[M + b.dn];
                    ptot.x -= bb.x, ptot.y -= bb.y;
                }
            }
            const int first = round * stride;           // (a lane beyond the last whole chip re-does the last one)
            const uint32_t* lq;                         // replica entry of the block's anchor chip
            if constexpr (kStatic)                      // (... against three zero words: it adds nothing, whatever its phasor)
                lq = first + lane <= last_idx ? lut + q_lane + first : zero_lq;
            else
                lq = lut + q_lane + (first + lane <= last_idx ? first : last_idx - lane);
            if constexpr (KI != 0) {
                // every tap sees the whole block on one chip: turn the block's sum once, add it per tap
                const double xr = __builtin_fma(-sbk, ptot.y, cbk * ptot.x);
                const double xi = __builtin_fma(sbk, ptot.x, cbk * ptot.y);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double c = __hiloint2double((int)lq[(t - A) * KI], 0);
                    accr[t] = __builtin_fma(c, xr, accr[t]);
                    acci[t] = __builtin_fma(c, xi, acci[t]);
                }
                return;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double gr, gi;
                if (t == A) {
                    const double c = __hiloint2double((int)lq[0], 0);
                    gr = c * ptot.x;
                    gi = c * ptot.y;
                } else {
                    double2 ps;
                    if constexpr (kStatic)
                        ps = b.ds[NT == 3 ? 0 : t] ? make_double2(capr[2], capi[2]) : make_double2(capr[0], capi[0]);
                    else {
                        ps = strip[rank[t] + b.ds[t]];
                        if constexpr (SDR_BIASED_CVT) {
                            const double2 bb = (rot + kChipMax)[m[t] + b.ds[t]];
                            ps.x -= bb.x, ps.y -= bb.y;
                        }
                    }
                    const int jt = kStatic ? (t < A ? -1 : 0) : J[t];      // (KS: checked when the epoch was set up)
                    const double ca = __hiloint2double((int)lq[jt], 0);
                    const double cbn = __hiloint2double((int)lq[jt + 1], 0);
                    const double diff = ca - cbn;
                    gr = __builtin_fma(diff, ps.x, cbn * ptot.x);
                    gi = __builtin_fma(diff, ps.y, cbn * ptot.y);
                }
                accr[t] = __builtin_fma(-sbk, gi, __builtin_fma(cbk, gr, accr[t]));
                acci[t] = __builtin_fma(sbk, gr, __builtin_fma(cbk, gi, acci[t]));
            }
        };

        auto wave_has_work = [&](int round) { return SINGLE_WAVE || round * stride + (lane - wlane) <= last_idx; };
        // rotation to the next block of this lane
        auto advance = [&](const ChipBlock<NT>& from, const ChipBlock<NT>& to, int to_round) {
            const unsigned dd = (unsigned)(to.S - from.S - Dmin);
            const bool alive = to_round * stride + lane <= last_idx;
            bad = bad || (alive && dd > 1u);
            const double rc_ = dd ? rd1c : rd0c, rs_ = dd ? rd1s : rd0s;
            const double cbn = __builtin_fma(cb, rc_, -sb * rs_);
            const double sbn = __builtin_fma(sb, rc_, cb * rs_);
            if constexpr (kStatic) {
                cb = cbn, sb = sbn;                     // (a lane without a block correlates against zero replica words)
            } else {
                cb = alive ? cbn : 0.0;
                sb = alive ? sbn : 0.0;
            }
        };

        for (int it = 0; it < rounds / 2; ++it) {
            const double sb0 = sb, cb0 = cb;
            u_cur += (uint64_t)stride_fx;
            prepare(2 * it + 1, u_cur, blk_b);
            advance(blk_a, blk_b, 2 * it + 1);
            if (wave_has_work(2 * it)) process(blk_a, 2 * it, sb0, cb0);
            const double sb1 = sb, cb1 = cb;
            if (2 * it + 2 < rounds) {
                u_cur += (uint64_t)stride_fx;
                prepare(2 * it + 2, u_cur, blk_a);
                advance(blk_b, blk_a, 2 * it + 2);
            }
            if (wave_has_work(2 * it + 1)) process(blk_b, 2 * it + 1, sb1, cb1);
        }
        if ((rounds & 1) && wave_has_work(rounds - 1)) process(blk_a, rounds - 1, sb, cb);
    }

    if (__any(bad)) return false;
    if (edge_lane >= 0) {
        if (head_end + (n - tail_start) > 64) {
            for (int off = 0; off < head_end + (n - tail_start); off += 64)
                edge_samples<SDR_FMT_CI8, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane + off, head_end, tail_start, accr, acci, base);
        } else {
            edge_samples<SDR_FMT_CI8, NT>(ring, capacity, ep, dphi, shift, step, lut, edge_lane, head_end, tail_start, accr, acci, base);
        }
    }
    return true;
}

// True when the epoch can go through the chip-aligned routine: the code step in its range and no ring wrap within
// the epoch plus the slack the 56-byte windows may touch.
__device__ __forceinline__ bool chip_variant_applies(const EpochParams& ep, int64_t capacity) {
    return ep.code_step >= kChipMinCodeStep && ep.code_step <= kChipMaxCodeStep &&
           ep.start_sample % capacity + ep.n + 32 <= capacity;
}

}  // namespace sdr
