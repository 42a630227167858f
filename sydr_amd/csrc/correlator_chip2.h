// Two chips per lane: the straight-line form of correlator_chip.h with a block of TWO whole chips of the prompt tap, three
// taps half a chip apart, ci8 ring.  Compiled for the boundary positions P0 < P1 < P2 < P3 (each "or one later"):
//      <4, 9, 14, 19>     chips of 9.5 .. 10 samples: a C/A code at the reference's shipped 10 MHz (config/receiver.ini:18-20)
//      <5, 11, 17, 23>    11.5 .. 12 samples: 12 MHz
// (<12, 24, 36, 48>, the headline 25 MHz, works too and was measured: no faster than the one-chip form there.)
// At the low rates one chip is too short for a lane (the per-block work of correlator_chip.h would be paid every ~10
// samples), and 8 consecutive samples per lane (correlator.h's boundary variant) cost ~30 issue slots per sample.  A lane
// owns chips q and q + 1: P3 or P3 + 1 samples in which
//      the early and late tap change chips   P0 | P0 + 1   samples in      (half a chip: E from q - 1 to q, L from q to q + 1)
//      the prompt tap changes chips          P1 | P1 + 1                   (q to q + 1)
//      early and late change again           P2 | P2 + 1                   (E to q + 1, L to q + 2)
//      the block ends                        P3 | P3 + 1
// -- every position wave-uniform up to + 1, as in the one-chip forms.  The block is summed as FOUR segments (samples
// [0, P0], (P0, P1], (P1, P2], (P2, P3]) that each start at rotation 0 (the longest segment's rotations in scalar
// registers); a segment keeps its sum before and after its last sample, a lane picks the member its boundary needs, and the
// prefix sums at the four boundaries follow by turning the segments with exp(-1j*start_g*dphi):
//      Q1 = P_(s1) = sel_0      Q2 = P_(sP) = f_0 + T1*sel_1      Q3 = P_(s2) = f_0 + T1*f_1 + T2*sel_2      Q4 = P_n = ... + T3*sel_3
// With the replica words c(q - 1), c(q), c(q + 1), c(q + 2) the taps' shares of the block are
//      E = (c(q-1) - c(q))*Q1 + (c(q) - c(q+1))*Q3 + c(q+1)*Q4
//      P =                      (c(q) - c(q+1))*Q2 + c(q+1)*Q4
//      L = (c(q) - c(q+1))*Q1 + (c(q+1) - c(q+2))*Q3 + c(q+2)*Q4
// Samples are built with the one-instruction biased conversion from the flipped ring image (correlator_chip.h); what the
// offset puts into a segment sum of L samples is one complex constant per length and epoch.  Block boundaries come from
// the same Q32.32 line; a boundary within 2^-16 of a sample sends the wave through exact evaluations of the reference
// expression; a lane that meets a position outside its pair flags the epoch, which is redone per sample.  Everything
// wave-uniform that is not data -- tap constants, geometry, rotations -- is worked out by the host when the plan is made
// (Chip2Setup, one per item), as for the one-chip forms.
#pragma once

#include "correlator_chip.h"

#pragma clang fp contract(off)

namespace sdr {

template <int P0, int P1, int P2, int P3>
struct Chip2Shape {
    static_assert(0 < P0 && P0 < P1 && P1 < P2 && P2 < P3, "boundary positions in order");
    static constexpr int start(int g) { return g == 0 ? 0 : (g == 1 ? P0 + 1 : (g == 2 ? P1 + 1 : P2 + 1)); }
    static constexpr int last(int g) { return g == 0 ? P0 : (g == 1 ? P1 : (g == 2 ? P2 : P3)); }   // the segment's last sample (when its boundary is "one later")
    static constexpr int len(int g) { return last(g) + 1 - start(g); }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int kMaxLen = cmax(cmax(len(0), len(1)), cmax(len(2), len(3)));
    static constexpr int kSamples = P3 + 1;                     // a block holds P3 or P3 + 1 samples
    static constexpr int kRawDwords = (kSamples + 1) / 2;
    static constexpr int segment_of(int k) { return k <= P0 ? 0 : (k <= P1 ? 1 : (k <= P2 ? 2 : 3)); }
};

template <int P0, int P1, int P2, int P3>
struct Chip2Setup {
    using Shape = Chip2Shape<P0, P1, P2, P3>;
    double dphi;                       // carrier_step(carrier_hz, fs)
    double shift[3], step[3];          // the taps' np.linspace constants (exact re-evaluations, edge samples)
    int64_t base;                      // start_sample % capacity; < 0: this routine does not serve the epoch
    int64_t Tfx, Ufx;                  // samples per chip and the prompt line's offset, Q32.32
    uint64_t delta;                    // the outer taps' first switch, samples after the block start (Q32.32; E's -- L's agrees to 2^-20)
    int q0, F2;                        // block b < F2 holds chips q0 + 1 + 2b and q0 + 2 + 2b of the prompt tap
    int head_end, tail_start;          // samples [0, head_end) and [tail_start, n) are correlated one per lane
    int Dmin, pad;
    double rc[Shape::kMaxLen], rs[Shape::kMaxLen];   // exp(-1j*k*dphi), k < the longest segment
    double tc[4], ts[4];               // exp(-1j*start_g*dphi)
    double rd0c, rd0s, rd1c, rd1s;     // over the Dmin / Dmin + 1 samples to a lane's next block
    double bc[Shape::kMaxLen + 1], bs[Shape::kMaxLen + 1];   // the biased conversion's share of a segment sum of L samples, L <= the longest
};

// Host side (sdr_epl_plan_create): false when the two-chip scheme does not cover the item.
template <int P0, int P1, int P2, int P3>
__host__ inline bool chip2_setup(int n, int64_t start_sample, int64_t capacity, double carrier_hz, double rem_code, double code_step,
                                 const double* spacing, double fs, Chip2Setup<P0, P1, P2, P3>& S) {
    using Shape = Chip2Shape<P0, P1, P2, P3>;
    S = Chip2Setup<P0, P1, P2, P3>{};
    S.base = -1;
    S.dphi = carrier_step(carrier_hz, fs);
    double inv[3];
    const double nd = (double)n;
    for (int t = 0; t < 3; ++t) {
        const double shift = rem_code + spacing[t];
        double stop = code_step * nd;
        stop = stop + shift;
        S.shift[t] = shift;
        S.step[t] = (stop - shift) / nd;
        inv[t] = 1.0 / S.step[t];
    }
    ChipGeom<3> g;
    chip_geometry<3, 0, 0, 0>(n, S.shift, S.step, inv, g);
    const int64_t base = start_sample % capacity;
    const int64_t T = g.Tfx;
    const uint64_t dE = g.delta[0], dL = g.delta[2];
    const uint64_t gap = dE > dL ? dE - dL : dL - dE;
    const bool ok = !(T < ((int64_t)1 << 32)) && g.F >= 2 && g.F <= 32768 && base + n + 32 <= capacity &&
                    (int)((2 * T) >> 32) == P3 && (int)(T >> 32) == P1 &&
                    g.m[0] == P0 && g.m[2] == P0 && g.J[0] == -1 && g.J[2] == 0 && gap < ((uint64_t)1 << 12) &&
                    (int)((dE + (uint64_t)T) >> 32) == P2 &&
                    // (chip_geometry's own block-length test is for one chip per lane: only its other findings count)
                    g.delta[0] < (uint64_t)T && g.delta[2] < (uint64_t)T;
    if (!ok) return false;
    S.Tfx = T, S.Ufx = g.Ufx, S.delta = dE;
    S.q0 = g.q0;
    S.F2 = g.F / 2;
    S.head_end = g.head_end;
    S.tail_start = g.tail_start;
    if (g.F & 1) {   // an odd whole chip is left behind the last pair: it goes with the last partial chip
        const int q_left = g.q0 + g.F;          // chips q0 + 1 .. q0 + F are whole: the first sample with y > q_left - 1 starts it
        bool nr = false;
        S.tail_start = chip_first_above_exact(chip_first_above((double)(q_left - 1), S.shift[1], inv[1], nr), (double)(q_left - 1),
                                              S.step[1], S.shift[1]);
    }
    if (S.head_end + (n - S.tail_start) > 128) return false;
    S.Dmin = (int)((128 * T) >> 32);            // a lane's blocks are 64 pairs of chips apart
    for (int k = 0; k < Shape::kMaxLen; ++k) sincos_reduced(-(double)k * S.dphi, &S.rs[k], &S.rc[k]);
    for (int gseg = 0; gseg < 4; ++gseg) sincos_reduced(-(double)Shape::start(gseg) * S.dphi, &S.ts[gseg], &S.tc[gseg]);
    sincos_reduced(-(double)S.Dmin * S.dphi, &S.rd0s, &S.rd0c);
    sincos_reduced(-(double)(S.Dmin + 1) * S.dphi, &S.rd1s, &S.rd1c);
    double pc = 0.0, ps = 0.0;
    S.bc[0] = S.bs[0] = 0.0;
    for (int k = 0; k < Shape::kMaxLen; ++k) {
        pc += S.rc[k], ps += S.rs[k];
        S.bc[k + 1] = (pc - ps) * kCvtBias, S.bs[k + 1] = (pc + ps) * kCvtBias;
    }
    S.base = base;
    return true;
}

template <int RAW>
struct Chip2Block {
    uint32_t raw[RAW];         // the block's samples (and what follows them in the dwords)
    int S;                     // first sample (epoch-relative)
    int d1, dP, d2, dn;        // position - P_g of the four boundaries: 0 or 1
};

// Returns false when a lane met a block the scheme does not cover (the caller redoes the epoch per sample).
// zero_words: four zero words of LDS (the replica of a lane without a block).
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ bool correlate_epoch_chip2(const void* __restrict__ ring, const void* __restrict__ ring_flipped,
                                                      const EpochParams& ep, const Chip2Setup<P0, P1, P2, P3>& P, const uint32_t* lut,
                                                      const uint32_t* zero_words, int lane, double* accr, double* acci) {
    using Shape = Chip2Shape<P0, P1, P2, P3>;
    constexpr int NT = 3, A = 1;
    constexpr int kRaw = Shape::kRawDwords;
    constexpr int kMaxLen = Shape::kMaxLen;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;
    const double dphi_u = P.dphi, rem_carrier_u = uniform(ep.rem_carrier);
    const int q0 = P.q0, F2 = P.F2, head_end = P.head_end, tail_start = P.tail_start;
    const int64_t base = P.base;
    const char* ring_base = static_cast<const char*>(ring_flipped) + base * 2;
    // (what feeds per-lane 64-bit arithmetic lives in vector registers: the scalar ones hold the rotations)
    double shift[NT], step[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        shift[t] = P.shift[t], step[t] = P.step[t];
        asm volatile("" : "+v"(shift[t]), "+v"(step[t]));
    }
    int64_t Tfx = P.Tfx, Ufx = P.Ufx;
    uint64_t delta = P.delta;
    int64_t stride_fx = 128 * P.Tfx;
    const int Dmin = P.Dmin;
    asm volatile("" : "+v"(Tfx), "+v"(Ufx), "+v"(delta), "+v"(stride_fx));
    double rd0c = P.rd0c, rd0s = P.rd0s, rd1c = P.rd1c, rd1s = P.rd1s;
    asm volatile("" : "+v"(rd0c), "+v"(rd0s), "+v"(rd1c), "+v"(rd1s));
    double rc[kMaxLen], rs[kMaxLen], tc[4], ts[4];
#pragma unroll
    for (int k = 1; k < kMaxLen; ++k) rc[k] = P.rc[k], rs[k] = P.rs[k];
#pragma unroll
    for (int g = 1; g < 4; ++g) tc[g] = P.tc[g], ts[g] = P.ts[g];

    bool bad = false;
    const int rounds = (F2 + 63) >> 6;
    if (rounds > 0) {
        const int last_idx = F2 - 1;
        const int64_t two32 = (int64_t)1 << 32;
        uint64_t u_cur = (uint64_t)(Ufx + (int64_t)q0 * Tfx + two32) + (uint64_t)((int64_t)(2 * lane) * Tfx);
        auto prepare = [&](int round, uint64_t u0, Chip2Block<kRaw>& b) {
            const int idx = round * 64 + lane;
            const bool inside = idx <= last_idx;
            const uint64_t uS = inside ? u0 : (uint64_t)(Ufx + (int64_t)(q0 + 2 * last_idx) * Tfx + two32);
            const uint64_t u1 = uS + delta, uP = uS + (uint64_t)Tfx, u2 = u1 + (uint64_t)Tfx, uE = uP + (uint64_t)Tfx;
            int S = (int)(uS >> 32);
            int s1 = (int)(u1 >> 32) - S, sP = (int)(uP >> 32) - S, s2 = (int)(u2 >> 32) - S, nn = (int)(uE >> 32) - S;
            auto near_sample = [](uint64_t u) { return (uint32_t)u + 0x10000u < 0x20000u; };
            const bool near = near_sample(uS) || near_sample(u1) || near_sample(uP) || near_sample(u2) || near_sample(uE);
            if (__builtin_expect(__any(near), 0)) {
                const int q = q0 + 1 + 2 * (inside ? idx : last_idx);       // the block's first chip
                S = chip_first_above_exact(S, (double)(q - 1), step[A], shift[A]);
                const int bP = chip_first_above_exact(S + sP, (double)q, step[A], shift[A]);
                const int bE = chip_first_above_exact(S + nn, (double)(q + 1), step[A], shift[A]);
                int b1[2], b2[2];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int t = o ? 2 : 0;
                    const int p0 = q + (o ? 0 : -1);                         // the chip the tap sits on at the block's start
                    b1[o] = chip_first_above_exact(S + s1, (double)p0, step[t], shift[t]);
                    b2[o] = chip_first_above_exact(S + s2, (double)(p0 + 1), step[t], shift[t]);
                    double y = (double)S * step[t];
                    y = y + shift[t];
                    bad = bad || (int)ceil(y) != p0;
                }
                bad = bad || b1[0] != b1[1] || b2[0] != b2[1];
                s1 = b1[0] - S, sP = bP - S, s2 = b2[0] - S, nn = bE - S;
            }
            b.S = S;
            b.d1 = s1 - P0, b.dP = sP - P1, b.d2 = s2 - P2, b.dn = nn - P3;
            bad = bad || (unsigned)b.d1 > 1u || (unsigned)b.dP > 1u || (unsigned)b.d2 > 1u || (unsigned)b.dn > 1u;
            const char* src = ring_base + (int64_t)S * 2;                  // the block's dwords from a 2-byte aligned address
            static_for<0, kRaw / 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const uint4 w = *reinterpret_cast<const uint4*>(src + 16 * i);
                b.raw[4 * i] = w.x, b.raw[4 * i + 1] = w.y, b.raw[4 * i + 2] = w.z, b.raw[4 * i + 3] = w.w;
            });
            if constexpr (kRaw % 4 >= 2) {
                const uint2 w = *reinterpret_cast<const uint2*>(src + 16 * (kRaw / 4));
                b.raw[kRaw / 4 * 4] = w.x, b.raw[kRaw / 4 * 4 + 1] = w.y;
            }
            if constexpr (kRaw % 2 == 1) b.raw[kRaw - 1] = *reinterpret_cast<const uint32_t*>(src + 4 * (kRaw - 1));
        };

        Chip2Block<kRaw> blk_a, blk_b;
        prepare(0, u_cur, blk_a);
        double sb, cb;
        sincos_reduced(__builtin_fma(-(double)blk_a.S, dphi_u, rem_carrier_u), &sb, &cb);
        const int q_lane = q0 + 1 + 2 * lane + SDR_LUT_PAD;
        sdr_u32x2 zI = {0u, 0u}, zQ = {0u, 0u};
        asm volatile("" : "+v"(zI), "+v"(zQ));

        auto process = [&](const Chip2Block<kRaw>& b, int round, double sbk, double cbk) {
            uint32_t hi_const = 0x40B00000u;
            asm volatile("" : "+v"(hi_const));
            double pr = 0.0, pi = 0.0;
            double capr[4], capi[4], fr[4], fi[4];
            static_for<0, Shape::kSamples>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int g = Shape::segment_of(k), j = k - Shape::start(g);
                const uint32_t w = b.raw[k >> 1];
                const double ar = biased_sample(zI, w, cvt_selector((k & 1) ? 2 : 0), hi_const);
                const double ai = biased_sample(zQ, w, cvt_selector((k & 1) ? 3 : 1), hi_const);
                if constexpr (k == Shape::last(g)) capr[g] = pr, capi[g] = pi;      // the segment's sum before its last sample
                if constexpr (j == 0) {
                    pr = ar, pi = ai;
                } else {
                    pr = __builtin_fma(-ai, rs[j], __builtin_fma(ar, rc[j], pr));
                    pi = __builtin_fma(ai, rc[j], __builtin_fma(ar, rs[j], pi));
                }
                if constexpr (k == Shape::last(g)) fr[g] = pr, fi[g] = pi;
                asm volatile("" : "+v"(pr), "+v"(pi), "+v"(zI), "+v"(zQ));
            });
            // the offset's shares out; every boundary picks the sum before or after its segment's last sample
            const int dsel[4] = {b.d1, b.dP, b.d2, b.dn};
            double sr[4], si[4];
            static_for<0, 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int L = Shape::len(g);
                fr[g] -= P.bc[L], fi[g] -= P.bs[L];
                const double cr = capr[g] - P.bc[L - 1], ci = capi[g] - P.bs[L - 1];
                sr[g] = dsel[g] ? fr[g] : cr;
                si[g] = dsel[g] ? fi[g] : ci;
            });
            auto turned = [&](int g, double xr, double xi, double& outr, double& outi, double addr, double addi) {
                outr = __builtin_fma(-xi, ts[g], __builtin_fma(xr, tc[g], addr));
                outi = __builtin_fma(xi, tc[g], __builtin_fma(xr, ts[g], addi));
            };
            double q1r = sr[0], q1i = si[0], q2r, q2i, q3r, q3i, q4r, q4i, f01r, f01i, f012r, f012i;
            turned(1, sr[1], si[1], q2r, q2i, fr[0], fi[0]);
            turned(1, fr[1], fi[1], f01r, f01i, fr[0], fi[0]);
            turned(2, sr[2], si[2], q3r, q3i, f01r, f01i);
            turned(2, fr[2], fi[2], f012r, f012i, f01r, f01i);
            turned(3, sr[3], si[3], q4r, q4i, f012r, f012i);
            // replica words c(q - 1) .. c(q + 2); a lane without a block reads zeros
            const int first = round * 128;
            const uint32_t* lq = round * 64 + lane <= last_idx ? lut + q_lane + first : zero_words + 1;
            const double cm1 = __hiloint2double((int)lq[-1], 0), c0 = __hiloint2double((int)lq[0], 0);
            const double c1 = __hiloint2double((int)lq[1], 0), c2 = __hiloint2double((int)lq[2], 0);
            const double d0 = cm1 - c0, d1 = c0 - c1, d2 = c1 - c2;
            double gr[NT], gi[NT];
            gr[0] = __builtin_fma(d0, q1r, __builtin_fma(d1, q3r, c1 * q4r));
            gi[0] = __builtin_fma(d0, q1i, __builtin_fma(d1, q3i, c1 * q4i));
            gr[1] = __builtin_fma(d1, q2r, c1 * q4r);
            gi[1] = __builtin_fma(d1, q2i, c1 * q4i);
            gr[2] = __builtin_fma(d1, q1r, __builtin_fma(d2, q3r, c2 * q4r));
            gi[2] = __builtin_fma(d1, q1i, __builtin_fma(d2, q3i, c2 * q4i));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                accr[t] = __builtin_fma(-sbk, gi[t], __builtin_fma(cbk, gr[t], accr[t]));
                acci[t] = __builtin_fma(sbk, gr[t], __builtin_fma(cbk, gi[t], acci[t]));
            }
        };
        auto advance = [&](const Chip2Block<kRaw>& from, const Chip2Block<kRaw>& to, int to_round) {
            const unsigned dd = (unsigned)(to.S - from.S - Dmin);
            const bool alive = to_round * 64 + lane <= last_idx;
            bad = bad || (alive && dd > 1u);
            const double rc_ = dd ? rd1c : rd0c, rs_ = dd ? rd1s : rd0s;
            const double cbn = __builtin_fma(cb, rc_, -sb * rs_);
            const double sbn = __builtin_fma(sb, rc_, cb * rs_);
            cb = cbn, sb = sbn;                         // (a lane without a block correlates against zero replica words)
        };
        for (int it = 0; it < rounds / 2; ++it) {
            const double sb0 = sb, cb0 = cb;
            u_cur += (uint64_t)stride_fx;
            prepare(2 * it + 1, u_cur, blk_b);
            advance(blk_a, blk_b, 2 * it + 1);
            process(blk_a, 2 * it, sb0, cb0);
            const double sb1 = sb, cb1 = cb;
            if (2 * it + 2 < rounds) {
                u_cur += (uint64_t)stride_fx;
                prepare(2 * it + 2, u_cur, blk_a);
                advance(blk_b, blk_a, 2 * it + 2);
            }
            process(blk_b, 2 * it + 1, sb1, cb1);
        }
        if (rounds & 1) process(blk_a, rounds - 1, sb, cb);
    }
    if (__any(bad)) return false;
    // (up to three partial or left-over chips: more than a wave of samples at 25 MHz now and then)
    for (int off = 0; off < head_end + (ep.n - tail_start); off += 64)
        edge_samples<SDR_FMT_CI8, NT>(ring, 0, ep, dphi_u, shift, step, lut, lane + off, head_end, tail_start, accr, acci, base);
    return true;
}

}  // namespace sdr
