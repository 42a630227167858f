// Several chips per lane: the straight-line form of correlator_chip.h with a block of CH whole chips of the prompt tap, three
// taps half a chip apart, ci8 ring.  Compiled for the 2*CH boundary positions P0 < P1 < ... (each "or one sample later"):
//      <4, 9, 14, 19>             two chips of 9.5 .. 10 samples: a C/A code at the reference's shipped 10 MHz (config/receiver.ini:18-20)
//      (<4, 9, 14, 19, 24, 29>, three of them, works and was measured: it spills at three waves per SIMD)
//      <5, 11, 17, 23>            two chips of 11.5 .. 12 samples: 12 MHz
//      <1, 3, 5, 7, 9, 11, 13, 15>  FOUR chips of 3.75 .. 4 samples: the reference's 4 MHz (BASELINE configs[0]; round 6) -- eight
//                                 segments of two samples; the per-sample kernel spends 43 instructions per sample there
// At these rates one chip is too short for a lane (the per-block work of correlator_chip.h would be paid every ~10
// samples), and 8 consecutive samples per lane (correlator.h's boundary variant) cost ~30 issue slots per sample.  A lane
// owns chips q .. q + CH - 1: P_last or P_last + 1 samples in which, for c = 0 .. CH - 1,
//      the early and late tap change chips     P_(2c) | P_(2c) + 1     samples in   (E from q + c - 1 to q + c, L from q + c to q + c + 1)
//      the prompt tap changes chips            P_(2c+1) | P_(2c+1) + 1              (q + c to q + c + 1; the last one is the block's end)
// -- every position wave-uniform up to + 1, as in the one-chip forms.  The block is summed as 2*CH segments (samples
// (P_(g-1), P_g]) that each start at rotation 0 (the longest segment's rotations in scalar registers); a segment keeps its sum
// before and after its last sample and a lane picks the member its boundary needs.  With T_g = exp(-1j*start_g*dphi), the
// running prefix F_g = F_(g-1) + T_g*f_g and the prefix at boundary g, Q_g = F_(g-1) + T_g*sel_g, the taps' shares follow
// boundary by boundary (w_k = c(q + k), the replica words):
//      g even (= 2c):      E += (w_(c-1) - w_c) * Q_g        L += (w_c - w_(c+1)) * Q_g
//      g odd  (= 2c + 1):  P += (w_c - w_(c+1)) * Q_g
//      the block's end:    E += w_(CH-1) * Q      P += w_(CH-1) * Q      L += w_CH * Q
// so that only F, the three taps' sums and the current segment are alive.  Samples are built with the one-instruction
// biased conversion from the flipped ring image (correlator_chip.h); what the offset puts into a segment sum of L samples
// is one complex constant per length and epoch.  Block boundaries come from the same Q32.32 line; a boundary within
// 2^-16 of a sample sends the wave through exact evaluations of the reference expression; a lane that meets a position
// outside its pair flags the epoch, which is redone per sample.  Everything wave-uniform that is not data -- tap
// constants, geometry, rotations -- is worked out when the plan is made, one thread per item (ChipNSetup), as for the
// one-chip forms.
#pragma once

#include "correlator_chip.h"

#pragma clang fp contract(off)

namespace sdr {

template <int... P>
struct ChipNShape {
    static constexpr int NB = sizeof...(P);                 // boundaries = segments
    static constexpr int CH = NB / 2;                       // chips of the prompt tap per block
    static_assert(NB >= 4 && NB % 2 == 0, "an even number of boundaries: one of the outer taps and one of the prompt tap per chip");
    static constexpr int pos(int g) {
        constexpr int a[] = {P...};
        return a[g];
    }
    static constexpr int start(int g) { return g == 0 ? 0 : pos(g - 1) + 1; }
    static constexpr int len(int g) { return pos(g) + 1 - start(g); }       // (when the boundary is "one later")
    static constexpr int max_len() {
        int m = 0;
        for (int g = 0; g < NB; ++g) m = len(g) > m ? len(g) : m;
        return m;
    }
    static constexpr int kMaxLen = max_len();
    static constexpr int kSamples = pos(NB - 1) + 1;        // a block holds P_last or P_last + 1 samples
    static constexpr int kRawDwords = (kSamples + 1) / 2;
    static constexpr int segment_of(int k) {
        int g = 0;
        while (k > pos(g)) ++g;
        return g;
    }
};

template <int... P>
struct ChipNSetup {
    using Shape = ChipNShape<P...>;
    double dphi;                       // carrier_step(carrier_hz, fs)
    double shift[3], step[3];          // the taps' np.linspace constants (exact re-evaluations, edge samples)
    int64_t base;                      // start_sample % capacity; < 0: this routine does not serve the epoch
    int64_t Tfx, Ufx;                  // samples per chip and the prompt line's offset, Q32.32
    uint64_t delta;                    // the outer taps' first switch, samples after the block start (Q32.32; E's -- L's agrees to 2^-20)
    int q0, FB;                        // block b < FB holds chips q0 + 1 + CH*b .. q0 + CH*(b + 1) of the prompt tap
    int head_end, tail_start;          // samples [0, head_end) and [tail_start, n) are correlated one per lane
    int Dmin, pad;
    double rc[Shape::kMaxLen], rs[Shape::kMaxLen];   // exp(-1j*k*dphi), k < the longest segment
    double tc[Shape::NB], ts[Shape::NB];             // exp(-1j*start_g*dphi)
    double rd0c, rd0s, rd1c, rd1s;     // over the Dmin / Dmin + 1 samples to a lane's next block
    double bc[Shape::kMaxLen + 1], bs[Shape::kMaxLen + 1];   // the biased conversion's share of a segment sum of L samples
};

// One thread per item of a plan (epl.hip: chipn_setup_kernel; the host builds of the tests): false when the scheme does not
// cover the item.
template <int... P>
__host__ __device__ inline bool chipn_setup(int n, int64_t start_sample, int64_t capacity, double carrier_hz, double rem_code, double code_step,
                                 const double* spacing, double fs, ChipNSetup<P...>& S) {
    using Shape = ChipNShape<P...>;
    constexpr int CH = Shape::CH;
    S = ChipNSetup<P...>{};
    S.base = -1;
    // (4 .. 64 samples per chip, far around anything the shapes cover: a step outside it -- a denormal passes the list's
    // "positive and finite" -- must not reach the fixed-point conversions below: undefined on the host, `make check-sanitize`)
    if (!(code_step >= 1.0 / 64.0 && code_step <= 0.3) || n < 1) return false;
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
    // (chip_geometry's own block-length test is for one chip per lane: only its other findings count)
    bool ok = !(T < ((int64_t)1 << 32)) && g.F >= CH && g.F <= 32768 && base + n + 32 <= capacity && g.J[0] == -1 && g.J[2] == 0 &&
              gap < ((uint64_t)1 << 12) && dE < (uint64_t)T && dL < (uint64_t)T;
    for (int c = 0; ok && c < CH; ++c)
        ok = (int)((dE + (uint64_t)c * (uint64_t)T) >> 32) == Shape::pos(2 * c) &&
             (int)(((uint64_t)(c + 1) * (uint64_t)T) >> 32) == Shape::pos(2 * c + 1);
    if (!ok) return false;
    S.Tfx = T, S.Ufx = g.Ufx, S.delta = dE;
    S.q0 = g.q0;
    S.FB = g.F / CH;
    S.head_end = g.head_end;
    S.tail_start = g.tail_start;
    if (g.F % CH) {   // whole chips left behind the last block: they go with the last partial chip
        const int q_left = g.q0 + S.FB * CH + 1;     // the first left-over chip: the first sample with y > q_left - 1 starts it
        bool nr = false;
        S.tail_start = chip_first_above_exact(chip_first_above((double)(q_left - 1), S.shift[1], inv[1], nr), (double)(q_left - 1),
                                              S.step[1], S.shift[1]);
    }
    if (S.head_end + (n - S.tail_start) > 128) return false;
    S.Dmin = (int)(((int64_t)(64 * CH) * T) >> 32);            // a lane's blocks are 64 blocks of CH chips apart
    for (int k = 0; k < Shape::kMaxLen; ++k) sincos_reduced(-(double)k * S.dphi, &S.rs[k], &S.rc[k]);
    for (int gseg = 0; gseg < Shape::NB; ++gseg) sincos_reduced(-(double)Shape::start(gseg) * S.dphi, &S.ts[gseg], &S.tc[gseg]);
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
struct ChipNBlock {
    uint32_t raw[RAW];         // the block's samples (and what follows them in the dwords)
    int S;                     // first sample (epoch-relative)
    unsigned later;            // bit g: boundary g sits at P_g + 1 (else at P_g)
};

// Returns false when a lane met a block the scheme does not cover (the caller redoes the epoch per sample).
// zero_words: CH + 2 zero words of LDS (the replica of a lane without a block).
template <int... P>
__device__ __forceinline__ bool correlate_epoch_chipn(const void* __restrict__ ring, const void* __restrict__ ring_flipped,
                                                      const EpochParams& ep, const ChipNSetup<P...>& S_, const uint32_t* lut,
                                                      const uint32_t* zero_words, int lane, double* accr, double* acci) {
    using Shape = ChipNShape<P...>;
    constexpr int NT = 3, A = 1;
    constexpr int NB = Shape::NB, CH = Shape::CH;
    constexpr int kRaw = Shape::kRawDwords;
    constexpr int kMaxLen = Shape::kMaxLen;
#pragma unroll
    for (int t = 0; t < NT; ++t) accr[t] = acci[t] = 0.0;
    const double dphi_u = S_.dphi, rem_carrier_u = uniform(ep.rem_carrier);
    const int q0 = S_.q0, FB = S_.FB, head_end = S_.head_end, tail_start = S_.tail_start;
    const int64_t base = S_.base;
    const char* ring_base = static_cast<const char*>(ring_flipped) + base * 2;
    // (what feeds per-lane 64-bit arithmetic lives in vector registers: the scalar ones hold the rotations)
    double shift[NT], step[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        shift[t] = S_.shift[t], step[t] = S_.step[t];
        asm volatile("" : "+v"(shift[t]), "+v"(step[t]));
    }
    int64_t Tfx = S_.Tfx, Ufx = S_.Ufx;
    uint64_t delta = S_.delta;
    int64_t stride_fx = (int64_t)(64 * CH) * S_.Tfx;
    const int Dmin = S_.Dmin;
    asm volatile("" : "+v"(Tfx), "+v"(Ufx), "+v"(delta), "+v"(stride_fx));
    double rd0c = S_.rd0c, rd0s = S_.rd0s, rd1c = S_.rd1c, rd1s = S_.rd1s;
    asm volatile("" : "+v"(rd0c), "+v"(rd0s), "+v"(rd1c), "+v"(rd1s));
    double rc[kMaxLen], rs[kMaxLen], tc[NB], ts[NB];
#pragma unroll
    for (int k = 1; k < kMaxLen; ++k) rc[k] = S_.rc[k], rs[k] = S_.rs[k];
#pragma unroll
    for (int g = 1; g < NB; ++g) tc[g] = S_.tc[g], ts[g] = S_.ts[g];

    bool bad = false;
    const int rounds = (FB + 63) >> 6;
    if (rounds > 0) {
        const int last_idx = FB - 1;
        const int64_t two32 = (int64_t)1 << 32;
        uint64_t u_cur = (uint64_t)(Ufx + (int64_t)q0 * Tfx + two32) + (uint64_t)((int64_t)(CH * lane) * Tfx);
        auto prepare = [&](int round, uint64_t u0, ChipNBlock<kRaw>& b) {
            const int idx = round * 64 + lane;
            const bool inside = idx <= last_idx;
            const uint64_t uS = inside ? u0 : (uint64_t)(Ufx + (int64_t)(q0 + CH * last_idx) * Tfx + two32);
            auto near_sample = [](uint64_t u) { return (uint32_t)u + 0x10000u < 0x20000u; };
            int S = (int)(uS >> 32);
            int at[NB];                                    // boundary g, samples after the block's first one
            bool near = near_sample(uS);
            {
                uint64_t uo = uS + delta, up = uS;         // the outer taps' line and the prompt tap's, chip by chip
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    up += (uint64_t)Tfx;
                    near = near || near_sample(uo) || near_sample(up);
                    at[2 * c] = (int)(uo >> 32) - S;
                    at[2 * c + 1] = (int)(up >> 32) - S;
                    uo += (uint64_t)Tfx;
                }
            }
            if (__builtin_expect(__any(near), 0)) {
                const int q = q0 + 1 + CH * (inside ? idx : last_idx);       // the block's first chip
                const int S_pred = S;
                S = chip_first_above_exact(S_pred, (double)(q - 1), step[A], shift[A]);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    // the prompt tap leaves chip q + c; E leaves q + c - 1 and L leaves q + c, half a chip earlier
                    const int bp = chip_first_above_exact(S_pred + at[2 * c + 1], (double)(q + c), step[A], shift[A]);
                    const int be = chip_first_above_exact(S_pred + at[2 * c], (double)(q + c - 1), step[0], shift[0]);
                    const int bl = chip_first_above_exact(S_pred + at[2 * c], (double)(q + c), step[2], shift[2]);
                    bad = bad || be != bl;
                    at[2 * c] = be - S, at[2 * c + 1] = bp - S;
                }
#pragma unroll
                for (int o = 0; o < 2; ++o) {              // each outer tap on its chip at the block's first sample
                    const int t = o ? 2 : 0;
                    double y = (double)S * step[t];
                    y = y + shift[t];
                    bad = bad || (int)ceil(y) != q + (o ? 0 : -1);
                }
            }
            b.S = S;
            // (every boundary at P_g or one later: d_g = 0 / 1 is bit g; anything else, in any of them, shows in their union)
            unsigned later = 0, any_d = 0;
            static_for<0, NB>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const unsigned d = (unsigned)(at[g] - Shape::pos(g));
                any_d |= d;
                later |= d << g;
            });
            bad = bad || any_d > 1u;
            b.later = later;
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

        ChipNBlock<kRaw> blk_a, blk_b;
        prepare(0, u_cur, blk_a);
        double sb, cb;
        sincos_reduced(__builtin_fma(-(double)blk_a.S, dphi_u, rem_carrier_u), &sb, &cb);
        const int q_lane = q0 + 1 + CH * lane + SDR_LUT_PAD;
        sdr_u32x2 zI = {0u, 0u}, zQ = {0u, 0u};
        asm volatile("" : "+v"(zI), "+v"(zQ));

        auto process = [&](const ChipNBlock<kRaw>& b, int round, double sbk, double cbk) {
            uint32_t hi_const = 0x40B00000u;
            asm volatile("" : "+v"(hi_const));
            // replica words w_k = c(q + k), k = -1 .. CH; a lane without a block reads zeros
            const uint32_t* lq = round * 64 + lane <= last_idx ? lut + q_lane + round * (64 * CH) : zero_words + 1;
            double w[CH + 2];
#pragma unroll
            for (int k = 0; k < CH + 2; ++k) w[k] = __hiloint2double((int)lq[k - 1], 0);       // w[k] holds w_(k-1)
            double pr = 0.0, pi = 0.0, capr = 0.0, capi = 0.0, Fr = 0.0, Fi = 0.0;
            double gr[NT] = {0.0, 0.0, 0.0}, gi[NT] = {0.0, 0.0, 0.0};
            static_for<0, Shape::kSamples>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int g = Shape::segment_of(k), j = k - Shape::start(g);
                const uint32_t wd = b.raw[k >> 1];
                const double ar = biased_sample(zI, wd, cvt_selector((k & 1) ? 2 : 0), hi_const);
                const double ai = biased_sample(zQ, wd, cvt_selector((k & 1) ? 3 : 1), hi_const);
                if constexpr (k == Shape::pos(g)) capr = pr, capi = pi;          // the segment's sum before its last sample
                if constexpr (j == 0) {
                    pr = ar, pi = ai;
                } else {
                    pr = __builtin_fma(-ai, rs[j], __builtin_fma(ar, rc[j], pr));
                    pi = __builtin_fma(ai, rc[j], __builtin_fma(ar, rs[j], pi));
                }
                asm volatile("" : "+v"(pr), "+v"(pi), "+v"(zI), "+v"(zQ));
                if constexpr (k == Shape::pos(g)) {
                    // segment g is complete: the offset's share out, the member its boundary needs, the prefix at the boundary
                    constexpr int L = Shape::len(g);
                    const double fr = pr - S_.bc[L], fi = pi - S_.bs[L];
                    const double cr = capr - S_.bc[L - 1], ci = capi - S_.bs[L - 1];
                    const bool one_later = (b.later >> g) & 1u;
                    const double sr = one_later ? fr : cr, si = one_later ? fi : ci;
                    double qr, qi;
                    if constexpr (g == 0) {
                        qr = sr, qi = si;
                        Fr = fr, Fi = fi;
                    } else {
                        qr = __builtin_fma(-si, ts[g], __builtin_fma(sr, tc[g], Fr));
                        qi = __builtin_fma(si, tc[g], __builtin_fma(sr, ts[g], Fi));
                        if constexpr (g < NB - 1) {
                            const double nr = __builtin_fma(-fi, ts[g], __builtin_fma(fr, tc[g], Fr));
                            Fi = __builtin_fma(fi, tc[g], __builtin_fma(fr, ts[g], Fi));
                            Fr = nr;
                        }
                    }
                    if constexpr (g == NB - 1) {           // the block's end
                        gr[0] = __builtin_fma(w[CH], qr, gr[0]), gi[0] = __builtin_fma(w[CH], qi, gi[0]);            // E: w_(CH-1)
                        gr[1] = __builtin_fma(w[CH], qr, gr[1]), gi[1] = __builtin_fma(w[CH], qi, gi[1]);            // P: w_(CH-1)
                        gr[2] = __builtin_fma(w[CH + 1], qr, gr[2]), gi[2] = __builtin_fma(w[CH + 1], qi, gi[2]);    // L: w_CH
                    } else if constexpr (g % 2 == 0) {     // E and L change chips
                        constexpr int c = g / 2;
                        const double de = w[c] - w[c + 1], dl = w[c + 1] - w[c + 2];                                // w_(c-1) - w_c, w_c - w_(c+1)
                        gr[0] = __builtin_fma(de, qr, gr[0]), gi[0] = __builtin_fma(de, qi, gi[0]);
                        gr[2] = __builtin_fma(dl, qr, gr[2]), gi[2] = __builtin_fma(dl, qi, gi[2]);
                    } else {                               // the prompt tap changes chips
                        constexpr int c = g / 2;
                        const double dp = w[c + 1] - w[c + 2];                                                      // w_c - w_(c+1)
                        gr[1] = __builtin_fma(dp, qr, gr[1]), gi[1] = __builtin_fma(dp, qi, gi[1]);
                    }
                }
            });
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                accr[t] = __builtin_fma(-sbk, gi[t], __builtin_fma(cbk, gr[t], accr[t]));
                acci[t] = __builtin_fma(sbk, gr[t], __builtin_fma(cbk, gi[t], acci[t]));
            }
        };
        auto advance = [&](const ChipNBlock<kRaw>& from, const ChipNBlock<kRaw>& to, int to_round) {
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
    // (the partial first and last chip and up to CH - 1 left-over ones: more than a wave of samples now and then)
    for (int off = 0; off < head_end + (ep.n - tail_start); off += 64)
        edge_samples<SDR_FMT_CI8, NT>(ring, 0, ep, dphi_u, shift, step, lut, lane + off, head_end, tail_start, accr, acci, base);
    return true;
}

}  // namespace sdr
