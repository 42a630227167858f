// One workgroup per (PRN, bin) inverse transform of the map-free PCPS search at N = 25 000 = 125 x 200 (25 MHz, 1 ms:
// BASELINE configs[1]): the four-step intermediate never leaves the compute unit.  Included by pcps_fused.hip alone (its own
// translation unit: built without MachineLICM, see the Makefile).
//
// The two-kernel four-step of pcps_fast.h writes the 400 KB intermediate of every transform to memory and reads it back
// (1.07 GB per 32-PRN x 41-bin call; a kernel's dirty L2 lines are written back when it ends).  Here a 512-thread
// workgroup -- alone on its CU: 2 waves per SIMD, up to 256 registers per lane, all 160 KiB of LDS -- keeps it:
//
//   column stage  (length 125 = 25 x 5; five threads per column, 100 columns at a time, two items per thread):
//       A_r[k'] = sum_m x[r + 5m] w25^(m k'),  B_r[k'] = A_r[k'] w125^(r k')       25-point transform in registers
//     and the 25 000 values B are PARKED: k' = kA + 5 kB belongs to round kA; rounds 0 and 1 go straight to the two LDS
//     buffers (80 000 B each), rounds 2-4 stay in registers (30 points = 120 registers per thread) and move into a
//     buffer as soon as its round has been consumed.
//   round rho = 0..4 (25 rows k1 = rho + 5 s + 25 q of the intermediate, 80 KB, in place in one buffer):
//       Y[k' + 25 q] = sum_r B_r[k'] w5^(r q)                                       thread (s, column): k' = rho + 5 s
//       row stage (length 200 = 10 x 20, twenty threads per row, ten points each):
//       z[n2] = Y[k1][n2] w_N^(n2 k1)                                               four-step twiddle: base * step^m
//       A_e[k''] = sum_m z[e + 20m] w10^(m k''),  B_e[k''] = A_e[k''] w200^(e k'')  thread (row, e)
//       X[k1 + 125 (k'' + 10 (2p + h))] = sum_t (B_t[k''] +- B_(t+10)[k'']) w20^(h t) w10^(t p)
//                                                                                   thread (row, k'', h): radix-2 DIF on
//                                                                                   the way in, then a 10-point transform
//     and only the running (|.|/N, first index) maximum survives: one 16-byte record per wave and transform.
//   (w = conjugated table entries: inverse transforms, unnormalised; 1/N enters with |.| as in pcps_fast.h.)
//
// The workgroups are persistent: workgroup (XCD x, slot s) walks its XCD's share of a host-made work list in steps of
// 32, and the list is ordered in blocks of 4 bins x 8 PRNs so that the operands of the transforms an XCD has in flight
// (12 arrays of 400 KB) mostly sit in its L2.
//
// N = 50 000 (50 MHz, BASELINE configs[3]/[4]; TERMS = 2): 800 KB of state do not fit a compute unit, two transforms of
// 25 000 do.  One radix-2 decimation-in-frequency step in FRONT of the kernel above -- with P = spectrum x conj(fft(code)),
// M = 25 000, w = exp(-2 pi i / 50 000):
//     x[2m]     = IFFT_M( P[k] + P[k + M] )[m]                x[2m + 1] = IFFT_M( (P[k] - P[k + M]) w^-k )[m]
// and the twiddle of the odd half is folded into a second image of the PRN's code spectrum, made with the spectrum and
// cached with it (pcps.hip code_parity_kernel: C1[k] = C[k] w^-k, C1[k + M] = -C[k + M] w^-k), so that BOTH halves are
//     IFFT_M( F[k] Cp[k] + F[k + M] Cp[k + M] ),   p = parity of the output sample,
// the same unit with two operand terms per point instead of one: four loads per point where the 25 000-point search has
// two, everything behind the operand stage unchanged.  A unit is (PRN, bin, parity); the work list sees 2 nbins "virtual
// bins" (parity-major: a block of 4 virtual bins x 8 PRNs still shares 4 + 8 operand arrays); records carry the index in
// the 50 000-sample row, 2m + p.
#pragma once

// Build-time switches of the A/B variants (tools/build_variant.sh <tag> pcps_fused -DFUSED_...=0|1); measured on one
// box, 32 PRNs x 41 bins, ms per call: all off 0.266; buffer loads 0.267; merged Y step 0.291; twiddles a round ahead
// 0.273; reads first 0.267 (gpurun_out/r04_fused_var9.txt) -- the defaults are the fastest (the last two switches went
// with round 6's lane roles)
#ifndef FUSED_BUFLOAD
#define FUSED_BUFLOAD 0      // operand loads as buffer loads (scalar descriptor + one per-lane offset) instead of 64-bit addresses
#endif
#ifndef FUSED_SPLIT_S2Y
#define FUSED_SPLIT_S2Y 0    // whole transforms: waves 0-3 do BOTH halves of a round's second row stage while waves 4-7 do the next round's Y step
                             // (measured, same box, ms per 32 x 41 search: 0.2077 against 0.2039 at 25 MHz, 0.4514 against 0.4482 at 50 MHz --
                             // a SIMD's one arithmetic wave issues a dependent fp64 instruction every ~7 cycles, two share the pipe at 4: the
                             // stage wants both of a SIMD's waves in it; not kept)
#endif
#ifndef FUSED_MERGE_Y
#define FUSED_MERGE_Y 0      // the next round's Y-in-place step inside this round's second row stage (3 barriers per round, not 4)
#endif

namespace fused25k {

using fast25k::cmul_conj;
using fast25k::cmulf;
using fast25k::ibf5;

constexpr int N1 = 125, N2 = 200, N = 25000;
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kBuf = 25 * N2;                 // double2 per round buffer
constexpr int kTab = 240;                     // w125^e (e <= 96) during the column stage, w200^e (e <= 171) during the rows
constexpr size_t kLdsBytes = (size_t)(2 * kBuf + kTab) * sizeof(double2);
static_assert(kLdsBytes == 160 * 1024, "the whole LDS of a CU");
constexpr int kRecordsPerTransform = kWaves;
constexpr int kSlotsPerXcd = 32;

// exp(-2 pi i k / n) for the transforms' internal twiddles: literals (a wave-uniform table read costs scalar registers the
// kernel does not have; the compiler materialises a literal where it is used)
constexpr double kW25X[17] = {1.0, 0.9685831611286311, 0.8763066800438636, 0.7289686274214116, 0.5358267949789967, 0.30901699437494745, 0.06279051952931337, -0.18738131458572463, -0.42577929156507266, -0.6374239897486897, -0.8090169943749475, -0.9297764858882515, -0.9921147013144779, -0.9921147013144779, -0.9297764858882515, -0.8090169943749475, -0.6374239897486897};
constexpr double kW25Y[17] = {0.0, -0.2486898871648548, -0.48175367410171527, -0.6845471059286887, -0.8443279255020151, -0.9510565162951535, -0.9980267284282716, -0.9822872507286887, -0.9048270524660196, -0.7705132427757893, -0.5877852522924731, -0.368124552684678, -0.12533323356430426, 0.12533323356430426, 0.368124552684678, 0.5877852522924731, 0.7705132427757893};
constexpr double kW10X[5] = {1.0, 0.8090169943749475, 0.30901699437494745, -0.30901699437494745, -0.8090169943749475};
constexpr double kW10Y[5] = {0.0, -0.5877852522924731, -0.9510565162951535, -0.9510565162951535, -0.5877852522924731};
constexpr double kW20X[10] = {1.0, 0.9510565162951535, 0.8090169943749475, 0.5877852522924731, 0.30901699437494745, 0.0, -0.30901699437494745, -0.5877852522924731, -0.8090169943749475, -0.9510565162951535};
constexpr double kW20Y[10] = {0.0, -0.30901699437494745, -0.5877852522924731, -0.8090169943749475, -0.9510565162951535, -1.0, -0.9510565162951535, -0.8090169943749475, -0.5877852522924731, -0.30901699437494745};

// 10-point inverse transform of u[r], r = r1 + 2 r2: result X[qA + 5 qB] in u[2 qA + qB] (fast25k::idft10 with literal twiddles)
__device__ __forceinline__ void idft10c(double2* u) {
#pragma unroll
    for (int r1 = 0; r1 < 2; ++r1) {
        double2 t[5] = {u[r1], u[r1 + 2], u[r1 + 4], u[r1 + 6], u[r1 + 8]};
        ibf5(t);
#pragma unroll
        for (int k = 0; k < 5; ++k) u[r1 + 2 * k] = t[k];
    }
#pragma unroll
    for (int qA = 1; qA < 5; ++qA) u[1 + 2 * qA] = cmul_conj(u[1 + 2 * qA], make_double2(kW10X[qA], kW10Y[qA]));
#pragma unroll
    for (int qA = 0; qA < 5; ++qA) {
        const double2 a = u[2 * qA], b = u[2 * qA + 1];
        u[2 * qA] = cadd(a, b);
        u[2 * qA + 1] = csub(a, b);
    }
}

// 16-byte buffer load: wave-uniform descriptor (base, bytes) + per-lane 32-bit byte offset + wave-uniform byte offset --
// no 64-bit address per load in vector registers and no vector arithmetic for the 50 distances of an item
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
#if FUSED_BUFLOAD
struct Operand {
    __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ Operand make_operand(const double2* base) { return {make_rsrc(base, N * 16)}; }
__device__ __forceinline__ double2 ldb(const Operand& o, unsigned voff, unsigned soff) {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(o.rs, (int)voff, (int)soff, 0);
    double2 d;
    d.x = __hiloint2double((int)q.y, (int)q.x);
    d.y = __hiloint2double((int)q.w, (int)q.z);
    return d;
}
#else
struct Operand {
    const char* base;
};
__device__ __forceinline__ Operand make_operand(const double2* base) { return {reinterpret_cast<const char*>(base)}; }
__device__ __forceinline__ double2 ldb(const Operand& o, unsigned voff, unsigned soff) {
    return *reinterpret_cast<const double2*>((o.base + soff) + voff);
}
#endif

// One unit of work of a persistent workgroup: a whole (PRN, bin) transform, or -- for the transforms left over when the
// search is not a whole number of rounds of 256 -- ONE of a transform's five rounds: the column stage then computes that
// round's fifth of its outputs only (all operands are still read), so that the tail of the launch is a fifth of a round
// on 5 x as many compute units.  `record`: where its kRecordsPerTransform records go (PRN-major for the peak kernel).
struct WorkItem {
    int prn, bin, round, record;
};

struct Args {
    const double2* spec;       // [nbins][N] forward spectra of the Doppler-mixed block
    // Doppler bins whose frequencies differ by a whole number of transform bins (250 Hz steps over 1 ms: every fourth) have
    // the SAME spectrum, circularly shifted (pcps.hip: shared spectra): when given, bin b's spectrum starts spec_off[b]
    // elements into `spec` -- a row of its class with the shift taken out of a halo in front of it.
    const long long* spec_off;
    const double2* code_spec;  // [n_prn][N]
    const double2* tw;         // exp(-2 pi i m / N)
    const WorkItem* work;      // in processing order
    int xcd_first[9];          // XCD x owns work[xcd_first[x] .. xcd_first[x + 1])
    double scale;
    Best* partials;            // [work item's record slot][kRecordsPerTransform]
    // theta[p]: bits of the largest |.|/N any workgroup has recorded for PRN p so far in THIS launch (0 at its start).  A
    // value of PRN p's map below it can neither be the map's maximum nor tie with it, so a round whose ten new values per
    // lane all lie below it (with a margin of 2^-40) skips the candidate bookkeeping: after the first round of workgroups
    // nearly every one does.  theta_next: the next launch's array, zeroed by this one.
    unsigned long long* theta;
    unsigned long long* theta_next;
    int n_prn;
    int nbins;                 // TERMS = 2: real Doppler bins (a unit's virtual bin is parity * nbins + bin)
};

// Diagnostic build (-DSDR_FUSED_STAMPS, tools/pcps_fused_phases.py): wave 0 of every workgroup adds up the shader cycles
// between phase boundaries; read back through sdr_debug_fused_stamps.  No stamp executes in the regular build.
#ifdef SDR_FUSED_STAMPS
__device__ unsigned long long g_fused_stamps[256][8];
#define FUSED_STAMP(slot)                                                  \
    do {                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        if (tid == 0) g_fused_stamps[blockIdx.x & 255][slot] += now_ - stamp_;   /* (the second sweep at N = 50 000 has 320 workgroups) */   \
        stamp_ = now_;                                                     \
    } while (0)
#define FUSED_DFLT(x)                 /* (the stamp reference follows these parameters: no defaults in this build) */
#else
#define FUSED_STAMP(slot) do { } while (0)
#define FUSED_DFLT(x) = x
#endif

// FUSED_PREFETCH_UNIT (N = 25 000, whole transforms; round 6): the first three of item 0's five operand groups of the NEXT
// unit are requested during THIS unit's last two rounds -- into the registers the parked rounds have left by then -- and
// consumed by the next unit's column stage: 30 of a unit's 100 operand loads, and the ~2 k cycles until a fresh stream's
// first load is back, move under the rounds, where the vector-memory path is idle.
// OFF: built and measured (VERDICT r5 item 1's "operands of transform t + 1 loaded while t is in its row stages").  The
// values cross the persistent loop's back edge, and the register allocator carries them through scratch memory there
// (48 / 136 / 383 spilled registers for 1 / 2 / 3 groups, whatever is done to shorten their live ranges): same box, ms per
// 32 x 41 search, 0.2034 without, 0.215 with one group, 0.220 with two.
#ifndef FUSED_PREFETCH_UNIT
#define FUSED_PREFETCH_UNIT 0    // groups requested a unit ahead (0: none; 1..3)
#endif
constexpr int kPfGroups = FUSED_PREFETCH_UNIT > 0 ? FUSED_PREFETCH_UNIT : 1;
struct UnitPrefetch {
    double2 x[kPfGroups][5], c[kPfGroups][5];    // item 0, groups m1 = 0 .. kPfGroups - 1 (points m1 + 5 m2), spectrum and code spectrum
    bool have;                   // (uniform) x / c hold THIS unit's operands
    bool next_whole;             // (uniform) the unit after this one is a whole transform: its groups are requested in round 3
    int next_prn, next_bin;
};

// One unit of work (a whole transform, or one or two of its five rounds) by the 512 threads of the workgroup.
// !WHOLE: `mode` = r0 | r1 << 4, the unit's rounds (r1 = 15: one round only); round r0 lives in the first LDS buffer, r1 in
// the second -- nothing is parked in registers.
// SECOND: the unit belongs to the second sweep -- the winning Doppler row of a PRN again, the maximum over the columns
// TwoCorrelationPeakComparison allows: [0, a1) U [b0, b1) (acquisition.py:98-111); flat index = the code phase;
// no bound from the first sweep (the second peak lies below it).
// TERMS = 2: the unit is one parity of a 50 000-point transform (header comment): `bin` = the real bin, `par` the parity.
template <bool WHOLE, bool SECOND = false, int TERMS = 1, bool PF = false>
__device__ __forceinline__ void one_unit(const Args& a, double2* lds4, const int tid, const int prn, const int bin, const int mode,
                                         const int rec_slot, const int a1 FUSED_DFLT(0), const int b0 FUSED_DFLT(0), const int b1 FUSED_DFLT(0),
                                         const int par FUSED_DFLT(0), UnitPrefetch* const pf_ FUSED_DFLT(nullptr)
#ifdef SDR_FUSED_STAMPS
                                         , unsigned long long& stamp_
#endif
) {
    constexpr int NF = TERMS * N;                       // samples of the row the unit's outputs belong to
    constexpr bool kPf = PF && FUSED_PREFETCH_UNIT > 0 && WHOLE && !SECOND && TERMS == 1;
    UnitPrefetch dummy_pf_;                             // (!kPf: never touched)
    UnitPrefetch* const pf = kPf ? pf_ : &dummy_pf_;     // (kPf: the kernel's own object -- never null, so that it lives in registers)
    const int r0 = mode & 15, r1 = mode >> 4;           // (!WHOLE)
    double2* const tab = lds4 + 2 * kBuf;
    // (TERMS = 2: the engine's table is exp(-2 pi i m / 50 000); the 25 000-point transform's own twiddles are its even entries)
    const double2* __restrict__ tw = a.tw;
    auto twi = [&](int m) -> double2 { return tw[TERMS * m]; };
    // Thread roles, re-derived per transform from an opaque copy of the thread number: left loop-invariant the compiler
    // keeps every address of the loop body in registers across the whole loop -- and spills them.
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    // column stage / first half of a round: thread (r, c) of 5 x 100
    const bool live = t_ < 500;
    const int r = live ? t_ / 100 : 0, c = live ? t_ - 100 * r : 0;
    const int cb = r * N2 + c;                          // its slot in a buffer row group
    // Row stages: WHO reads WHAT is laid out for the LDS banks (round 6; tools/lds_conflicts_fused.py is the bank model, exact
    // against SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT: with threads numbered row-major -- 20, then 10 per row -- 38 % of
    // the kernel's LDS cycles were conflict cycles, 12.2 k of 32 k per transform).  A ds_read_b128 is served in four
    // groups of sixteen lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32, conflict-free when the sixteen
    // 16-byte elements differ modulo 16; a ds_write_b128 in eight groups of eight contiguous lanes, modulo 8.
    // (gq, gi): this lane's read group within the wave and its place in it.
    const int l5 = t_ & 31;
    const int gq = 2 * ((t_ >> 5) & 1) + ((l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28 ? 1 : 0);
    const int gi = l5 < 4 ? l5 : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : (l5 < 28 ? l5 - 12 : l5 - 16)));
    // row stage 1: thread (row i, e) of 25 x 20.  Read groups 0-24 of the workgroup: e = 0..15 of ONE row (sixteen
    // consecutive elements); groups 25-31: e = 16..19 of four rows each.  The two groups of a 32-lane half hold rows
    // whose exchange swizzles differ in bit 0 (below): their lanes share the write groups.
    const int g1 = 4 * (t_ >> 6) + gq;
    int role1, role2;      // (row, e | k'', swizzle) packed, < 0: the lane has no part in that stage -- unpacked where a round
                           // needs them (kept apart the seven values live through the column stage and spill)
    {
        int ri, re;
        bool live1 = true;
        if (g1 <= 24) {
            const int n = g1 >> 1;
            ri = 4 * (n >> 1) + (n & 1) + 2 * (g1 & 1);
            re = gi;
        } else {
            const int u = g1 - 25, q = gi >> 2;
            ri = 8 * (u >> 1) + ((u & 1) ? 0 : 2) + (q & 1) + 4 * (q >> 1);
            re = 16 + (gi & 3);
            if (g1 == 31) ri = 24, live1 = q == 0;
        }
        // The exchange between the row stages goes through the row IN PLACE, element (e, k'') at (10 e + k'') ^ swz(row),
        // swz(row) = 2 ((row >> 1) & 3) + ((row >> 1) & 1): the low three bits of the place only (200 = 25 x 8: closed).
        auto swz = [](int row) { return 2 * ((row >> 1) & 3) + ((row >> 1) & 1); };
        // (with what a round derives from them: k1b -- the row's k1 = rho + k1b -- and the first output index of stage 2)
        role1 = live1 ? (ri | re << 5 | swz(ri) << 10 | (5 * (ri / 5) + 25 * (ri % 5)) << 13) : -1;
        // row stage 2: waves 0-3 take the even outputs (h = 0), waves 4-7 the odd ones: thread (row, k'') of 25 x 10.  Read
        // groups 0-11 of a half: k'' = 0..7 of rows 2g, 2g + 1 (same swizzle, bases 8 elements apart modulo 16); 12-14:
        // k'' = 8, 9 of eight rows (eight different (row parity, swizzle >> 1)); 15: row 24.
        const int g2 = 4 * ((t_ >> 6) & 3) + gq;
        int si, sk;
        bool live2 = true;
        if (g2 < 12) si = 2 * g2 + (gi >> 3), sk = gi & 7;
        else if (g2 < 15) si = 8 * (g2 - 12) + (gi >> 1), sk = 8 + (gi & 1);
        else si = 24, sk = gi, live2 = gi < 10;
        role2 = live2 ? (si | sk << 5 | swz(si) << 10 | (5 * (si / 5) + 25 * (si % 5) + N1 * (sk + 10 * (tid >> 8))) << 13) : -1;
    }
    const int h = tid >> 8;

    if (tid < 97) tab[tid] = twi((N / 125) * tid);
    __syncthreads();   // the table; and every reader of the previous transform's last rounds is done with the buffers
    FUSED_STAMP(0);

    // ---- column stage: two items of 25 points; rounds 0 / 1 to the buffers, rounds 2-4 parked.  Register budget
    // (256 per lane, nothing may spill): a group of five points is multiplied and put through the first radix-5
    // stage while the NEXT group's ten loads are in flight; while item 1 is transformed, ten of item 0's fifteen
    // parked points wait in the LDS slots item 1 will fill at its end (a thread's own slots: no barrier).
    double2 park[2][15];
    const Operand xs_u = make_operand(a.spec + (a.spec_off ? (size_t)a.spec_off[bin] : (size_t)bin * NF));
    const Operand cs_u = make_operand(a.code_spec + ((size_t)prn * TERMS + par) * NF);
    const unsigned toff = (unsigned)cb * 16u;
    long long next_spec = 0;                                // (kPf) where the NEXT unit's spectrum starts: read now, used in round 3
    if constexpr (kPf)
        if (pf->next_whole && a.spec_off) next_spec = a.spec_off[pf->next_bin];
    // FUSED_EARLY_ITEM1 = 1 | 2 (N = 25 000, whole transforms): item 1's first one or two groups of operands are requested BEFORE
    // item 0's second radix-5 stage (~600 instructions per lane with nothing in flight for this wave, then the ~2 k cycles
    // until the first load of a fresh stream is back).  Measured, same box, ms per 32 x 41 search: off 0.2055, one group
    // 0.2038, two 0.2066 -- the other waves cover most of that gap already.
#ifndef FUSED_EARLY_ITEM1
#define FUSED_EARLY_ITEM1 1
#endif
    constexpr bool kEarly = FUSED_EARLY_ITEM1 && TERMS == 1 && WHOLE;       // (the short units keep nothing parked, but their bodies are not where the time is)
    constexpr int kEarlyGroups = FUSED_EARLY_ITEM1;         // 1 or 2
    constexpr int kRowBytes = N2 * 16;                      // one n1 step
    double2 xe[2][5], ce[2][5];                             // (kEarly) item 1's groups 0 and 1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double2* const eP = lds4 + cb + 100 * j;            // E slots of this (r, column): + 5 N2 kB; round 1: + kBuf
        double2 v[25];
        if (live) {
            double2 xa[5], ca[5];
#pragma unroll
            for (int m2 = 0; m2 < 5; ++m2) {
                if (kPf && j == 0) {
                    xa[m2] = ca[m2] = make_double2(0.0, 0.0);       // (unused: item 0 reads the prefetched groups)
                } else if (kEarly && j == 1) {
                    xa[m2] = xe[0][m2];
                    ca[m2] = ce[0][m2];
                } else {
                    xa[m2] = ldb(xs_u, toff, 1600 * j + kRowBytes * 25 * m2);
                    ca[m2] = ldb(cs_u, toff, 1600 * j + kRowBytes * 25 * m2);
                }
            }
            if (j == 1 && WHOLE) {
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    eP[5 * N2 * g] = park[0][g];
                    eP[kBuf + 5 * N2 * g] = park[0][5 + g];
                }
            }
            if (kPf && j == 0) {
                // (FUSED_PREFETCH_UNIT) the first kPfGroups groups were requested by the unit before (or are requested now: the
                // workgroup's first whole unit); the others a group ahead of their use, as in the plain loop
                if (!pf->have) {
#pragma unroll
                    for (int g = 0; g < kPfGroups; ++g)
#pragma unroll
                        for (int m2 = 0; m2 < 5; ++m2) {
                            pf->x[g][m2] = ldb(xs_u, toff, kRowBytes * 5 * (g + 5 * m2));
                            pf->c[g][m2] = ldb(cs_u, toff, kRowBytes * 5 * (g + 5 * m2));
                        }
                }
                double2 xn[5], cn[5];                        // the group after the one being multiplied
#pragma unroll
                for (int m1 = 0; m1 < 5; ++m1) {
                    double2 xc[5], cc[5];
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        xc[m2] = m1 < kPfGroups ? pf->x[m1][m2] : xn[m2];
                        cc[m2] = m1 < kPfGroups ? pf->c[m1][m2] : cn[m2];
                    }
                    if (m1 + 1 < 5 && m1 + 1 >= kPfGroups) {
#pragma unroll
                        for (int m2 = 0; m2 < 5; ++m2) {
                            xn[m2] = ldb(xs_u, toff, kRowBytes * 5 * (m1 + 1 + 5 * m2));
                            cn[m2] = ldb(cs_u, toff, kRowBytes * 5 * (m1 + 1 + 5 * m2));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double2 t[5];
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) t[m2] = cmulf(xc[m2], cc[m2]);
                    ibf5(t);
                    v[m1] = t[0];
#pragma unroll
                    for (int kA = 1; kA < 5; ++kA)
                        v[m1 + 5 * kA] = m1 ? cmul_conj(t[kA], make_double2(kW25X[m1 * kA], kW25Y[m1 * kA])) : t[kA];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (TERMS == 1) {
#pragma unroll
                for (int m1 = 0; m1 < 5; ++m1) {
                    double2 xb[5], cb_[5];
                    if (m1 < 4) {
#pragma unroll
                        for (int m2 = 0; m2 < 5; ++m2) {
                            if (kEarly && kEarlyGroups > 1 && j == 1 && m1 == 0) {
                                xb[m2] = xe[1][m2];
                                cb_[m2] = ce[1][m2];
                            } else {
                                xb[m2] = ldb(xs_u, toff, 1600 * j + kRowBytes * 5 * (m1 + 1 + 5 * m2));
                                cb_[m2] = ldb(cs_u, toff, 1600 * j + kRowBytes * 5 * (m1 + 1 + 5 * m2));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double2 t[5];
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) t[m2] = cmulf(xa[m2], ca[m2]);
                    ibf5(t);
                    v[m1] = t[0];
#pragma unroll
                    for (int kA = 1; kA < 5; ++kA)
                        v[m1 + 5 * kA] = m1 ? cmul_conj(t[kA], make_double2(kW25X[m1 * kA], kW25Y[m1 * kA])) : t[kA];
                    __builtin_amdgcn_sched_barrier(0);
                    if (m1 < 4) {
#pragma unroll
                        for (int m2 = 0; m2 < 5; ++m2) {
                            xa[m2] = xb[m2];
                            ca[m2] = cb_[m2];
                        }
                    }
                }
            } else {
                // two operand terms per point: P = F[k] C[k] + F[k + M] C[k + M].  Ten half-steps, two register sets that
                // take turns (no copies): while one term of a group of five points is multiplied, the ten loads of the next
                // half-step are in flight.  The products are PINNED where they are written (an empty volatile asm that
                // "modifies" them): left free, the arithmetic of all ten half-steps sinks below the last load -- the
                // scheduling barriers order memory operations, not pure arithmetic -- and 100 loads x 4 registers are
                // alive at once (the 25 000-point kernel lives with exactly that: its 50 loads fit).
                constexpr unsigned kHalf = (unsigned)N * 16u;      // k + M
                auto pin = [](double2& d) { asm volatile("" : "+v"(d.x), "+v"(d.y)); };
#pragma unroll
                for (int m1 = 0; m1 < 5; ++m1) {
                    double2 xb[5], cb_[5];
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        xb[m2] = ldb(xs_u, toff, kHalf + 1600 * j + kRowBytes * 5 * (m1 + 5 * m2));
                        cb_[m2] = ldb(cs_u, toff, kHalf + 1600 * j + kRowBytes * 5 * (m1 + 5 * m2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double2 t[5];
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        t[m2] = cmulf(xa[m2], ca[m2]);
                        pin(t[m2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (m1 < 4) {
#pragma unroll
                        for (int m2 = 0; m2 < 5; ++m2) {
                            xa[m2] = ldb(xs_u, toff, 1600 * j + kRowBytes * 5 * (m1 + 1 + 5 * m2));
                            ca[m2] = ldb(cs_u, toff, 1600 * j + kRowBytes * 5 * (m1 + 1 + 5 * m2));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        t[m2].x = __builtin_fma(-xb[m2].y, cb_[m2].y, __builtin_fma(xb[m2].x, cb_[m2].x, t[m2].x));
                        t[m2].y = __builtin_fma(xb[m2].y, cb_[m2].x, __builtin_fma(xb[m2].x, cb_[m2].y, t[m2].y));
                    }
                    ibf5(t);
                    v[m1] = t[0];
#pragma unroll
                    for (int kA = 1; kA < 5; ++kA)
                        v[m1 + 5 * kA] = m1 ? cmul_conj(t[kA], make_double2(kW25X[m1 * kA], kW25Y[m1 * kA])) : t[kA];
#pragma unroll
                    for (int kA = 0; kA < 5; ++kA) pin(v[m1 + 5 * kA]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kEarly && j == 0) {
#pragma unroll
                for (int g = 0; g < kEarlyGroups; ++g)
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        xe[g][m2] = ldb(xs_u, toff, 1600 + kRowBytes * 5 * (g + 5 * m2));
                        ce[g][m2] = ldb(cs_u, toff, 1600 + kRowBytes * 5 * (g + 5 * m2));
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // second stage of the 25-point transform and the twiddle between the column's two levels
#pragma unroll
            for (int kA = 0; kA < 5; ++kA) {
                if (WHOLE || r0 == kA || r1 == kA) {          // (a unit of one or two rounds: their outputs alone, straight into the buffers)
                    double2 t[5] = {v[5 * kA], v[5 * kA + 1], v[5 * kA + 2], v[5 * kA + 3], v[5 * kA + 4]};
                    ibf5(t);
#pragma unroll
                    for (int kB = 0; kB < 5; ++kB) {
                        const int kp = kA + 5 * kB;
                        v[5 * kA + kB] = kp ? cmul_conj(t[kB], tab[r * kp]) : t[kB];
                    }
                    if (!WHOLE) {
#pragma unroll
                        for (int kB = 0; kB < 5; ++kB) eP[(kA == r1 ? kBuf : 0) + 5 * N2 * kB] = v[5 * kA + kB];
                    }
                }
            }
            if (WHOLE) {
                if (j == 1) {
#pragma unroll
                    for (int g = 0; g < 5; ++g) {
                        park[0][g] = eP[5 * N2 * g];
                        park[0][5 + g] = eP[kBuf + 5 * N2 * g];
                    }
                }
#pragma unroll
                for (int kB = 0; kB < 5; ++kB) {
                    eP[5 * N2 * kB] = v[kB];
                    eP[kBuf + 5 * N2 * kB] = v[5 + kB];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 15; ++g) park[j][g] = v[10 + g];
        __builtin_amdgcn_sched_barrier(0);
        FUSED_STAMP(1 + j);
    }
    __syncthreads();
    FUSED_STAMP(3);

    // Y[k' + 25 q] = sum_r B_r[k'] w5^(r q) of one round, in place (a thread reads and writes its own five slots)
    auto y_in_place = [&](double2* X) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double2* const col = X + 5 * cb - 4 * c + 100 * j;      // (r * 5) * N2 + c
            double2 t5[5];
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) t5[rr] = col[rr * N2];
            ibf5(t5);
#pragma unroll
            for (int q = 0; q < 5; ++q) col[q * N2] = t5[q];
        }
    };
    // FUSED_SPLIT_S2Y (whole transforms; round 6; OFF: see the switch).  The Y step is LDS traffic with a sliver of arithmetic, the second row stage
    // arithmetic with a sliver of LDS traffic, and as phases of their own -- every wave in both, a barrier between -- neither
    // covers the other (2.5 k + 2.1 k cycles per round).  Round rho + 1's Y step needs nothing of round rho but its first
    // stage's stores into the other buffer: waves 4-7 do it (four butterflies per thread: 250 threads x 4 = the 1000 of a
    // round) WHILE waves 0-3 do both halves of round rho's second stage one after the other -- every SIMD holds one wave
    // of each kind.
    constexpr bool kSplit = FUSED_SPLIT_S2Y && WHOLE && !FUSED_MERGE_Y;
    auto y_by_upper_waves = [&](double2* X) {
        const int ty = t_ - 256;
        if (ty >= 0 && ty < 250) {
            const int yr = ty / 50, yc = ty - 50 * yr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double2* const col = X + (5 * yr) * N2 + yc + 50 * j;
                double2 t5[5];
#pragma unroll
                for (int rr = 0; rr < 5; ++rr) t5[rr] = col[rr * N2];
                ibf5(t5);
#pragma unroll
                for (int q = 0; q < 5; ++q) col[q * N2] = t5[q];
            }
        }
    };
    // the row's four-step twiddle w_N^(k1 (e + 20 m)) = base * step^m: two scattered table reads per round, requested
    // a phase ahead
#if FUSED_MERGE_Y
    if (live) y_in_place(lds4);
    if (tid < 180) tab[tid] = twi((N / 200) * ((tid % 20) * (tid / 20 + 1)));   // (every w125 read lies before the last barrier)
    __syncthreads();
    FUSED_STAMP(4);
#endif

    double best_sq = -1.0, best_x = 0.0, best_y = 0.0;
    int best_k = -1;
    // (squared, unscaled, a little low: the bound the lanes' squared magnitudes are screened against)
    double floor_sq = 0.0;
    if constexpr (!SECOND) {
        const double th = __longlong_as_double((long long)__hip_atomic_load(&a.theta[prn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * (double)NF;
        floor_sq = th * th * (1.0 - 0x1p-40);
    }
    // (unrolled: as a run-time loop -- FUSED_UNROLL_ROUNDS 1: buffer offsets and the parked registers' cases decided per round,
    // half the code -- a 32-PRN call measured 0.250 against 0.237 ms)
#ifndef FUSED_UNROLL_ROUNDS
#define FUSED_UNROLL_ROUNDS 5
#endif
#pragma unroll FUSED_UNROLL_ROUNDS
    for (int rho = 0; rho < 5; ++rho) {
        if (!WHOLE && rho != r0 && rho != r1) continue;
        double2* const X = lds4 + ((WHOLE ? (rho & 1) : rho == r1) ? kBuf : 0);
        double2* const Xo = lds4 + ((rho & 1) ? 0 : kBuf);
        int r1_ = role1, r2_ = role2;
        asm volatile("" : "+v"(r1_), "+v"(r2_));            // (this round's own unpacking: nothing of it lives across rounds)
        // (a lane without a part -- role < 0 -- unpacks to in-range garbage and is masked off wherever it would touch memory)
        const bool live1 = r1_ >= 0, live2 = r2_ >= 0;
        const int ri = r1_ & 31, re = (r1_ >> 5) & 31, sw1 = (r1_ >> 10) & 7, k1b = (r1_ >> 13) & 127;
        const int si = r2_ & 31, sk = (r2_ >> 5) & 15, sw2 = (r2_ >> 10) & 7, kf0 = (r2_ >> 13) & 4095;
        const double2 tw_base = twi((rho + k1b) * re), tw_step = twi(20 * (rho + k1b));
#if !FUSED_MERGE_Y
        // (kSplit: rounds 1-4 had their Y step done by waves 4-7 during the round before's second row stage, barrier included)
        if (!kSplit || rho == 0) {
            if (live) y_in_place(X);
            // w200^(e k''), k'' = 1..9, stored [k'' - 1][e]: the sixteen lanes of a read group take sixteen consecutive entries
            // (indexed e * k'' the strides 2, 4, 6, 8 cost 2-, 4-, 2-, 8-way conflicts: 1.6 k cycles per transform)
            if ((rho == 0 || !WHOLE) && tid < 180) tab[tid] = twi((N / 200) * ((tid % 20) * (tid / 20 + 1)));   // (every w125 read lies before the last barrier)
            __syncthreads();
        }
        FUSED_STAMP(4);
#endif
        // ---- rows, first stage (reads first; then the parked round rho + 1 moves into the buffer round rho - 1 has
        // left -- its stores drain while the transform computes)
        double2 z[10];
        if (WHOLE && live) {
#pragma unroll
            for (int rr = 1; rr <= 3; ++rr) {          // (wave-uniform cases: the register index must be a constant)
                if (rho == rr) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int kB = 0; kB < 5; ++kB) Xo[cb + 100 * j + 5 * N2 * kB] = park[j][5 * (rr - 1) + kB];
                }
            }
        }
        if (live1) {
            const double2* __restrict__ rowz = X + ri * N2 + re;
#pragma unroll
            for (int m = 0; m < 10; ++m) z[m] = rowz[20 * m];
            double2 t = tw_base;
            z[0] = cmul_conj(z[0], t);
#pragma unroll
            for (int m = 1; m < 10; ++m) {
                t = cmulf(t, tw_step);
                z[m] = cmul_conj(z[m], t);
            }
            idft10c(z);
#pragma unroll
            for (int g = 1; g < 10; ++g) {
                const int kpp = g / 2 + 5 * (g % 2);
                z[g] = cmul_conj(z[g], tab[20 * (kpp - 1) + re]);
            }
        }
        __syncthreads();   // every read of the row is done: the exchange goes in place
        FUSED_STAMP(5);
        if (live1) {
            // (byte arithmetic on the 32-bit LDS address by hand: the row starts on a multiple of 128 bytes -- 200 elements
            // of 16 -- so the swizzle, bits 4..6 of the byte address, is one exclusive or behind one addition per store)
            typedef __attribute__((address_space(3))) char lds_char;
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
            const unsigned roww = (unsigned)(size_t)((lds_char*)(X + ri * N2)) + 160u * (unsigned)re;
            const unsigned sw16 = (unsigned)sw1 << 4;
#pragma unroll
            for (int g = 0; g < 10; ++g) {
                const int kpp = g / 2 + 5 * (g % 2);
                *(lds_f64x2*)(size_t)((roww + 16u * kpp) ^ sw16) = f64x2{z[g].x, z[g].y};
            }
        }
        __syncthreads();
        FUSED_STAMP(6);
        if constexpr (kPf) {
            // (FUSED_PREFETCH_UNIT) every parked round has moved into its buffer: 120 registers are free from here to the
            // unit's end -- the next whole unit's first three operand groups go into them
            if (rho == 3 && pf->next_whole) {                  // (uniform: a per-lane condition would keep the old values alive in the lanes it leaves out; the twelve spare lanes read column 0)
                const Operand nxs = make_operand(a.spec + (a.spec_off ? (size_t)next_spec : (size_t)pf->next_bin * NF));
                const Operand ncs = make_operand(a.code_spec + (size_t)pf->next_prn * NF);
#pragma unroll
                for (int g = 0; g < kPfGroups; ++g)
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2) {
                        pf->x[g][m2] = ldb(nxs, toff, kRowBytes * 5 * (g + 5 * m2));
                        pf->c[g][m2] = ldb(ncs, toff, kRowBytes * 5 * (g + 5 * m2));
                    }
                __builtin_amdgcn_sched_barrier(0);
            } else if (rho == 3) {
                // (no next whole unit: the registers are "defined" all the same -- an empty statement -- so that on no path
                // this unit's consumed operands count as alive from the column stage to the next unit)
#pragma unroll
                for (int g = 0; g < kPfGroups; ++g)
#pragma unroll
                    for (int m2 = 0; m2 < 5; ++m2)
                        asm volatile("" : "=v"(pf->x[g][m2].x), "=v"(pf->x[g][m2].y), "=v"(pf->c[g][m2].x), "=v"(pf->c[g][m2].y));
            }
        }
        // ---- rows, second stage: 20 = 2 x 10, radix-2 decimation in frequency while reading; and the next round's Y
        // in place in the other buffer (its stores overlap this stage's arithmetic)
        double2 u[10];
        if (kSplit && tid >= 256) {
            if (rho < 4) y_by_upper_waves(Xo);              // (waves 4-7: the next round's Y step, in the other buffer)
        } else {
#pragma unroll 1
            for (int hh = 0; hh < (kSplit ? 2 : 1); ++hh) {  // (kSplit: both halves by the same lane, one after the other)
                const int hcur = kSplit ? hh : h;
            if (live2) {
                // element (e, k'') of the row sits at (10 e + k'') ^ swizzle: with e = 4 a + q that is ((k'' + 2 q) ^ swizzle) + 40 a + 8 q
                // -- four per-lane places, the rest a constant
                const double2* __restrict__ const row = X + si * N2;
                const double2* __restrict__ const rowq[4] = {row + (sk ^ sw2), row + ((sk + 2) ^ sw2), row + ((sk + 4) ^ sw2), row + ((sk + 6) ^ sw2)};
#pragma unroll
                for (int t0 = 0; t0 < 10; t0 += 5) {     // (five pairs of reads in flight: registers)
                    double2 lo[5], hi[5];
#pragma unroll
                    for (int t = 0; t < 5; ++t) {
                        const int e_lo = t0 + t, e_hi = t0 + t + 10;
                        lo[t] = rowq[e_lo & 3][10 * e_lo - 2 * (e_lo & 3)];
                        hi[t] = rowq[e_hi & 3][10 * e_hi - 2 * (e_hi & 3)];
                    }
#pragma unroll
                    for (int t = 0; t < 5; ++t) u[t0 + t] = hcur ? csub(lo[t], hi[t]) : cadd(lo[t], hi[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if FUSED_MERGE_Y
            if (rho < 4 && live) y_in_place(Xo);
#ifdef FUSED_MERGE_SB
            __builtin_amdgcn_sched_barrier(0);
#endif
#endif
            if (live2) {
                if (hcur) {
#pragma unroll
                    for (int t = 1; t < 10; ++t) u[t] = cmul_conj(u[t], make_double2(kW20X[t], kW20Y[t]));
                }
                idft10c(u);
                const int k_first = kf0 + rho + (kSplit ? N1 * 10 * hh : 0);
                double sqv[10];
#pragma unroll
                for (int g = 0; g < 10; ++g) {
                    sqv[g] = __builtin_fma(u[g].x, u[g].x, u[g].y * u[g].y);
                    if constexpr (SECOND) {        // (a column the peak comparison excludes never becomes a candidate: -2 loses against -1)
                        const int k = TERMS * (k_first + 20 * N1 * (g / 2 + 5 * (g % 2))) + par;     // (sample of the whole row)
                        sqv[g] = (k < a1 || (k >= b0 && k < b1)) ? sqv[g] : -2.0;
                    }
                }
                const double m01 = fmax(sqv[0], sqv[1]), m23 = fmax(sqv[2], sqv[3]), m45 = fmax(sqv[4], sqv[5]), m67 = fmax(sqv[6], sqv[7]),
                             m89 = fmax(sqv[8], sqv[9]);
                const double round_max = fmax(fmax(fmax(m01, m23), fmax(m45, m67)), m89);
                // (the lane's own best with the same margin: a value within 2^-48 of it must still meet it below)
                const bool candidate = round_max >= fmax(floor_sq, best_sq * (1.0 - 0x1p-40));
                if (__any(candidate)) {
#ifdef SDR_FUSED_STAMPS
                    if (tid == 0) g_fused_stamps[blockIdx.x & 255][6] += 1000000;    // (diagnostic: rounds of wave 0 that keep books)
#endif
                    // Ordering by the squared magnitude.  A candidate within 2^-48 of the lane's best has to be compared through
                    // the scaled hypot -- the reference's np.abs -- with an exact tie keeping the smaller index (pcps_fast.h): the
                    // round is then redone from the state it started with by ONE run-time loop over a copy of the ten values
                    // (a private array: scratch memory, touched on this path alone).  Inlined per candidate that comparison
                    // made the kernel several times the instruction cache.
                    const double sq0 = best_sq, x0 = best_x, y0 = best_y;
                    const int k0 = best_k;
                    bool tie = false;
#pragma unroll
                    for (int g = 0; g < 10; ++g) {
                        const int p = g / 2 + 5 * (g % 2);
                        const int k = k_first + 20 * N1 * p;              // code phase = position in the transform
                        const double sq = sqv[g];
                        const bool take = sq > best_sq;
                        tie |= sq >= 0.0 && fabs(sq - best_sq) <= best_sq * 0x1p-48;
                        best_sq = take ? sq : best_sq;
                        best_x = take ? u[g].x : best_x;
                        best_y = take ? u[g].y : best_y;
                        best_k = take ? k : best_k;
                    }
                    if (__builtin_expect(__any(tie), 0)) {
                        double2 copy[10];
#pragma unroll
                        for (int g = 0; g < 10; ++g) copy[g] = u[g];
                        best_sq = sq0;
                        best_x = x0;
                        best_y = y0;
                        best_k = k0;
#pragma unroll 1
                        for (int g = 0; g < 10; ++g) {
                            const int p = g / 2 + 5 * (g % 2);
                            const int k = k_first + 20 * N1 * p;
                            const double2 x = copy[g];
                            double sq = __builtin_fma(x.x, x.x, x.y * x.y);
                            if (SECOND && !(TERMS * k + par < a1 || (TERMS * k + par >= b0 && TERMS * k + par < b1))) sq = -2.0;
                            bool take = sq > best_sq;
                            if (sq >= 0.0 && fabs(sq - best_sq) <= best_sq * 0x1p-48) {
                                const double m_new = hypot(x.x * a.scale, x.y * a.scale), m_old = hypot(best_x * a.scale, best_y * a.scale);
                                take = m_new > m_old || (m_new == m_old && k < best_k);
                            }
                            best_sq = take ? sq : best_sq;
                            best_x = take ? x.x : best_x;
                            best_y = take ? x.y : best_y;
                            best_k = take ? k : best_k;
                        }
                    }
                }
            }
            }
        }
        if (kSplit && rho < 4) __syncthreads();             // (the Y step's stores; and this round's readers are done with X, which round rho + 1 overwrites)
#if FUSED_MERGE_Y
        if (rho < 4) __syncthreads();   // (the last round's readers meet the next transform's first barrier)
#endif
        FUSED_STAMP(7);
    }
    if constexpr (kPf)
        pf->have = pf->next_whole;                              // (what was requested in round 3 is the next unit's)
    int best_i = 0x7fffffff;
    double best_v = -1.0;
    if (role2 >= 0 && best_k >= 0) {
        best_i = (SECOND ? 0 : bin * NF) + TERMS * best_k + par;
        best_v = 0.0 + hypot(best_x * a.scale, best_y * a.scale);   // (0.0 + |.|: the map's own rounding)
    }
    wave_best(best_v, best_i);
    if ((tid & 63) == 63) {
        Best rec = {best_v, (long long)best_i};
        a.partials[(size_t)rec_slot * kRecordsPerTransform + (tid >> 6)] = rec;
        if constexpr (!SECOND)
            if (best_v > 0.0) atomicMax(&a.theta[prn], (unsigned long long)__double_as_longlong(best_v));   // (positive doubles order as integers)
    }
}

template <int TERMS>
__global__ __launch_bounds__(kThreads) void ifft_max_kernel(const Args a) {
    extern __shared__ double2 lds4[];
#ifdef SDR_FUSED_STAMPS
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#define FUSED_STAMP_ARG , 0, 0, 0, par, nullptr, stamp_
#define FUSED_STAMP_ARG_PF , 0, 0, 0, par, &pf, stamp_
#else
#define FUSED_STAMP_ARG , 0, 0, 0, par
#define FUSED_STAMP_ARG_PF , 0, 0, 0, par, &pf
#endif
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (blockIdx.x == 0)
        for (int p = tid; p < a.n_prn; p += kThreads) a.theta_next[p] = 0ull;
    const int w_end = a.xcd_first[xcd + 1];
    // (requesting the NEXT unit's list entry a unit ahead by itself measured 0.2037 against 0.2023 ms per search at 25 MHz: the
    // other waves cover those latencies; it is read ahead here because FUSED_PREFETCH_UNIT needs to know the next unit)
    UnitPrefetch pf;
    pf.have = false;
    for (int w = a.xcd_first[xcd] + slot; w < w_end; w += kSlotsPerXcd) {
        // (wave-uniform: scalar base addresses in the unit)
        const int prn = __builtin_amdgcn_readfirstlane(a.work[w].prn);
        int bin = __builtin_amdgcn_readfirstlane(a.work[w].bin), par = 0;
        if (TERMS == 2 && bin >= a.nbins) bin -= a.nbins, par = 1;            // (virtual bin = parity * nbins + bin)
        const int mode = __builtin_amdgcn_readfirstlane(a.work[w].round);     // -1: the whole transform; else its rounds r0 | r1 << 4
        const int rec_slot = __builtin_amdgcn_readfirstlane(a.work[w].record);
        pf.next_whole = false;
        pf.next_prn = pf.next_bin = 0;
        if (FUSED_PREFETCH_UNIT > 0 && TERMS == 1 && w + kSlotsPerXcd < w_end) {
            pf.next_whole = __builtin_amdgcn_readfirstlane(a.work[w + kSlotsPerXcd].round) < 0;
            pf.next_prn = __builtin_amdgcn_readfirstlane(a.work[w + kSlotsPerXcd].prn);
            pf.next_bin = __builtin_amdgcn_readfirstlane(a.work[w + kSlotsPerXcd].bin);
        }
        if (mode < 0) {
            one_unit<true, false, TERMS, true>(a, lds4, tid, prn, bin, mode, rec_slot FUSED_STAMP_ARG_PF);
        } else {
            // (nothing requested ahead lives across a short unit: "defined" here by an empty statement, so that the register
            // allocator does not carry thirty operands through this body)
#pragma unroll
            for (int g = 0; g < kPfGroups; ++g)
#pragma unroll
                for (int m2 = 0; m2 < 5; ++m2)
                    asm volatile("" : "=v"(pf.x[g][m2].x), "=v"(pf.x[g][m2].y), "=v"(pf.c[g][m2].x), "=v"(pf.c[g][m2].y));
            one_unit<false, false, TERMS>(a, lds4, tid, prn, bin, mode, rec_slot FUSED_STAMP_ARG);
            pf.have = false;
        }
    }
}

// The second sweep of a map-free search in one launch of n_prn x 5 workgroups: workgroup (PRN p, round rho) finds PRN p's first
// peak from the records of the first sweep (every one of the five the same few KB: cheaper than a launch in front of it;
// the round-0 workgroup hands it on), then does round rho of that PRN's winning row with the maximum over the allowed
// columns -- five records sets of kRecordsPerTransform per PRN in `seconds` for the ratio kernel.  Replaces the peak
// kernel and the general four-step pair (three launches, ~26 us of a 32-PRN call).
struct SecondArgs {
    Args a;                    // spec / code_spec / tw / scale as in the first sweep; a.partials = seconds
    const Best* recs;          // [n_prn][per_prn] records of the first sweep
    int per_prn, n_prn;
    int spc;                   // samples per chip (the exclusion window's half width)
    Best* tops;                // [n_prn] out
    long long* out_bin;        // [n_prn] out
    long long* out_code;
    // the ratio of the two peaks by whichever of the PRN's workgroups finishes LAST (a ticket per PRN; that workgroup
    // sets it back to zero): no launch of its own for 32 divisions
    unsigned* tickets;         // [n_prn], zero between launches
    long long* res_bin;        // [n_prn] the caller's results (page-locked memory or the device copies themselves)
    long long* res_code;
    double* res_ratio;
    unsigned* done;            // (nullable) [n_prn] page-locked: raised to done_seq behind a PRN's results
    unsigned done_seq;
};

template <int TERMS>
__global__ __launch_bounds__(kThreads) void ifft_second_kernel(const SecondArgs s) {
    extern __shared__ double2 lds4[];
    constexpr int NF = TERMS * N;
    const int tid = threadIdx.x;
    // (the workgroups of a PRN read the same arrays: block numbers equal modulo 8 share an XCD's L2)
    // TERMS = 1: five workgroups per PRN, one round each.  TERMS = 2: three per parity -- rounds {0, 1}, {2, 3}, {4} -- six per
    // PRN: 192 workgroups for 32 PRNs, ONE wave of workgroups on the 256 compute units where ten per PRN were a wave and a quarter
    constexpr int kUnits = TERMS == 2 ? 6 : 5;
    const int prn8 = (int)(gridDim.x / kUnits);
    const int unit = blockIdx.x / prn8, prn = blockIdx.x - unit * prn8;
    const int par = TERMS == 2 ? unit / 3 : 0;
    const int piece = TERMS == 2 ? unit - 3 * par : unit;
    const int rounds = TERMS == 2 ? (piece == 2 ? (4 | 15 << 4) : (2 * piece | (2 * piece + 1) << 4)) : (piece | 15 << 4);
    if (prn >= s.n_prn) return;
    // first peak: larger value, smaller flat index on ties (np.argmax's first occurrence)
    Best* const sh = reinterpret_cast<Best*>(lds4);
    {
        double v = -1.0;
        long long i = 0x7fffffffffffffffLL;
        for (int k = tid; k < s.per_prn; k += kThreads) {
            const Best b = s.recs[(size_t)prn * s.per_prn + k];
            if (b.v > v || (b.v == v && b.i < i)) v = b.v, i = b.i;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(v, off, 64);
            const long long oi = __shfl_down(i, off, 64);
            if (ov > v || (ov == v && oi < i)) v = ov, i = oi;
        }
        if ((tid & 63) == 0) sh[tid >> 6] = Best{v, i};
    }
    __syncthreads();
    Best top = sh[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w)
        if (sh[w].v > top.v || (sh[w].v == top.v && sh[w].i < top.i)) top = sh[w];
    if (top.v < 0.0) top.i = 0;          // (no record at all: never an index a row is read with)
    const int bin = __builtin_amdgcn_readfirstlane((int)(top.i / NF));
    const int code = __builtin_amdgcn_readfirstlane((int)(top.i - (long long)bin * NF));
    if (unit == 0 && tid == 0) {
        s.tops[prn] = top;
        s.out_bin[prn] = bin;
        s.out_code[prn] = code;
    }
    int a1 = 0, b0 = 0, b1 = 0;
    {
        const int e0 = code - s.spc, e1 = code + s.spc;
        if (e0 < 1) {
            b0 = e1;
            b1 = NF - 1;
        } else if (e1 >= NF) {
            a1 = e0;
        } else {
            a1 = e0;
            b0 = e1;
            b1 = NF - 1;
        }
    }
    __syncthreads();                     // (the record scratch is the first buffer)
#ifdef SDR_FUSED_STAMPS
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    one_unit<false, true, TERMS>(s.a, lds4, tid, prn, bin, rounds, prn * kUnits + unit, a1, b0, b1, par, nullptr, stamp_);
#else
    one_unit<false, true, TERMS>(s.a, lds4, tid, prn, bin, rounds, prn * kUnits + unit, a1, b0, b1, par);
#endif
    // ---- TwoCorrelationPeakComparison's ratio (acquisition.py:113) by the PRN's last workgroup.  Every workgroup makes its
    // records visible device-wide (release), takes a ticket; the one that drew the last ticket reads all of the PRN's records
    // (loads that go to memory, not to this XCD's L2) and divides.  No workgroup waits for another: nothing can hang.
    // (ONE release per workgroup, no acquire: an acquire invalidates the XCD's L2 under the workgroups still streaming their
    // operands from it -- with a fence per thread and an acq_rel ticket the call measured 0.250 ms against 0.228 at N = 25 000)
    __syncthreads();
    unsigned* const sh_ticket = reinterpret_cast<unsigned*>(lds4);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        *sh_ticket = __hip_atomic_fetch_add(&s.tickets[prn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*sh_ticket != (unsigned)(kUnits - 1) || tid >= 64) return;
    constexpr int kRec = kUnits * kRecordsPerTransform;
    double v = -1.0;
    long long i = 0x7fffffffffffffffLL;
    for (int k = tid; k < kRec; k += 64) {
        const Best* const rec = s.a.partials + (size_t)prn * kRec + k;
        const double rv = __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(&rec->v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const long long ri = __hip_atomic_load(&rec->i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rv > v || (rv == v && ri < i)) v = rv, i = ri;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off, 64);
        const long long oi = __shfl_down(i, off, 64);
        if (ov > v || (ov == v && oi < i)) v = ov, i = oi;
    }
    if (tid == 0) {
        s.res_ratio[prn] = v >= 0.0 ? top.v / v : __builtin_nan("");
        if (s.res_bin != s.out_bin) {
            s.res_bin[prn] = bin;
            s.res_code[prn] = code;
        }
        __hip_atomic_store(&s.tickets[prn], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s.done) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // (the PRN's results are in the host's memory ...)
            __hip_atomic_store(&s.done[prn], s.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before this word
        }
    }
}

// Processing order and record slots of a search over bins [0, nbins) x n_prn PRNs.  The first `bins_whole` bins are done
// as whole transforms: blocks of 4 bins x 8 PRNs (32 transforms = one round of an XCD's 32 workgroups, so that the
// operands an XCD has in flight mostly sit in its L2), dealt to the XCDs in contiguous eighths; the transforms of the
// remaining bins are cut into their five rounds, dealt out behind the whole ones (a workgroup's last, short unit).
// Record slots are PRN-major: PRN p owns slots [p * per_prn, (p + 1) * per_prn), per_prn = bins_whole + 5 (nbins - bins_whole).
inline int make_work_list(int n_prn, int nbins, int bins_whole, std::vector<WorkItem>& order, int first[9], int pieces = 5,
                          int block_bins = 4) {
    // block_bins: bins of a block of 32 whole transforms (x 32 / block_bins PRNs).  4 x 8 shares 4 spectra + 8 code spectra;
    // with SHARED spectra (pcps.hip: a handful of class arrays serve every bin, and stay in the L2s) 16 x 2 shares the classes
    // + 2 code spectra, 32 x 1 the classes + 1: half the operand arrays an XCD has in flight, or fewer.
    const int block_prns = kSlotsPerXcd / block_bins;
    // pieces: 5 = one unit per round; 3 = rounds {0, 1}, {2, 3}, {4} (N = 50 000: the operand stage of a unit costs what four
    // rounds do, and 64 left-over transforms x 3 fit ONE wave of workgroups where x 5 need a wave and a quarter)
    const int per_prn = bins_whole + pieces * (nbins - bins_whole);
    std::vector<WorkItem> whole, parts;
    for (int b0 = 0; b0 < bins_whole; b0 += block_bins)
        for (int p0 = 0; p0 < n_prn; p0 += block_prns)
            for (int b = b0; b < b0 + block_bins && b < bins_whole; ++b)
                for (int p = p0; p < p0 + block_prns && p < n_prn; ++p) whole.push_back({p, b, -1, p * per_prn + b});
    for (int b = bins_whole; b < nbins; ++b)
        for (int p = 0; p < n_prn; ++p)
            for (int u = 0; u < pieces; ++u) {
                const int rounds = pieces == 5 ? (u | 15 << 4) : (u == 2 ? (4 | 15 << 4) : (2 * u | (2 * u + 1) << 4));
                parts.push_back({p, b, rounds, p * per_prn + bins_whole + pieces * (b - bins_whole) + u});
            }
    // XCD x: its eighth of the short units, then its eighth of the whole transforms (its workgroups walk the list in steps
    // of 32)
    order.clear();
    const int nw = (int)whole.size(), np = (int)parts.size();
    first[0] = 0;
    // The short units go FIRST (round 5): three workgroups in four start with one and run the rest of the launch ~3/4 of a
    // unit out of step with the others -- the operand stages of an XCD's 32 workgroups no longer all fall into the same
    // moments -- and no workgroup is left holding a short unit while the others have finished.  Measured, one box, ms of
    // kernels per 32-PRN call: N = 50 000 0.532 -> 0.511, N = 25 000 0.235 -> 0.228 (gpurun_out/r05_partsfirst.txt).
    constexpr bool parts_first = true;
    for (int x = 0; x < 8; ++x) {
        const int p_lo = (int)(((long long)np * x) / 8), p_hi = (int)(((long long)np * (x + 1)) / 8);
        const int w_lo = (int)(((long long)nw * x) / 8), w_hi = (int)(((long long)nw * (x + 1)) / 8);
        if (parts_first) {
            for (int i = p_lo; i < p_hi; ++i) order.push_back(parts[i]);
            // the XCD's 32 workgroups walk the list 32 items at a time, and the whole transforms come in blocks of 32 that share
            // their operand arrays: the step the short units leave incomplete is filled with the LAST whole transforms, so that
            // every later step is one block again (with the blocks straddling two steps the launch fetched 1.34 instead of
            // 0.87 GB through the fabric at N = 50 000)
            int pad = (kSlotsPerXcd - (p_hi - p_lo) % kSlotsPerXcd) % kSlotsPerXcd;
            if (pad > w_hi - w_lo) pad = w_hi - w_lo;
            for (int i = w_hi - pad; i < w_hi; ++i) order.push_back(whole[i]);
            for (int i = w_lo; i < w_hi - pad; ++i) order.push_back(whole[i]);
        } else {
            for (int i = w_lo; i < w_hi; ++i) order.push_back(whole[i]);
            for (int i = p_lo; i < p_hi; ++i) order.push_back(parts[i]);
        }
        first[x + 1] = (int)order.size();
    }
    return per_prn;
}

}  // namespace fused25k
