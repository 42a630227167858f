// On-device loop closure: persistent workgroups run
// correlate -> discriminators -> loop filters -> NCO update for n_epochs without leaving the
// GPU.  A channel is served by a CLUSTER of `parts` workgroups (1, 2, 4 or 8,
// chosen so that the launch fills the GPU: 32 channels x 8 parts = 256 CUs): every epoch each part
// correlates its share of the samples, publishes six partial sums through global memory, waits for
// its peers' and adds all of them in the same fixed order -- so every part holds bit-identical
// totals and runs the scalar loop update redundantly; there is no second exchange.  The PRN replica
// stays in LDS for the whole run; the loop arithmetic is fp64, spread over three waves by dependency
// (carrier loop / code loop / lock indicators + state machine + bit decisions) and, inside each,
// over lanes for the divisions, roots and arctangents -- following the two reference plugins
// statement by statement:
//   kind 0  Borre  : channel_l1ca_borre.py:333-451  (DLL NNEML + Costas PLL, Borre filters, np.pi NCO)
//   kind 1  Kaplan : channel_l1ca_kaplan.py:342-619 (FLL-assisted 2nd-order PLL, lock-state machine,
//                    GPS-ICD pi in the NCO and the discriminators: SURVEY.md T3)
// built on sydr/dsp/tracking.py:120-186,246-279 and sydr/dsp/lockindicator.py:6-122.
#include "correlator.h"

#ifdef SDR_TRACE_TRACK
// Debug build only (tools/track_phases.py): per-phase clock totals of channel 0's epoch loop.
__device__ unsigned long long g_track_phase[32];
extern "C" int sdr_debug_track_phases(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_track_phase), sizeof(g_track_phase));
}
#define TRACK_MARK(k)                                                     \
    do {                                                                  \
        if (tid == 0 && ch == 0 && part == 0) {                           \
            const unsigned long long now_ = wall_clock64();               \
            g_track_phase[k] += now_ - mark_;                             \
            mark_ = now_;                                                 \
        }                                                                 \
    } while (0)
#else
#define TRACK_MARK(k) ((void)0)
#endif

namespace {

using namespace sdr;

constexpr int kTaps = 3;
constexpr int kMaxParts = 8;
constexpr int red_doubles(int threads) { return (threads / 64) * 2 * kTaps > 64 ? (threads / 64) * 2 * kTaps : 64; }
constexpr int kXchgWords = 16;             // 12 tagged half-values + padding: one 128-byte line per part and parity
constexpr long kSpinLimit = 1L << 20;      // peer polls before a part gives up (about a second): never hang the GPU

constexpr double kGpsPi = 3.1415926535898;  // sydr/utils/constants.py:4
constexpr double kGpsTwoPi = kGpsPi * 2.0;
constexpr double kGpsHalfPi = kGpsPi / 2.0;
constexpr double kChips = 1023.0;
constexpr int kMsPerBit = 20;
constexpr double kW0Bw1 = 0.25, kW0Bw2 = 0.53, kW0A2 = 1.414;

enum { FLAG_CODE_LOCK = 1, FLAG_BIT_SYNC = 2 };
enum { LOCK_PULL_IN = 1, LOCK_WIDE = 2, LOCK_NARROW = 3 };

// Python / NumPy float modulo (result takes the sign of the divisor).
__device__ __forceinline__ double py_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}
__device__ __forceinline__ double np_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : (x == 0.0 ? 0.0 : x)); }

// The discriminators and filters of sydr/dsp/tracking.py -- DLL NNEML (:120-129), Costas PLL (:133-142),
// FLL atan (:156-176), BorreLoopFilter (:180-186) -- are evaluated inside the kernel's loop update, their
// divisions / square roots / arctangents spread over lanes (see there).

struct alignas(16) EpochShared {  // (size a multiple of 16: the replica behind it is copied with 16-byte stores)
    EpochParams ep;
    double spacing[kTaps];
    double dphi;
    int epochs_done;
    int fault;                 // a peer part never showed up: leave the epoch loop (reported to the host)
    // what the three update roles hand to each other (written before an epoch's first barrier, read after it)
    double corr[2 * kTaps];    // this epoch's correlator totals, for the roles on waves 1 and 2
    double fll_bw, pll_bw;     // Kaplan bandwidths chosen by the lock-state machine, for the carrier loop
    int lock_state;            // lock state the NEXT epoch's discriminators run under
    int c_code_counter, l_code_counter, l_bits_run;  // private copies of the roles on waves 0 and 2
    int stop_code, stop_carrier;  // the next epoch would leave the staged replica / the ring, or the carrier NCO is not finite
    double smin, smax;         // extreme tap offsets over both tap sets (constant for the run)
    double l_ipp, l_qpp;       // the lock role's copy of the previous prompt (the carrier role owns st.i/q_prompt_prev)
    sdr_track_state st;        // the loop state; each update role owns a disjoint set of its fields
    sdr_loop_cfg cfg;
};

// WAVES: resident waves per SIMD the register allocation has to leave room for (2 = two 256-thread
// workgroups can share a CU, at the price of a few spills).
template <int FMT, int THREADS, int WAVES>
__global__ __launch_bounds__(THREADS, WAVES) void track_kernel(const void* __restrict__ ring, int64_t capacity,
                                                        sdr_track_state* __restrict__ states,
                                                        const sdr_loop_cfg* __restrict__ cfg_ptr,
                                                        int n_epochs, sdr_track_epoch* __restrict__ traj,
                                                        int keep_traj, int8_t* __restrict__ nav_bits, int max_bits,
                                                        int32_t* __restrict__ n_bits,
                                                        const uint32_t* __restrict__ luts,
                                                        int lut_words, int lut_stride, int use_prefix,
                                                        int n_ch, int parts, unsigned long long* xchg,
                                                        int* __restrict__ fault) {
    extern __shared__ double smem[];
    double* red = smem;                                   // kWaves * 6 wave sums; reused by the cluster exchange
    EpochShared* sh = reinterpret_cast<EpochShared*>(red + red_doubles(THREADS));
    double2* prefix = reinterpret_cast<double2*>(sh + 1);  // THREADS * kPrefixSlots, when the launcher found room
    uint32_t* lut = reinterpret_cast<uint32_t*>(prefix + (use_prefix ? THREADS * kPrefixSlots : 0));

    const int tid = threadIdx.x;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8): the parts of one channel are
    // blockIdx-es with the same residue, so a cluster shares one XCD's L2 for its exchange lines.
    int ch, part;
    if (parts == 1 || gridDim.x % (8 * parts) != 0) {
        ch = blockIdx.x / parts;
        part = blockIdx.x % parts;
    } else {
        const int per_xcd = gridDim.x / 8;               // workgroups per XCD = channels per XCD * parts
        const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
        ch = xcd * (per_xcd / parts) + q / parts;
        part = q % parts;
    }
    const int lane_global = part * THREADS + tid;          // index among the cluster's lanes
    const int cluster_lanes = parts * THREADS;
    const bool edge_wave = lane_global >= cluster_lanes - 64;
    const int edge_lane = edge_wave ? lane_global - (cluster_lanes - 64) : -1;
    if (tid == 0) {
        sh->st = states[ch];
        sh->cfg = *cfg_ptr;
        sh->epochs_done = 0;
        sh->fault = 0;
        sh->fll_bw = states[ch].fll_bw;
        sh->pll_bw = states[ch].pll_bw;
        sh->lock_state = states[ch].lock_state;
    }
    const int slot = states[ch].code_slot;
    stage_lut<THREADS>(lut, luts + (size_t)slot * lut_stride, lut_words, tid);
    const double fs = cfg_ptr->fs;
    sdr_track_state& st = sh->st;
    const sdr_loop_cfg& cfg = sh->cfg;
    const bool writer = part == 0;                         // one part records trajectory, bits and the end state

#ifdef SDR_TRACE_TRACK
    unsigned long long mark_ = wall_clock64();
    if (tid == 0 && ch == 0 && part == 0) for (int k = 0; k < 32; ++k) g_track_phase[k] = 0;
    __syncthreads();
#endif
    // The loop update is split over three waves by dependency (see below); each role owns a disjoint set of fields of
    // the state's LDS copy (loaded into registers for the duration of its update only: carried across the correlation
    // they cost ~50 VGPRs and spill) and publishes its part of the next epoch's parameters.
    const int role = tid >> 6, rlane = tid & 63;
    const sdr_track_state s_init = states[ch];
    // The replica LUT and the ring bound what an epoch may touch; a loop that has run away (loss of lock) stops
    // instead of reading out of range.  Checked by the role that produces the values, for the epoch it announces.
    auto code_out_of_range = [&](int64_t start, int n, double rem_code, double code_step) {
        const double lo = ceil(rem_code + sh->smin);
        const double hi = ceil(code_step * (double)n + rem_code + sh->smax);
        return !(n > 0 && (int64_t)n <= capacity && code_step > 0.0 && lo >= -(double)SDR_LUT_PAD &&
                 hi <= (double)(lut_words - SDR_LUT_PAD - 2) && start >= 0);
    };
    auto carrier_bad = [&](double hz) { return !(hz == hz && fabs(hz) < 1e9); };
    if (tid == 0) {  // parameters of the first epoch
        double smin = cfg_ptr->spacing_wide[0], smax = smin;
        for (int t = 0; t < kTaps; ++t) {
            smin = fmin(smin, fmin(cfg_ptr->spacing_wide[t], cfg_ptr->spacing_narrow[t]));
            smax = fmax(smax, fmax(cfg_ptr->spacing_wide[t], cfg_ptr->spacing_narrow[t]));
        }
        sh->smin = smin;
        sh->smax = smax;
        sh->stop_code = code_out_of_range(s_init.current_sample, s_init.n_samples, s_init.rem_code, s_init.code_step) ? 1 : 0;
        sh->stop_carrier = carrier_bad(s_init.carrier_hz) ? 1 : 0;
        const double* sp = s_init.spacing_sel ? cfg_ptr->spacing_narrow : cfg_ptr->spacing_wide;
        sh->ep.start_sample = s_init.current_sample;
        sh->ep.n = s_init.n_samples;
        sh->ep.carrier_hz = s_init.carrier_hz;
        sh->ep.rem_carrier = s_init.rem_carrier;
        sh->ep.rem_code = s_init.rem_code;
        sh->ep.code_step = s_init.code_step;
        for (int t = 0; t < kTaps; ++t) sh->spacing[t] = sp[t];
        sh->dphi = carrier_step(s_init.carrier_hz, fs);
        sh->c_code_counter = sh->l_code_counter = s_init.code_counter;
        sh->l_ipp = s_init.i_prompt_prev;
        sh->l_qpp = s_init.q_prompt_prev;
        sh->l_bits_run = 0;
    }
    for (int epoch = 0; epoch < n_epochs; ++epoch) {
        TRACK_MARK(5);
        __syncthreads();
        TRACK_MARK(0);
#ifdef SDR_TRACE_TRACK
        const unsigned long long wave_mark_ = wall_clock64();
#endif
        if (sh->fault | sh->stop_code | sh->stop_carrier) break;
        // What comes out of LDS is the same in every lane, but only readfirstlane tells the compiler so:
        // as scalars the epoch parameters (and everything derived from them: group counts, ring positions,
        // linspace constants) live in SGPRs and are computed on the scalar unit -- ~100 VGPRs per lane.
        EpochParams ep;
        {
            const EpochParams v = sh->ep;
            ep.start_sample = ((int64_t)__builtin_amdgcn_readfirstlane((int)(v.start_sample >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)v.start_sample);
            ep.n = __builtin_amdgcn_readfirstlane(v.n);
            ep.carrier_hz = uniform(v.carrier_hz);
            ep.rem_carrier = uniform(v.rem_carrier);
            ep.rem_code = uniform(v.rem_code);
            ep.code_step = uniform(v.code_step);
        }
        const double dphi = uniform(sh->dphi);
        // what the carrier loop needs from the state machine's previous decision (captured now: the lock role
        // rewrites these while the carrier role is still running)
        const double cur_fll_bw = uniform(sh->fll_bw), cur_pll_bw = uniform(sh->pll_bw);
        const int cur_lock_state = __builtin_amdgcn_readfirstlane(sh->lock_state);
        EpochConsts<kTaps> K;
        compute_constants<kTaps>(K, ep, sh->spacing, dphi, cluster_lanes);
        TRACK_MARK(1);

        double accr[kTaps], acci[kTaps];
        const bool boundary_ok = use_prefix && ep.code_step >= kFastMinCodeStep && !epoch_wraps(ep, capacity);  // (uniform)
        if (boundary_ok && ep.code_step <= kFastMaxCodeStep)         // 16-sample boundary variant above ~17 MHz
            correlate_epoch_wide<FMT, kTaps, false, 16>(ring, capacity, ep, dphi, K, lut, prefix, tid, lane_global, cluster_lanes, edge_lane, accr, acci);
        else if (boundary_ok && ep.code_step <= kFastMaxCodeStep8)   // 8-sample boundary variant above ~8.2 MHz
            correlate_epoch_wide<FMT, kTaps, false, 8>(ring, capacity, ep, dphi, K, lut, prefix, tid, lane_global, cluster_lanes, edge_lane, accr, acci);
        else
            correlate_epoch<FMT, kTaps>(ring, capacity, ep, dphi, K, lut, lane_global, cluster_lanes, edge_lane, accr, acci);
#ifdef SDR_TRACE_TRACK
        if ((tid & 63) == 0 && ch == 0) g_track_phase[8 + part * (THREADS / 64) + (tid >> 6)] += wall_clock64() - wave_mark_;
#endif
        TRACK_MARK(2);
        double total = reduce_taps<kTaps, THREADS>(accr, acci, red, tid);
        TRACK_MARK(3);

        // Cluster exchange (wave 0): publish this part's six sums, collect the peers', add all parts in
        // part order.  No fences, no separate flag: every 64-bit word carries half a double and the epoch
        // tag (epoch+1), is written and read whole, and validates itself (the "LL" idea of collective
        // libraries) -- an agent-scope release/acquire pair would write back and invalidate the whole L2
        // every epoch (measured: 2.9 us); a relaxed device-scope word costs ~0.4 us one way
        // (tools/ubench_xchg.hip), on the same XCD or across XCDs.
        // Lines are double-buffered by epoch parity: a part can run at most one exchange ahead of a peer.
        if (parts > 1 && tid < 64) {
            unsigned long long* lines = xchg + ((size_t)ch * 2 + (epoch & 1)) * kMaxParts * kXchgWords;
            const unsigned long long tag = (unsigned long long)(unsigned)(epoch + 1) << 32;
            if (tid < 4 * kTaps) {  // lane 2v+h publishes half h of value v (lanes 0..5 hold the values)
                const double v = __shfl(total, tid >> 1, 64);
                const unsigned half = (tid & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
                __hip_atomic_store(lines + part * kXchgWords + tid, tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // lane l polls word l%16 of parts l/16 and l/16+4 (words 12..15 of a line are padding)
            const int k = tid & 15, p0 = tid >> 4, p1 = p0 + 4;
            const bool want0 = k < 4 * kTaps && p0 < parts, want1 = k < 4 * kTaps && p1 < parts;
            const unsigned long long* a0 = lines + (want0 ? p0 : part) * kXchgWords + (want0 ? k : 0);
            const unsigned long long* a1 = lines + (want1 ? p1 : part) * kXchgWords + (want1 ? k : 0);
            unsigned long long w0 = 0, w1 = 0;
            bool done = false;
            for (long spins = 0; spins < kSpinLimit; ++spins) {
                w0 = __hip_atomic_load(a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1 = __hip_atomic_load(a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (!want0 || (w0 >> 32 << 32) == tag) && (!want1 || (w1 >> 32 << 32) == tag);
                if (__all(ok)) {
                    done = true;
                    break;
                }
            }
            if (!done) {
                if (tid == 0) {
                    sh->fault = 1;
                    *fault = 1;
                }
            } else {
                unsigned* halves = reinterpret_cast<unsigned*>(red);  // (the wave sums in `red` have been consumed)
                halves[tid] = (unsigned)w0;            // [p*16 + k], p < 4
                halves[64 + tid] = (unsigned)w1;       // p >= 4
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS serves a wave's operations in order)
                __builtin_amdgcn_wave_barrier();
                if (tid < 2 * kTaps) {
                    double sum = 0.0;
                    for (int p = 0; p < parts; ++p)
                        sum += __hiloint2double((int)halves[p * 16 + 2 * tid + 1], (int)halves[p * 16 + 2 * tid]);
                    total = sum;
                }
            }
        }
        TRACK_MARK(6);

        // Loop update.  The reference's per-epoch sequence is scalar arithmetic whose cost is instruction LATENCY
        // (~15 fp64 divisions, two square roots, two arctangents, a float modulo, ~1250 instructions when one lane
        // does it all).  It splits into three chains that only meet through the previous epoch's results:
        //   wave 0  carrier loop : FLL/PLL discriminators, carrier filter, carrier NCO        (kaplan:405-447,506-534)
        //   wave 1  code loop    : DLL discriminator, code filter, code NCO, next epoch length (kaplan:451-461,506-534)
        //   wave 2  lock role    : lock indicators, C/N0, lock-state machine, flags, bit sync, nav bits (kaplan:465-619)
        // Each wave evaluates its divisions / roots / arctangents side by side, one per lane (same IEEE operations on
        // the same operands as the reference's statements: bit-identical), then lane 0 runs the rest of its chain and
        // publishes its share of the next epoch's parameters.
        if (tid < 2 * kTaps) sh->corr[tid] = total;
        __syncthreads();
        TRACK_MARK(7);
        if (!sh->fault && role < 3) {
            const double ie = sh->corr[0], qe = sh->corr[1], ip = sh->corr[2], qp = sh->corr[3], il = sh->corr[4], ql = sh->corr[5];
            const int n = ep.n;
            const bool kaplan = cfg.loop_kind != 0;
            sdr_track_epoch* rec = (writer && keep_traj) ? traj + ((size_t)ch * n_epochs + epoch) : nullptr;
            if (role == 0) {
                // ------------------------------------------------------------------ carrier loop
                double num = 0.0, den = 1.0;
                switch (rlane) {
                    case 0: num = qp, den = ip; break;                    // atan(qP/iP): Costas PLL, FLL (tracking.py:133-176)
                    case 1: num = st.q_prompt_prev, den = st.i_prompt_prev; break;  // atan(qP'/iP')
                    case 2: num = cur_fll_bw, den = kW0Bw1; break;
                    case 3: num = cur_pll_bw, den = kW0Bw2; break;
                    case 4:                                               // carrier NCO advance over the epoch
                        num = kaplan ? ep.carrier_hz * kGpsTwoPi * (double)n : ep.carrier_hz * 2.0 * M_PI * (double)n;
                        den = fs;
                        break;
                    case 5: num = cfg.pll_tau2, den = cfg.pll_tau1; break;  // (Borre PLL filter, tracking.py:180-186)
                    case 6: num = cfg.pll_pdi, den = cfg.pll_tau1; break;
                    default: break;
                }
                const double quot = num / den;
                const double at = atan(quot);
                const double at_now = lane_value(at, 0), at_prev = lane_value(at, 1);
                double fll_err = at_now - at_prev;
                if (fll_err != fll_err) fll_err = 0.0;
                if (fll_err >= kGpsHalfPi) fll_err = fll_err - kGpsPi;
                else if (fll_err <= -kGpsHalfPi) fll_err = fll_err + kGpsPi;
                // lane 0: atan/2pi (pll_costas), lane 1: err/dt, then lane 1: (err/dt)/2pi (fll_atan)
                const double q4 = (rlane == 1 ? fll_err : at_now) / (rlane == 1 ? 1e-3 : kGpsTwoPi);
                const double q5 = q4 / kGpsTwoPi;
                const double costas = lane_value(q4, 0), fll_full = lane_value(q5, 1);
                const double w0f = lane_value(quot, 2), w0p = lane_value(quot, 3), carrier_adv = lane_value(quot, 4);
                const double pll_r1 = lane_value(quot, 5), pll_r2 = lane_value(quot, 6);
                __builtin_amdgcn_wave_barrier();  // (every lane has read the previous prompt before lane 0 replaces it)
                if (rlane == 0) {
                    double c_pll_mem = st.pll_mem;
                    const int c_code_counter = sh->c_code_counter;
                    double rem_carrier = ep.rem_carrier, carrier_hz = ep.carrier_hz;
                    double rec_pll, rec_fll, rec_carrier_err;
                    if (!kaplan) {  // Borre: channel_l1ca_borre.py:364-429
                        rem_carrier -= carrier_adv;
                        rem_carrier = py_mod(rem_carrier, 2.0 * M_PI);
                        const double phase_err = costas;
                        double nco_carrier = pll_r1 * (phase_err - c_pll_mem);
                        nco_carrier += pll_r2 * phase_err;
                        c_pll_mem = phase_err;
                        carrier_hz += nco_carrier;
                        rec_pll = nco_carrier, rec_fll = 0.0, rec_carrier_err = phase_err;
                    } else {        // Kaplan: runDiscriminators / runCarrierFrequencyFilter / postTrackingUpdate
                        double fll_d = 0.0, pll_d = 0.0;
                        if (cur_lock_state == LOCK_PULL_IN) {
                            if (c_code_counter > 1) fll_d = fll_full;
                        } else {
                            fll_d = fll_full;
                            pll_d = costas;
                        }
                        const double upd = (pll_d * (w0p * w0p) + fll_d * w0f) * 1e-3;  // FLLassistedPLL_2ndOrder (tracking.py:246-279)
                        double carrier_err = upd + c_pll_mem;
                        c_pll_mem = upd;
                        carrier_err += pll_d * kW0A2 * w0p;
                        rem_carrier -= carrier_adv;
                        rem_carrier = py_mod(rem_carrier, kGpsTwoPi);
                        carrier_hz += carrier_err;
                        rec_pll = pll_d, rec_fll = fll_d, rec_carrier_err = carrier_err;
                    }
                    st.pll_mem = c_pll_mem;
                    st.i_prompt_prev = ip;
                    st.q_prompt_prev = qp;
                    sh->c_code_counter = c_code_counter + 1;
                    sh->ep.carrier_hz = carrier_hz;
                    sh->stop_carrier = carrier_bad(carrier_hz) ? 1 : 0;
                    sh->ep.rem_carrier = rem_carrier;
                    sh->dphi = carrier_step(carrier_hz, fs);
                    if (rec) {
                        rec->carrier_hz_in = ep.carrier_hz;
                        rec->rem_carrier_in = ep.rem_carrier;
                        rec->pll = rec_pll;
                        rec->fll = rec_fll;
                        rec->carrier_err = rec_carrier_err;
                        rec->carrier_hz = carrier_hz;
                    }
                }
            } else if (role == 1) {
                // ------------------------------------------------------------------ code loop
                const double env = sqrt(rlane == 1 ? il * il + ql * ql : ie * ie + qe * qe);  // DLL NNEML envelopes (tracking.py:120-129)
                const double env_e = lane_value(env, 0), env_l = lane_value(env, 1);
                double num = 0.0, den = 1.0;
                switch (rlane) {
                    case 0: num = env_e - env_l, den = env_e + env_l; break;
                    case 1: num = cfg.dll_tau2, den = cfg.dll_tau1; break;              // BorreLoopFilter (tracking.py:180-186)
                    case 2: num = kaplan ? cfg.dll_pdi * 1.0 : cfg.dll_pdi, den = cfg.dll_tau1; break;
                    default: break;
                }
                const double quot = num / den;
                const double dll_nn = lane_value(quot, 0), dll_r1 = lane_value(quot, 1), dll_r2 = lane_value(quot, 2);
                if (rlane == 0) {
                    const double dll_d = dll_nn;
                    double code_err = dll_r1 * (dll_d - st.dll_mem);
                    code_err += dll_r2 * dll_d;
                    st.dll_mem = dll_d;
                    st.code_counter += 1;
                    const double k_code_hz = st.code_hz - code_err;
                    st.code_hz = k_code_hz;
                    double rem_code = ep.rem_code;
                    rem_code += (double)n * ep.code_step - kChips;
                    const double code_step = k_code_hz / fs;
                    const int64_t next_start = ep.start_sample + n;
                    const int next_n = (int)ceil((kChips - rem_code) / code_step);
                    sh->stop_code = code_out_of_range(next_start, next_n, rem_code, code_step) ? 1 : 0;
                    sh->ep.start_sample = next_start;
                    sh->ep.n = next_n;
                    sh->ep.rem_code = rem_code;
                    sh->ep.code_step = code_step;
                    sh->epochs_done = epoch + 1;
                    if (rec) {
                        rec->start_sample = ep.start_sample;
                        rec->n_samples = n;
                        rec->rem_code_in = ep.rem_code;
                        rec->code_step_in = ep.code_step;
                        for (int k = 0; k < 2 * SDR_MAX_TAPS; ++k) rec->corr[k] = k < 2 * kTaps ? sh->corr[k] : 0.0;
                        // Kaplan records the discriminator and the filter output; Borre the NCO command and the error
                        rec->dll = kaplan ? dll_d : code_err;
                        rec->code_err = kaplan ? code_err : dll_d;
                        rec->code_hz = k_code_hz;
                    }
                }
            } else {
                // ------------------------------------------------------------------ lock indicators, state machine, bits
                const double pw = ip * ip + qp * qp;
                double num = 0.0, den = 1.0;
                switch (rlane) {
                    case 0: {                                             // FLL lock (lockindicator.py:6-18)
                        const double l_ipp = sh->l_ipp, l_qpp = sh->l_qpp;
                        double v = ip * l_ipp - qp * l_qpp;
                        v *= np_sign(ip * l_ipp + qp * l_qpp);
                        num = v, den = pw;
                        break;
                    }
                    case 1: num = ip * ip - qp * qp, den = pw; break;     // PLL lock (:22-36)
                    case 2: {                                             // C/N0 (Beaulieu) ratio term (kaplan:488)
                        const double d = fabs(ip) - fabs(qp);
                        num = pw, den = d * d;
                        break;
                    }
                    default: break;
                }
                const double quot = num / den;
                const double fll_lock_v = lane_value(quot, 0), pll_lock_v = lane_value(quot, 1), cn0_term = lane_value(quot, 2);
                if (rlane == 0) {
                    double l_fll_lock = st.fll_lock, l_pll_lock = st.pll_lock, l_cn0 = st.cn0, l_ratio_acc = st.cn0_ratio_acc;
                    double l_ipp = sh->l_ipp, l_qpp = sh->l_qpp, l_fll_bw = st.fll_bw, l_pll_bw = st.pll_bw, l_nav_sum = st.nav_prompt_sum;
                    int l_accum = st.accum_counter, l_lock_state = st.lock_state, l_time_in_state = st.time_in_state;
                    int l_spacing_sel = st.spacing_sel, l_flags = st.track_flags, l_code_counter = sh->l_code_counter;
                    int l_nav_count = st.nav_sum_counter, l_bits_emitted = st.nav_bits_emitted, l_bits_run = sh->l_bits_run;
                    int nav_bit = -1;
                    if (!kaplan) {
                        // Borre bit sync: first prompt sign flip after MIN_CONVERGENCE_TIME = 100 epochs (borre:384-391)
                        if (!(l_flags & FLAG_BIT_SYNC) && (l_flags & FLAG_CODE_LOCK) && l_code_counter > 100 &&
                            np_sign(l_ipp) != np_sign(ip))
                            l_flags |= FLAG_BIT_SYNC;
                        l_flags |= FLAG_CODE_LOCK;
                        l_ipp = ip;
                        l_qpp = qp;
                        l_code_counter += 1;
                    } else {
                        // runCorrelators bookkeeping (kaplan:392-399)
                        if (l_accum == kMsPerBit) l_accum = 0;
                        l_accum += 1;
                        // runLoopIndicators (:465-502)
                        if (l_code_counter != 0) {
                            const double v = fabs(fll_lock_v);
                            l_fll_lock = (1.0 - 0.005) * l_fll_lock + 0.005 * v;
                            if (l_lock_state > LOCK_PULL_IN) l_pll_lock = (1.0 - 0.005) * l_pll_lock + 0.005 * pll_lock_v;
                            l_ratio_acc += cn0_term;
                            if (l_accum == kMsPerBit) {
                                const double lam = 1.0 / (l_ratio_acc / (double)l_accum);
                                const double c = lam * (1.0 / ((double)l_accum * 1e-3));
                                l_cn0 = (1.0 - 0.1) * l_cn0 + 0.1 * c;
                                l_ratio_acc = 0.0;
                            }
                        }
                        l_code_counter += 1;
                        // trackingStateUpdate (:538-619)
                        if (l_lock_state != LOCK_PULL_IN && l_cn0 > cfg.dll_threshold && !(l_flags & FLAG_CODE_LOCK))
                            l_flags |= FLAG_CODE_LOCK;
                        else if (l_cn0 < cfg.dll_threshold && (l_flags & FLAG_CODE_LOCK))
                            l_flags ^= FLAG_CODE_LOCK;
                        if ((l_flags & FLAG_CODE_LOCK) && !(l_flags & FLAG_BIT_SYNC)) {
                            if (np_sign(l_ipp) != np_sign(ip)) {
                                l_flags |= FLAG_BIT_SYNC;
                                l_accum = 1;
                                l_ratio_acc = 0.0;
                            }
                        }
                        l_ipp = ip;
                        l_qpp = qp;
                        if (l_lock_state != LOCK_NARROW && l_fll_lock >= cfg.fll_thr_narrow && l_pll_lock >= cfg.pll_thr_narrow) {
                            l_lock_state = LOCK_NARROW;
                            l_fll_bw = cfg.fll_bw_narrow;
                            l_pll_bw = cfg.pll_bw_narrow;
                            l_spacing_sel = 1;
                            l_time_in_state = 0;
                        } else if (l_lock_state != LOCK_WIDE && l_fll_lock >= cfg.fll_thr_wide && l_fll_lock < cfg.fll_thr_narrow) {
                            l_lock_state = LOCK_WIDE;
                            l_fll_bw = cfg.fll_bw_wide;
                            l_pll_bw = cfg.pll_bw_wide;
                            l_spacing_sel = 0;
                            l_time_in_state = 0;
                        } else if (l_lock_state != LOCK_PULL_IN && l_fll_lock <= cfg.fll_thr_wide) {
                            l_lock_state = LOCK_PULL_IN;
                            l_fll_bw = cfg.fll_bw_pullin;
                            l_pll_bw = 0.0;
                            l_spacing_sel = 0;
                            l_time_in_state = 0;
                        } else {
                            l_time_in_state += 1;
                        }
                    }
                    // decodeBit (kaplan:728-754, borre:470-491): 20 prompts after bit sync -> one bit (Prompt2Bit)
                    if (!(l_flags & FLAG_BIT_SYNC)) {
                        l_nav_sum = 0.0;
                        l_nav_count = 0;
                    } else {
                        l_nav_sum += ip;
                        l_nav_count += 1;
                        if (l_nav_count == kMsPerBit) {
                            nav_bit = l_nav_sum > 0.0 ? 1 : 0;
                            if (writer && nav_bits && l_bits_run < max_bits) nav_bits[(size_t)ch * max_bits + l_bits_run] = (int8_t)nav_bit;
                            l_bits_run += 1;
                            l_bits_emitted += 1;
                            l_nav_sum = 0.0;
                            l_nav_count = 0;
                        }
                    }
                    st.fll_lock = l_fll_lock, st.pll_lock = l_pll_lock, st.cn0 = l_cn0, st.cn0_ratio_acc = l_ratio_acc;
                    sh->l_ipp = l_ipp, sh->l_qpp = l_qpp, st.fll_bw = l_fll_bw, st.pll_bw = l_pll_bw, st.nav_prompt_sum = l_nav_sum;
                    st.accum_counter = l_accum, st.lock_state = l_lock_state, st.time_in_state = l_time_in_state;
                    st.spacing_sel = l_spacing_sel, st.track_flags = l_flags, sh->l_code_counter = l_code_counter;
                    st.nav_sum_counter = l_nav_count, st.nav_bits_emitted = l_bits_emitted, sh->l_bits_run = l_bits_run;
                    // hand-over to the other roles / the next epoch
                    const double* sp = l_spacing_sel ? cfg.spacing_narrow : cfg.spacing_wide;
                    for (int t = 0; t < kTaps; ++t) sh->spacing[t] = sp[t];
                    sh->fll_bw = l_fll_bw;
                    sh->pll_bw = l_pll_bw;
                    sh->lock_state = l_lock_state;
                    if (rec) {
                        rec->cn0 = kaplan ? l_cn0 : 0.0;
                        rec->pll_lock = kaplan ? l_pll_lock : 0.0;
                        rec->fll_lock = kaplan ? l_fll_lock : 0.0;
                        rec->lock_state = l_lock_state;
                        rec->track_flags = l_flags;
                        rec->nav_bit = nav_bit;
                    }
                }
            }
        }
        TRACK_MARK(4);
        // the next iteration's first barrier orders lane 0's LDS writes against everyone's reads
    }
    // End state: the roles kept the LDS copy of the state current; one lane of the recording part writes it out.
    __syncthreads();
    if (tid == 0 && writer) {
        const int epochs_done = sh->epochs_done;
        // the NCO values of the next epoch are the ones the roles published last
        st.current_sample = sh->ep.start_sample;
        st.n_samples = sh->ep.n;
        st.carrier_hz = sh->ep.carrier_hz;
        st.rem_carrier = sh->ep.rem_carrier;
        st.rem_code = sh->ep.rem_code;
        st.code_step = sh->ep.code_step;
        if (epochs_done < n_epochs) {
            st.n_samples = -1 - epochs_done;  // stopped early: -(1 + epochs completed)
            if (keep_traj)
                for (int k = epochs_done; k < n_epochs; ++k) traj[(size_t)ch * n_epochs + k].n_samples = 0;
        }
        states[ch] = st;
        if (n_bits) n_bits[ch] = sh->l_bits_run < max_bits ? sh->l_bits_run : max_bits;
    }
}

}  // namespace

#ifdef SDR_TRACK_DENSE_TU
// This translation unit (track_dense.hip) carries only the variant of the kernel for more channels than CUs:
// 256-thread workgroups capped at 168 registers so that THREE share a CU.  It is compiled with
// -mllvm -disable-machine-licm: hoisting the fp64 polynomial constants of sincos / atan / division out of the
// epoch loop parks ~80 of them in VGPRs for the whole kernel (256 instead of 173 registers).
hipError_t sdr_track_dense_launch(int fmt, int n_ch, size_t shmem, hipStream_t stream, void** args) {
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        return hipLaunchKernel((const void*)kernel, dim3(n_ch), dim3(256), args, shmem, stream);
    };
    switch (fmt) {
        case SDR_FMT_CI8: return launch(track_kernel<SDR_FMT_CI8, 256, 3>);
        case SDR_FMT_CI16: return launch(track_kernel<SDR_FMT_CI16, 256, 3>);
        case SDR_FMT_CF32: return launch(track_kernel<SDR_FMT_CF32, 256, 3>);
        default: return launch(track_kernel<SDR_FMT_CF64, 256, 3>);
    }
}
#else
hipError_t sdr_track_dense_launch(int fmt, int n_ch, size_t shmem, hipStream_t stream, void** args);  // track_dense.hip

extern "C" {

int sdr_track_cluster(sdr_engine* e, int parts) {
    if (!e) return sdr_fail(SDR_ERR_INVALID, "null engine");
    if (parts != 0 && parts != 1 && parts != 2 && parts != 4 && parts != 8)
        return sdr_fail(SDR_ERR_INVALID, "cluster size %d: must be 0 (automatic), 1, 2, 4 or 8", parts);
    e->track_force_parts = parts;
    return SDR_OK;
}

int sdr_track_closed_loop(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfg, int n_epochs,
                          sdr_track_epoch* traj) {
    return sdr_track_closed_loop_bits(e, n_ch, st, cfg, n_epochs, traj, nullptr, 0, nullptr);
}

int sdr_track_closed_loop_bits(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfg, int n_epochs,
                               sdr_track_epoch* traj, int8_t* nav_bits, int max_bits, int32_t* n_bits) {
    if (int rc = sdr_set_device(e)) return rc;
    if ((nav_bits && (max_bits < 1 || !n_bits)) || (!nav_bits && n_bits))
        return sdr_fail(SDR_ERR_INVALID, "nav_bits, max_bits and n_bits go together");
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if (!st || !cfg || n_ch < 1 || n_epochs < 1) return sdr_fail(SDR_ERR_INVALID, "bad closed-loop request");
    if (cfg->n_taps != kTaps) return sdr_fail(SDR_ERR_UNSUPPORTED, "closed-loop tracking uses 3 taps (E/P/L), got %d", cfg->n_taps);
    if (cfg->loop_kind != 0 && cfg->loop_kind != 1) return sdr_fail(SDR_ERR_INVALID, "loop_kind %d is neither 0 (Borre) nor 1 (Kaplan)", cfg->loop_kind);
    if (!(cfg->fs > 0.0)) return sdr_fail(SDR_ERR_INVALID, "fs must be positive");
    int maxlen = 0;
    for (int c = 0; c < n_ch; ++c) {
        const int slot = st[c].code_slot;
        if (slot < 0 || slot >= e->n_slots || e->code_len_host[slot] <= 0)
            return sdr_fail(SDR_ERR_INVALID, "channel %d: code slot %d is not staged", c, slot);
        if (st[c].n_samples <= 0) return sdr_fail(SDR_ERR_INVALID, "channel %d: n_samples must be positive", c);
        if (e->code_len_host[slot] > maxlen) maxlen = e->code_len_host[slot];
    }
    const size_t traj_bytes = traj ? (size_t)n_ch * n_epochs * sizeof(sdr_track_epoch) : 0;
    int rc = sdr_devbuf_reserve(e, &e->track_state, (size_t)n_ch * sizeof(sdr_track_state));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->track_cfg, sizeof(sdr_loop_cfg));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->track_traj, traj ? traj_bytes : sizeof(sdr_track_epoch));
    const size_t bits_bytes = nav_bits ? (size_t)n_ch * max_bits : 0;
    if (!rc && nav_bits) rc = sdr_devbuf_reserve(e, &e->track_bits, bits_bytes + (size_t)n_ch * sizeof(int32_t) + 16);
    if (rc) return rc;
    int32_t* d_nbits = nav_bits ? (int32_t*)e->track_bits.ptr : nullptr;
    int8_t* d_bits = nav_bits ? (int8_t*)e->track_bits.ptr + (((size_t)n_ch * sizeof(int32_t) + 15) & ~(size_t)15) : nullptr;
    if (nav_bits) SDR_HIP(hipMemsetAsync(e->track_bits.ptr, 0, e->track_bits.bytes, e->stream));
    SDR_HIP(hipMemcpyAsync(e->track_state.ptr, st, (size_t)n_ch * sizeof(sdr_track_state), hipMemcpyHostToDevice, e->stream));
    SDR_HIP(hipMemcpyAsync(e->track_cfg.ptr, cfg, sizeof(sdr_loop_cfg), hipMemcpyHostToDevice, e->stream));
    (void)maxlen;
    const int lut_words = e->lut_stride;  // the whole staged row (every code period the slots were sized for)
    sdr_track_state* d_st = (sdr_track_state*)e->track_state.ptr;
    const sdr_loop_cfg* d_cfg = (const sdr_loop_cfg*)e->track_cfg.ptr;
    sdr_track_epoch* d_traj = (sdr_track_epoch*)e->track_traj.ptr;
    int keep = traj ? 1 : 0;
    // exchange lines [n_ch][2 parities][8 parts][16 words] (tags zeroed: epoch tags start at 1), then the fault word
    const size_t xchg_bytes = (size_t)n_ch * 2 * kMaxParts * kXchgWords * sizeof(unsigned long long);
    if (int rc2 = sdr_devbuf_reserve(e, &e->track_xchg, xchg_bytes + 16)) return rc2;
    SDR_HIP(hipMemsetAsync(e->track_xchg.ptr, 0, xchg_bytes + 16, e->stream));
    unsigned long long* d_xchg = (unsigned long long*)e->track_xchg.ptr;
    int* d_fault = (int*)((char*)e->track_xchg.ptr + xchg_bytes);
    const void* d_iq = e->iq;
    int64_t cap = e->iq_capacity;
    const uint32_t* d_luts = e->luts;
    int lw = lut_words, ls = e->lut_stride, nch = n_ch, n_ep = n_epochs, mb = max_bits;

    // One attempt with `parts` workgroups per channel.
    auto attempt = [&](int parts, bool* too_big) -> hipError_t {
        // more channels than CUs: smaller workgroups, two or three of which share a CU, so that one channel's
        // loop update overlaps another's correlation (measured: 512 channels 20.2 -> 14.6 us per epoch,
        // 768 channels 22.3 -> 18.6)
        const bool dense = parts == 1 && n_ch > e->n_cus;
        const int threads = (parts >= 2 || dense) ? 256 : 512;
        const size_t shmem_base = (size_t)red_doubles(threads) * sizeof(double) + sizeof(EpochShared) +
                                  (size_t)((lut_words + 3) & ~3) * sizeof(uint32_t);
        const size_t prefix_bytes = (size_t)threads * kPrefixSlots * sizeof(double2);
        // The boundary variant of the correlator needs a 144-byte LDS strip per lane; long multi-period
        // replicas that leave no room for it are tracked with the per-sample variant.
        int up = shmem_base + prefix_bytes <= 160u * 1024u && e->lut_stride < kFastMaxLutWords ? 1 : 0;
        const size_t shmem = shmem_base + (up ? prefix_bytes : 0);
        *too_big = shmem > 160u * 1024u;
        if (*too_big) return hipSuccess;
        hipError_t err = hipSuccess;
        auto launch = [&](auto kernel) {
            // more than 64 KB of dynamic LDS has to be granted per kernel
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            void* args[] = {&d_iq, &cap, &d_st, &d_cfg, &n_ep, &d_traj, &keep, &d_bits, &mb, &d_nbits, &d_luts,
                            &lw, &ls, &up, &nch, &parts, &d_xchg, &d_fault};
            if (parts > 1)  // the parts of a cluster wait for each other: all workgroups must be resident
                err = hipLaunchCooperativeKernel((const void*)kernel, dim3(n_ch * parts), dim3(threads), args,
                                                 (unsigned)shmem, e->stream);
            else
                err = hipLaunchKernel((const void*)kernel, dim3(n_ch), dim3(threads), args, shmem, e->stream);
        };
        if (dense) {
            void* args[] = {&d_iq, &cap, &d_st, &d_cfg, &n_ep, &d_traj, &keep, &d_bits, &mb, &d_nbits, &d_luts,
                            &lw, &ls, &up, &nch, &parts, &d_xchg, &d_fault};
            return sdr_track_dense_launch(e->iq_fmt, n_ch, shmem, e->stream, args);
        }
        auto by_threads = [&](auto fmt) {
            constexpr int F = decltype(fmt)::value;
            if (threads == 256) launch(track_kernel<F, 256, 1>);
            else launch(track_kernel<F, 512, 2>);
        };
        switch (e->iq_fmt) {
            case SDR_FMT_CI8: by_threads(std::integral_constant<int, SDR_FMT_CI8>{}); break;
            case SDR_FMT_CI16: by_threads(std::integral_constant<int, SDR_FMT_CI16>{}); break;
            case SDR_FMT_CF32: by_threads(std::integral_constant<int, SDR_FMT_CF32>{}); break;
            default: by_threads(std::integral_constant<int, SDR_FMT_CF64>{}); break;
        }
        return err;
    };

    // Cluster size: as many workgroups per channel as the GPU has room for (1 per CU), up to 8.  When the
    // cooperative launch is refused (GPU shared or partitioned: not every workgroup could be resident) the
    // automatic choice halves the cluster until the launch goes through; a forced size fails instead.
    int parts = 1;
    if (e->track_force_parts) {
        parts = e->track_force_parts;
    } else {
        while (parts < kMaxParts && (long)n_ch * parts * 2 <= (long)e->n_cus) parts *= 2;
    }
    hipError_t launch_err = hipSuccess;
    {
        ProfScope ps(e, "track_kernel");
        for (;;) {
            bool too_big = false;
            launch_err = attempt(parts, &too_big);
            if (too_big) return sdr_fail(SDR_ERR_RANGE, "closed-loop tracking: code table does not fit the LDS");
            if (launch_err == hipSuccess || e->track_force_parts || parts == 1) break;
            (void)hipGetLastError();
            parts /= 2;
        }
    }
    if (launch_err != hipSuccess)
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking launch (%d channels x %d parts) failed: %s", n_ch, parts,
                        hipGetErrorString(launch_err));
    int fault_host = 0;
    SDR_HIP(hipMemcpyAsync(&fault_host, d_fault, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipMemcpyAsync(st, e->track_state.ptr, (size_t)n_ch * sizeof(sdr_track_state), hipMemcpyDeviceToHost, e->stream));
    if (traj) SDR_HIP(hipMemcpyAsync(traj, e->track_traj.ptr, traj_bytes, hipMemcpyDeviceToHost, e->stream));
    if (nav_bits) {
        SDR_HIP(hipMemcpyAsync(nav_bits, d_bits, bits_bytes, hipMemcpyDeviceToHost, e->stream));
        SDR_HIP(hipMemcpyAsync(n_bits, d_nbits, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    }
    SDR_HIP(hipStreamSynchronize(e->stream));
    if (fault_host)
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking: a workgroup of a %d-part cluster never published its sums", parts);
    for (int c = 0; c < n_ch; ++c)
        if (st[c].n_samples < 0)
            return sdr_fail(SDR_ERR_RANGE, "channel %d stopped after %d epochs: NCO state left the staged replica / ring",
                            c, -1 - st[c].n_samples);
    return SDR_OK;
}

}  // extern "C"
#endif  // SDR_TRACK_DENSE_TU
