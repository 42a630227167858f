// On-device loop closure: one persistent 512-lane workgroup per channel runs
// correlate -> discriminators -> loop filters -> NCO update for n_epochs without leaving the
// GPU.  The PRN replica stays in LDS for the whole run; the scalar loop
// arithmetic runs on lane 0 in fp64, following the two reference plugins statement by statement:
//   kind 0  Borre  : channel_l1ca_borre.py:333-451  (DLL NNEML + Costas PLL, Borre filters, np.pi NCO)
//   kind 1  Kaplan : channel_l1ca_kaplan.py:342-619 (FLL-assisted 2nd-order PLL, lock-state machine,
//                    GPS-ICD pi in the NCO and the discriminators: SURVEY.md T3)
// built on sydr/dsp/tracking.py:120-186,246-279 and sydr/dsp/lockindicator.py:6-122.
#include "correlator.h"

namespace {

using namespace sdr;

constexpr int kTrackThreads = 512;
constexpr int kTrackWaves = kTrackThreads / 64;
constexpr int kTaps = 3;

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

__device__ __forceinline__ double dll_nneml(double ie, double qe, double il, double ql) {  // tracking.py:120-129
    const double e = sqrt(ie * ie + qe * qe), l = sqrt(il * il + ql * ql);
    return (e - l) / (e + l);
}
__device__ __forceinline__ double pll_costas(double ip, double qp) {  // tracking.py:133-142
    return atan(qp / ip) / kGpsTwoPi;
}
__device__ __forceinline__ double fll_atan(double ip, double qp, double ipp, double qpp, double dt) {  // :156-176
    double err = atan(qp / ip) - atan(qpp / ipp);
    if (err != err) err = 0.0;
    if (err >= kGpsHalfPi) err = err - kGpsPi;
    else if (err <= -kGpsHalfPi) err = err + kGpsPi;
    err = err / dt;
    return err / kGpsTwoPi;
}
__device__ __forceinline__ double borre_filter(double x, double mem, double tau1, double tau2, double pdi) {  // :180-186
    double out = tau2 / tau1 * (x - mem);
    out += pdi / tau1 * x;
    return out;
}

struct alignas(16) EpochShared {  // (size a multiple of 16: the replica behind it is copied with 16-byte stores)
    EpochParams ep;
    double spacing[kTaps];
    double dphi;
    int stop;
    int epochs_done;
    int bits_this_run;
    int pad_;
    sdr_track_state st;  // lane 0's working copy lives in LDS, not in 1024 x VGPRs
    sdr_loop_cfg cfg;
};

template <int FMT>
__global__ __launch_bounds__(kTrackThreads) void track_kernel(const void* __restrict__ ring, int64_t capacity,
                                                              sdr_track_state* __restrict__ states,
                                                              const sdr_loop_cfg* __restrict__ cfg_ptr,
                                                              int n_epochs, sdr_track_epoch* __restrict__ traj,
                                                              int keep_traj, int8_t* __restrict__ nav_bits, int max_bits,
                                                              int32_t* __restrict__ n_bits,
                                                              const uint32_t* __restrict__ luts,
                                                              int lut_words, int lut_stride, int use_prefix) {
    extern __shared__ double smem[];
    double* red = smem;                                   // kTrackWaves * 6
    EpochShared* sh = reinterpret_cast<EpochShared*>(red + kTrackWaves * 2 * kTaps);
    double2* prefix = reinterpret_cast<double2*>(sh + 1);  // kTrackThreads * kPrefixSlots, when the launcher found room
    uint32_t* lut = reinterpret_cast<uint32_t*>(prefix + (use_prefix ? kTrackThreads * kPrefixSlots : 0));

    const int tid = threadIdx.x;
    const int ch = blockIdx.x;
    if (tid == 0) {
        sh->st = states[ch];
        sh->cfg = *cfg_ptr;
        sh->epochs_done = 0;
        sh->bits_this_run = 0;
    }
    const int slot = states[ch].code_slot;
    stage_lut<kTrackThreads>(lut, luts + (size_t)slot * lut_stride, lut_words, tid);
    const double fs = cfg_ptr->fs;
    sdr_track_state& st = sh->st;
    const sdr_loop_cfg& cfg = sh->cfg;

    for (int epoch = 0; epoch < n_epochs; ++epoch) {
        if (tid == 0) {
            const double* sp = st.spacing_sel ? cfg.spacing_narrow : cfg.spacing_wide;
            double smin = sp[0], smax = sp[0];
            for (int t = 1; t < kTaps; ++t) {
                smin = fmin(smin, sp[t]);
                smax = fmax(smax, sp[t]);
            }
            // The replica LUT and the ring bound what an epoch may touch; a loop that has run
            // away (loss of lock) stops here instead of reading out of range.
            const double lo = ceil(st.rem_code + smin);
            const double hi = ceil(st.code_step * (double)st.n_samples + st.rem_code + smax);
            const bool ok = st.n_samples > 0 && (int64_t)st.n_samples <= capacity && st.code_step > 0.0 &&
                            lo >= -(double)SDR_LUT_PAD && hi <= (double)(lut_words - SDR_LUT_PAD - 2) &&
                            st.carrier_hz == st.carrier_hz && fabs(st.carrier_hz) < 1e9 && st.current_sample >= 0;
            sh->stop = ok ? 0 : 1;
            sh->ep.start_sample = st.current_sample;
            sh->ep.n = st.n_samples;
            sh->ep.carrier_hz = st.carrier_hz;
            sh->ep.rem_carrier = st.rem_carrier;
            sh->ep.rem_code = st.rem_code;
            sh->ep.code_step = st.code_step;
            for (int t = 0; t < kTaps; ++t) sh->spacing[t] = sp[t];
            sh->dphi = carrier_step(st.carrier_hz, fs);
        }
        __syncthreads();
        if (sh->stop) break;
        const EpochParams ep = sh->ep;
        const double dphi = sh->dphi;
        EpochConsts<kTaps> K;
        compute_constants<kTaps, kTrackThreads>(K, ep, sh->spacing, dphi);

        double accr[kTaps], acci[kTaps];
        if (use_prefix && ep.code_step <= kFastMaxCodeStep && ep.code_step >= kFastMinCodeStep && !epoch_wraps(ep, capacity))   // uniform branch: 16-sample boundary variant above ~17 MHz
            correlate_epoch_wide<FMT, kTaps, kTrackThreads>(ring, capacity, ep, dphi, K, lut, prefix, tid, accr, acci);
        else
            correlate_epoch<FMT, kTaps, kTrackThreads>(ring, capacity, ep, dphi, K, lut, tid, accr, acci);
        const double total = reduce_taps<kTaps, kTrackThreads>(accr, acci, red, tid);

        // lanes 0..5 of wave 0 hold [IE,QE,IP,QP,IL,QL]; hand them to lane 0 without a barrier
        double corr[2 * kTaps];
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < 2 * kTaps; ++k) corr[k] = __shfl(total, k, 64);
        }

        if (tid == 0) {
            const double ie = corr[0], qe = corr[1], ip = corr[2], qp = corr[3], il = corr[4], ql = corr[5];
            const int n = st.n_samples;
            // keep_traj = 0: one record per channel, overwritten every epoch (nobody reads it)
            sdr_track_epoch& rec = traj[keep_traj ? (size_t)ch * n_epochs + epoch : (size_t)ch];
            rec.start_sample = st.current_sample;
            rec.n_samples = n;
            rec.carrier_hz_in = st.carrier_hz;
            rec.rem_carrier_in = st.rem_carrier;
            rec.rem_code_in = st.rem_code;
            rec.code_step_in = st.code_step;
            for (int k = 0; k < 2 * SDR_MAX_TAPS; ++k) rec.corr[k] = k < 2 * kTaps ? corr[k] : 0.0;
            rec.nav_bit = -1;

            if (cfg.loop_kind == 0) {
                // ---- Borre: channel_l1ca_borre.py:364-429
                st.rem_carrier -= st.carrier_hz * 2.0 * M_PI * (double)n / fs;
                st.rem_carrier = py_mod(st.rem_carrier, 2.0 * M_PI);
                const double code_err = dll_nneml(ie, qe, il, ql);
                const double nco_code = borre_filter(code_err, st.dll_mem, cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi);
                st.dll_mem = code_err;
                const double phase_err = pll_costas(ip, qp);
                const double nco_carrier = borre_filter(phase_err, st.pll_mem, cfg.pll_tau1, cfg.pll_tau2, cfg.pll_pdi);
                st.pll_mem = phase_err;
                // bit sync: first prompt sign flip after MIN_CONVERGENCE_TIME = 100 epochs (borre:384-391)
                if (!(st.track_flags & FLAG_BIT_SYNC) && (st.track_flags & FLAG_CODE_LOCK) && st.code_counter > 100 &&
                    np_sign(st.i_prompt_prev) != np_sign(ip))
                    st.track_flags |= FLAG_BIT_SYNC;
                st.track_flags |= FLAG_CODE_LOCK;
                st.i_prompt_prev = ip;
                st.q_prompt_prev = qp;
                st.code_counter += 1;
                st.code_hz -= nco_code;
                st.carrier_hz += nco_carrier;
                st.rem_code += (double)n * st.code_step - kChips;
                st.code_step = st.code_hz / fs;
                st.current_sample += n;
                st.n_samples = (int)ceil((kChips - st.rem_code) / st.code_step);
                rec.dll = nco_code;
                rec.pll = nco_carrier;
                rec.fll = 0.0;
                rec.carrier_err = phase_err;
                rec.code_err = code_err;
                rec.cn0 = 0.0;
                rec.pll_lock = 0.0;
                rec.fll_lock = 0.0;
            } else {
                // ---- Kaplan: runCorrelators bookkeeping (channel_l1ca_kaplan.py:392-399)
                if (st.accum_counter == kMsPerBit) st.accum_counter = 0;
                st.accum_counter += 1;
                // runDiscriminators (:405-430)
                double fll_d = 0.0, pll_d = 0.0, dll_d;
                if (st.lock_state == LOCK_PULL_IN) {
                    if (st.code_counter > 1) fll_d = fll_atan(ip, qp, st.i_prompt_prev, st.q_prompt_prev, 1e-3);
                    dll_d = dll_nneml(ie, qe, il, ql);
                } else {
                    fll_d = fll_atan(ip, qp, st.i_prompt_prev, st.q_prompt_prev, 1e-3);
                    pll_d = pll_costas(ip, qp);
                    dll_d = dll_nneml(ie, qe, il, ql);
                }
                // FLLassistedPLL_2ndOrder (tracking.py:246-279) via runCarrierFrequencyFilter (:434-447)
                const double w0f = st.fll_bw / kW0Bw1, w0p = st.pll_bw / kW0Bw2;
                const double upd = (pll_d * (w0p * w0p) + fll_d * w0f) * 1e-3;
                double carrier_err = upd + st.pll_mem;
                st.pll_mem = upd;
                carrier_err += pll_d * kW0A2 * w0p;
                // BorreLoopFilter via runCodeFrequencyFilter (:451-461)
                const double code_err = borre_filter(dll_d, st.dll_mem, cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi * 1.0);
                // runLoopIndicators (:465-502)
                if (st.code_counter != 0) {
                    double v = ip * st.i_prompt_prev - qp * st.q_prompt_prev;
                    v *= np_sign(ip * st.i_prompt_prev + qp * st.q_prompt_prev);
                    v /= (ip * ip + qp * qp);
                    v = fabs(v);
                    st.fll_lock = (1.0 - 0.005) * st.fll_lock + 0.005 * v;
                    if (st.lock_state > LOCK_PULL_IN) {
                        const double nbd = ip * ip - qp * qp, nbp = ip * ip + qp * qp;
                        st.pll_lock = (1.0 - 0.005) * st.pll_lock + 0.005 * (nbd / nbp);
                    }
                    const double d = fabs(ip) - fabs(qp);
                    st.cn0_ratio_acc += (ip * ip + qp * qp) / (d * d);
                    if (st.accum_counter == kMsPerBit) {
                        const double lam = 1.0 / (st.cn0_ratio_acc / (double)st.accum_counter);
                        const double c = lam * (1.0 / ((double)st.accum_counter * 1e-3));
                        st.cn0 = (1.0 - 0.1) * st.cn0 + 0.1 * c;
                        st.cn0_ratio_acc = 0.0;
                    }
                }
                // postTrackingUpdate (:506-534)
                st.code_counter += 1;
                st.dll_mem = dll_d;
                st.rem_carrier -= st.carrier_hz * kGpsTwoPi * (double)n / fs;
                st.rem_carrier = py_mod(st.rem_carrier, kGpsTwoPi);
                st.code_hz -= code_err;
                st.carrier_hz += carrier_err;
                st.rem_code += (double)n * st.code_step - kChips;
                st.code_step = st.code_hz / fs;
                st.current_sample += n;
                st.n_samples = (int)ceil((kChips - st.rem_code) / st.code_step);
                // trackingStateUpdate (:538-619)
                if (st.lock_state != LOCK_PULL_IN && st.cn0 > cfg.dll_threshold && !(st.track_flags & FLAG_CODE_LOCK))
                    st.track_flags |= FLAG_CODE_LOCK;
                else if (st.cn0 < cfg.dll_threshold && (st.track_flags & FLAG_CODE_LOCK))
                    st.track_flags ^= FLAG_CODE_LOCK;
                if ((st.track_flags & FLAG_CODE_LOCK) && !(st.track_flags & FLAG_BIT_SYNC)) {
                    if (np_sign(st.i_prompt_prev) != np_sign(ip)) {
                        st.track_flags |= FLAG_BIT_SYNC;
                        st.accum_counter = 1;
                        st.cn0_ratio_acc = 0.0;
                    }
                }
                st.i_prompt_prev = ip;
                st.q_prompt_prev = qp;
                if (st.lock_state != LOCK_NARROW && st.fll_lock >= cfg.fll_thr_narrow && st.pll_lock >= cfg.pll_thr_narrow) {
                    st.lock_state = LOCK_NARROW;
                    st.fll_bw = cfg.fll_bw_narrow;
                    st.pll_bw = cfg.pll_bw_narrow;
                    st.spacing_sel = 1;
                    st.time_in_state = 0;
                } else if (st.lock_state != LOCK_WIDE && st.fll_lock >= cfg.fll_thr_wide && st.fll_lock < cfg.fll_thr_narrow) {
                    st.lock_state = LOCK_WIDE;
                    st.fll_bw = cfg.fll_bw_wide;
                    st.pll_bw = cfg.pll_bw_wide;
                    st.spacing_sel = 0;
                    st.time_in_state = 0;
                } else if (st.lock_state != LOCK_PULL_IN && st.fll_lock <= cfg.fll_thr_wide) {
                    st.lock_state = LOCK_PULL_IN;
                    st.fll_bw = cfg.fll_bw_pullin;
                    st.pll_bw = 0.0;
                    st.spacing_sel = 0;
                    st.time_in_state = 0;
                } else {
                    st.time_in_state += 1;
                }
                rec.dll = dll_d;
                rec.pll = pll_d;
                rec.fll = fll_d;
                rec.carrier_err = carrier_err;
                rec.code_err = code_err;
                rec.cn0 = st.cn0;
                rec.pll_lock = st.pll_lock;
                rec.fll_lock = st.fll_lock;
            }
            // decodeBit (kaplan:728-754, borre:470-491): 20 prompts after bit sync -> one bit (Prompt2Bit)
            if (!(st.track_flags & FLAG_BIT_SYNC)) {
                st.nav_prompt_sum = 0.0;
                st.nav_sum_counter = 0;
            } else {
                st.nav_prompt_sum += ip;
                st.nav_sum_counter += 1;
                if (st.nav_sum_counter == kMsPerBit) {
                    const int bit = st.nav_prompt_sum > 0.0 ? 1 : 0;
                    rec.nav_bit = bit;
                    if (nav_bits && sh->bits_this_run < max_bits) nav_bits[(size_t)ch * max_bits + sh->bits_this_run] = (int8_t)bit;
                    sh->bits_this_run += 1;
                    st.nav_bits_emitted += 1;
                    st.nav_prompt_sum = 0.0;
                    st.nav_sum_counter = 0;
                }
            }
            rec.carrier_hz = st.carrier_hz;
            rec.code_hz = st.code_hz;
            rec.lock_state = st.lock_state;
            rec.track_flags = st.track_flags;
            sh->epochs_done = epoch + 1;
        }
        // the next iteration's first barrier orders lane 0's LDS writes against everyone's reads
    }
    if (tid == 0) {
        const int epochs_done = sh->epochs_done;
        if (epochs_done < n_epochs) {
            st.n_samples = -1 - epochs_done;  // stopped early: -(1 + epochs completed)
            if (keep_traj)
                for (int k = epochs_done; k < n_epochs; ++k) traj[(size_t)ch * n_epochs + k].n_samples = 0;
        }
        states[ch] = st;
        if (n_bits) n_bits[ch] = sh->bits_this_run < max_bits ? sh->bits_this_run : max_bits;
    }
}

}  // namespace

extern "C" {

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
    if (!rc) rc = sdr_devbuf_reserve(e, &e->track_traj, traj ? traj_bytes : (size_t)n_ch * sizeof(sdr_track_epoch));
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
    const size_t shmem_base = (size_t)(kTrackWaves * 2 * kTaps) * sizeof(double) + sizeof(EpochShared) +
                              (size_t)((lut_words + 3) & ~3) * sizeof(uint32_t);
    const size_t prefix_bytes = (size_t)kTrackThreads * kPrefixSlots * sizeof(double2);
    // The boundary variant of the correlator needs a 272-byte LDS strip per lane; long multi-period
    // replicas that leave no room for it are tracked with the per-sample variant.
    const int use_prefix = shmem_base + prefix_bytes <= 160u * 1024u && e->lut_stride < kFastMaxLutWords ? 1 : 0;
    const size_t shmem = shmem_base + (use_prefix ? prefix_bytes : 0);
    sdr_track_state* d_st = (sdr_track_state*)e->track_state.ptr;
    const sdr_loop_cfg* d_cfg = (const sdr_loop_cfg*)e->track_cfg.ptr;
    sdr_track_epoch* d_traj = (sdr_track_epoch*)e->track_traj.ptr;
    const int keep = traj ? 1 : 0;
    if (shmem > 160u * 1024u) return sdr_fail(SDR_ERR_RANGE, "closed-loop tracking: code table does not fit the LDS");
    {
        ProfScope ps(e, "track_kernel");
        auto launch = [&](auto kernel) {
            // more than 64 KB of dynamic LDS has to be granted per kernel
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            hipLaunchKernelGGL(kernel, dim3(n_ch), dim3(kTrackThreads), shmem, e->stream, e->iq, e->iq_capacity, d_st,
                               d_cfg, n_epochs, d_traj, keep, d_bits, max_bits, d_nbits, e->luts, lut_words,
                               e->lut_stride, use_prefix);
        };
        switch (e->iq_fmt) {
            case SDR_FMT_CI8: launch(track_kernel<SDR_FMT_CI8>); break;
            case SDR_FMT_CI16: launch(track_kernel<SDR_FMT_CI16>); break;
            case SDR_FMT_CF32: launch(track_kernel<SDR_FMT_CF32>); break;
            default: launch(track_kernel<SDR_FMT_CF64>); break;
        }
    }
    SDR_HIP(hipGetLastError());
    SDR_HIP(hipMemcpyAsync(st, e->track_state.ptr, (size_t)n_ch * sizeof(sdr_track_state), hipMemcpyDeviceToHost, e->stream));
    if (traj) SDR_HIP(hipMemcpyAsync(traj, e->track_traj.ptr, traj_bytes, hipMemcpyDeviceToHost, e->stream));
    if (nav_bits) {
        SDR_HIP(hipMemcpyAsync(nav_bits, d_bits, bits_bytes, hipMemcpyDeviceToHost, e->stream));
        SDR_HIP(hipMemcpyAsync(n_bits, d_nbits, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    }
    SDR_HIP(hipStreamSynchronize(e->stream));
    for (int c = 0; c < n_ch; ++c)
        if (st[c].n_samples < 0)
            return sdr_fail(SDR_ERR_RANGE, "channel %d stopped after %d epochs: NCO state left the staged replica / ring",
                            c, -1 - st[c].n_samples);
    return SDR_OK;
}

}  // extern "C"
