/*
 * sydr_amd.h -- C-ABI of the MI355X (gfx950) GNSS correlator engine.
 *
 * This is the drop-in boundary for the one hot path of aproposorg/sydr:
 * PCPS acquisition, the E/P/L tracking correlators and PRN replica
 * generation.  It supersedes the reference's legacy native layer
 * (sydr/c_functions/tracking.c, acquisition.c, loaded through ctypes by
 * sydr/old/tracking/tracking_epl_c.py:31 and
 * sydr/old/acquisition/acquisition_pcps_c.py:32) and is what the live NumPy
 * hot path (sydr/dsp/acquisition.py:9-115, sydr/dsp/tracking.py:92-116,
 * sydr/signal/gnsssignal.py:9-58) is replaced with.
 *
 * Conventions (kept from the reference's ctypes layer):
 *   - plain C linkage, plain pointers and sizes, caller allocates every output;
 *   - complex numbers are interleaved (re, im);
 *   - 2-D arrays are C-contiguous row-major;
 *   - the library keeps no caller pointer after a call returns.
 * Added over the reference (which returned void and had no error path):
 *   - every call returns 0 or a negative sdr_status, text via sdr_last_error();
 *   - one opaque engine per GPU; calls on one engine are serialised by the caller.
 *
 * All arithmetic that decides an INTEGER result (code-chip index, peak
 * indices) is IEEE fp64 in the reference's operation order.
 */
#ifndef SYDR_AMD_H
#define SYDR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: what this header declares -- and nothing else -- is in its dynamic
 * symbol table (sydr/c_functions/Makefile:1-12: one .so, only the bound symbols matter); tests/test_abi.py holds
 * `nm -D --defined-only` against these declarations. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SDR_ABI_VERSION 5   /* 5: + sdr_bank_tick_mirrored_begin / _end, sdr_iq_upload_queue, sdr_host_alloc / _free, options "tick_server" + sdr_tick_server_stats, "bind_thread_to_device" (additive); 4: + sdr_build_id, sdr_epl_plan_create_dev, sdr_bank_tick_mirrored, sdr_iq_upload_begin, sdr_block_schedule, sdr_bank_step_begin / _end (additive) */

typedef struct sdr_engine sdr_engine;

enum sdr_status {
    SDR_OK = 0,
    SDR_ERR_INVALID = -1,     /* bad argument                                   */
    SDR_ERR_HIP = -2,         /* a HIP runtime call failed (text has the cause) */
    SDR_ERR_NOMEM = -3,       /* host or device allocation failed               */
    SDR_ERR_UNSUPPORTED = -4, /* size / format not supported by the kernels     */
    SDR_ERR_RANGE = -5,       /* request reaches outside the IQ ring / code LUT */
    SDR_ERR_STATE = -6        /* call made before the state it needs exists     */
};

/* IQ sample formats held in HBM.  ci8 is the reference's file format
 * (sydr/signal/rfsignal.py:36-37,127-130: interleaved int8 I,Q). */
enum sdr_iq_format {
    SDR_FMT_CI8 = 0,  /* int8  I, int8  Q  (2 B / sample) */
    SDR_FMT_CI16 = 1, /* int16 I, int16 Q  (4 B / sample) */
    SDR_FMT_CF32 = 2, /* float I, float Q  (8 B / sample) */
    SDR_FMT_CF64 = 3  /* double I, double Q (16 B / sample) = numpy complex128 */
};

#define SDR_MAX_TAPS 8
#define SDR_GPS_L1CA_CHIPS 1023

/* ---------------------------------------------------------------- library */
const char* sdr_last_error(void);
int sdr_abi_version(void);
/* First 16 hex digits of the SHA-256 over the library's sources as they were when it was built (the .hip / .h files of
 * sydr_amd/csrc in name order, then this header): what a profile or a counter file names as the build it was taken on. */
const char* sdr_build_id(void);
/* Number of visible GPUs (0 on a CPU-only host; never an error there). */
int sdr_device_count(int* n);

/* ----------------------------------------------------------------- engine */
int sdr_engine_create(int device_id, sdr_engine** out);
void sdr_engine_destroy(sdr_engine* e);
/* Wait for everything queued on the engine's stream. */
int sdr_engine_sync(sdr_engine* e);
/* HIP-event timing.  enable=1 brackets every stage of a call (named scopes: "epl_kernel", "pcps_fwd_fft", ...) with
 * hipEvents on the launch stream; enable=2 brackets each whole call instead ("call_pcps": one event pair around
 * everything sdr_pcps launches -- the per-stage pairs each cost the stream a few microseconds); sdr_prof_read drains
 * them (it syncs).  While either is on, sdr_pcps keeps to one stream. */
int sdr_prof_enable(sdr_engine* e, int enable);
/* Sum/count of launch durations of kernels whose name starts with `prefix`
 * ("" = all) since the last sdr_prof_reset. */
int sdr_prof_read(sdr_engine* e, const char* prefix, double* total_ms, int64_t* launches);
int sdr_prof_reset(sdr_engine* e);

/* Diagnostic switches (tests, A/B timing): "pcps_materialise_map" = 1 writes the whole correlation map even when the
 * caller asks for indices and ratio only; "pcps_radix_passes" = 1 runs one kernel per radix pass instead of the
 * four-step transform; "pcps_general_kernels" = 1 keeps the general four-step kernels where the register-resident
 * 125 x 200 ones would run; "pcps_prn_chunk" = n searches n PRNs per inverse sweep; "pcps_one_stream" = 1 keeps the sweeps of a map-free search on one stream;
 * "pcps_fused" = 0 keeps a map-free search at 25 MHz on the two-kernel sweeps where one launch of persistent workgroups, one
 * (PRN, bin) transform per workgroup, would run (256 transforms or more); "pcps_no_spectra_cache" = 1
 * recomputes conj(fft(code)) in every search, as the reference does (channel_l1ca_kaplan.py:184-185), instead of keeping
 * the spectra of the staged codes; "ingest_by_copy_command" = 1 moves the slabs of sdr_iq_upload_begin / sdr_bank_tick
 * into the ring with a copy command instead of the ingest kernel; "track_one_launch_tick" = 1 runs a one-epoch step
 * (sdr_bank_tick*, sdr_bank_step with n_epochs = 1) as one workgroup per channel instead of on the cluster a block of
 * epochs would use (other order of additions); "track_two_launch_tick" = 1 runs that cluster's one-epoch step as two
 * launches cut at the exchange of the parts' sums instead of one (the part that draws its channel's last ticket
 * collects: same bits); "ingest_with_tick" = 0 gives the slab of sdr_iq_upload_begin a launch of its own again instead of
 * workgroups of the tick's launch; "pcps_no_shared_spectra" = 1 transforms the Doppler-mixed millisecond once per bin
 * even where bins a whole number of FFT bins apart could share a spectrum; "epl_no_chip_variant",
 * "epl_no_split_variant", "epl_no_half_chip_view" = 1 keep the E/P/L correlator from its chip-aligned core, from the
 * kernel with the tap switch positions compiled in, from the half-chip view of 32-52 samples per chip.  Integer
 * results do not depend on any of them, floating ones to rounding. */
int sdr_set_option(sdr_engine* e, const char* name, int value);

/* Measured HBM copy rate of THIS GPU: a hand-written 16-byte-per-lane grid-stride copy kernel moves n_bytes
 * from one buffer to another `reps` times; *gbps = (bytes read + bytes written) / time.  The second denominator
 * beside the 8 TB/s datasheet peak; no counterpart in the reference. */
int sdr_hbm_copy_rate(sdr_engine* e, int64_t n_bytes, int reps, double* gbps);

/* ---------------------------------------------------------------- IQ ring
 * Replaces the host shm ring (sydr/channel/channelManager.py:57-61,
 * sydr/utils/circularbuffer.py:21-137): capacity_samples slots in HBM,
 * addressed modulo capacity.  capacity must be a multiple of 8 samples. */
int sdr_iq_alloc(sdr_engine* e, int64_t capacity_samples, int fmt);
/* Copy n_samples host samples (in the ring's format) to ring_offset.. ,
 * wrapping at the end of the ring (CircularBuffer.shift, circularbuffer.py:54-82). */
int sdr_iq_upload(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset);
/* Copy ring samples back to the host in the ring's format (tests / oracle). */
int sdr_iq_download(sdr_engine* e, void* iq, int64_t n_samples, int64_t ring_offset);

/* Synthetic multi-satellite IQ written straight into the ring:
 *   x[n] = sum_s amp * c_s(chip_s(n)) * d_s(n) * exp(j*2*pi*(doppler_s*n/fs + phase_s)) + CN(0, sigma^2)
 * rounded and clipped to the ring's integer format.  chip_s(n) =
 * code_phase_s + n * 1.023e6*(1+doppler_s/1575.42e6)/fs (chips, mod 1023); d_s is a
 * seeded +-1 sequence changing every 20 code periods.  Deterministic in (seed, n). */
#define SDR_SYNTH_CODE_SLOT 1 /* `prn` is a staged code slot (any length), not a GPS PRN number      */
#define SDR_SYNTH_BOC11 2     /* multiply by the BOC(1,1) square sub-carrier (flip every half chip)  */
typedef struct sdr_synth_sat {
    int32_t prn;        /* GPS PRN 1..210, or a code slot with SDR_SYNTH_CODE_SLOT */
    int32_t flags;
    double doppler_hz;  /* carrier Doppler                             */
    double code_phase;  /* chips into the code at ring sample 0        */
    double carrier_phase; /* cycles at ring sample 0                   */
    double amplitude;   /* per-axis amplitude in LSB                   */
} sdr_synth_sat;
int sdr_iq_synth(sdr_engine* e, const sdr_synth_sat* sats, int n_sats, double fs,
                 double noise_sigma, uint64_t seed, int64_t first_sample, int64_t n_samples);

/* ------------------------------------------------------------ PRN replicas
 * Code slots are device-resident +-1 chip tables staged for the correlators.
 * sdr_code_gps_l1ca generates the C/A Gold code ON DEVICE (G1/G2 LFSRs, G2
 * delay table) and replaces GenerateGPSGoldCode (sydr/signal/gnsssignal.py:9-31
 * -> sydr/signal/ca.py:70-112; chip mapping bit 1 -> +1, bit 0 -> -1). */
int sdr_code_slots(sdr_engine* e, int n_slots, int max_chips);
/* Same, with the replicas staged over max_periods code periods so that one correlator epoch may span
 * several periods (e.g. 4 ms of C/A code) or taps may sit many chips out; chips*periods <= 32768. */
int sdr_code_slots_ex(sdr_engine* e, int n_slots, int max_chips, int max_periods);
int sdr_code_gps_l1ca(sdr_engine* e, int slot, int prn);
/* Stage an arbitrary +-1 code (e.g. a synthetic 4092-chip E1-like code). */
int sdr_code_custom(sdr_engine* e, int slot, const int8_t* chips, int n_chips);
/* Read a staged code back: out_chips[n_chips] in {-1,+1}. */
int sdr_code_read(sdr_engine* e, int slot, int8_t* out_chips, int max_chips, int* n_chips);
/* UpsampleCode (sydr/signal/gnsssignal.py:35-58): out[k] = code[trunc((ts*k)/tc)],
 * k < n_samples = round(fs/1000) for GPS L1 C/A; computed on device. */
int sdr_code_upsample(sdr_engine* e, int slot, double fs, int64_t n_samples, int8_t* out);

/* ------------------------------------------------ E/P/L tracking correlators
 * One item = one call of EPL (sydr/dsp/tracking.py:92-116) = one channel-epoch:
 *   replica_i = exp(1j*(-(carrier_hz*2.0*pi*(i/fs)) + rem_carrier)),  i = 0..n-1
 *   idx_i     = ceil(linspace(rem_code+spacing, code_step*n+rem_code+spacing, n, endpoint=False))
 *   I_tap, Q_tap = sum code[idx_i]*Re/Im(replica_i*x_i)
 * with code the padded table [c[L-1], c[0..L-1], c[0]] (channel_l1ca_kaplan.py:104-107),
 * extended periodically here so that any tap spacing can be served. */
typedef struct sdr_epl_item {
    int32_t code_slot;    /* staged PRN replica                                   */
    int32_t n_samples;    /* samples in this epoch (track_requiredSamples)        */
    int64_t start_sample; /* ring index of the first sample (currentSample)       */
    double carrier_hz;    /* carrierFrequency                                     */
    double rem_carrier;   /* remainingCarrier [rad]                               */
    double rem_code;      /* remainingCode [chips]                                */
    double code_step;     /* codeStep [chips/sample]                              */
} sdr_epl_item;

/* Synchronous convenience call: out[n_items][2*n_taps] = I,Q per tap, in tap order
 * (the reference's [IE,QE,IP,QP,IL,QL] for taps (-0.5,0,0.5)). */
int sdr_epl_batch(sdr_engine* e, const sdr_epl_item* items, int n_items, const double* spacing,
                  int n_taps, double fs, double* out);

/* Resident plans: items and outputs stay in HBM so a timed region holds only
 * kernel launches.  run is asynchronous on the engine stream.
 * Creating a plan checks every item against the ring and the staged replicas (host), uploads the list and, for lists the
 * straight-line kernels serve (ci8 ring; 24-25 samples per (half-)chip with taps +-0.5 chip or whole chips apart; 9.5-10 or
 * 11.5-12 samples per chip with taps +-0.5 chip), works out each epoch's setup -- tap constants, chip geometry, carrier
 * rotations -- in one launch with a thread per item: 400-560 bytes of device memory per item, so that the correlator
 * starts an epoch with scalar loads instead of ~450 vector instructions repeated in all 64 lanes. */
typedef struct sdr_epl_plan sdr_epl_plan;
int sdr_epl_plan_create(sdr_engine* e, const sdr_epl_item* items, int n_items,
                        const double* spacing, int n_taps, double fs, sdr_epl_plan** out);
/* The same for a list that is already in DEVICE memory (a closed-loop launch's trajectory turned into items, a list another
 * kernel produced): copied device to device -- the plan owns its items either way -- and checked there, one thread per item;
 * nothing crosses the host but the verdict.  (The host form checks long lists the same way, behind their upload.) */
int sdr_epl_plan_create_dev(sdr_engine* e, const sdr_epl_item* items_dev, int n_items,
                            const double* spacing, int n_taps, double fs, sdr_epl_plan** out);
int sdr_epl_plan_run(sdr_engine* e, sdr_epl_plan* p);
/* Launch only items [first, first+count) of the plan (e.g. one second of a long stream). */
int sdr_epl_plan_run_range(sdr_engine* e, sdr_epl_plan* p, int64_t first, int64_t count);
int sdr_epl_plan_fetch(sdr_engine* e, sdr_epl_plan* p, double* out); /* syncs */
void sdr_epl_plan_destroy(sdr_engine* e, sdr_epl_plan* p);

/* --------------------------------------------------------- PCPS acquisition
 * PCPS (sydr/dsp/acquisition.py:9-74) + TwoCorrelationPeakComparison (:78-115)
 * for n_prn staged codes over the same ring slice:
 *   bins = arange(-doppler_range, doppler_range+1, doppler_step)
 *   map[prn][b][:] = sum_noncoh | sum_coh ifft( fft(x_ms * exp(-1j*(if_hz-bins[b])*k*2*pi/fs)) * conj(fft(code)) ) |
 * peak_bin/peak_code: first global maximum in row-major order;
 * peak_ratio: peak / second peak in the same row outside +-samplesPerChip,
 * never looking at the last code sample (reference behaviour).
 * corr_map may be NULL (indices and ratio only). n_bins_out (nullable) gets len(bins). */
int sdr_pcps(sdr_engine* e, const int32_t* code_slots, int n_prn, int64_t start_sample, double fs,
             double if_hz, double doppler_range, double doppler_step, int coh, int noncoh,
             int64_t* peak_bin, int64_t* peak_code, double* peak_ratio, double* corr_map,
             int* n_bins_out);
/* Same search with caller-supplied code spectra: code_spectra[n_prn][n_code] interleaved
 * complex128 = the `codeFFT` argument of the reference's PCPS() (acquisition.py:9; built as
 * conj(fft(UpsampleCode(code))) at channel_l1ca_kaplan.py:184-185).  Function-level drop-in. */
int sdr_pcps_spectra(sdr_engine* e, const double* code_spectra, int n_prn, int n_code,
                     int64_t start_sample, double fs, double if_hz, double doppler_range,
                     double doppler_step, int coh, int noncoh, int64_t* peak_bin, int64_t* peak_code,
                     double* peak_ratio, double* corr_map, int* n_bins_out);
/* len(np.arange(-range, range+1, step)) for float range/step. */
int sdr_pcps_bins(double doppler_range, double doppler_step);
/* TwoCorrelationPeakComparison alone on a caller-supplied map[n_bins][n_code] (row-major
 * f64), as the legacy twoCorrelationPeakComparison symbol offered (acquisition.c:181-244);
 * the search itself runs on the device. */
int sdr_two_peak_compare(sdr_engine* e, const double* corr_map, int n_bins, int n_code,
                         int samples_per_chip, int64_t* peak_bin, int64_t* peak_code,
                         double* peak_ratio);

/* SerialSearch acquisition (sydr/dsp/acquisition.py:119-155; plugin sydr/channel/channel_l1ca_kaplan_ss.py):
 *   map[prn][b][k] = sum over `noncoh` successive code periods of
 *                    |sum_n x[n]*exp(+1j*bins[b]*n*2*pi/fs) * code[(trunc((ts*n)/tc) - k) mod L]|^2
 * for the L circular chip shifts k, and TwoCorrelationPeakComparison_SS (:159-193) on each map
 * (second peak = maximum outside the 3x3 block around the first, with Python's slice semantics).
 * corr_map (nullable) is [n_prn][bins][L]. */
int sdr_serial_search(sdr_engine* e, const int32_t* code_slots, int n_prn, int64_t start_sample, double fs,
                      double doppler_range, double doppler_step, int noncoh, int64_t* peak_bin,
                      int64_t* peak_code, double* peak_ratio, double* corr_map, int* n_bins_out);
int sdr_two_peak_compare_ss(sdr_engine* e, const double* corr_map, int n_rows, int n_cols,
                            int64_t* peak_bin, int64_t* peak_code, double* peak_ratio);

/* ------------------------------------------------- closed-loop tracking
 * On-device loop closure: one persistent workgroup per
 * channel runs n_epochs of correlate -> discriminators -> loop filters -> NCO
 * update without leaving the GPU.  loop_kind selects the reference plugin whose
 * arithmetic is followed: 0 = Borre (channel_l1ca_borre.py:333-451),
 * 1 = Kaplan (channel_l1ca_kaplan.py:342-619). */
typedef struct sdr_track_state {
    int32_t code_slot;
    int32_t n_samples;        /* track_requiredSamples of the NEXT epoch            */
    int64_t current_sample;   /* currentSample (ring index)                         */
    double carrier_hz;        /* carrierFrequency                                   */
    double code_hz;           /* codeFrequency                                      */
    double rem_carrier;       /* remainingCarrier / NCO_remainingCarrier            */
    double rem_code;          /* remainingCode / NCO_remainingCode                  */
    double code_step;         /* codeStep                                           */
    double dll_mem;           /* Borre: NCO_codeError; Kaplan: dllDiscrim           */
    double pll_mem;           /* Borre: NCO_carrierError; Kaplan: fll_vel_memory    */
    double i_prompt_prev;     /* Kaplan iPromptPrev                                 */
    double q_prompt_prev;     /* Kaplan qPromptPrev                                 */
    double fll_lock;          /* Kaplan fllLockIndicator                            */
    double pll_lock;          /* Kaplan pllLockIndicator                            */
    double cn0;               /* Kaplan cn0 (= dllLockIndicator)                    */
    double cn0_ratio_acc;     /* Kaplan cn0_PdPnRatio                               */
    double fll_bw;            /* Kaplan fllBandwidth                                */
    double pll_bw;            /* Kaplan pllBandwidth                                */
    int32_t code_counter;     /* codeCounter                                        */
    int32_t accum_counter;    /* Kaplan correlatorsAccumCounter                     */
    int32_t lock_state;       /* Kaplan LoopLockState (1 PULL_IN, 2 WIDE, 3 NARROW) */
    int32_t track_flags;      /* TrackingFlags bit set                              */
    int32_t time_in_state;    /* Kaplan timeSinceLastState                          */
    int32_t spacing_sel;      /* Kaplan: 0 = wide taps, 1 = narrow taps             */
    /* navigation-bit accumulation on device: decodeBit of
     * channel_l1ca_kaplan.py:728-754 / channel_l1ca_borre.py:470-491 + Prompt2Bit (dsp/decoding.py:16-27) */
    double nav_prompt_sum;    /* navPromptSum                                       */
    int32_t nav_sum_counter;  /* navPromptSumCounter                                */
    int32_t nav_bits_emitted; /* bits produced so far (navBitsCounter without the reference's flushes) */
} sdr_track_state;

typedef struct sdr_loop_cfg {
    int32_t loop_kind;        /* 0 Borre, 1 Kaplan                                  */
    int32_t n_taps;           /* 3 (E/P/L, the reference) or 5 (VE/E/P/L/VL); the discriminators use the centre
                               * tap as prompt and its two neighbours as early / late                           */
    double fs;
    double spacing_wide[SDR_MAX_TAPS];
    double spacing_narrow[SDR_MAX_TAPS];
    double dll_tau1, dll_tau2, dll_pdi;
    double pll_tau1, pll_tau2, pll_pdi;       /* Borre                              */
    double dll_threshold;                     /* Kaplan                             */
    double fll_bw_pullin, fll_bw_wide, fll_bw_narrow, fll_thr_wide, fll_thr_narrow;
    double pll_bw_wide, pll_bw_narrow, pll_thr_wide, pll_thr_narrow;
    /* generalisation beyond the reference's GPS L1 C/A epoch (BASELINE configs 4-5; no reference counterpart):
     * chips per correlator epoch (code length x code periods; 0 = 1023, the reference's GPS_L1CA_CODE_SIZE_BITS at
     * channel_l1ca_kaplan.py:529-532) and epochs per navigation symbol (0 = 20 = LNAV_MS_PER_BIT). */
    double epoch_chips;
    int32_t epochs_per_bit;
    int32_t reserved;
    /* epoch duration the Kaplan discriminators / filters / C/N0 estimator are scaled with; 0 = 1e-3 s, the constant
     * the reference hard-codes (channel_l1ca_kaplan.py:417,425,443,494).  4e-3 for the 4 ms epochs of configs 4-5. */
    double epoch_seconds;
} sdr_loop_cfg;

/* Per-epoch record written when traj != NULL (what the reference's tracking
 * packet carries, channel_l1ca_kaplan.py:653-676, plus the NCO inputs used). */
typedef struct sdr_track_epoch {
    int64_t start_sample;
    int32_t n_samples;
    int32_t lock_state;
    double carrier_hz_in, rem_carrier_in, rem_code_in, code_step_in; /* inputs of the epoch */
    double corr[2 * SDR_MAX_TAPS];
    double dll, pll, fll;
    double carrier_err, code_err;
    double carrier_hz, code_hz;           /* after the update */
    double cn0, pll_lock, fll_lock;
    int32_t track_flags;
    int32_t nav_bit;                      /* -1: none this epoch; 0/1: bit closed by this epoch (20 prompts) */
} sdr_track_epoch;

/* A channel is tracked by a cluster of 1, 2, 4 or 8 cooperating workgroups (one per compute unit; the
 * partial sums of an epoch are added in a fixed order, so a given cluster size always gives the same
 * bits).  parts = 0 (default) lets the library fill the GPU: min(8, CUs / n_ch) rounded down to a power
 * of two.  The reference has no counterpart: its channels are one Python process each
 * (sydr/channel/channel.py:90-151). */
int sdr_track_cluster(sdr_engine* e, int parts);
int sdr_track_closed_loop(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfg,
                          int n_epochs, sdr_track_epoch* traj /* [n_ch][n_epochs], nullable */);
/* Same run; additionally the navigation bits decided during it leave the device as one byte each
 * (only 1 bit per 20 ms per channel has to cross PCIe): nav_bits[n_ch][max_bits] (0/1), bit k of
 * channel c at nav_bits[c*max_bits + k]; n_bits[c] = number written for channel c. */
int sdr_track_closed_loop_bits(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfg,
                               int n_epochs, sdr_track_epoch* traj, int8_t* nav_bits, int max_bits,
                               int32_t* n_bits);

/* Same run with a per-channel outcome instead of one status for the call: epochs_done[c] = epochs channel c
 * completed (< n_epochs when its NCO left the staged replica / the ring: that channel stops, st[c] holds its last
 * valid state, the others are unaffected -- the reference likewise lets a channel that lost lock run on alone,
 * sydr/channel/channel.py:121-160).  cfgs: one sdr_loop_cfg per channel when cfg_per_channel != 0, else cfgs[0]
 * serves all (channelManager.addChannel takes a configuration per call, channelManager.py:70-93). */
int sdr_track_closed_loop_ex(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfgs,
                             int cfg_per_channel, int n_epochs, sdr_track_epoch* traj, int8_t* nav_bits,
                             int max_bits, int32_t* n_bits, int32_t* epochs_done);

/* ------------------------------------------------- device-resident channel bank
 * The tracking state of every channel of one GPU lives in HBM (sdr_track_state + sdr_loop_cfg per channel) and
 * is advanced there; the host keeps views.  This replaces the reference's one-process-per-channel fan-out with
 * its per-millisecond Event barrier (sydr/channel/channelManager.py:70-127,149-188, sydr/channel/channel.py:121-160):
 * one sdr_bank_step = "every listed channel runs n_epochs tracking epochs", one sdr_bank_tick = one iteration of the
 * receiver's outer loop (receiver.py:120-131: addNewRFData of one slab, then run) in a single call.
 * Channels are independent; a call on one stream never touches channels that are not listed. */
typedef struct sdr_bank sdr_bank;
int sdr_bank_create(sdr_engine* e, int max_channels, sdr_bank** out);
void sdr_bank_destroy(sdr_engine* e, sdr_bank* b);
/* Host -> HBM: (re)initialise channel ch (after acquisition: postAcquisitionUpdate, channel_l1ca_kaplan.py:217-235). */
int sdr_bank_put(sdr_engine* e, sdr_bank* b, int ch, const sdr_track_state* st, const sdr_loop_cfg* cfg);
/* HBM -> host. */
int sdr_bank_get(sdr_engine* e, sdr_bank* b, int ch, sdr_track_state* st);
/* Advance the listed channels by n_epochs each.  records[n_ch][n_epochs] (nullable) receives the per-epoch
 * packets' contents, states_out[n_ch] (nullable) the states after the run, epochs_done[n_ch] (nullable) the epochs
 * each channel completed, nav_bits/n_bits as in sdr_track_closed_loop_bits.  stream_id: 0 = the engine's stream,
 * else a stream from sdr_stream_create (one stream per channel batch).  Synchronous on that stream. */
int sdr_bank_step(sdr_engine* e, sdr_bank* b, const int32_t* channels, int n_ch, int n_epochs,
                  sdr_track_epoch* records, sdr_track_state* states_out, int32_t* epochs_done,
                  int8_t* nav_bits, int max_bits, int32_t* n_bits, int stream_id);
/* sdr_bank_step (on the engine's stream, records and states always produced) in two halves: _begin queues the launch and
 * the copies of its results into page-locked memory and returns at once; _end waits and hands them out.  One step may be
 * in flight per bank; work queued on the engine's stream in between (a tick, an upload) runs after it.  Lets a receiver
 * that tracks ahead (sydr_amd/channel/readahead.py) have its next block computed while it still hands out the current
 * one's packets.  SDR_ERR_STATE: _begin with a step in flight, _end without one. */
int sdr_bank_step_begin(sdr_engine* e, sdr_bank* b, const int32_t* channels, int n_ch, int n_epochs);
int sdr_bank_step_end(sdr_engine* e, sdr_bank* b, sdr_track_epoch* records /* [n_ch][n_epochs] */, sdr_track_state* states_out,
                      int32_t* epochs_done);
/* One receiver tick: copy n_samples new host samples into the ring at ring_offset (CircularBuffer.shift), then
 * one epoch for the listed channels (n_ch may be 0: ingest only).  One stream synchronisation in all. */
int sdr_bank_tick(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                  const int32_t* channels, int n_ch, sdr_track_epoch* records, sdr_track_state* states_out,
                  int32_t* epochs_done);

/* The same tick with the reference's per-tick bookkeeping done here instead of in the caller's language: which
 * channels are ready (channel.py:137-146 -- the ring holds their next epoch completely: getNbUnreadSamples >=
 * track_requiredSamples), one epoch for those, and the caller's MIRRORS of the bank brought up to date in place --
 * what ChannelManager.run() needs to make its TRACKING_UPDATE and CHANNEL_UPDATE packets (channelManager.py:149-188,
 * channel.py:205-228) is in `records` / `updates` when the call returns.  Every array has max_channels rows and is
 * owned by the caller; rows of channels that do not run are not touched. */
typedef struct sdr_tick_update {      /* one CHANNEL_UPDATE (channel.py:205-228) */
    int32_t channel;
    int32_t track_flags;              /* the device's TrackingFlags bits | host_flags[channel]            */
    int64_t unread;                   /* unprocessed_samples: getNbUnreadSamples(currentSample) afterwards */
    int64_t epochs_since_tow;         /* code_since_tow                                                    */
} sdr_tick_update;
typedef struct sdr_tick_mirror {
    int32_t max_channels;             /* rows of every array = the bank's max_channels                     */
    int32_t reserved;
    sdr_track_state* states;          /* in/out: state of every channel as last put / advanced             */
    sdr_track_epoch* last;            /* out: the newest epoch record per channel                          */
    int64_t* epochs_since_tow;        /* in/out (nullable): += 1 per epoch run (codeSinceTOW)              */
    const uint8_t* tracking;          /* in: channel is in ChannelState.TRACKING                           */
    uint8_t* lost;                    /* in/out: the device parked the channel (its NCO left the replica / the ring) */
    const int64_t* host_flags;        /* in (nullable): TrackingFlags bits the host owns (the decoder's)   */
    int32_t* ran;                     /* out: channels that completed an epoch in this tick, ascending     */
    sdr_track_epoch* records;         /* out: their records, same order                                    */
    sdr_tick_update* updates;         /* out: one row per channel with tracking != 0, ascending            */
    int32_t n_ran, n_updates;         /* out                                                               */
    int32_t n_nav_bits;               /* out: records of this tick with nav_bit >= 0                       */
    int32_t n_lost;                   /* out: channels parked by this tick                                 */
    int64_t max_unread;               /* out: largest `unread` among the channels still running            */
} sdr_tick_mirror;
/* iq may be NULL / n_samples 0 when the slab was handed over with sdr_iq_upload_begin already; write_index = the
 * ring's write index AFTER this tick's slab (CircularBuffer.idxWrite). */
int sdr_bank_tick_mirrored(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                           int64_t write_index, sdr_tick_mirror* m);
/* The same tick in two halves, so that ONE host thread drives the banks of several devices the way the reference's
 * manager drives its channel processes -- start every one, then wait for each (channelManager.py:164-171: eventRun.set()
 * for all, then eventDone.wait() for all).  _begin decides who is ready and queues their epoch on the engine's stream
 * (nothing is waited for; the bank's own page-locked block takes the results, so other calls on the engine may come in
 * between, but none that touches THIS bank: they return SDR_ERR_STATE); _end waits, absorbs the results into the mirrors
 * and writes the tick's rows.  sdr_bank_tick_mirrored == _begin + _end.  Same `m` in both halves. */
int sdr_bank_tick_mirrored_begin(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                                 int64_t write_index, sdr_tick_mirror* m);
int sdr_bank_tick_mirrored_end(sdr_engine* e, sdr_bank* b, sdr_tick_mirror* m);
/* sdr_set_option(e, "tick_server", 1): the steady tick (sdr_bank_tick_mirrored*, the slab handed over by
 * sdr_iq_upload_begin) is answered by RESIDENT kernels instead of launches and a stream synchronisation -- what the
 * reference's manager gets from channel processes that wait on an Event between ticks (channel.py:121-160).  The cluster form
 * of the tracking kernel stays on the device; eight more workgroups, the doormen, watch a 64-byte request line in page-locked
 * memory (the whole request in one access over the link), pull an eighth of the slab each into the ring and release the
 * channels; every channel ANSWERS THE HOST ITSELF -- state, record and the request's number written straight into page-locked
 * memory by one of its waves -- and the host spins on those words; who is ready is decided on the device by the arithmetic of
 * channel.py:137-146 and must agree with the caller's mirror (SDR_ERR_STATE otherwise).  Same clusters and order of additions
 * as the plain tick: same bits.  Served: banks whose tracking channels run one tap count and number at most a quarter of the
 * compute units (64), after eight such ticks in a row with no other call on the engine in between (a server takes ~25 ms to
 * start); anything else takes the plain path.  Any other call on the engine (a search, a put, an upload by another route,
 * sdr_engine_destroy) tells the server to leave first and waits for it; the next steady tick starts a new one.  Nothing on the
 * device waits without a bound: the server leaves by itself after 0.2 s without a request; the host waits at most 0.25 s for
 * an answer, then reports SDR_ERR_HIP and goes back to plain ticks.
 * out4: {a server is resident now, requests answered, servers started, the engine went back to plain ticks for good}.
 *
 * sdr_set_option(e, "bind_thread_to_device", 1): the CALLING thread is restricted to the CPUs next to the engine's GPU (the
 * local_cpulist of its PCI function; never outside the thread's present mask), 0 gives it its old mask back; SDR_ERR_UNSUPPORTED
 * where sysfs does not say.  A served tick is a handful of round trips through page-locked words: from the other socket of a
 * two-socket host each crosses the sockets' interconnect too (21 us per tick against 27, examples/receiver_loop.c).  Opt-in: a
 * library does not move its caller's threads unasked. */
int sdr_tick_server_stats(sdr_engine* e, int64_t* out4);
/* Where the answered requests' time went ON THE DEVICE, microseconds summed over them (the first doorman's wall-clock stamps):
 * {slab pulled into the ring, channels released, every channel has answered -- as the doorman sees it, which keeps quiet
 * while the channels work: up to a microsecond late --, the request closed}. */
int sdr_tick_server_phases(sdr_engine* e, double* out4);
/* ... and channel 0's own tick (lane 0 of its first part): {release seen, samples visible, correlated, sums exchanged, loops
 * updated, answer written}. */
int sdr_tick_server_tracker_phases(sdr_engine* e, double* out6);
/* sdr_iq_upload without the wait: the samples are copied out of `iq` before the call returns (the caller may reuse
 * its buffer), their transfer into the ring is queued on the engine's stream and ordered before everything queued
 * there afterwards (slabs above 1 MiB are uploaded synchronously).  sdr_engine_sync completes it for readers on
 * other streams.  Lets addNewRFData start the transfer while the caller is still on its way to run().  Any number of
 * slabs may be outstanding: the engine stages them in two page-locked halves and waits, before it overwrites a half, for
 * the transfer that read it (an event per half) -- a caller that queues a third slab while the first has not reached the
 * ring blocks in this call until it has; nothing is lost or reordered.  A tick in which no channel was ready
 * (sdr_bank_tick_mirrored with n_ran == 0 and n_samples == 0) launches nothing and waits for nothing: the slab is then
 * still in flight when the call returns.  Between receiver ticks the slab only waits in its staging half: the next
 * sdr_bank_tick_mirrored's ONE launch begins with workgroups that pull it into the ring while the trackers behind them set
 * up (or the resident tick server's doormen pull it); every other call on the engine puts it into the ring first, in order.
 * ONE exception to "copied before the call returns": a slab that lies in page-locked memory from sdr_host_alloc, on a 16-byte
 * boundary and a whole number of 16-byte granules long, is read IN PLACE by whoever pulls it into the ring (no staging copy:
 * ~2 us of a 50 KB slab) -- the caller leaves it unchanged until the next sdr_bank_tick* of the engine, or sdr_engine_sync,
 * has returned. */
int sdr_iq_upload_begin(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset);
/* A chunk of a recording (any size) queued for the ring WITHOUT being copied first: one asynchronous copy command on the
 * engine's stream, ordered like everything else queued there; the caller keeps `iq` valid and unchanged until
 * sdr_engine_sync (or a later synchronous call on the engine) returns.  From page-locked memory (sdr_host_alloc) the call
 * returns at once and the transfer runs beside whatever other streams compute -- how a file reader feeds the ring a second
 * of samples at a time while the previous second is correlated (rfsignal.py:58-132 reads the file chunk by chunk; bench.py
 * `host_fed`); from pageable memory the runtime stages the chunk through its own buffers and the call returns when the last
 * piece has been handed over.  n_samples up to the ring's capacity, wrapping at its end. */
int sdr_iq_upload_queue(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset);
/* Page-locked host memory for recordings that are fed with sdr_iq_upload_queue (hipHostMalloc / hipHostFree on the engine's
 * device).  The block is the caller's until sdr_host_free; the engine keeps no pointer to it. */
int sdr_host_alloc(sdr_engine* e, size_t bytes, void** out);
int sdr_host_free(sdr_engine* e, void* block);

/* Host-only helper of a receiver that tracks ahead (no device work): `records[n_ch][n_cols]` hold `done[r]` epochs per channel
 * computed in one sdr_bank_step while the host still feeds its per-millisecond loop (receiver.py:120-131); this works out
 * which tick releases which epoch -- the first tick k whose slab completes it (channel.py:137-146): unread_now[r] +
 * (k + 1) * samples_per_tick >= the samples up to its end, one epoch per channel and tick (channelManager.py:149-188) -- and
 * what every tick's CHANNEL_UPDATE reports (channel.py:205-228).  Outputs (caller-allocated): first[n_ch][n_cols] = the tick
 * of each epoch (-1: not run); *n_ticks; the epochs in tick order (channels ascending inside a tick) as order_rows /
 * order_cols / records_sorted [sum of done], tick k's slice being [starts[k], starts[k + 1]) (starts: max_ticks + 1 entries);
 * last_records[n_ch] = each channel's newest record; per tick and channel [max_ticks][n_ch] (row stride n_ch): unread samples
 * after the tick, the device's TrackingFlags bits (flags0 before the channel's first epoch) and code_since0 + epochs released
 * so far; last_tick[n_ch] = the tick of each channel's last epoch (-1: none); the navigation bits the block decided, channel by
 * channel in epoch order: bit_rows / bit_cols / bit_values [up to sum of done], *n_bits of them.  SDR_ERR_RANGE when an epoch
 * would fall beyond max_ticks. */
int sdr_block_schedule(const sdr_track_epoch* records, int n_ch, int n_cols, const int32_t* done, const int64_t* unread_now,
                       int64_t samples_per_tick, const int64_t* flags0, const int64_t* code_since0, int max_ticks,
                       int32_t* first, int32_t* n_ticks, int32_t* order_rows, int32_t* order_cols, int32_t* starts,
                       sdr_track_epoch* records_sorted, sdr_track_epoch* last_records, int64_t* unread, int64_t* dev_flags,
                       int64_t* code_count, int32_t* last_tick, int32_t* bit_rows, int32_t* bit_cols, int32_t* bit_values,
                       int32_t* n_bits);

/* ------------------------------------------------- streams (one per channel batch)
 * north_star: "one HIP stream per channel batch".  Stream ids are small positive integers owned by the engine;
 * 0 always names the engine's default stream. */
int sdr_stream_create(sdr_engine* e, int* stream_id);
int sdr_stream_sync(sdr_engine* e, int stream_id);
/* sdr_epl_plan_run_range on a chosen stream (asynchronous; sdr_stream_sync or sdr_epl_plan_fetch completes it). */
int sdr_epl_plan_run_range_on(sdr_engine* e, sdr_epl_plan* p, int64_t first, int64_t count, int stream_id);
/* Diagnostics: which correlator variant the plan's items selected -- 0 per-sample, 8 / 16 boundary variant with that
 * many samples per lane, 26 chip-aligned; + KM when the block length is compiled in (every epoch KM.x samples per chip,
 * KM = 16 .. 25; 15 on the half-chip view) and with it + 256 * floor(KM / 2) when the outer taps' switch position is too
 * (three taps half a chip apart: both switch floor(KM / 2).x samples into the prompt tap's chip) or + 4096 when the taps sit
 * whole (half-)chips apart; 26 + 16 alone: two block lengths compiled in (every epoch 15.x or 16.x samples per chip:
 * 16.368 MHz); + 8192 * k: several chips per lane (k = 1, 2: two chips of 9.5 - 10 / 11.5 - 12 samples; k = 3: four of 3.75 - 4);
 * + 65536 when the plan runs on the half-chip view of its replicas (32-52 samples per chip: every chip twice).  The
 * results do not depend on it beyond the tolerance of the free arithmetic. */
int sdr_epl_plan_variant(const sdr_epl_plan* p);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* SYDR_AMD_H */
