// Internal declarations shared by the HIP translation units of libsydr_amd.so.
// gfx950 only; no other backend exists.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sydr_amd.h"

// Thread-local error text behind sdr_last_error().
void sdr_set_error(const char* fmt, ...);
int sdr_fail(int status, const char* fmt, ...);

#define SDR_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t _err = (call);                                                              \
        if (_err != hipSuccess)                                                                \
            return sdr_fail(SDR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), \
                            __FILE__, __LINE__);                                               \
    } while (0)

// Code LUT layout in HBM and LDS: lut[q + SDR_LUT_PAD] = chip[(q - 1) mod L] for
// q in [-SDR_LUT_PAD, L + SDR_LUT_PAD], i.e. the reference's padded table
// [c[L-1], c[0..L-1], c[0]] (channel_l1ca_kaplan.py:104-107) extended periodically.
#define SDR_LUT_PAD 4

struct ProfRecord {
    const char* name;
    hipEvent_t start, stop;
};

// Grow-only device buffer.
struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// What one HIP stream of the engine needs for itself so that launches on different streams (one per channel
// batch) never share a scratch buffer.
struct StreamCtx {
    hipStream_t stream = nullptr;
    uint64_t iq_waited_seq = 0;  // the engine's ring writes this stream has been ordered behind (sdr_iq_order_reader)
    DevBuf traj, bits, xchg;     // closed-loop launch scratch: epoch records, [list][n_bits][done] + bits, cluster exchange lines
    void* xchg_tagged = nullptr; // the exchange-line buffer the two-launch ticks' tags refer to (zeroed when it changes)
    unsigned tick_seq = 0;       // sequence number in those tags
    unsigned ingest_launches = 0;// one-launch ticks that brought their slab along since the counters were zeroed
    unsigned done_seq = 0;       // what the channels of the last launch with results in page-locked memory raise their done words to
    void* pinned = nullptr;      // page-locked host staging for the small per-step results
    size_t pinned_bytes = 0;
};

struct sdr_engine {
    int device = 0;
    hipStream_t stream = nullptr;   // == ctx0.stream
    StreamCtx ctx0;                 // stream id 0
    std::vector<StreamCtx*> streams;  // ids 1..
    int64_t code_generation = 0;    // bumped whenever the code slots are re-allocated (live plans are then stale)

    // IQ ring
    void* iq = nullptr;
    int64_t iq_capacity = 0;  // samples
    // ci8 rings: a second image of the ring with the sign bit of every byte flipped (x + 128 as an unsigned byte) -- what
    // the straight-line E/P/L kernels build their doubles from (correlator_chip.h: one v_perm_b32 per component, no
    // v_xor per dword).  Allocated when such a kernel first runs; brought up to date from the ring's dirty range.
    int iq_fmt = SDR_FMT_CI8;

    // code slots: int8 chips, row stride = code_stride bytes, plus per-slot length
    int8_t* codes = nullptr;  // [n_slots][code_stride] raw chips (+-1), no padding
    int32_t* code_len = nullptr;  // device [n_slots]
    std::vector<int32_t> code_len_host;
    std::vector<int64_t> code_stamp;      // per slot: value of code_stamp_counter when the slot was last staged
    int64_t code_stamp_counter = 0;
    std::vector<int64_t> pcps_spec_key;   // what pcps_code currently holds: [N, fs bits, (slot, stamp) ...]
    int n_slots = 0;
    int code_stride = 0;
    // per slot: the replica as the kernels want it in LDS -- uint32 high words of +-1.0 with the periodic
    // padding already applied (lut[q] = chip[(q - SDR_LUT_PAD - 1) mod L]); copied with 16-byte loads
    uint32_t* luts = nullptr;  // [n_slots][lut_stride]
    int lut_stride = 0;
    // the same replicas with every chip twice (lut2[h + PAD] = lut[ceil(h / 2) + PAD]): the "half-chip view" E/P/L
    // plans of 32-52 samples per chip run on (epl.hip); built on demand, rebuilt when a slot is re-staged
    uint32_t* luts2 = nullptr;  // [n_slots][lut2_stride]
    int lut2_stride = 0;
    int64_t luts2_generation = -1, luts2_stamp = -1;

    // workspaces
    DevBuf ws_items, ws_out, ws_spacing, ws_setups, ws_stats;
    // Device buffers of destroyed E/P/L plans, kept for the next plan (epl.hip plan_take / plan_give): a stream correlated
    // segment by segment makes and drops a plan per segment, and hipMalloc / hipFree cost more than the segment's setups --
    // hipFree also waits for the whole device, i.e. for the segment still running on another stream.
    std::vector<DevBuf> plan_pool;
    DevBuf pcps_fwd, pcps_a, pcps_b, pcps_code, pcps_tw, pcps_map, pcps_csum, pcps_part, pcps_res;
    DevBuf pcps_code2;            // N = 50 000, fused search: [prn][parity][N] -- the spectra and their image with the odd half's twiddle (pcps_fused.h)
    bool pcps_code2_ok = false;   // ... made from what pcps_code holds now
    bool pcps_shared = false;     // the search in flight read shared spectra (its second sweep reads them the same way)
    const long long* pcps_shared_off = nullptr;
    DevBuf pcps_spec_off;         // shared spectra: every bin's element offset into the padded class spectra (pcps.hip)
    std::vector<int64_t> pcps_spec_off_key;
    bool pcps_no_shared_spectra = false;   // "pcps_no_shared_spectra": one forward transform per bin, as before round 5
    DevBuf pcps_tickets;          // one word per PRN: the second sweep's last workgroup of a PRN divides the two peaks (pcps_fused.h)
    int pcps_tickets_n = 0;
    DevBuf track_state, track_cfg;
    int n_cus = 0;              // compute units of the device (sizes the closed-loop clusters)
    int track_force_parts = 0;  // diagnostics / tests: 0 = choose, else 1, 2, 4 or 8 workgroups per channel
    void* pcps_res_direct = nullptr;  // during sdr_pcps: page-locked block the peak kernels write their results into
    const void* pcps_slots_pinned = nullptr;   // ... the caller's slot numbers in it (copied to the device when spectra have to be made)
    unsigned* pcps_done = nullptr;    // ... one word per PRN the fused second sweep raises to pcps_done_seq behind the PRN's results
    unsigned pcps_done_seq = 0, pcps_done_counter = 0;
    bool pcps_done_used = false;      // the search ended in a kernel that raises them
    void* slab_pinned = nullptr;  // page-locked staging of the slab a receiver tick brings (sdr_bank_tick)
    size_t slab_bytes = 0;        // bytes of ONE of its two halves
    int slab_flip = 0;
    hipEvent_t slab_done[2] = {nullptr, nullptr};   // recorded behind the transfer that reads a half: a half is reused only
    bool slab_busy[2] = {false, false};             // after its event has completed (any number of slabs may be outstanding)
    int64_t pcps_tw_n = 0;
    // chirp-z (Bluestein) plan for code lengths the mixed-radix planner cannot factor: [chirp N][B_fwd M][B_inv M][tw M]
    DevBuf pcps_blu, pcps_blu_x, pcps_blu_a, pcps_blu_b;
    int64_t pcps_blu_n = 0;
    bool epl_no_chip = false;        // diagnostics: keep the 16-sample boundary variant where the chip-aligned one would run
    bool epl_no_double = false;      // diagnostics: keep the boundary variant where the half-chip view would run
    bool prof_calls_only = false;    // sdr_prof_enable(e, 2): only the "call_*" scopes record
    bool epl_no_chip2 = false;       // diagnostics: keep the 8-sample boundary variant where the two-chip kernel would run
    bool epl_no_split = false;       // diagnostics: keep the run-time switch positions where the KS = 12 kernel would run
    int pcps_prn_chunk = 0;          // diagnostics: PRNs per inverse sweep (0 = as many as the work buffers hold)
    bool pcps_force_map = false;     // diagnostics / tests: materialise the map even when the caller does not ask for it
    bool pcps_no_overlap = false;    // diagnostics: one stream for the inverse sweeps of a map-free search
    hipStream_t pcps_aux = nullptr;  // second stream of the map-free search (odd sweeps), its two ordering events
    hipEvent_t pcps_ev[2] = {nullptr, nullptr};
    bool pcps_no_fast = false;       // diagnostics: keep the general four-step kernels where the N = 125 x 200 ones would run
    bool track_one_launch_tick = false;  // "track_one_launch_tick": a one-epoch step as one workgroup per channel in one launch
    bool track_two_launch_tick = false;  // "track_two_launch_tick": the cluster's one-epoch step as two launches cut at the exchange (A/B)
    // "tick_server": the steady receiver tick served by a resident kernel (track.hip).  srv_running: one is resident NOW --
    // every call on the engine but the tick's own stops it first (sdr_set_device); a slab handed over by sdr_iq_upload_begin
    // while it runs waits in its staging half for the next request (srv_slab_*).
    bool tick_server_opt = false;
    bool srv_running = false;
    int srv_steady_ticks = 0;     // steady ticks in a row with no other call on the engine in between (a server starts at 8)
    struct TickServerState* srv = nullptr;
    // Writes into the ring are queued on `stream`; a reader launched on ANOTHER stream of the engine (one stream per channel
    // batch) is ordered behind them by an event: iq_write_seq counts the writes queued (sdr_iq_mark_written), iq_written is
    // recorded on `stream` behind write number iq_recorded_seq, a stream waits for it once (StreamCtx::iq_waited_seq).
    uint64_t iq_write_seq = 0, iq_recorded_seq = 0;
    hipEvent_t iq_written = nullptr;
    bool inplace_slab_in_flight = false; // an ingest kernel on `stream` may still be reading a slab out of the CALLER's page-locked block
    bool srv_slab_pending = false;       // ... or, without a server, for the tick's own launch (ingest_with_tick): whoever needs the ring
                                         // first flushes it the ordinary way (sdr_set_device, the tick itself)
    bool ingest_with_tick = true;        // "ingest_with_tick": a receiver tick's slab is pulled into the ring by workgroups of the tick's
                                         // own launch instead of a launch of its own in front of it
    bool last_tick_took_slab = false;    // the previous tick's launch was of that form: the next slab waits for the next tick's
    int srv_slab_half = 0;               // (-1: the slab lies in the caller's page-locked memory -- sdr_host_alloc -- and is read in place)
    const void* srv_slab_src = nullptr;  // where it lies: a staging half or the caller's block
    std::vector<std::pair<const char*, size_t>> host_blocks;   // what sdr_host_alloc has handed out
    int64_t srv_slab_off = 0, srv_slab_n = 0;
    bool ingest_by_copy = false;     // "ingest_by_copy_command": queued slabs go into the ring by hipMemcpyAsync, not by the ingest kernel
    bool pcps_no_spec_cache = false; // "pcps_no_spectra_cache": conj(fft(code)) recomputed by every search, as the reference does (kaplan:184-185)
    bool pcps_force_passes = false;  // diagnostics: use the one-kernel-per-radix-pass transform instead of the four-step one
    bool pcps_slow_second = false;   // "pcps_general_second_sweep": peak kernel + general four-step pair where the fused second sweep would run
    bool pcps_fused = true;          // map-free search at N = 125 x 200 of a round of 256 transforms or more: one workgroup per (PRN, bin) transform (pcps_fused.h); "pcps_fused" = 0: the two-kernel sweeps
    DevBuf pcps_work;                // its work list (transform numbers in processing order)
    int pcps_work_prn = 0, pcps_work_bins = 0, pcps_work_block = 0;   // ... and the grid / block shape it was made for
    int pcps_work_first[9] = {0};
    DevBuf pcps_theta;               // its two per-PRN bound arrays (pcps_fused.h Args::theta), the PRN count they are for, the one in use
    int pcps_theta_prn = 0, pcps_theta_flip = 0;

    // profiling
    bool prof = false;
    std::vector<ProfRecord> prof_records;
    std::vector<hipEvent_t> prof_pool;
};

int sdr_devbuf_reserve(sdr_engine* e, DevBuf* b, size_t bytes);
// Same for a buffer only ever used on `stream` (waits for that stream alone before a re-allocation).
int sdr_devbuf_reserve_on(sdr_engine* e, hipStream_t stream, DevBuf* b, size_t bytes);
int sdr_pinned_reserve(sdr_engine* e, StreamCtx* ctx, size_t bytes);
StreamCtx* sdr_stream_ctx(sdr_engine* e, int stream_id);   // nullptr when the id does not exist
// Queue the ring write of sdr_iq_upload on the engine's stream without waiting for it.
int sdr_iq_upload_async(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset);
// hipSetDevice WITHOUT stopping a resident tick server (the tick's own entry points); sdr_set_device stops it.
int sdr_set_device_keep(sdr_engine* e);
// track.hip: tell the resident tick server to leave and wait for it (bounded); a slab it had not pulled yet goes into the ring
// the ordinary way.  No-op when none runs.
int sdr_tick_server_stop(sdr_engine* e);
void sdr_tick_server_free(sdr_engine* e);
// engine.hip: the slab waiting in a staging half (srv_slab_*) into the ring by the ingest kernel, on the engine's stream
int sdr_iq_flush_server_slab(sdr_engine* e);
// Ring samples [offset, offset + n) (modulo capacity) are being written on e->stream: the flipped image is stale there.
void sdr_iq_mark_written(sdr_engine* e, int64_t ring_offset, int64_t n_samples);
// The flipped image of a ci8 ring, up to date with everything queued on e->stream, usable from `stream`.
int sdr_iq_flipped(sdr_engine* e, hipStream_t stream, const void** out);
// A launch on ctx's stream that reads the ring: behind every ring write queued on the engine's own stream so far.
int sdr_iq_order_reader(sdr_engine* e, StreamCtx* ctx);

static inline size_t sdr_fmt_bytes(int fmt) {
    switch (fmt) {
        case SDR_FMT_CI8: return 2;
        case SDR_FMT_CI16: return 4;
        case SDR_FMT_CF32: return 8;
        case SDR_FMT_CF64: return 16;
    }
    return 0;
}

// RAII bracket: records hipEvents around a launch when profiling is on.
struct ProfScope {
    sdr_engine* e;
    ProfRecord rec;
    bool active;
    hipStream_t stream;
    ProfScope(sdr_engine* eng, const char* name, hipStream_t on = nullptr);
    ~ProfScope();
};

int sdr_set_device(sdr_engine* e);

// epl_straight.hip: the straight-line E/P/L kernels (ci8, three taps) of the block lengths epl.hip does not instantiate itself;
// nullptr for a length there is none for.  Arguments: those of epl_kernel (epl_kernel.h).
const void* sdr_epl_ks_kernel(int km);   // outer taps switching floor(KM / 2).x samples into the block
const void* sdr_epl_ki_kernel(int km);   // three taps whole (half-)chips apart
const void* sdr_epl_ki5_kernel(int km);  // five taps whole (half-)chips apart
const void* sdr_epl_km_kernel(int km);   // three taps, the block length alone compiled in (tap positions at run time)

// pcps_fused.hip: every (PRN, bin) inverse transform of a map-free search at N = 25 000 in one launch; leaves
// SDR_PCPS_FUSED_RECORDS (value, index) records per transform in `partials` ([transform][record], 16 bytes each).
// N = 50 000 (terms = 2): a unit is one parity of a transform -- twice the records -- and C is the [prn][parity][N] image.
#define SDR_PCPS_FUSED_RECORDS 8
// spec_off (nullable): element offset of every bin's spectrum inside F (shared spectra: pcps.hip); else bin b is at b * N.
int sdr_pcps_fused_sweep(sdr_engine* e, const void* F, const void* spec_off, const void* C, const void* tw, int n_prn, int nbins, int N,
                         void* partials);
// records per PRN that sweep leaves (PRN-major: [prn][records]); a search that is not a whole number of rounds of its 256
// workgroups has its last transforms cut into five units of SDR_PCPS_FUSED_RECORDS records each
int sdr_pcps_fused_records_per_prn(int n_prn, int nbins, int terms);
// pcps_fused10k.h: a search at N = 10 000 that wants indices and ratio only (coh = 1, any number of non-coherent blocks): one
// workgroup per (PRN, bin) keeps the transform in its LDS and the non-coherent sum in registers; F_all = [noncoh][nbins][N]
// forward spectra, records = n_prn * nbins * SDR_PCPS_FUSED10K_RECORD_BYTES bytes of scratch; the results go to out_*.
#define SDR_PCPS_FUSED10K_RECORD_BYTES 32
// (spec_off / blk_stride: shared spectra, as above; nullptr / nbins * N: one spectrum per bin)
int sdr_pcps_fused10k_search(sdr_engine* e, const void* F_all, const void* spec_off, long long blk_stride, const void* C, const void* tw, int n_prn,
                             int nbins, int noncoh, int N, int spc, void* records, void* out_bin, void* out_code, void* out_ratio);
// The second sweep of such a search in one launch: the first peaks from `recs` ([n_prn][per_prn] records) into tops / dev_bin /
// dev_code, and 5 x SDR_PCPS_FUSED_RECORDS records per PRN of its winning row's allowed columns into `seconds`.
// ... and TwoCorrelationPeakComparison's results (bin, code phase, ratio of the two peaks) into res_*.
int sdr_pcps_fused_second(sdr_engine* e, const void* F, const void* spec_off, const void* C, const void* tw, int n_prn, int N, int spc, const void* recs,
                          int per_prn, void* tops, void* dev_bin, void* dev_code, void* seconds, void* res_bin, void* res_code,
                          void* res_ratio);
