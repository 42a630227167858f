// On-device loop closure: persistent workgroups run
// correlate -> discriminators -> loop filters -> NCO update for n_epochs without leaving the
// GPU.  A channel is served by ONE workgroup (512 threads; or 256 threads capped at
// 168 registers so that three share a CU when there are more channels than CUs) or by a CLUSTER of 2, 4 or 8
// workgroups (chosen so that the launch fills the GPU: 32 channels x 8 parts = 256 CUs): every epoch each part
// correlates its share of the samples, publishes its partial sums through global memory, collects its peers' and
// adds all of them in the same fixed order -- so every part holds bit-identical totals and runs the scalar loop
// update redundantly; there is no second exchange.  The PRN replica stays in LDS for the whole run; the loop
// arithmetic is fp64, spread over four waves by dependency (carrier loop / code loop / lock indicators + state
// machine + bit decisions / carrier phase over the epoch) and, inside each, over lanes for the divisions, roots
// and arctangents -- following the two reference plugins statement by statement:
//   kind 0  Borre  : channel_l1ca_borre.py:333-451  (DLL NNEML + Costas PLL, Borre filters, np.pi NCO)
//   kind 1  Kaplan : channel_l1ca_kaplan.py:342-619 (FLL-assisted 2nd-order PLL, lock-state machine,
//                    GPS-ICD pi in the NCO and the discriminators: SURVEY.md T3)
// built on sydr/dsp/tracking.py:120-186,246-279 and sydr/dsp/lockindicator.py:6-122, generalised by configuration
// only (taps, chips per epoch, epochs per symbol, epoch duration: BASELINE configs 4-5).  The second half of the
// file is the host side: launch geometry, the sdr_track_closed_loop* entry points and the device-resident channel
// bank (sdr_bank_*).
#include "correlator.h"
#include "correlator_chip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef SDR_TRACE_TRACK
// Debug build only (tools/track_phases.py): per-phase clock totals of channel 0's epoch loop.
__device__ unsigned long long g_track_phase[64];   // [0,8) phases, [8,40) per-wave arrival (8 parts x 4 waves), [48,52) per-role, [62,64) clocks
extern "C" __attribute__((visibility("default"))) int sdr_debug_track_phases(unsigned long long* dst) {
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

constexpr int kMaxParts = 8;

// ------------------------------------------------------------------------------------------------ the tick server
// The receiver's per-millisecond loop (receiver.py:120-131: addNewRFData(1 ms); run()) pays a kernel launch and a stream
// synchronisation per tick: ~13 of a tick's ~29 us in the library (tools/ubench_pingpong.hip: a round trip host -> resident
// workgroup -> host through page-locked words takes 1.8 us, an empty launch + synchronisation 12.5).  With
// sdr_set_option("tick_server", 1) the cluster form of the kernel stays RESIDENT between ticks, and eight more workgroups, the
// doormen, watch two words in page-locked memory.  A SLAB (sdr_iq_upload_begin) is announced at once: every doorman pulls
// an eighth of it out of the staging block into the ring and says so in device memory -- while the host is still on its way to
// the tick.  A REQUEST (sdr_bank_tick_mirrored) carries the ring's write index and the slab it needs; the first doorman waits
// for the eight shares and releases the channels, each through a word of its own; every channel whose next epoch is complete
// runs it exactly as a block launch would (same cluster, same order of additions: the bits of a plain tick) and ANSWERS THE
// HOST ITSELF: one wave writes state and record into page-locked memory and, behind a system-wide release, the request's
// number into the channel's done word; the host spins on those words.  The doorman only counts the answers (its clock bounds
// them) and stamps the request.
// Nothing waits without a bound: the doormen give up after `idle_ticks` without a request (the host starts a new server
// with the next tick), a tracker after twice that without a release, the doorman after `busy_ticks` without the channels'
// answers (fault), the exchange as ever after kSpinLimit polls; the host waits a bounded time for the done words and falls
// back to plain launches.  Every other call on the engine stops the server first (sdr_set_device).
constexpr unsigned kServerStop = 0xFFFFFFFFu;
constexpr int kGoStride = 32;      // unsigned words between two channels' release words: a 128-byte line each
// The request line: 64 bytes, 64-byte aligned, read by the doormen in ONE access (eight lanes x 8 bytes).  The host writes
// words 1-5, then word 7, then word 0 (both hold the same numbers); a reader takes the line when words 0 and 7 agree -- each
// 32-byte half of the access is then at least as new as its number, should the access ever be split.
enum { kReqSeqs = 0, kReqWriteIndex = 1, kReqSlabSrc16 = 2, kReqSlabN16 = 3, kReqSlabFirst16 = 4, kReqNeedPull = 5, kReqSeqsCopy = 7 };
struct TickServerHost {            // page-locked: the words the host and the doormen share
    // [0] / [7]: request number (low half: 1, 2, 3, ...; kServerStop = leave) and slab number (high half: 1, 2, 3, ...: "pull
    // this slab into the ring"); [1] the ring's write index after the slab; [2] where the slab lies in page-locked memory
    // (its address / 16: a staging half of the engine's or the caller's own block), [3] its granules, [4] its first ring granule; [5] the slab the request needs in the ring before the channels
    // are released (0: none)
    unsigned long long line[8];
    unsigned done_seq;             // doorman: the last request it has seen answered by every channel (stamps below are that request's)
    unsigned alive;                // doorman: 1 while the server runs
    unsigned fault;                // doorman: why it gave up (1 idle, 2 a channel never answered, 3 exchange fault)
    unsigned pad_;
    // doorman: wall-clock stamps (100 MHz) of the last request -- seen, slab in the ring, channels released, all channels
    // answered, stamps written (sdr_tick_server_phases: where a served tick's time goes on the device)
    unsigned long long stamps[6];
    unsigned long long tracker[12]; // ... and channel 0's own: release seen, samples visible, correlated, exchanged, updated, answered
};
struct TickServerDev {             // device memory: the words the doormen and the trackers share
    unsigned done_count;           // trackers: channels that have answered, all requests together (the doorman's watch on them)
    unsigned stop;                 // doorman: it has left (the helpers follow)
    unsigned pulled[8];            // doormen: the last slab whose share each has put into the ring
    long long write_index;
    int fault;                     // a cluster exchange timed out
    unsigned long long t[12];      // channel 0, part 0, lane 0: wall-clock stamps of its last tick (sdr_tick_server_phases); [8..11]: trace build
#ifdef SDR_SRV_TRACE
    unsigned long long seen_at;          // the doorman: when it saw the current request
    unsigned long long ch_gate[64], ch_done[64], ch_n[64];   // per channel, summed over requests: request seen -> release seen / answered (recording part)
    unsigned ch_where[64];               // ... and where that part runs (XCC_ID << 16 | HW_ID's CU bits)
    unsigned door_where[8];              // ... and the doormen
#endif
};
struct TickServer {
    TickServerHost* host;          // nullptr: not a server launch
    TickServerDev* dev;
    unsigned* go;                  // device [n_ch * kGoStride]: channel c's release word (the request its cluster may work on; kServerStop: leave)
    uint4* ring16;
    unsigned long long ring_n16;
    unsigned ring_flip;            // what a slab's dwords are xor-ed with on their way into the ring (ci8: every sign bit, correlator.h kCi8Flip)
    sdr_track_epoch* rec_out;      // device [n_ch]: where the roles write the epoch's record
    // page-locked, written by the channels themselves: the answer (state, record, 1 ran / 0 not ready / -1 stopped), then -- behind a
    // system-wide release -- the number of the request it answers
    int* h_ran;
    sdr_track_state* h_st;
    sdr_track_epoch* h_rec;
    unsigned* h_done;
    unsigned long long idle_ticks, busy_ticks;   // of wall_clock64() (100 MHz)
    // NOT the server's (host == nullptr): a plain launch whose results go straight into page-locked memory has every channel
    // raise done_words[its position in the list] = done_seq behind them -- the host reads the results when the words are
    // there instead of waiting for the stream's signal, which follows the kernel's last store by ~9 us (bank_collect)
    unsigned* done_words;
    unsigned done_seq;
    // Likewise a plain launch's (the one-launch receiver tick): its first kTickIngestGroups workgroups pull the tick's slab out
    // of page-locked memory (address / 16 = ingest_src16, ingest_n16 granules) into the ring (`ring16`, from ingest_first16) and
    // count themselves in at *ingest_count; the trackers stage their tables and parameters meanwhile and read their first
    // sample when the count has reached ingest_target.  ingest_n16 = 0: no slab with this launch.
    unsigned long long ingest_src16, ingest_n16, ingest_first16;
    unsigned* ingest_count;
    unsigned ingest_target;
};
// In front of the exchange lines (one-launch ticks): 64 ticket counters (a cluster of two parts or more means 64 channels at
// most), then the ingest workgroups' counter.
constexpr int kXchgHeadBytes = 512;
constexpr int kTickIngestGroups = 16;    // (a multiple of 8: the trackers' blockIdx % 8 -- their XCD -- is what it was without them)

// The doormen: kDoorGroups workgroups, a launch of their own beside the trackers' (the cluster of 32 channels x 8 parts fills
// the cooperative launch's 256 workgroups; a doorman needs a few registers and no LDS to speak of, and shares a compute unit
// with one of them).
// (eight waves, two per SIMD, few registers: it has to fit beside a tracker workgroup that holds 256 registers per lane on
// every SIMD of its compute unit -- sixteen waves did not.  A 50 KB slab is seven 16-byte loads per lane, four in flight.)
constexpr int kDoorThreads = 512;
// One workgroup reads the host's memory at a few GB/s (a 50 KB slab took it 19 us): kDoorGroups workgroups pull an equal share
// each.  All of them watch the host's words; the first is the doorman proper (requests are its business alone), the others
// leave when it does (or by their own, longer, clock).
constexpr int kDoorGroups = 8;
static_assert(kDoorGroups == 8, "TickServerDev::pulled has eight words");
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long x, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __attribute__((unused)) void tick_server_doorman(const TickServer& s, const int n_ch, const int tid, unsigned* sh_words,
                                                            unsigned long long* sh_q, const int group) {
    unsigned served = 0, pulled = 0, requests = 0;
    unsigned quiet = 0;                // long sleeps before the next look at the host's line (see below)
    unsigned long long t_last = wall_clock64();
    unsigned why = 0;
    const bool helper = group != 0;
    if (tid == 0 && !helper) __hip_atomic_store(&s.host->alive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef SDR_SRV_TRACE
    if (tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        s.dev->door_where[group] = (xcc << 16) | (hw & 0xffff);
    }
#endif
    for (;;) {
        if (tid < 64) {
            // A compute unit returns its waves' loads IN ORDER: while a doorman's look at the host's line is on its way over the
            // link (~1.5 us), every load of the tracker workgroup it shares the compute unit with waits behind it -- measured
            // per channel (-DSDR_SRV_TRACE): the channels with a part beside a doorman answered 5-7 us after the others, and the
            // tick is as long as its last channel.  So the doormen keep QUIET while the channels work: a helper for ~12 us after
            // its share of a slab (the request it came with takes longer than that), the doorman proper between the release
            // and the time the first answers are due (its count of the answers is not on the host's path any more).
            for (unsigned k = 0; k < quiet; ++k) __builtin_amdgcn_s_sleep(127);      // (8128 clocks each)
            quiet = 0;
            // (one look per turn of the loop, a short sleep between turns: eight workgroups reading the host's line back to back
            // slowed the trackers' own traffic -- the channels' answers took 14.9 instead of 12.0 us.  The whole request comes
            // with the look: eight lanes, 8 bytes each, one access over the link -- no second round trip for its words)
            unsigned long long w = 0;
            if (tid < 8) w = __hip_atomic_load(&s.host->line[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long w0 = lane_u64(w, kReqSeqs), w7 = lane_u64(w, kReqSeqsCopy);
            const unsigned long long q_wi = lane_u64(w, kReqWriteIndex), q_src = lane_u64(w, kReqSlabSrc16), q_n = lane_u64(w, kReqSlabN16),
                                     q_first = lane_u64(w, kReqSlabFirst16), q_need = lane_u64(w, kReqNeedPull);
            if (tid == 0) {
                const bool whole = w0 == w7;
                const unsigned seq = whole ? (unsigned)w0 : served, pseq = whole ? (unsigned)(w0 >> 32) : pulled;
                unsigned act = 0;                            // 1: a slab to pull, 2: a request (doorman proper), 4: leave
                if ((unsigned)w0 == kServerStop || (unsigned)w7 == kServerStop) {
                    act = 4;
                } else {
                    if (pseq != pulled) act |= 1;
                    if (!helper && seq != served) act |= 2;
                }
                if (!act && wall_clock64() - t_last > (helper ? 2 * s.idle_ticks : s.idle_ticks)) act = 4, sh_words[1] = 1;
                if (helper && !act && __hip_atomic_load(&s.dev->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) act = 4;   // (the doorman has left)
                if (act & 3) sh_q[0] = q_wi, sh_q[1] = q_n, sh_q[2] = q_src, sh_q[3] = q_first, sh_words[6] = (unsigned)q_need;
                sh_words[0] = act, sh_words[4] = seq, sh_words[5] = pseq;
            }
        }
        __syncthreads();
        const unsigned act = sh_words[0], seq = sh_words[4], pseq = sh_words[5], need_pull = sh_words[6];
        const long long wi = (long long)sh_q[0];
        const unsigned long long n16 = sh_q[1], src16 = sh_q[2], first16 = sh_q[3];
        __syncthreads();
        if (!act) {
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        if (act & 4) {
            why = sh_words[1];
            break;
        }
        unsigned long long stamp[6];
        stamp[0] = wall_clock64();
        if (act & 1) {
            // (the staging block is the host's memory, written since this compute unit last read it: nothing of it may come out
            // of a cache -- one invalidation by the first wave, the barrier hands it on)
            if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            __syncthreads();
            // this workgroup's share of the slab: granules [lo, hi)
            const unsigned long long lo = n16 * (unsigned long long)group / kDoorGroups, hi = n16 * (unsigned long long)(group + 1) / kDoorGroups;
            const uint4* const src = reinterpret_cast<const uint4*>(static_cast<uintptr_t>(src16) << 4);   // (an address in the host's memory / 16)
            for (unsigned long long i0 = lo + tid; i0 < hi; i0 += 4 * kDoorThreads) {     // four loads per lane in flight, then their stores
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + (unsigned long long)k * kDoorThreads < hi) v[k] = src[i0 + (unsigned long long)k * kDoorThreads];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long i = i0 + (unsigned long long)k * kDoorThreads;
                    if (i < hi) {
                        unsigned long long d = first16 + i;
                        if (d >= s.ring_n16) d -= s.ring_n16;
                        uint4 o = v[k];
                        o.x ^= s.ring_flip, o.y ^= s.ring_flip, o.z ^= s.ring_flip, o.w ^= s.ring_flip;
                        s.ring16[d] = o;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every wave: its stores have left (the barrier orders them ...)
            __syncthreads();                                          // ... before lane 0's device-wide release below)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(&s.dev->pulled[group], pseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pulled = pseq;
            t_last = wall_clock64();
            if (helper) quiet = 3;
        }
        if (!(act & 2)) continue;
        // ---- a request (the doorman proper)
        ++requests;
        if (tid == 0) {
            s.dev->write_index = wi;
#ifdef SDR_SRV_TRACE
            s.dev->seen_at = stamp[0];
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            // the eight shares of the slab it needs (bounded: a helper that never shows up is a fault like a channel that never answers)
            unsigned ok_pull = 1;
            if (need_pull) {
                const unsigned long long t0 = wall_clock64();
                for (int g = 0; g < kDoorGroups && ok_pull; ++g) {
                    while ((int)(__hip_atomic_load(&s.dev->pulled[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need_pull) < 0) {
                        if (wall_clock64() - t0 > s.busy_ticks) {
                            ok_pull = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            sh_words[3] = ok_pull;
            stamp[1] = wall_clock64();
        }
        __syncthreads();
        // the release: every channel's word (n_ch <= 64: one store of the first wave)
        if (sh_words[3] && tid < n_ch) __hip_atomic_store(&s.go[(size_t)tid * kGoStride], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            stamp[2] = wall_clock64();
            const unsigned target = requests * (unsigned)n_ch;
            const unsigned long long t0 = wall_clock64();
            unsigned ok = 1;
            __builtin_amdgcn_s_sleep(127);       // (quiet while the channels work: see the top of the loop)
            __builtin_amdgcn_s_sleep(127);
#ifdef SDR_SRV_TRACE
            unsigned long long seen[4] = {0, 0, 0, 0};      // first sight of 1, n/2, n - 1, n answers
#endif
            for (;;) {
                const unsigned c = __hip_atomic_load(&s.dev->done_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SDR_SRV_TRACE
                const unsigned have = c - (target - (unsigned)n_ch);
                const unsigned long long now = wall_clock64();
                if (have >= 1 && !seen[0]) seen[0] = now;
                if (have >= (unsigned)n_ch / 2 && !seen[1]) seen[1] = now;
                if (have >= (unsigned)n_ch - 1 && !seen[2]) seen[2] = now;
                if (have >= (unsigned)n_ch && !seen[3]) seen[3] = now;
#endif
                if (c == target) break;
                if (wall_clock64() - t0 > s.busy_ticks) {
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
#ifdef SDR_SRV_TRACE
            for (int k = 0; k < 4; ++k) s.dev->t[8 + k] = seen[k] - t0;
#endif
            sh_words[2] = ok && sh_words[3];
            stamp[3] = wall_clock64();
            if (sh_words[2]) {
                // (the channels have answered the host themselves; what is left is the request's bookkeeping)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stamp[4] = wall_clock64();
                for (int k = 0; k < 5; ++k) s.host->stamps[k] = stamp[k];
                for (int k = 0; k < 12; ++k) s.host->tracker[k] = s.dev->t[k];
                if (__hip_atomic_load(&s.dev->fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    __hip_atomic_store(&s.host->fault, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                __hip_atomic_store(&s.host->done_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        if (!sh_words[2]) {
            why = 2;
            break;
        }
        served = seq;
        t_last = wall_clock64();
    }
    if (!helper) {
        // the channels (their release words) and the helpers (the stop word) follow; then the host is told
        if (tid < n_ch) __hip_atomic_store(&s.go[(size_t)tid * kGoStride], kServerStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&s.dev->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (why) __hip_atomic_store(&s.host->fault, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&s.host->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#ifndef SDR_TRACK_DENSE_TU
__global__ __launch_bounds__(kDoorThreads) void tick_doorman_kernel(const TickServer srv, int n_ch) {
    __shared__ unsigned words[8];
    __shared__ unsigned long long q[4];
    if (threadIdx.x < 8) words[threadIdx.x] = 0;
    if (threadIdx.x < 4) q[threadIdx.x] = 0;
    __syncthreads();
    tick_server_doorman(srv, n_ch, (int)threadIdx.x, words, q, (int)blockIdx.x);
}
#endif
// Exchange line of one part and parity: 4*NT tagged half-values padded to whole 128-byte lines (16 words for E/P/L,
// 32 for five taps).
constexpr int xchg_words(int nt) { return 4 * nt <= 16 ? 16 : 32; }
constexpr int kXchgWordsMax = 32;
constexpr long kSpinLimit = 1L << 20;      // peer polls before a part gives up (about a second): never hang the GPU
#ifndef SDR_XCHG_SLEEP
#define SDR_XCHG_SLEEP 16
#endif
constexpr int kXchgSleep = SDR_XCHG_SLEEP;             // x 64 cycles between publishing a part's sums and the first look at the peers'

constexpr double kGpsPi = 3.1415926535898;  // sydr/utils/constants.py:4
constexpr double kGpsTwoPi = kGpsPi * 2.0;
constexpr double kGpsHalfPi = kGpsPi / 2.0;
constexpr double kDefaultEpochChips = 1023.0;  // GPS_L1CA_CODE_SIZE_BITS (kaplan:529-532)
constexpr int kDefaultEpochsPerBit = 20;       // LNAV_MS_PER_BIT
constexpr double kDefaultEpochSeconds = 1e-3;  // the dt the Kaplan plugin hard-codes (kaplan:417,425,443,494)
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

// The lock role's share of the loop state, held in registers of its lane 0 for the whole run (19 LDS reads and as many
// writes per epoch otherwise -- and this role is the longest of the four: tools/track_phases.py).  Written back to
// the LDS copy of the state once, after the last epoch.
struct LockRegs {
    double fll_lock, pll_lock, cn0, ratio_acc, ipp, qpp, fll_bw, pll_bw, nav_sum;
    int accum, lock_state, time_in_state, spacing_sel, flags, code_counter, nav_count, bits_emitted, bits_run;
};
// The run's constants the update roles divide and scale with.  The forms with registers to spare keep them in registers
// (UpdateCtx); the dense form (three workgroups per compute unit, 168 registers) reads them from LDS when a role needs them.
struct UpdateConsts {
    double fs, chips, dt;
    double by_fs_b, by_fs_y, by_dt_b, by_dt_y, by_2pi_b, by_2pi_y;
    int epochs_per_bit, ok_bits;       // ok_bits: InvDen::ok of by_fs | by_dt << 1 | by_2pi << 2
};

struct alignas(16) EpochShared {  // (size a multiple of 16: the replica behind it is copied with 16-byte stores)
    EpochParams ep;
    double spacing[SDR_MAX_TAPS];
    double dphi;
    int epochs_done;
    int fault;                 // a peer part never showed up: leave the epoch loop (reported to the host)
    // what the update roles hand to each other (written before an epoch's first barrier, read after it)
    double corr[2 * SDR_MAX_TAPS];  // this epoch's correlator totals (one-workgroup kernels: for the roles on waves 1 and 2)
    double fll_bw, pll_bw;     // Kaplan bandwidths chosen by the lock-state machine, for the carrier loop
    int lock_state;            // lock state the NEXT epoch's discriminators run under
    int c_code_counter, l_code_counter, l_bits_run;  // private copies of the roles on waves 0 and 2
    int stop_code, stop_carrier;  // the next epoch would leave the staged replica / the ring, or the carrier NCO is not finite
    double smin, smax;         // extreme tap offsets over both tap sets (constant for the run)
    double l_ipp, l_qpp;       // the lock role's copy of the previous prompt (the carrier role owns st.i/q_prompt_prev)
    // quotients that only change with the configuration or the lock state, kept instead of re-divided every epoch
    // (same operands => same bits): atan(qP'/iP') of the previous epoch, the Kaplan natural frequencies for the
    // bandwidths they were computed from, the Borre filter ratios
    double at_prev, w0f, w0p, w0f_bw, w0p_bw, pll_r1, pll_r2, dll_r1, dll_r2;
    unsigned gate;             // (tick server) the release the workgroup's lane 0 saw at the top of the tick
    unsigned pad_sh_;          // (keeps the struct a multiple of 16 bytes)
    long long gate_wi;         // ... and the request's write index
    long long pad_sh2_;
    sdr_track_state st;        // the loop state; each update role owns a disjoint set of its fields
    sdr_loop_cfg cfg;
    // the dense form: what the other forms carry in registers across the whole epoch -- the lock role's state and the run's
    // constants.  At 168 registers per lane the compiler spilled them to scratch memory and the roles fetched them back one
    // dependent load at a time, on the epoch's critical path (76 scratch loads behind the reduction barrier, round 5).
    LockRegs lk;
    UpdateConsts uc;
    long long pad_dense_;
};
static_assert(sizeof(EpochShared) % 16 == 0, "the replica behind it is copied with 16-byte stores");

// a / b for a denominator that does not change during the run (fs, 2*pi, the epoch duration), given y = RN(1/b):
// two Newton corrections of the quotient with exact FMA residuals.  The second one rounds correctly (Markstein's
// theorem: y within half an ulp of 1/b and q within one ulp of a/b => RN(q + r*y) = RN(a/b), b's significand not
// all ones) -- the SAME bits as the reference's division, on a dependent chain of 5 operations instead of the ~12
// of v_div_scale / v_rcp / v_fma... / v_div_fmas / v_div_fixup.  tests/test_div_by_constant.py checks the identity
// on 10^8 operands per denominator.  ok == false (significand all ones, never the case for a sampling rate):
// plain division.
struct InvDen {
    double b, y;
    bool ok;
};
__device__ __forceinline__ InvDen inv_den(double b) {
    InvDen d;
    d.b = b;
    d.y = 1.0 / b;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
    d.ok = (bits & 0xFFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFFull && b == b && fabs(b) > 1e-290 && fabs(b) < 1e290;
    return d;
}
__device__ __forceinline__ double div_by(double a, const InvDen& d) {
    if (!d.ok) return a / d.b;
    const double q0 = a * d.y;
    const double r0 = __builtin_fma(-d.b, q0, a);
    const double q1 = __builtin_fma(r0, d.y, q0);
    const double r1 = __builtin_fma(-d.b, q1, a);
    return __builtin_fma(r1, d.y, q1);
}

__device__ __forceinline__ LockRegs lock_regs_from(const sdr_track_state& s) {
    LockRegs r;
    r.fll_lock = s.fll_lock, r.pll_lock = s.pll_lock, r.cn0 = s.cn0, r.ratio_acc = s.cn0_ratio_acc;
    r.ipp = s.i_prompt_prev, r.qpp = s.q_prompt_prev, r.fll_bw = s.fll_bw, r.pll_bw = s.pll_bw, r.nav_sum = s.nav_prompt_sum;
    r.accum = s.accum_counter, r.lock_state = s.lock_state, r.time_in_state = s.time_in_state;
    r.spacing_sel = s.spacing_sel, r.flags = s.track_flags, r.code_counter = s.code_counter;
    r.nav_count = s.nav_sum_counter, r.bits_emitted = s.nav_bits_emitted, r.bits_run = 0;
    return r;
}
__device__ __forceinline__ void lock_regs_store(const LockRegs& r, EpochShared* sh);

// What the update roles need besides the shared state: the run's constants and this epoch's inputs (all wave-uniform).
struct UpdateCtx {
    double fs, chips, dt;          // sampling rate, chips per epoch, epoch duration the Kaplan filters are scaled with
    int epochs_per_bit;
    InvDen by_fs, by_dt, by_2pi;   // 1/fs, 1/dt, 1/(GPS 2 pi)
    int64_t capacity;
    int lut_words;
    EpochParams ep;                // the epoch that has just been correlated
    double cur_fll_bw, cur_pll_bw; // the state machine's previous decision (captured at the top of the epoch)
    int cur_lock_state;
    int epoch, n_epochs, ch;
    bool writer;
    sdr_track_epoch* rec;          // this epoch's record (recording part only, trajectory requested) or nullptr
    int8_t* nav_bits;
    int max_bits;
};

__device__ __forceinline__ bool carrier_bad(double hz) { return !(hz == hz && fabs(hz) < 1e9); }

// The replica LUT and the ring bound what an epoch may touch; a loop that has run away (loss of lock) stops
// instead of reading out of range.  Checked by the role that produces the values, for the epoch it announces.
__device__ __forceinline__ bool code_out_of_range(const EpochShared* sh, int64_t capacity, int lut_words, int64_t start,
                                                  int n, double rem_code, double code_step) {
    const double lo = ceil(rem_code + sh->smin);
    const double hi = ceil(code_step * (double)n + rem_code + sh->smax);
    return !(n > 0 && (int64_t)n <= capacity && code_step > 0.0 && lo >= -(double)SDR_LUT_PAD &&
             hi <= (double)(lut_words - SDR_LUT_PAD - 2) && start >= 0);
}

// Loop update.  The reference's per-epoch sequence is scalar arithmetic whose cost is instruction LATENCY
// (~15 fp64 divisions, two square roots, two arctangents, a float modulo, ~1250 instructions when one lane
// does it all).  It splits into four chains that only meet through the previous epoch's results:
//   wave 0  carrier loop : FLL/PLL discriminators, carrier filter, carrier frequency        (kaplan:405-447,506-534)
//   wave 1  code loop    : DLL discriminator, code filter, code NCO, next epoch length      (kaplan:451-461,506-534)
//   wave 2  lock role    : lock indicators, C/N0, lock-state machine, flags, bit sync, nav bits (kaplan:465-619)
//   wave 3  carrier phase: remCarrier advanced over the epoch, modulo 2 pi -- depends on nothing this epoch
//                          measured (kaplan:523-524, borre:364-365)
// Each wave evaluates its divisions / roots / arctangents side by side, one per lane (same IEEE operations on
// the same operands as the reference's statements: bit-identical), then lane 0 runs the rest of its chain and
// publishes its share of the next epoch's parameters.  corr[2*NT]: this epoch's correlator totals.
template <int NT, bool DENSE = false>
__device__ __forceinline__ void loop_update(EpochShared* sh, const UpdateCtx& u, const double* corr, int role, int rlane,
                                            LockRegs& lk_regs) {
    constexpr int kTaps = NT;
    constexpr int kPrompt = NT / 2;                       // centre tap; its neighbours are early and late
    sdr_track_state& st = sh->st;
    const sdr_loop_cfg& cfg = sh->cfg;
    const EpochParams& ep = u.ep;
    double fs = u.fs, kChips = u.chips, kDt = u.dt;
    int kMsPerBit = u.epochs_per_bit;
    InvDen by_fs = u.by_fs, by_dt = u.by_dt, by_2pi = u.by_2pi;
    LockRegs lk_local;
    if constexpr (DENSE) {
        // (through a pointer the compiler cannot see through: read HERE, every epoch -- hoisted out of the epoch loop they
        // would be registers again, and spilled again)
        const UpdateConsts* c = &sh->uc;
        asm volatile("" : "+v"(c));
        fs = c->fs, kChips = c->chips, kDt = c->dt, kMsPerBit = c->epochs_per_bit;
        const int okb = c->ok_bits;
        by_fs = InvDen{c->by_fs_b, c->by_fs_y, (okb & 1) != 0};
        by_dt = InvDen{c->by_dt_b, c->by_dt_y, (okb & 2) != 0};
        by_2pi = InvDen{c->by_2pi_b, c->by_2pi_y, (okb & 4) != 0};
        if (role == 2 && rlane == 0) lk_local = sh->lk;
    }
    LockRegs& lk = DENSE ? lk_local : lk_regs;
    const double ie = corr[2 * kPrompt - 2], qe = corr[2 * kPrompt - 1], ip = corr[2 * kPrompt], qp = corr[2 * kPrompt + 1],
                 il = corr[2 * kPrompt + 2], ql = corr[2 * kPrompt + 3];
    const int n = ep.n;
    const bool kaplan = cfg.loop_kind != 0;
    sdr_track_epoch* rec = u.rec;
    if (role == 0) {
        // ------------------------------------------------------------------ carrier loop
        // (every lane evaluates the same chain on the same wave-uniform operands: no lane specialisation, no
        // v_readlane -- a lone wave pays per instruction, not per lane)
        const double at_now = atan(qp / ip);                      // atan(qP/iP): Costas PLL, FLL (tracking.py:133-176)
        const double at_prev = sh->at_prev;                       // atan(qP'/iP') as the previous epoch computed it
        double fll_err = at_now - at_prev;
        if (fll_err != fll_err) fll_err = 0.0;
        if (fll_err >= kGpsHalfPi) fll_err = fll_err - kGpsPi;
        else if (fll_err <= -kGpsHalfPi) fll_err = fll_err + kGpsPi;
        const double costas = div_by(at_now, by_2pi);                          // atan/2pi (pll_costas)
        const double fll_full = div_by(div_by(fll_err, by_dt), by_2pi);       // (err/dt)/2pi (fll_atan)
        double w0f = sh->w0f, w0p = sh->w0p;
        if (kaplan && (sh->w0f_bw != u.cur_fll_bw || sh->w0p_bw != u.cur_pll_bw)) {   // (the lock state changed the bandwidths)
            w0f = u.cur_fll_bw / kW0Bw1;
            w0p = u.cur_pll_bw / kW0Bw2;
            if (rlane == 0) {
                sh->w0f = w0f, sh->w0p = w0p;
                sh->w0f_bw = u.cur_fll_bw, sh->w0p_bw = u.cur_pll_bw;
            }
        }
        const double pll_r1 = sh->pll_r1, pll_r2 = sh->pll_r2;  // Borre PLL filter ratios tau2/tau1, pdi/tau1 (tracking.py:180-186)
        __builtin_amdgcn_wave_barrier();  // (every lane has read the previous prompt before lane 0 replaces it)
        if (rlane == 0) {
            double c_pll_mem = st.pll_mem;
            const int c_code_counter = sh->c_code_counter;
            double carrier_hz = ep.carrier_hz;
            double rec_pll, rec_fll, rec_carrier_err;
            if (!kaplan) {  // Borre: channel_l1ca_borre.py:364-429
                const double phase_err = costas;
                double nco_carrier = pll_r1 * (phase_err - c_pll_mem);
                nco_carrier += pll_r2 * phase_err;
                c_pll_mem = phase_err;
                carrier_hz += nco_carrier;
                rec_pll = nco_carrier, rec_fll = 0.0, rec_carrier_err = phase_err;
            } else {        // Kaplan: runDiscriminators / runCarrierFrequencyFilter / postTrackingUpdate
                double fll_d = 0.0, pll_d = 0.0;
                if (u.cur_lock_state == LOCK_PULL_IN) {
                    if (c_code_counter > 1) fll_d = fll_full;
                } else {
                    fll_d = fll_full;
                    pll_d = costas;
                }
                const double upd = (pll_d * (w0p * w0p) + fll_d * w0f) * kDt;  // FLLassistedPLL_2ndOrder (tracking.py:246-279)
                double carrier_err = upd + c_pll_mem;
                c_pll_mem = upd;
                carrier_err += pll_d * kW0A2 * w0p;
                carrier_hz += carrier_err;
                rec_pll = pll_d, rec_fll = fll_d, rec_carrier_err = carrier_err;
            }
            st.pll_mem = c_pll_mem;
            st.i_prompt_prev = ip;
            st.q_prompt_prev = qp;
            sh->at_prev = at_now;
            sh->c_code_counter = c_code_counter + 1;
            sh->ep.carrier_hz = carrier_hz;
            sh->stop_carrier = carrier_bad(carrier_hz) ? 1 : 0;
            sh->dphi = div_by((carrier_hz * 2.0) * M_PI, by_fs);  // carrier_step(): tracking.py:102 uses np.pi
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
        const double env_e = sqrt(ie * ie + qe * qe), env_l = sqrt(il * il + ql * ql);  // DLL NNEML envelopes (tracking.py:120-129)
        const double dll_nn = (env_e - env_l) / (env_e + env_l);
        const double dll_r1 = sh->dll_r1, dll_r2 = sh->dll_r2;   // BorreLoopFilter ratios tau2/tau1, pdi/tau1 (tracking.py:180-186)
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
            const double code_step = div_by(k_code_hz, by_fs);
            const int64_t next_start = ep.start_sample + n;
            const int next_n = (int)ceil((kChips - rem_code) / code_step);
            sh->stop_code = code_out_of_range(sh, u.capacity, u.lut_words, next_start, next_n, rem_code, code_step) ? 1 : 0;
            sh->ep.start_sample = next_start;
            sh->ep.n = next_n;
            sh->ep.rem_code = rem_code;
            sh->ep.code_step = code_step;
            sh->epochs_done = u.epoch + 1;
            if (rec) {
                rec->start_sample = ep.start_sample;
                rec->n_samples = n;
                rec->rem_code_in = ep.rem_code;
                rec->code_step_in = ep.code_step;
                for (int k = 0; k < 2 * SDR_MAX_TAPS; ++k) rec->corr[k] = k < 2 * kTaps ? corr[k < 2 * kTaps ? k : 0] : 0.0;
                // Kaplan records the discriminator and the filter output; Borre the NCO command and the error
                rec->dll = kaplan ? dll_d : code_err;
                rec->code_err = kaplan ? code_err : dll_d;
                rec->code_hz = k_code_hz;
            }
        }
    } else if (role == 2) {
        // ------------------------------------------------------------------ lock indicators, state machine, bits
        const double pw = ip * ip + qp * qp;
        double num = 0.0, den = 1.0;
        switch (rlane) {
            case 0: {                                             // FLL lock (lockindicator.py:6-18)
                const double l_ipp = lk.ipp, l_qpp = lk.qpp;
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
            double l_fll_lock = lk.fll_lock, l_pll_lock = lk.pll_lock, l_cn0 = lk.cn0, l_ratio_acc = lk.ratio_acc;
            double l_ipp = lk.ipp, l_qpp = lk.qpp, l_fll_bw = lk.fll_bw, l_pll_bw = lk.pll_bw, l_nav_sum = lk.nav_sum;
            int l_accum = lk.accum, l_lock_state = lk.lock_state, l_time_in_state = lk.time_in_state;
            int l_spacing_sel = lk.spacing_sel, l_flags = lk.flags, l_code_counter = lk.code_counter;
            int l_nav_count = lk.nav_count, l_bits_emitted = lk.bits_emitted, l_bits_run = lk.bits_run;
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
                        const double c = lam * (1.0 / ((double)l_accum * kDt));
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
                    if (u.writer && u.nav_bits && l_bits_run < u.max_bits) u.nav_bits[(size_t)u.ch * u.max_bits + l_bits_run] = (int8_t)nav_bit;
                    l_bits_run += 1;
                    l_bits_emitted += 1;
                    l_nav_sum = 0.0;
                    l_nav_count = 0;
                }
            }
            // hand-over to the other roles / the next epoch: only what changed (taps and bandwidths move with the lock state)
            if (l_spacing_sel != lk.spacing_sel) {
                const double* sp = l_spacing_sel ? cfg.spacing_narrow : cfg.spacing_wide;
                for (int t = 0; t < kTaps; ++t) sh->spacing[t] = sp[t];
            }
            if (l_lock_state != lk.lock_state || l_fll_bw != lk.fll_bw || l_pll_bw != lk.pll_bw) {
                sh->fll_bw = l_fll_bw;
                sh->pll_bw = l_pll_bw;
                sh->lock_state = l_lock_state;
            }
            lk.fll_lock = l_fll_lock, lk.pll_lock = l_pll_lock, lk.cn0 = l_cn0, lk.ratio_acc = l_ratio_acc;
            lk.ipp = l_ipp, lk.qpp = l_qpp, lk.fll_bw = l_fll_bw, lk.pll_bw = l_pll_bw, lk.nav_sum = l_nav_sum;
            lk.accum = l_accum, lk.lock_state = l_lock_state, lk.time_in_state = l_time_in_state;
            lk.spacing_sel = l_spacing_sel, lk.flags = l_flags, lk.code_counter = l_code_counter;
            lk.nav_count = l_nav_count, lk.bits_emitted = l_bits_emitted, lk.bits_run = l_bits_run;
            if constexpr (DENSE) sh->lk = lk;
            if (rec) {
                rec->cn0 = kaplan ? l_cn0 : 0.0;
                rec->pll_lock = kaplan ? l_pll_lock : 0.0;
                rec->fll_lock = kaplan ? l_fll_lock : 0.0;
                rec->lock_state = l_lock_state;
                rec->track_flags = l_flags;
                rec->nav_bit = nav_bit;
            }
        }
    } else if (role == 3 && rlane == 0) {
        // ------------------------------------------------------------------ carrier phase over the epoch
        // remCarrier -= f * 2 pi * n / fs; remCarrier %= 2 pi -- Kaplan with the GPS-ICD pi (kaplan:523-524), Borre with
        // np.pi (borre:364-365): SURVEY.md T3.  Needs nothing this epoch measured, so it is off the carrier role's chain.
        const double adv = (kaplan ? ep.carrier_hz * kGpsTwoPi * (double)n : ep.carrier_hz * 2.0 * M_PI * (double)n) / fs;
        double rem_carrier = ep.rem_carrier;
        rem_carrier -= adv;
        sh->ep.rem_carrier = py_mod(rem_carrier, kaplan ? kGpsTwoPi : 2.0 * M_PI);
    }
}

__device__ __forceinline__ void lock_regs_store(const LockRegs& r, EpochShared* sh) {
    sdr_track_state& st = sh->st;
    st.fll_lock = r.fll_lock, st.pll_lock = r.pll_lock, st.cn0 = r.cn0, st.cn0_ratio_acc = r.ratio_acc;
    st.fll_bw = r.fll_bw, st.pll_bw = r.pll_bw, st.nav_prompt_sum = r.nav_sum;
    st.accum_counter = r.accum, st.lock_state = r.lock_state, st.time_in_state = r.time_in_state;
    st.spacing_sel = r.spacing_sel, st.track_flags = r.flags;
    st.nav_sum_counter = r.nav_count, st.nav_bits_emitted = r.bits_emitted;
    sh->l_bits_run = r.bits_run;
}

// Wave-uniform copy of the epoch parameters in LDS.  What comes out of LDS is the same in every lane, but only
// readfirstlane tells the compiler so: as scalars the epoch parameters (and everything derived from them: group
// counts, ring positions, linspace constants) live in SGPRs and are computed on the scalar unit -- ~100 VGPRs per lane.
__device__ __forceinline__ EpochParams uniform_params(const EpochParams& v) {
    EpochParams ep;
    ep.start_sample = ((int64_t)__builtin_amdgcn_readfirstlane((int)(v.start_sample >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)v.start_sample);
    ep.n = __builtin_amdgcn_readfirstlane(v.n);
    ep.carrier_hz = uniform(v.carrier_hz);
    ep.rem_carrier = uniform(v.rem_carrier);
    ep.rem_code = uniform(v.rem_code);
    ep.code_step = uniform(v.code_step);
    return ep;
}

// LDS layout in doubles: [0, 256) workgroup reduction scratch, [256, 640) three role waves x 256 words of exchange staging
constexpr int kRedDoubles = 640;
constexpr int kXchgStageWords = 256;   // per role wave: up to 4 wave-wide loads of 64 words

// WAVES: resident waves per SIMD the register allocation has to leave room for (2 = two 256-thread
// workgroups can share a CU, at the price of a few spills).
// states / cfgs are indexed through ch_map when it is given (the device-resident channel bank: the launch serves
// the listed channels of a larger array); everything this launch produces (trajectory, bits, epochs_done, states_copy)
// is indexed by the position in the list.  Any of these outputs, ch_map and the fault word may live in page-locked
// host memory (a receiver tick reads its few KB of results without a single copy command).
// THREADS == 256 && WAVES == 1 is the CLUSTER form (parts >= 2 workgroups per channel, cooperative launch).
// A ONE-EPOCH step of that form (a receiver tick) needs no cooperative launch when it is cut at the exchange: phase 1 =
// the cluster's workgroups correlate and publish their sums, then end; phase 2 = one workgroup per channel collects the
// parts' sums (same order, same bits as the cooperative kernel) and runs the loop update.  Nobody waits for a peer inside
// a kernel, so two plain launches do (phase = 0: the whole epoch loop in one launch).  tag_base: added to the exchange
// words' epoch tag, so that lines left by earlier launches cannot validate (the phases do not zero the lines).
template <int FMT, int THREADS, int WAVES, int NT>
__global__ __launch_bounds__(THREADS, WAVES) void track_kernel(const void* __restrict__ ring, int64_t capacity,
                                                        sdr_track_state* __restrict__ states,
                                                        sdr_track_state* __restrict__ states_copy,
                                                        const int32_t* __restrict__ ch_map,
                                                        const sdr_loop_cfg* __restrict__ cfgs, int cfg_stride,
                                                        int n_epochs, sdr_track_epoch* __restrict__ traj,
                                                        int keep_traj, int8_t* __restrict__ nav_bits, int max_bits,
                                                        int32_t* __restrict__ n_bits,
                                                        int32_t* __restrict__ epochs_done_out,
                                                        const uint32_t* __restrict__ luts,
                                                        int lut_words, int lut_stride, int use_prefix,
                                                        int n_ch, int parts, unsigned long long* xchg,
                                                        int* __restrict__ fault, int phase, unsigned tag_base, const TickServer srv) {
    constexpr int kTaps = NT;
    constexpr int kXchgWords = xchg_words(NT);
    constexpr bool kCluster = THREADS == 256 && WAVES == 1;
    extern __shared__ double smem[];
    double* red = smem;                                   // kWaves * 2*NT wave sums, then the exchange staging
    EpochShared* sh = reinterpret_cast<EpochShared*>(red + kRedDoubles);
    double2* prefix = reinterpret_cast<double2*>(sh + 1);  // THREADS * kPrefixSlots, when the launcher found room
    uint32_t* lut = reinterpret_cast<uint32_t*>(prefix + (use_prefix ? THREADS * kPrefixSlots : 0));

    const int tid = threadIdx.x;
    // (uniform) a tick-server launch of the cluster form: resident, every tick behind the doorman's release
    const bool server = kCluster && srv.host != nullptr;
    int n_wg = (int)gridDim.x;
    int bid = (int)blockIdx.x;
    // (uniform) a one-launch receiver tick that brings its slab along: the launch's first workgroups are the ingest
    const bool with_slab = kCluster && !server && srv.ingest_n16 != 0;
    if constexpr (kCluster) {
        if (with_slab) {
            if (bid < kTickIngestGroups) {
                const uint4* const src = reinterpret_cast<const uint4*>(static_cast<uintptr_t>(srv.ingest_src16) << 4);
                for (unsigned long long i = (unsigned long long)bid * THREADS + tid; i < srv.ingest_n16; i += (unsigned long long)kTickIngestGroups * THREADS) {
                    unsigned long long d = srv.ingest_first16 + i;
                    if (d >= srv.ring_n16) d -= srv.ring_n16;
                    uint4 o = src[i];
                    o.x ^= srv.ring_flip, o.y ^= srv.ring_flip, o.z ^= srv.ring_flip, o.w ^= srv.ring_flip;
                    srv.ring16[d] = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every wave: its stores have left (the barrier orders them ...)
                __syncthreads();                                          // ... before lane 0's device-wide release)
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(srv.ingest_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            bid -= kTickIngestGroups;
            n_wg -= kTickIngestGroups;
        }
    }
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8): the parts of one channel are
    // blockIdx-es with the same residue, so a cluster shares one XCD's L2 for its exchange lines.
    int ch, part;
    const bool collect_only = kCluster && phase == 2;      // (uniform) second half of a two-launch tick: no correlation here
    const bool publish_only = kCluster && phase == 1;      // (uniform) first half: ends after publishing its sums
    // (uniform) the two halves in ONE plain launch: every part publishes its sums and draws a ticket; the part that draws its
    // channel's last one carries on as phase 2 would (it polls the lines, its own among them, and adds them in part order:
    // the same bits), the others end.  Nobody waits for a peer that may not be resident: a line that is still on its way is
    // a store already issued by a workgroup that has run.
    const bool last_collects = kCluster && phase == 3;
    if (collect_only) {
        ch = bid;
        part = 0;
    } else if (parts == 1 || n_wg % (8 * parts) != 0) {
        ch = bid / parts;
        part = bid % parts;
    } else {
        const int per_xcd = n_wg / 8;                    // workgroups per XCD = channels per XCD * parts
        const int xcd = bid % 8, q = bid / 8;
        ch = xcd * (per_xcd / parts) + q / parts;
        part = q % parts;
    }
    const int lane_global = part * THREADS + tid;          // index among the cluster's lanes
    const int cluster_lanes = parts * THREADS;
    const bool edge_wave = lane_global >= cluster_lanes - 64;
    const int edge_lane = edge_wave ? lane_global - (cluster_lanes - 64) : -1;
    const int sidx = ch_map ? ch_map[ch] : ch;             // where this channel's state and configuration live
    const sdr_loop_cfg* __restrict__ cfg_ptr = cfgs + (size_t)sidx * cfg_stride;
    {   // the channel's state and configuration into LDS, one 64-bit word per lane (a single thread copying the 472
        // bytes is ~60 loads one after the other: microseconds of a one-epoch receiver tick)
        constexpr int kStWords = (int)(sizeof(sdr_track_state) / 8), kCfgWords = (int)(sizeof(sdr_loop_cfg) / 8);
        static_assert(sizeof(sdr_track_state) % 8 == 0 && sizeof(sdr_loop_cfg) % 8 == 0 && kStWords <= 64 && kCfgWords <= 64,
                      "state / configuration are copied as 64-bit words by the first two waves");
        if (tid < kStWords)
            reinterpret_cast<unsigned long long*>(&sh->st)[tid] = reinterpret_cast<const unsigned long long*>(states + sidx)[tid];
        else if (tid >= 64 && tid < 64 + kCfgWords)
            reinterpret_cast<unsigned long long*>(&sh->cfg)[tid - 64] = reinterpret_cast<const unsigned long long*>(cfg_ptr)[tid - 64];
    }
    if (tid == 0) {
        sh->epochs_done = 0;
        sh->fault = 0;
    }
    const int slot = states[sidx].code_slot;
    if (!collect_only) stage_lut<THREADS>(lut, luts + (size_t)slot * lut_stride, lut_words, tid);
    const double fs = cfg_ptr->fs;
    UpdateCtx u;
    u.fs = fs;
    u.chips = cfg_ptr->epoch_chips > 0.0 ? cfg_ptr->epoch_chips : kDefaultEpochChips;
    u.epochs_per_bit = cfg_ptr->epochs_per_bit > 0 ? cfg_ptr->epochs_per_bit : kDefaultEpochsPerBit;
    u.dt = cfg_ptr->epoch_seconds > 0.0 ? cfg_ptr->epoch_seconds : kDefaultEpochSeconds;
    u.by_fs = inv_den(fs);
    u.by_dt = inv_den(u.dt);
    u.by_2pi = inv_den(kGpsTwoPi);
    // the one-workgroup forms (three 256-thread workgroups per compute unit at 168 registers per lane; 512 threads, two waves
    // per SIMD): no registers to spare for what only the roles read
    constexpr bool kDense = !kCluster;
    if constexpr (kDense) {
        if (tid == 0) {
            UpdateConsts c;
            c.fs = u.fs, c.chips = u.chips, c.dt = u.dt, c.epochs_per_bit = u.epochs_per_bit;
            c.by_fs_b = u.by_fs.b, c.by_fs_y = u.by_fs.y, c.by_dt_b = u.by_dt.b, c.by_dt_y = u.by_dt.y;
            c.by_2pi_b = u.by_2pi.b, c.by_2pi_y = u.by_2pi.y;
            c.ok_bits = (u.by_fs.ok ? 1 : 0) | (u.by_dt.ok ? 2 : 0) | (u.by_2pi.ok ? 4 : 0);
            sh->uc = c;
        }
    } else {
        // (vector registers: the scalar file is full of per-epoch constants, and spilled SGPRs come back as v_readlane)
        asm volatile("" : "+v"(u.fs), "+v"(u.chips), "+v"(u.dt));
        asm volatile("" : "+v"(u.by_fs.b), "+v"(u.by_fs.y), "+v"(u.by_dt.b), "+v"(u.by_dt.y), "+v"(u.by_2pi.b), "+v"(u.by_2pi.y));
    }
    u.capacity = capacity;
    u.lut_words = lut_words;
    u.n_epochs = n_epochs;
    u.ch = ch;
    u.nav_bits = nav_bits;
    u.max_bits = max_bits;
    sdr_track_state& st = sh->st;
    bool writer = part == 0;                               // one part records trajectory, bits and the end state (phase 3: the
    u.writer = writer;                                     // part that drew the channel's last ticket; part 0 of a stopped channel)

#ifdef SDR_TRACE_TRACK
    unsigned long long mark_ = wall_clock64();
    const unsigned long long clk0_ = clock64(), wall0_ = wall_clock64();   // shader clock the kernel really runs at
    if (tid == 0 && ch == 0 && part == 0) for (int k = 0; k < 64; ++k) g_track_phase[k] = 0;
    __syncthreads();
#endif
    // Each update role owns a disjoint set of fields of the state's LDS copy (loaded into registers for the
    // duration of its update only: carried across the correlation they cost ~50 VGPRs and spill) and publishes its
    // part of the next epoch's parameters.
    const int role = tid >> 6, rlane = tid & 63;
    const sdr_track_state s_init = states[sidx];
    if (tid == 0) {  // parameters of the first epoch
        double smin = cfg_ptr->spacing_wide[0], smax = smin;
        for (int t = 0; t < kTaps; ++t) {
            smin = fmin(smin, fmin(cfg_ptr->spacing_wide[t], cfg_ptr->spacing_narrow[t]));
            smax = fmax(smax, fmax(cfg_ptr->spacing_wide[t], cfg_ptr->spacing_narrow[t]));
        }
        sh->smin = smin;
        sh->smax = smax;
        sh->stop_code = code_out_of_range(sh, capacity, lut_words, s_init.current_sample, s_init.n_samples, s_init.rem_code,
                                          s_init.code_step) ? 1 : 0;
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
        sh->fll_bw = s_init.fll_bw;
        sh->pll_bw = s_init.pll_bw;
        sh->lock_state = s_init.lock_state;
        sh->c_code_counter = sh->l_code_counter = s_init.code_counter;
        sh->l_ipp = s_init.i_prompt_prev;
        sh->l_qpp = s_init.q_prompt_prev;
        sh->l_bits_run = 0;
        // the quotients the roles keep instead of re-dividing (see EpochShared)
        sh->at_prev = atan(s_init.q_prompt_prev / s_init.i_prompt_prev);
        sh->w0f_bw = s_init.fll_bw, sh->w0p_bw = s_init.pll_bw;
        sh->w0f = s_init.fll_bw / kW0Bw1, sh->w0p = s_init.pll_bw / kW0Bw2;
        sh->pll_r1 = cfg_ptr->pll_tau2 / cfg_ptr->pll_tau1, sh->pll_r2 = cfg_ptr->pll_pdi / cfg_ptr->pll_tau1;
        sh->dll_r1 = cfg_ptr->dll_tau2 / cfg_ptr->dll_tau1;
        sh->dll_r2 = (cfg_ptr->loop_kind != 0 ? cfg_ptr->dll_pdi * 1.0 : cfg_ptr->dll_pdi) / cfg_ptr->dll_tau1;
    }
    // (cluster form) this lane's 16-sample group of the current epoch and of the next one: where epoch k+1 starts is
    // known when epoch k starts, so its samples are requested a whole epoch ahead
    Raw8<FMT> cur[2], nxt[2];
    bool have_next = false;
    LockRegs lk = lock_regs_from(s_init);                  // (meaningful in lane 0 of the lock role's wave only)
    if constexpr (kDense)
        if (tid == 0) sh->lk = lk;                         // (the dense form keeps it in LDS: see EpochShared)
    // ring position of the current epoch's first sample: start % capacity once, then += n (minus the capacity when it
    // passes it) -- the 64-bit modulo is not paid three times per epoch
    int64_t ring_pos;
    {
        const int64_t p0 = s_init.current_sample % capacity;
        ring_pos = ((int64_t)__builtin_amdgcn_readfirstlane((int)(p0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);
    }
    // the channel's state as the roles' LDS copy and their last announcement give it (lane 0 of the recording part)
    auto compose_state = [&]() {
        st.current_sample = sh->ep.start_sample;
        st.n_samples = sh->ep.n;
        st.carrier_hz = sh->ep.carrier_hz;
        st.rem_carrier = sh->ep.rem_carrier;
        st.rem_code = sh->ep.rem_code;
        st.code_step = sh->ep.code_step;
    };
    unsigned server_tick = 0;                              // (server) requests this workgroup has seen
#ifndef SDR_DENSE_STAGGER
#define SDR_DENSE_STAGGER 0
#endif
    if constexpr (!kCluster && THREADS == 256 && SDR_DENSE_STAGGER != 0) {
        // Three workgroups share a compute unit, and workgroups that start together stay together: all three correlate at
        // once (the vector pipes shared three ways), then all three close their loops at once (one wave per role, the pipes
        // nearly idle).  The dispatcher fills the 256 compute units once before it gives any a second workgroup, so the
        // launch's thirds ARE the slots: the second and third start a third / two thirds of an epoch later and their
        // role phases fall into the others' correlation.
#ifndef SDR_DENSE_STAGGER_MODE
#define SDR_DENSE_STAGGER_MODE 0
#endif
        // (mode 1: the dispatcher packs a compute unit before it moves on -- an XCD's workgroups 3 j, 3 j + 1, 3 j + 2 share one)
        const int slot_on_cu = SDR_DENSE_STAGGER_MODE == 0 ? ((int)blockIdx.x / 256) % 3 : ((int)blockIdx.x / 8) % 3;
        for (int k = 0; k < slot_on_cu * SDR_DENSE_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);      // (8128 cycles = 3.4 us each)
    }
    for (int epoch = 0; epoch < n_epochs; ++epoch) {
        TRACK_MARK(5);
        __syncthreads();
        TRACK_MARK(0);
#ifdef SDR_TRACE_TRACK
        const unsigned long long wave_mark_ = wall_clock64();
#endif
        if constexpr (kCluster) {
            if (server) {
                // ---- the gate: wait for the doorman's release of the next request (bounded), see the samples it brought
                if (tid == 0) {
                    const unsigned want = server_tick + 1;
                    const unsigned long long t0 = wall_clock64();
                    unsigned g;
                    // (four looks in flight, a quarter of a round trip apart: a look that left just before the release landed
                    // is followed by one that sees it a quarter of a round trip later, not a whole one)
                    unsigned* const go = &srv.go[(size_t)ch * kGoStride];
                    auto look = [&]() { return __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                    auto hit = [&](unsigned v) { return v == want || v == kServerStop; };
                    unsigned l0 = look();
                    __builtin_amdgcn_s_sleep(3);
                    unsigned l1 = look();
                    __builtin_amdgcn_s_sleep(3);
                    unsigned l2 = look();
                    __builtin_amdgcn_s_sleep(3);
                    unsigned l3 = look();
                    for (;;) {
                        if (hit(l0)) { g = l0; break; }
                        l0 = look();
                        if (hit(l1)) { g = l1; break; }
                        l1 = look();
                        if (hit(l2)) { g = l2; break; }
                        l2 = look();
                        if (hit(l3)) { g = l3; break; }
                        l3 = look();
                        if (wall_clock64() - t0 > 2 * srv.idle_ticks) {
                            g = kServerStop;
                            break;
                        }
                    }
                    sh->gate = g;
#ifdef SDR_SRV_TRACE
                    if (writer) {
                        const unsigned long long at = __hip_atomic_load(&srv.dev->seen_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        srv.dev->ch_gate[ch] += wall_clock64() - at;
                        unsigned hw, xcc;
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                        srv.dev->ch_where[ch] = (xcc << 16) | (hw & 0xffff);
                    }
#endif
                    // (the request's write index now, past the L2: its round trip runs beside the invalidation below)
                    sh->gate_wi = __hip_atomic_load(&srv.dev->write_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ch == 0 && part == 0) srv.dev->t[0] = wall_clock64();
                }
                __syncthreads();
                if (sh->gate == kServerStop) break;
                ++server_tick;
                // (the slab was written through another XCD's L2: one invalidation serves the compute unit; the barrier hands it on)
                if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                if (tid == 0 && ch == 0 && part == 0) srv.dev->t[1] = wall_clock64();
                const bool dead = (sh->fault | sh->stop_code | sh->stop_carrier) != 0;
                bool ready = false;
                if (!dead) {
                    const int64_t wi = sh->gate_wi;
                    const int64_t unread = ring_pos <= wi ? wi - ring_pos : capacity - ring_pos + wi;   // circularbuffer.py:139-148
                    ready = unread >= (int64_t)sh->ep.n;
                }
                if (!ready) {       // (uniform over the channel's parts: same state, same write index)
                    if (tid == 0 && writer) {
                        __hip_atomic_store(&srv.h_ran[ch], dead ? -1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (acknowledged: in the host's memory -- see the answer below)
                        __hip_atomic_store(&srv.h_done[ch], server_tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_fetch_add(&srv.dev->done_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    --epoch;        // (this channel's next epoch is still the same one)
                    continue;
                }
            }
        }
        if constexpr (kCluster) {
            if (with_slab && epoch == 0) {
                // the slab this launch brought along: in the ring when the ingest workgroups have all counted themselves in (they
                // were dispatched in front of this one; bounded all the same), visible once this compute unit's caches are told
                if (tid == 0) {
                    const unsigned long long t0 = wall_clock64();
                    while ((int)(__hip_atomic_load(srv.ingest_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - srv.ingest_target) < 0) {
                        if (wall_clock64() - t0 > 200000ull) {      // 2 ms of the 100 MHz clock
                            sh->fault = 1;
                            *fault = 2;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
                if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
            }
        }
        if (sh->fault | sh->stop_code | sh->stop_carrier) break;
        const EpochParams ep = uniform_params(sh->ep);
        const double dphi = uniform(sh->dphi);
        // what the carrier loop needs from the state machine's previous decision (captured now: the lock role
        // rewrites these while the carrier role is still running)
        u.ep = ep;
        u.cur_fll_bw = uniform(sh->fll_bw), u.cur_pll_bw = uniform(sh->pll_bw);
        u.cur_lock_state = __builtin_amdgcn_readfirstlane(sh->lock_state);
        u.epoch = epoch;
        u.rec = (writer && keep_traj) ? traj + ((size_t)ch * n_epochs + epoch) : nullptr;
        if constexpr (kCluster)
            if (server) u.rec = writer ? srv.rec_out + ch : nullptr;

        double accr[kTaps], acci[kTaps];
#pragma unroll
        for (int t = 0; t < kTaps; ++t) accr[t] = acci[t] = 0.0;
        bool single = collect_only;                            // (phase 2: neither correlator runs)
        int64_t ring_pos_next = ring_pos + ep.n;               // (n <= capacity: checked by the role that announced it)
        if (ring_pos_next >= capacity) ring_pos_next -= capacity;
        if constexpr (kCluster) if (!collect_only) {
            const SingleGeometry geo = single_geometry(ring_pos, ep.n, capacity);
            single = use_prefix && ep.code_step >= kFastMinCodeStep && ep.code_step <= kFastMaxCodeStep && geo.fits &&
                     geo.groups <= cluster_lanes;
            if (single) {
                if (have_next) {
                    cur[0] = nxt[0];
                    cur[1] = nxt[1];
                } else {
                    single_load<FMT>(ring, single_load_pos(geo, lane_global, capacity), cur);
                }
                EpochConsts<kTaps> K;
                compute_constants<kTaps>(K, ep, sh->spacing, dphi, cluster_lanes);
                TRACK_MARK(1);
                correlate_epoch_single<FMT, kTaps>(cur, ep, dphi, K, lut, prefix, tid, lane_global, geo, accr, acci);
            }
        }
        if (!single) {
            const bool boundary_ok = use_prefix && ep.code_step >= kFastMinCodeStep && !epoch_wraps(ep, capacity);  // (uniform)
            EpochConsts<kTaps> K;
            // (uniform) the chip-aligned core will be tried: it reads the taps' constants only, not the in-group rotations
            bool try_chip = false;
#ifndef SDR_TRACK_NO_CHIP
            // (the 256-thread dense form only: in the 512-thread form a lane owns two chips and the routine's per-epoch part
            // outweighs them -- measured 10.1 against 9.4 us per epoch at 256 channels)
            if constexpr (FMT == SDR_FMT_CI8 && kTaps == 3 && !kCluster && THREADS == 256)
                try_chip = boundary_ok && ep.code_step >= kChipMinCodeStep && ep.code_step <= kChipMaxCodeStep && ring_pos + ep.n + 32 <= capacity;
#endif
            // (the in-group rotations are computed either way: taking compute_tap_constants() on the chip path and these only
            // on a fallback was measured at 26.5 instead of 15.8 us per epoch in the 168-register form -- it spills)
            compute_constants<kTaps>(K, ep, sh->spacing, dphi, cluster_lanes);
            TRACK_MARK(1);
            // one-workgroup forms, ci8, three taps, 24 / 25 samples per chip (the headline 25 MHz): lanes own whole chips
            // (correlator_chip.h, block length compiled in, tap positions at run time: ~14 instead of ~17.5 issue slots per
            // sample); its strips and rotations live where the boundary variants keep their prefix sums
            bool chip_done = false;
#ifndef SDR_TRACK_NO_CHIP
            if constexpr (FMT == SDR_FMT_CI8 && kTaps == 3 && !kCluster && THREADS == 256) {
                static_assert(THREADS * chip_strip_slots<3>() + (THREADS / 64) * kChipRotSlots <= THREADS * kPrefixSlots, "strips + rotations fit the prefix area");
                if (try_chip) {
                    ChipGeom<3> G;
                    chip_geometry<3, 24, 0, 0>(ep.n, K.shift, K.step, K.inv_step, G);
                    if (!__builtin_amdgcn_readfirstlane(G.bad))
                        chip_done = correlate_epoch_chip<3, false, 24, 0, 0>(ring, nullptr, capacity, ep, dphi, K, G, ring_pos, nullptr, lut, prefix,
                                                                             prefix + THREADS * chip_strip_slots<3>() + (tid >> 6) * kChipRotSlots,
                                                                             tid, lane_global, cluster_lanes, edge_lane, accr, acci);
                }
            }
#endif
            if (chip_done) {
            } else if (boundary_ok && ep.code_step <= kFastMaxCodeStep)         // 16-sample boundary variant above ~17 MHz
                correlate_epoch_wide<FMT, kTaps, false, 16>(ring, capacity, ep, dphi, K, lut, prefix, tid, lane_global, cluster_lanes, edge_lane, accr, acci);
            else if (boundary_ok && ep.code_step <= kFastMaxCodeStep8)   // 8-sample boundary variant above ~8.2 MHz
                correlate_epoch_wide<FMT, kTaps, false, 8>(ring, capacity, ep, dphi, K, lut, prefix, tid, lane_global, cluster_lanes, edge_lane, accr, acci);
            else
                correlate_epoch<FMT, kTaps>(ring, capacity, ep, dphi, K, lut, lane_global, cluster_lanes, edge_lane, accr, acci);
        }
#ifdef SDR_TRACE_TRACK
        if ((tid & 63) == 0 && ch == 0) g_track_phase[8 + part * (THREADS / 64) + (tid >> 6)] += wall_clock64() - wave_mark_;
#endif
        TRACK_MARK(2);
        if constexpr (kCluster)
            if (server && tid == 0 && ch == 0 && part == 0) srv.dev->t[2] = wall_clock64();
        // (cluster form: the totals go to wave 3, which publishes them while the three measuring roles already wait for
        // the peers' -- their chains are the epoch's critical path, the carrier-phase role's is short)
        double total;
#ifdef SDR_TRACE_TRACK
        unsigned long long role_mark_ = 0;
#endif
        if constexpr (kCluster) total = collect_only ? 0.0 : reduce_taps_rows<kTaps, THREADS, 3>(accr, acci, red, tid);   // value v in lanes v*G.. of wave 3
        else total = reduce_taps<kTaps, THREADS, 0>(accr, acci, red, tid);
        TRACK_MARK(3);
#ifdef SDR_TRACE_TRACK
        role_mark_ = wall_clock64();   // (all waves leave the reduction barrier together: per-role time from here to the end of its update)
#endif

        double corr[2 * kTaps];
        bool role_ok = true;
        if constexpr (kCluster) {
            // Cluster exchange: wave 3 publishes this part's sums; each of the three measuring roles collects the
            // parts' sums for itself and adds them in part order -- every role wave of every part holds bit-identical
            // totals, there is no second hand-over.  No fences, no separate flag: every 64-bit word carries half a
            // double and the epoch tag (epoch+1), is written and read whole, and validates itself (the "LL" idea of
            // collective libraries) -- an agent-scope release/acquire pair would write back and invalidate the whole
            // L2 every epoch (measured: 2.9 us); a relaxed device-scope word costs ~0.4 us one way
            // (tools/ubench_xchg.hip), on the same XCD or across XCDs.
            // Lines are double-buffered by epoch parity: a part can run at most one exchange ahead of a peer (its
            // wave 3 publishes epoch k+1 only after the workgroup's reduction barrier of that epoch, i.e. after
            // all its role waves have finished reading epoch k).
            unsigned long long* lines = xchg + ((size_t)ch * 2 + (epoch & 1)) * kMaxParts * kXchgWordsMax;
            const unsigned long long tag = (unsigned long long)((unsigned)(epoch + 1) + tag_base) << 32;
            if (role == 3 && !collect_only) {   // lane 2v+h publishes half h of value v (lanes 0..2*NT-1 hold the values)
                const double v = __shfl(total, ((rlane >> 1) & 15) * collector_group_lanes(NT), 64);
                if (rlane < 4 * kTaps) {
                    const unsigned half = (rlane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
                    __hip_atomic_store(lines + part * kXchgWords + rlane, tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (publish_only) return;   // (the whole workgroup: phase 2 takes it from here, in the next launch)
            if (last_collects) {
                // tickets: one counter per position in the launch's list, in front of the lines, never reset -- every launch of this
                // form adds `parts` to the counters it uses, so "the last of this launch" is the ticket that completes a multiple
                // of `parts`, whichever channel had the position the tick before
                if (tid == 192) {
                    unsigned* tickets = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(xchg) - kXchgHeadBytes);
                    sh->gate = __hip_atomic_fetch_add(tickets + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if ((sh->gate + 1u) % (unsigned)parts != 0u) return;      // (the whole workgroup: a later part collects)
                writer = true;
                u.writer = true;
                u.rec = keep_traj ? traj + ((size_t)ch * n_epochs + epoch) : nullptr;
            }
            // Request the next epoch's samples now -- after this epoch's last use of `cur`, before the wait for the
            // peers, so that nothing waits on them: they arrive while this wave sleeps (the counter a wave waits on
            // retires loads in order, and these are ~0.4 us older than the first poll).
            have_next = single && epoch + 1 < n_epochs && !server;   // (server: the next epoch's samples are not in the ring yet)
            if (have_next) {
                const SingleGeometry next = single_geometry(ring_pos_next, ep.n, capacity);
                single_load<FMT>(ring, single_load_pos(next, lane_global, capacity), nxt);
            }
            if (role < 3) {
                // lane l polls word l % W of parts l / W + j * (64 / W), j = 0 .. W/8 - 1 (words 4*NT.. of a line are padding)
                constexpr int kPerPass = 64 / kXchgWords;      // parts covered by one wave-wide load
                constexpr int kPasses = kMaxParts / kPerPass;  // 2 (16-word lines) or 4 (32-word lines)
                const int k = rlane & (kXchgWords - 1), pbase = rlane / kXchgWords;
                bool want[kPasses];
                const unsigned long long* addr[kPasses];
#pragma unroll
                for (int j = 0; j < kPasses; ++j) {
                    const int p = pbase + j * kPerPass;
                    want[j] = k < 4 * kTaps && p < parts;
                    addr[j] = lines + (want[j] ? p : part) * kXchgWords + (want[j] ? k : 0);
                }
                unsigned long long w[kPasses];
                bool done = false;
                // The peers' words cannot be there before a store has crossed to the L2 (~0.4 us): a poll issued at
                // once is wasted -- and 768 waves polling the lines their peers are storing to slow those stores
                // down (tools/ubench_sload.hip: a store -> load round trip is 1029 cycles alone, 1664 with every
                // workgroup polling).  Sleeping ~1000 cycles before the first poll: 5.2 -> 4.8 us per epoch at 32
                // channels (measured 4 / 8 / 12 / 16 / 20 / 28 x 64 cycles: 5.13, 4.97, 4.94, 4.82, 4.87, 5.04).
                // (phase 2: the words were there before this launch began; phase 3: every peer had stored its own before it drew
                // the ticket in front of this workgroup's)
                if (!collect_only && !last_collects) __builtin_amdgcn_s_sleep(kXchgSleep);
                for (long spins = 0; spins < kSpinLimit; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < kPasses; ++j) {
                        w[j] = __hip_atomic_load(addr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (!want[j] || (w[j] >> 32 << 32) == tag);
                    }
                    if (__all(ok)) {
                        done = true;
                        break;
                    }
                }
                if (!done) {
                    role_ok = false;
                    if (rlane == 0) {
                        sh->fault = 1;
                        *fault = 1;
                    }
                } else {
                    unsigned* halves = reinterpret_cast<unsigned*>(red + 256) + role * kXchgStageWords;  // this wave's staging
#pragma unroll
                    for (int j = 0; j < kPasses; ++j) halves[j * 64 + rlane] = (unsigned)w[j];  // [(p)*W + k], p = pbase + j*kPerPass
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS serves a wave's operations in order)
                    __builtin_amdgcn_wave_barrier();
                    double sum = 0.0;
                    if (rlane < 2 * kTaps) {
                        for (int p = 0; p < parts; ++p) {
                            const int at = ((p / kPerPass) * 64) + (p % kPerPass) * kXchgWords;
                            sum += __hiloint2double((int)halves[at + 2 * rlane + 1], (int)halves[at + 2 * rlane]);
                        }
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 2 * kTaps; ++k2) corr[k2] = lane_value(sum, k2);
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 2 * kTaps; ++k2) corr[k2] = 0.0;   // (the carrier-phase role measures nothing)
            }
            TRACK_MARK(6);
        } else {
            // one workgroup per channel: the totals reach the roles on waves 1 and 2 through LDS
            if (tid < 2 * kTaps) sh->corr[tid] = total;
            __syncthreads();
            TRACK_MARK(7);
#pragma unroll
            for (int k2 = 0; k2 < 2 * kTaps; ++k2) corr[k2] = uniform(sh->corr[k2]);
        }
        ring_pos = ring_pos_next;
        if constexpr (kCluster)
            if (server && tid == 0 && ch == 0 && part == 0) srv.dev->t[3] = wall_clock64();
        if (role_ok && role < 4) loop_update<kTaps, kDense>(sh, u, corr, role, rlane, lk);
#ifdef SDR_TRACE_TRACK
        if (rlane == 0 && role < 4 && ch == 0 && part == 0) g_track_phase[48 + role] += wall_clock64() - role_mark_;
#endif
        TRACK_MARK(4);
        if constexpr (kCluster) {
            if (server) {       // ---- the channel's answer, straight to the host: state and record, then the request's number
                if (tid == 0 && ch == 0 && part == 0) srv.dev->t[4] = wall_clock64();
                if (role == 2 && rlane == 0) lock_regs_store(lk, sh);
                __syncthreads();
                if (writer && tid < 64) {
                    if (tid == 0) compose_state();                          // (the NCO values the roles announced, into the LDS copy)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS serves a wave's operations in order)
                    __builtin_amdgcn_wave_barrier();
                    // one wave, 8 bytes per lane: the state out of LDS, the record the roles wrote (this workgroup's own stores:
                    // the barrier above orders them), the flag -- page-locked memory, one store instruction
                    constexpr int kStateWords = (int)(sizeof(sdr_track_state) / 8), kRecWords = (int)(sizeof(sdr_track_epoch) / 8);
                    static_assert(sizeof(sdr_track_state) % 8 == 0 && sizeof(sdr_track_epoch) % 8 == 0 && kStateWords + kRecWords < 64,
                                  "the answer is one 8-byte store per lane of one wave");
                    // (system-scope stores: past every cache, so that "the wave's stores are acknowledged" below means "they are in
                    // the host's memory".  A release FENCE here writes the whole XCD's L2 back -- the four channels that answer on
                    // one XCD queued behind each other for it: the last answer came 3 us after the median one)
                    if (tid < kStateWords)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(srv.h_st + ch) + tid,
                                           reinterpret_cast<const unsigned long long*>(&sh->st)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else if (tid < kStateWords + kRecWords)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(srv.h_rec + ch) + (tid - kStateWords),
                                           reinterpret_cast<const unsigned long long*>(srv.rec_out + ch)[tid - kStateWords], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                    else if (tid == 63)
                        __hip_atomic_store(&srv.h_ran[ch], sh->fault ? -2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (-2: a part of the cluster never published its sums)
                    if (tid == 0 && ch == 0) srv.dev->t[5] = wall_clock64();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the wave's stores have been acknowledged ...)
                    if (tid == 0) {
                        __hip_atomic_store(&srv.h_done[ch], server_tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // ... before this one
                        const unsigned before = __hip_atomic_fetch_add(&srv.dev->done_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ch == 0) srv.dev->t[6] = wall_clock64() + (before & 0);      // (stamped after the add has returned: the next tick reports it)
#ifdef SDR_SRV_TRACE
                        srv.dev->ch_done[ch] += wall_clock64() - __hip_atomic_load(&srv.dev->seen_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        srv.dev->ch_n[ch] += 1;
#endif
                    }
                }
            }
        }
        // the next iteration's first barrier orders the roles' LDS writes against everyone's reads
    }
#ifdef SDR_TRACE_TRACK
    if (tid == 0 && ch == 0 && part == 0) {
        g_track_phase[62] = clock64() - clk0_;
        g_track_phase[63] = wall_clock64() - wall0_;
    }
#endif
    // End state: the roles kept the LDS copy of the state current (the lock role hands its registers back now); one lane
    // of the recording part writes it out.
    if (publish_only) return;   // (only reached when the channel was stopped before its epoch: phase 2 reports that)
    if (last_collects && !writer) return;   // (likewise: a stopped channel's parts draw no tickets, part 0 reports)
    if (role == 2 && rlane == 0) {
        if constexpr (kDense) lk = sh->lk;
        lock_regs_store(lk, sh);
    }
    // (done words: what this wave wrote of the results -- records, bits -- has been acknowledged before the barrier lets the
    // recording lane raise the channel's word)
    if (srv.done_words) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && writer) {
        const int epochs_done = sh->epochs_done;
        // the NCO values of the next epoch are the ones the roles published last
        compose_state();
        // stopped early (the NCO left the staged replica / the ring, or a peer part never showed up): the state is
        // the one after the last completed epoch; the records of the epochs that did not run are marked empty
        if (epochs_done < n_epochs && keep_traj)
            for (int k = epochs_done; k < n_epochs; ++k) traj[(size_t)ch * n_epochs + k].n_samples = 0;
        states[sidx] = st;
        if (states_copy) states_copy[ch] = st;   // (position in the launch's list: what the host reads back)
        if (epochs_done_out) epochs_done_out[ch] = epochs_done;
        if (n_bits) n_bits[ch] = sh->l_bits_run < max_bits ? sh->l_bits_run : max_bits;
        if (srv.done_words) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // (everything above is in the host's memory ...)
            __hip_atomic_store(&srv.done_words[ch], srv.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before this
        }
    }
}

}  // namespace

#ifdef SDR_TRACK_DENSE_TU
// This translation unit (track_dense.hip) carries only the variant of the kernel for more channels than CUs:
// 256-thread workgroups capped at 168 registers so that THREE share a CU.  It is compiled with
// -mllvm -disable-machine-licm: hoisting the fp64 polynomial constants of sincos / atan / division out of the
// epoch loop parks ~80 of them in VGPRs for the whole kernel (256 instead of 173 registers).
hipError_t sdr_track_dense_launch(int fmt, int n_taps, int n_ch, size_t shmem, hipStream_t stream, void** args) {
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        return hipLaunchKernel((const void*)kernel, dim3(n_ch), dim3(256), args, shmem, stream);
    };
    auto by_taps = [&](auto fmt_c) {
        constexpr int F = decltype(fmt_c)::value;
        return n_taps == 5 ? launch(track_kernel<F, 256, 3, 5>) : launch(track_kernel<F, 256, 3, 3>);
    };
    switch (fmt) {
        case SDR_FMT_CI8: return by_taps(std::integral_constant<int, SDR_FMT_CI8>{});
        case SDR_FMT_CI16: return by_taps(std::integral_constant<int, SDR_FMT_CI16>{});
        case SDR_FMT_CF32: return by_taps(std::integral_constant<int, SDR_FMT_CF32>{});
        default: return by_taps(std::integral_constant<int, SDR_FMT_CF64>{});
    }
}
#else
hipError_t sdr_track_dense_launch(int fmt, int n_taps, int n_ch, size_t shmem, hipStream_t stream, void** args);  // track_dense.hip

namespace {

// Device-side operands of one closed-loop launch.
struct TrackRun {
    sdr_track_state* d_states = nullptr;  // indexed through d_map when given
    const int32_t* d_map = nullptr;
    const sdr_loop_cfg* d_cfgs = nullptr;
    int cfg_stride = 0;                   // 0: d_cfgs[0] serves every channel; 1: one per state index
    int n_ch = 0, n_epochs = 0, n_taps = 3;
    sdr_track_epoch* d_traj = nullptr;    // [n_ch][n_epochs] or one scratch record when keep == 0
    int keep = 0;
    int8_t* d_bits = nullptr;
    int max_bits = 0;
    int32_t* d_nbits = nullptr;
    int32_t* d_done = nullptr;            // [n_ch] epochs completed
    sdr_track_state* d_states_copy = nullptr;  // [n_ch] end states by position in the list (nullable)
    int* fault_word = nullptr;            // where the launch's fault flag lives (already zero); nullptr: behind the exchange lines
    int force_parts = 0;                  // 0: choose
    const TickServer* server = nullptr;   // a tick-server launch: cluster form + the doorman, resident until told to leave
    unsigned* done_words = nullptr;       // page-locked [n_ch] (zero): every channel raises its word to done_seq behind its results
    unsigned done_seq = 0;
};

// Enqueue one closed-loop launch on ctx's stream.  *d_fault_out points at the launch's fault word.
int launch_track(sdr_engine* e, StreamCtx* ctx, const TrackRun& r, int* parts_used, int** d_fault_out) {
    const int lut_words = e->lut_stride;  // the whole staged row (every code period the slots were sized for)
    // exchange lines [n_ch][2 parities][8 parts][32 words] (tags zeroed: epoch tags start at 1), then the fault word
    // [head: ticket counters, ingest counter (one-launch ticks)][lines][fault word]
    const size_t xchg_bytes = (size_t)r.n_ch * 2 * kMaxParts * kXchgWordsMax * sizeof(unsigned long long);
    if (int rc = sdr_devbuf_reserve_on(e, ctx->stream, &ctx->xchg, kXchgHeadBytes + xchg_bytes + 16)) return rc;
    unsigned long long* d_xchg = (unsigned long long*)((char*)ctx->xchg.ptr + kXchgHeadBytes);
    int* d_fault = r.fault_word ? r.fault_word : (int*)((char*)d_xchg + xchg_bytes);
    *d_fault_out = d_fault;
    const void* d_iq = e->iq;
    int64_t cap = e->iq_capacity;
    const uint32_t* d_luts = e->luts;
    int lw = lut_words, ls = e->lut_stride, nch = r.n_ch, n_ep = r.n_epochs, mb = r.max_bits, keep = r.keep;
    sdr_track_state* d_st = r.d_states;
    sdr_track_state* d_st_copy = r.d_states_copy;
    const int32_t* d_map = r.d_map;
    const sdr_loop_cfg* d_cfgs = r.d_cfgs;
    int cfg_stride = r.cfg_stride;
    sdr_track_epoch* d_traj = r.d_traj;
    int8_t* d_bits = r.d_bits;
    int32_t* d_nbits = r.d_nbits;
    int32_t* d_done = r.d_done;
    const int nt = r.n_taps;

    int phase = 0;
    unsigned tag_base = 0;
    bool take_slab = false;          // the one-launch tick below pulls the staged slab in itself
    unsigned ingest_target = 0;
    // One attempt with `parts` workgroups per channel.
    auto attempt = [&](int parts, bool* too_big) -> hipError_t {
        // more channels than CUs: smaller workgroups, two or three of which share a CU, so that one channel's
        // loop update overlaps another's correlation (measured: 512 channels 20.2 -> 14.6 us per epoch,
        // 768 channels 22.3 -> 18.6)
        const bool dense = parts == 1 && r.n_ch > e->n_cus;
        const int threads = (parts >= 2 || dense) ? 256 : 512;
        const size_t shmem_base = (size_t)kRedDoubles * sizeof(double) + sizeof(EpochShared) +
                                  (size_t)((lut_words + 3) & ~3) * sizeof(uint32_t);
        const size_t prefix_bytes = (size_t)threads * kPrefixSlots * sizeof(double2);
        // The boundary variant of the correlator needs a 144-byte LDS strip per lane; long multi-period
        // replicas that leave no room for it are tracked with the per-sample variant.
        int up = shmem_base + prefix_bytes <= 160u * 1024u && e->lut_stride < kFastMaxLutWords ? 1 : 0;
        const size_t shmem = shmem_base + (up ? prefix_bytes : 0);
        *too_big = shmem > 160u * 1024u;
        if (*too_big) return hipSuccess;
        if (parts > 1 && !phase)  // tags of a previous launch must not validate this one's polls
            if (hipError_t me = hipMemsetAsync(d_xchg, 0, xchg_bytes + 16, ctx->stream)) return me;
        TickServer srv_arg = {};
        if (r.server) srv_arg = *r.server;
        else srv_arg.done_words = r.done_words, srv_arg.done_seq = r.done_seq;
        if (take_slab && phase == 3) {
            const size_t sb = sdr_fmt_bytes(e->iq_fmt);
            srv_arg.ring16 = (uint4*)e->iq;
            srv_arg.ring_flip = e->iq_fmt == SDR_FMT_CI8 ? sdr::kCi8Flip : 0u;
            srv_arg.ring_n16 = (unsigned long long)((size_t)e->iq_capacity * sb / 16);
            srv_arg.ingest_src16 = (unsigned long long)((uintptr_t)e->srv_slab_src / 16);
            srv_arg.ingest_n16 = (unsigned long long)((size_t)e->srv_slab_n * sb / 16);
            srv_arg.ingest_first16 = (unsigned long long)((size_t)e->srv_slab_off * sb / 16);
            srv_arg.ingest_count = reinterpret_cast<unsigned*>(ctx->xchg.ptr) + 64;
            srv_arg.ingest_target = ingest_target;
        }
        const int extra_groups = srv_arg.ingest_n16 ? kTickIngestGroups : 0;
        void* args[] = {&d_iq, &cap, &d_st, &d_st_copy, &d_map, &d_cfgs, &cfg_stride, &n_ep, &d_traj, &keep, &d_bits, &mb,
                        &d_nbits, &d_done, &d_luts, &lw, &ls, &up, &nch, &parts, &d_xchg, &d_fault, &phase, &tag_base, &srv_arg};
        if (dense) return sdr_track_dense_launch(e->iq_fmt, nt, r.n_ch, shmem, ctx->stream, args);
        hipError_t err = hipSuccess;
        auto launch = [&](auto kernel) {
            // more than 64 KB of dynamic LDS has to be granted per kernel
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (phase)      // a two-launch tick: nobody waits for a peer inside either kernel
                err = hipLaunchKernel((const void*)kernel, dim3(phase == 2 ? r.n_ch : r.n_ch * parts + extra_groups), dim3(threads), args, shmem, ctx->stream);
            else if (parts > 1)  // the parts of a cluster wait for each other: all workgroups must be resident
                err = hipLaunchCooperativeKernel((const void*)kernel, dim3(r.n_ch * parts), dim3(threads), args,
                                                 (unsigned)shmem, ctx->stream);
            else
                err = hipLaunchKernel((const void*)kernel, dim3(r.n_ch), dim3(threads), args, shmem, ctx->stream);
        };
        auto by_shape = [&](auto fmt) {
            constexpr int F = decltype(fmt)::value;
            if (threads == 256) {
                if (nt == 5) launch(track_kernel<F, 256, 1, 5>);
                else launch(track_kernel<F, 256, 1, 3>);
            } else {
                if (nt == 5) launch(track_kernel<F, 512, 2, 5>);
                else launch(track_kernel<F, 512, 2, 3>);
            }
        };
        switch (e->iq_fmt) {
            case SDR_FMT_CI8: by_shape(std::integral_constant<int, SDR_FMT_CI8>{}); break;
            case SDR_FMT_CI16: by_shape(std::integral_constant<int, SDR_FMT_CI16>{}); break;
            case SDR_FMT_CF32: by_shape(std::integral_constant<int, SDR_FMT_CF32>{}); break;
            default: by_shape(std::integral_constant<int, SDR_FMT_CF64>{}); break;
        }
        return err;
    };

    // Cluster size: as many workgroups per channel as the GPU has room for (1 per CU), up to 8.  When the
    // cooperative launch is refused (GPU shared or partitioned: not every workgroup could be resident) the
    // automatic choice halves the cluster until the launch goes through; a forced size fails instead.
    // One epoch (a receiver tick) takes the same cluster as two plain launches (below: a cooperative launch costs the host
    // +15-19 us); two epochs or more take the cooperative launch -- so that what a channel's epochs add up to does not
    // depend on how they are batched into steps (a block of 49 + a step of 2 == a block of 51, bit for bit).
    const int forced = r.force_parts ? r.force_parts : e->track_force_parts;
    int parts = 1;
    if (r.server && (forced < 2 || r.n_epochs < 2)) return sdr_fail(SDR_ERR_INVALID, "tick server: cluster size not given");
    if (forced) {
        parts = forced;
    } else if (r.n_epochs > 1) {
        while (parts < kMaxParts && (long)r.n_ch * parts * 2 <= (long)e->n_cus) parts *= 2;
    }
    if (!r.fault_word) SDR_HIP(hipMemsetAsync(d_fault, 0, 16, ctx->stream));
    // A one-epoch step (a receiver tick) on the cluster a block of epochs would get -- the same partition, the same order of
    // additions, the same bits -- as TWO plain launches cut at the exchange (see track_kernel): 17.4 us of one workgroup
    // per channel became ~11 on eight, without the +15-19 us a cooperative launch costs the host.
    int p2 = 1;
    if (!forced && r.n_epochs == 1 && !e->track_one_launch_tick && !r.server)
        while (p2 < kMaxParts && (long)r.n_ch * p2 * 2 <= (long)e->n_cus) p2 *= 2;
    // A slab staged for "the next tick's launch" (sdr_iq_upload_async: ingest_with_tick): the one-launch tick below takes it
    // along; any other launch wants it in the ring first, the ordinary way.
    const bool tick_form = p2 > 1 && !e->track_two_launch_tick && e->ingest_with_tick && ctx->stream == e->stream;
    if (!r.server) {
        if (e->srv_slab_pending && !tick_form)
            if (int rc = sdr_iq_flush_server_slab(e)) return rc;
        e->last_tick_took_slab = tick_form;
    }
    if (!forced && r.n_epochs == 1 && !e->track_one_launch_tick && !r.server) {
        if (p2 > 1) {
            // the lines are not zeroed per tick: this launch's tag (bit 31 set: no epoch tag of a block launch has it) has
            // never been stored in them -- unless the buffer is new or the 31-bit sequence wrapped: zero it then
            if (ctx->xchg_tagged != ctx->xchg.ptr || ctx->tick_seq >= 0x7ffffff0u) {
                SDR_HIP(hipMemsetAsync(ctx->xchg.ptr, 0, kXchgHeadBytes + xchg_bytes + 16, ctx->stream));
                ctx->ingest_launches = 0;
                ctx->xchg_tagged = ctx->xchg.ptr;
                ctx->tick_seq = 0;
            }
            tag_base = 0x80000000u | ++ctx->tick_seq;
            hipError_t err = hipSuccess;
            bool too_big = false;
            {
                ProfScope ps(e, "track_kernel", ctx->stream);
                if (e->track_two_launch_tick) {
                    phase = 1;
                    err = attempt(p2, &too_big);
                    phase = 2;
                    if (err == hipSuccess && !too_big) err = attempt(p2, &too_big);
                } else {
                    phase = 3;      // (both halves in one plain launch: the part that draws a channel's last ticket collects)
                    take_slab = tick_form && e->srv_slab_pending;
                    if (take_slab) ingest_target = (unsigned)kTickIngestGroups * (ctx->ingest_launches + 1);
                    err = attempt(p2, &too_big);
                    if (take_slab && err == hipSuccess && !too_big) {
                        // (the launch reads the staging half: busy until it has run)
                        ctx->ingest_launches += 1;
                        e->srv_slab_pending = false;
                        const int half = e->srv_slab_half;
                        if (half >= 0) {      // (-1: the caller's own page-locked block, the caller's to keep until the tick returns)
                            if (!e->slab_done[half]) (void)hipEventCreateWithFlags(&e->slab_done[half], hipEventDisableTiming);
                            (void)hipEventRecord(e->slab_done[half], ctx->stream);
                            e->slab_busy[half] = true;
                        }
                    }
                    take_slab = false;
                }
                phase = 0;
            }
            if (too_big) return sdr_fail(SDR_ERR_RANGE, "closed-loop tracking: code table does not fit the LDS");
            if (err != hipSuccess)
                return sdr_fail(SDR_ERR_HIP, "closed-loop tick launch (%d channels x %d parts) failed: %s", r.n_ch, p2, hipGetErrorString(err));
            SDR_HIP(hipGetLastError());
            if (parts_used) *parts_used = p2;
            return SDR_OK;
        }
    }
    hipError_t launch_err = hipSuccess;
    {
        ProfScope ps(e, "track_kernel", ctx->stream);
        for (;;) {
            bool too_big = false;
            launch_err = attempt(parts, &too_big);
            if (too_big) return sdr_fail(SDR_ERR_RANGE, "closed-loop tracking: code table does not fit the LDS");
            if (launch_err == hipSuccess || forced || parts == 1) break;
            (void)hipGetLastError();
            parts /= 2;
        }
    }
    if (launch_err != hipSuccess)
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking launch (%d channels x %d parts) failed: %s", r.n_ch, parts,
                        hipGetErrorString(launch_err));
    SDR_HIP(hipGetLastError());
    if (parts_used) *parts_used = parts;
    return SDR_OK;
}

int check_cfg(const sdr_loop_cfg* cfg, int index) {
    if (cfg->n_taps != 3 && cfg->n_taps != 5)
        return sdr_fail(SDR_ERR_UNSUPPORTED, "channel %d: closed-loop tracking runs 3 (E/P/L) or 5 (VE/E/P/L/VL) taps, got %d",
                        index, cfg->n_taps);
    if (cfg->loop_kind != 0 && cfg->loop_kind != 1)
        return sdr_fail(SDR_ERR_INVALID, "channel %d: loop_kind %d is neither 0 (Borre) nor 1 (Kaplan)", index, cfg->loop_kind);
    if (!(cfg->fs > 0.0)) return sdr_fail(SDR_ERR_INVALID, "channel %d: fs must be positive", index);
    if (cfg->epoch_chips < 0.0 || cfg->epochs_per_bit < 0 || !(cfg->epoch_seconds >= 0.0))
        return sdr_fail(SDR_ERR_INVALID, "channel %d: negative epoch_chips / epochs_per_bit / epoch_seconds", index);
    return SDR_OK;
}

int check_state(sdr_engine* e, const sdr_track_state* st, int index) {
    const int slot = st->code_slot;
    if (slot < 0 || slot >= e->n_slots || e->code_len_host[slot] <= 0)
        return sdr_fail(SDR_ERR_INVALID, "channel %d: code slot %d is not staged", index, slot);
    if (st->n_samples <= 0) return sdr_fail(SDR_ERR_INVALID, "channel %d: n_samples must be positive", index);
    return SDR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------- channel bank
struct BankPending;
struct sdr_bank;
static void sdr_bank_free_pending(sdr_bank* b);
struct sdr_bank {
    int max_channels = 0;
    sdr_track_state* d_states = nullptr;  // [max_channels] -- the tracking state of this GPU's channels lives here
    sdr_loop_cfg* d_cfgs = nullptr;       // [max_channels]
    std::vector<int32_t> n_taps;          // host mirror of what was put: taps per channel, 0 = channel never put
    std::vector<int32_t> slot;
    int64_t code_generation = 0;
    // sdr_bank_step_begin / _end: scratch of their own (a tick between the two halves must not touch the results), on the engine's stream
    StreamCtx async_ctx;
    BankPending* pending = nullptr;
    // scratch of sdr_bank_tick_mirrored (kept between ticks: no allocation in the steady state)
    std::vector<int32_t> tick_list, tick_done;
    std::vector<sdr_track_state> tick_states;
    // a tick between its two halves (sdr_bank_tick_mirrored_begin / _end): one queued launch per tap group
    BankPending* tick_pending[2] = {nullptr, nullptr};
    StreamCtx tick_ctx[2];                // their scratch and page-locked blocks, the bank's own: other calls on the engine
                                          // between the two halves (an upload, a search) cannot move them
    bool tick_open = false, tick_two = false, tick_slab_queued = false;
    bool tick_served = false;             // ... answered by the resident tick server instead
    std::vector<int32_t> tick_candidates; // every tracking channel of the bank (what a tick server serves)
    int tick_groups = 0, tick_group_n[2] = {0, 0}, tick_rc = 0;
    int64_t tick_write_index = 0;
};

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
    if (n_ch < 1) return sdr_fail(SDR_ERR_INVALID, "bad closed-loop request");
    std::vector<int32_t> done((size_t)n_ch, 0);
    if (int rc = sdr_track_closed_loop_ex(e, n_ch, st, cfg, 0, n_epochs, traj, nav_bits, max_bits, n_bits, done.data()))
        return rc;
    // one status for the whole call (the per-channel outcome is what sdr_track_closed_loop_ex reports); the
    // states handed back are valid either way
    for (int c = 0; c < n_ch; ++c)
        if (done[c] < n_epochs)
            return sdr_fail(SDR_ERR_RANGE, "channel %d stopped after %d epochs: NCO state left the staged replica / ring",
                            c, done[c]);
    return SDR_OK;
}

int sdr_track_closed_loop_ex(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfgs,
                             int cfg_per_channel, int n_epochs, sdr_track_epoch* traj, int8_t* nav_bits,
                             int max_bits, int32_t* n_bits, int32_t* epochs_done) {
    if (int rc = sdr_set_device(e)) return rc;
    if ((nav_bits && (max_bits < 1 || !n_bits)) || (!nav_bits && n_bits))
        return sdr_fail(SDR_ERR_INVALID, "nav_bits, max_bits and n_bits go together");
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if (!st || !cfgs || n_ch < 1 || n_epochs < 1) return sdr_fail(SDR_ERR_INVALID, "bad closed-loop request");
    const int n_cfg = cfg_per_channel ? n_ch : 1;
    for (int c = 0; c < n_cfg; ++c) {
        if (int rc = check_cfg(&cfgs[c], c)) return rc;
        if (cfgs[c].n_taps != cfgs[0].n_taps)
            return sdr_fail(SDR_ERR_UNSUPPORTED, "channels of one launch must run the same number of taps (%d vs %d)",
                            cfgs[c].n_taps, cfgs[0].n_taps);
    }
    for (int c = 0; c < n_ch; ++c)
        if (int rc = check_state(e, &st[c], c)) return rc;
    StreamCtx* ctx = &e->ctx0;
    const size_t traj_bytes = traj ? (size_t)n_ch * n_epochs * sizeof(sdr_track_epoch) : 0;
    int rc = sdr_devbuf_reserve(e, &e->track_state, (size_t)n_ch * sizeof(sdr_track_state));
    if (!rc) rc = sdr_devbuf_reserve(e, &e->track_cfg, (size_t)n_cfg * sizeof(sdr_loop_cfg));
    if (!rc) rc = sdr_devbuf_reserve(e, &ctx->traj, traj ? traj_bytes : sizeof(sdr_track_epoch));
    const size_t bits_bytes = nav_bits ? (size_t)n_ch * max_bits : 0;
    const size_t head = ((size_t)2 * n_ch * sizeof(int32_t) + 15) & ~(size_t)15;   // [n_bits][epochs_done] then the bits
    if (!rc) rc = sdr_devbuf_reserve(e, &ctx->bits, head + bits_bytes + 16);
    if (rc) return rc;
    TrackRun r;
    r.d_states = (sdr_track_state*)e->track_state.ptr;
    r.d_cfgs = (const sdr_loop_cfg*)e->track_cfg.ptr;
    r.cfg_stride = cfg_per_channel ? 1 : 0;
    r.n_ch = n_ch, r.n_epochs = n_epochs, r.n_taps = cfgs[0].n_taps;
    r.d_traj = (sdr_track_epoch*)ctx->traj.ptr;
    r.keep = traj ? 1 : 0;
    r.d_nbits = nav_bits ? (int32_t*)ctx->bits.ptr : nullptr;
    r.d_done = (int32_t*)ctx->bits.ptr + n_ch;
    r.d_bits = nav_bits ? (int8_t*)ctx->bits.ptr + head : nullptr;
    r.max_bits = max_bits;
    SDR_HIP(hipMemsetAsync(ctx->bits.ptr, 0, head + bits_bytes, ctx->stream));
    SDR_HIP(hipMemcpyAsync(e->track_state.ptr, st, (size_t)n_ch * sizeof(sdr_track_state), hipMemcpyHostToDevice, ctx->stream));
    SDR_HIP(hipMemcpyAsync(e->track_cfg.ptr, cfgs, (size_t)n_cfg * sizeof(sdr_loop_cfg), hipMemcpyHostToDevice, ctx->stream));
    int parts = 1;
    int* d_fault = nullptr;
    if (int rc2 = launch_track(e, ctx, r, &parts, &d_fault)) return rc2;
    int fault_host = 0;
    std::vector<int32_t> done_host((size_t)n_ch, 0);
    SDR_HIP(hipMemcpyAsync(&fault_host, d_fault, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipMemcpyAsync(st, e->track_state.ptr, (size_t)n_ch * sizeof(sdr_track_state), hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipMemcpyAsync(done_host.data(), r.d_done, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (traj) SDR_HIP(hipMemcpyAsync(traj, ctx->traj.ptr, traj_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (nav_bits) {
        SDR_HIP(hipMemcpyAsync(nav_bits, r.d_bits, bits_bytes, hipMemcpyDeviceToHost, ctx->stream));
        SDR_HIP(hipMemcpyAsync(n_bits, r.d_nbits, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    if (fault_host)
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking: a workgroup of a %d-part cluster never published its sums", parts);
    if (epochs_done)
        for (int c = 0; c < n_ch; ++c) epochs_done[c] = done_host[c];
    return SDR_OK;
}

/* ------------------------------------------------------------------------------------------ channel bank */

int sdr_bank_create(sdr_engine* e, int max_channels, sdr_bank** out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (max_channels < 1 || max_channels > 65536) return sdr_fail(SDR_ERR_INVALID, "max_channels %d outside 1..65536", max_channels);
    sdr_bank* b = new (std::nothrow) sdr_bank();
    if (!b) return sdr_fail(SDR_ERR_NOMEM, "host allocation failed");
    b->max_channels = max_channels;
    b->n_taps.assign((size_t)max_channels, 0);
    b->slot.assign((size_t)max_channels, -1);
    hipError_t err = hipMalloc(&b->d_states, (size_t)max_channels * sizeof(sdr_track_state));
    if (err == hipSuccess) err = hipMalloc(&b->d_cfgs, (size_t)max_channels * sizeof(sdr_loop_cfg));
    if (err == hipSuccess) err = hipMemsetAsync(b->d_states, 0, (size_t)max_channels * sizeof(sdr_track_state), e->ctx0.stream);
    if (err == hipSuccess) err = hipMemsetAsync(b->d_cfgs, 0, (size_t)max_channels * sizeof(sdr_loop_cfg), e->ctx0.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->ctx0.stream);
    if (err != hipSuccess) {
        sdr_bank_destroy(e, b);
        return sdr_fail(err == hipErrorOutOfMemory ? SDR_ERR_NOMEM : SDR_ERR_HIP, "bank setup failed: %s", hipGetErrorString(err));
    }
    *out = b;
    return SDR_OK;
}

void sdr_bank_destroy(sdr_engine* e, sdr_bank* b) {
    if (!b) return;
    if (e) {
        (void)hipSetDevice(e->device);
        (void)sdr_tick_server_stop(e);      // (a resident tick server serves this bank: it leaves before the bank goes)
        (void)hipDeviceSynchronize();
    }
    if (b->d_states) (void)hipFree(b->d_states);
    if (b->d_cfgs) (void)hipFree(b->d_cfgs);
    for (StreamCtx* c : {&b->async_ctx, &b->tick_ctx[0], &b->tick_ctx[1]}) {
        for (DevBuf* d : {&c->traj, &c->bits, &c->xchg})
            if (d->ptr) (void)hipFree(d->ptr);
        if (c->pinned) (void)hipHostFree(c->pinned);
    }
    sdr_bank_free_pending(b);
    delete b;
}

int sdr_bank_put(sdr_engine* e, sdr_bank* b, int ch, const sdr_track_state* st, const sdr_loop_cfg* cfg) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b || !st || !cfg) return sdr_fail(SDR_ERR_INVALID, "NULL bank, state or configuration");
    if (ch < 0 || ch >= b->max_channels) return sdr_fail(SDR_ERR_RANGE, "channel %d outside the bank's %d", ch, b->max_channels);
    if (b->tick_open) return sdr_fail(SDR_ERR_STATE, "a tick of this bank is in flight: sdr_bank_tick_mirrored_end first");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if (int rc = check_cfg(cfg, ch)) return rc;
    if (int rc = check_state(e, st, ch)) return rc;
    hipStream_t s = e->ctx0.stream;
    SDR_HIP(hipMemcpyAsync(b->d_states + ch, st, sizeof(*st), hipMemcpyHostToDevice, s));
    SDR_HIP(hipMemcpyAsync(b->d_cfgs + ch, cfg, sizeof(*cfg), hipMemcpyHostToDevice, s));
    SDR_HIP(hipStreamSynchronize(s));
    b->n_taps[ch] = cfg->n_taps;
    b->slot[ch] = st->code_slot;
    return SDR_OK;
}

int sdr_bank_get(sdr_engine* e, sdr_bank* b, int ch, sdr_track_state* st) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b || !st) return sdr_fail(SDR_ERR_INVALID, "NULL bank or state");
    if (ch < 0 || ch >= b->max_channels) return sdr_fail(SDR_ERR_RANGE, "channel %d outside the bank's %d", ch, b->max_channels);
    if (!b->n_taps[ch]) return sdr_fail(SDR_ERR_STATE, "channel %d was never put into the bank", ch);
    SDR_HIP(hipMemcpyAsync(st, b->d_states + ch, sizeof(*st), hipMemcpyDeviceToHost, e->ctx0.stream));
    SDR_HIP(hipStreamSynchronize(e->ctx0.stream));
    return SDR_OK;
}

// Diagnostics build (-DSDR_TICK_TIMING): where the host side of a one-epoch step spends its time (stderr, every 128 calls).
#ifdef SDR_TICK_TIMING
static double g_tick_t[8], g_tick_sum[8];
static long g_tick_calls;
#define TICK_CLOCK(k) g_tick_t[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count()
#define TICK_REPORT()                                                                                                       \
    do {                                                                                                                    \
        for (int k_ = 1; k_ <= 4; ++k_) g_tick_sum[k_] += g_tick_t[k_] - g_tick_t[k_ - 1];                                  \
        if (++g_tick_calls % 128 == 0) {                                                                                    \
            fprintf(stderr, "[tick timing] prepare %.2f  launch %.2f  wait %.2f  copy-out %.2f us\n", g_tick_sum[1] / 128,   \
                    g_tick_sum[2] / 128, g_tick_sum[3] / 128, g_tick_sum[4] / 128);                                          \
            for (int k_ = 0; k_ < 8; ++k_) g_tick_sum[k_] = 0;                                                              \
        }                                                                                                                   \
    } while (0)
#else
#define TICK_CLOCK(k) ((void)0)
#define TICK_REPORT() ((void)0)
#endif

// A bank step whose launch and result copies are queued but not waited for (sdr_bank_step_begin / _end).
struct BankPending {
    bool active = false;
    StreamCtx* ctx = nullptr;
    int n_ch = 0, n_epochs = 0, parts = 1;
    int32_t* p_head = nullptr;
    sdr_track_state* p_states = nullptr;
    sdr_track_epoch* p_rec = nullptr;
    int8_t* p_bits = nullptr;
    size_t rec_bytes = 0, st_bytes = 0, bits_bytes = 0;
    volatile unsigned* done_words = nullptr;   // (results straight into page-locked memory) the channels' done words ...
    unsigned done_seq = 0;                     // ... and what they show when a channel's results are there
};

static void sdr_bank_free_pending(sdr_bank* b) {
    delete b->pending;
    b->pending = nullptr;
    for (int g = 0; g < 2; ++g) {
        delete b->tick_pending[g];
        b->tick_pending[g] = nullptr;
    }
}

// What is left of a step once everything is queued: wait, check, hand the results out of the page-locked block.
static int bank_collect(BankPending& P, sdr_track_epoch* records, sdr_track_state* states_out, int32_t* epochs_done, int8_t* nav_bits,
                        int32_t* n_bits) {
    P.active = false;
#ifdef SDR_TICK_TIMING
    {   // (how long before the stream's signal are the channels' results in page-locked memory?)
        static double early_sum; static long early_n;
        const double t0 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
        volatile int32_t* done = P.p_head + 4;
        long spins = 0;
        for (int c = 0; c < P.n_ch; ++c) while (!done[c] && ++spins < 2000000) {}
        const double t1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
        (void)hipStreamSynchronize(P.ctx->stream);
        const double t2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
        early_sum += t2 - t1;
        if (++early_n % 128 == 0) {
            fprintf(stderr, "[tick timing] results seen %.2f us after the wait began, the stream's signal %.2f us after that\n", t1 - t0, early_sum / 128);
            early_sum = 0;
        }
    }
#endif
    // The results are in page-locked memory ~9 us before the stream says so (the kernel's end, its release, the signal, the
    // runtime's wake-up: measured on the receiver tick, 18.6 against 27.8 us after the wait began): every channel raises a word
    // behind its results, and those are what is waited for.  Bounded: a launch that died never raises them -- the stream is
    // asked then, and says why.
    bool seen = false;
    if (P.done_words) {
        const auto t0 = std::chrono::steady_clock::now();
        int c = 0;
        for (long spins = 0;; ++spins) {
            while (c < P.n_ch && __atomic_load_n(&P.done_words[c], __ATOMIC_ACQUIRE) == P.done_seq) ++c;
            if (c == P.n_ch) {
                seen = true;
                break;
            }
            if ((spins & 4095) == 4095 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.02) break;
        }
    }
    if (!seen) {
        if (P.done_words) P.ctx->xchg_tagged = nullptr;      // (whatever kept the words from coming: the next tick zeroes lines and ticket counters)
        SDR_HIP(hipStreamSynchronize(P.ctx->stream));
    }
    TICK_CLOCK(3);
    if (P.p_head[0]) P.ctx->xchg_tagged = nullptr;     // (a part that never showed up has drawn no ticket: start the counters over)
    if (P.p_head[0] == 2)
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking: the slab the tick's launch was to pull into the ring never arrived");
    if (P.p_head[0])
        return sdr_fail(SDR_ERR_HIP, "closed-loop tracking: a workgroup of a %d-part cluster never published its sums", P.parts);
    if (epochs_done) memcpy(epochs_done, P.p_head + 4, (size_t)P.n_ch * sizeof(int32_t));
    if (nav_bits) {
        memcpy(n_bits, P.p_head + 4 + P.n_ch, (size_t)P.n_ch * sizeof(int32_t));
        memcpy(nav_bits, P.p_bits, P.bits_bytes);
    }
    if (states_out) memcpy(states_out, P.p_states, P.st_bytes);
    if (records) memcpy(records, P.p_rec, P.rec_bytes);
    TICK_CLOCK(4);
    TICK_REPORT();
    return SDR_OK;
}

// Shared by sdr_bank_step / sdr_bank_tick: everything between the optional ingest and the final synchronisation.
// defer != nullptr: queue only (records and states are produced whatever the pointers say) and describe the step in *defer.
static int bank_run(sdr_engine* e, sdr_bank* b, StreamCtx* ctx, const int32_t* channels, int n_ch, int n_epochs,
                    sdr_track_epoch* records, sdr_track_state* states_out, int32_t* epochs_done, int8_t* nav_bits,
                    int max_bits, int32_t* n_bits, BankPending* defer = nullptr) {
    TICK_CLOCK(0);
    if (defer) {   // (any non-null value: the queueing part below only asks whether they are wanted)
        records = reinterpret_cast<sdr_track_epoch*>(defer);
        states_out = reinterpret_cast<sdr_track_state*>(defer);
    }
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!e->codes) return sdr_fail(SDR_ERR_STATE, "code slots not allocated");
    if ((nav_bits && (max_bits < 1 || !n_bits)) || (!nav_bits && n_bits))
        return sdr_fail(SDR_ERR_INVALID, "nav_bits, max_bits and n_bits go together");
    if (!channels || n_ch < 1 || n_epochs < 1) return sdr_fail(SDR_ERR_INVALID, "bad bank step request");
    int nt = 0;
    for (int c = 0; c < n_ch; ++c) {
        const int ch = channels[c];
        if (ch < 0 || ch >= b->max_channels || !b->n_taps[ch])
            return sdr_fail(SDR_ERR_INVALID, "entry %d: channel %d is not in the bank", c, ch);
        if (b->slot[ch] >= e->n_slots || e->code_len_host[b->slot[ch]] <= 0)
            return sdr_fail(SDR_ERR_STATE, "entry %d: channel %d's code slot %d is no longer staged", c, ch, b->slot[ch]);
        if (nt && b->n_taps[ch] != nt)
            return sdr_fail(SDR_ERR_UNSUPPORTED, "channels of one step must run the same number of taps (%d vs %d)", b->n_taps[ch], nt);
        nt = b->n_taps[ch];
        for (int d = 0; d < c; ++d)
            if (channels[d] == ch) return sdr_fail(SDR_ERR_INVALID, "channel %d listed twice", ch);
    }
    const size_t rec_bytes = records ? (size_t)n_ch * n_epochs * sizeof(sdr_track_epoch) : 0;
    const size_t st_bytes = (size_t)n_ch * sizeof(sdr_track_state);
    const size_t bits_bytes = nav_bits ? (size_t)n_ch * max_bits : 0;
    const size_t head = ((size_t)3 * n_ch * sizeof(int32_t) + 15) & ~(size_t)15;  // [map][n_bits][epochs_done], then the bits
    // page-locked block: [fault (4 words)][epochs_done n][n_bits n][channel list n][done words n] [states][records][bits]
    const size_t pin_head = ((size_t)(4 + 4 * n_ch) * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t pin_bytes = pin_head + st_bytes + rec_bytes + bits_bytes;
    int rc = sdr_pinned_reserve(e, ctx, pin_bytes);
    if (rc) return rc;
    char* pin = (char*)ctx->pinned;
    int32_t* p_head = (int32_t*)pin;
    sdr_track_state* p_states = (sdr_track_state*)(pin + pin_head);
    sdr_track_epoch* p_rec = (sdr_track_epoch*)(pin + pin_head + st_bytes);
    int8_t* p_bits = (int8_t*)(pin + pin_head + st_bytes + rec_bytes);
    // A receiver tick (or any step with a few KB of results) has the kernel read its channel list from, and write its
    // outputs straight into, the page-locked block -- it is device-accessible -- so the call is one launch and one
    // synchronisation, with no copy commands around them (each costs the stream several microseconds).  Steps with
    // long trajectories keep device buffers and copy once at the end.
    const bool direct = pin_bytes <= 96u * 1024u;
    if (!direct) {
        rc = sdr_devbuf_reserve_on(e, ctx->stream, &ctx->traj, records ? rec_bytes : sizeof(sdr_track_epoch));
        if (!rc) rc = sdr_devbuf_reserve_on(e, ctx->stream, &ctx->bits, head + bits_bytes + 16);
    } else if (!records) {
        rc = sdr_devbuf_reserve_on(e, ctx->stream, &ctx->traj, sizeof(sdr_track_epoch));   // (the kernel's scratch record)
    }
    if (rc) return rc;
    memset(p_head, 0, pin_head);
    memcpy(p_head + 4 + 2 * n_ch, channels, (size_t)n_ch * sizeof(int32_t));  // (the caller's list is not kept past return)
    TrackRun r;
    r.d_states = b->d_states;
    r.d_cfgs = b->d_cfgs;
    r.cfg_stride = 1;
    r.n_ch = n_ch, r.n_epochs = n_epochs, r.n_taps = nt;
    r.keep = records ? 1 : 0;
    r.max_bits = max_bits;
    bool identity = true;             // channels 0 .. n-1 in order (a receiver's usual tick): the kernel needs no list
    for (int c = 0; c < n_ch && identity; ++c) identity = channels[c] == c;
    if (direct) {
        r.d_map = identity ? nullptr : p_head + 4 + 2 * n_ch;   // (a list in page-locked memory costs every workgroup a PCIe round trip)
        r.d_done = p_head + 4;
        r.d_nbits = nav_bits ? p_head + 4 + n_ch : nullptr;
        r.d_bits = nav_bits ? p_bits : nullptr;
        r.d_traj = records ? p_rec : (sdr_track_epoch*)ctx->traj.ptr;
        r.d_states_copy = states_out ? p_states : nullptr;
        r.fault_word = p_head;
        r.done_words = (unsigned*)(p_head + 4 + 3 * n_ch);      // (zeroed with the head above)
        r.done_seq = ++ctx->done_seq ? ctx->done_seq : ++ctx->done_seq;
        if (nav_bits) memset(p_bits, 0, bits_bytes);
    } else {
        int32_t* d_map = (int32_t*)ctx->bits.ptr;
        r.d_map = d_map;
        r.d_traj = (sdr_track_epoch*)ctx->traj.ptr;
        r.d_nbits = nav_bits ? d_map + n_ch : nullptr;
        r.d_done = d_map + 2 * n_ch;
        r.d_bits = nav_bits ? (int8_t*)ctx->bits.ptr + head : nullptr;
        SDR_HIP(hipMemsetAsync(ctx->bits.ptr, 0, head + bits_bytes, ctx->stream));
        SDR_HIP(hipMemcpyAsync(d_map, p_head + 4 + 2 * n_ch, (size_t)n_ch * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    int parts = 1;
    int* d_fault = nullptr;
    TICK_CLOCK(1);
    if (int rc2 = launch_track(e, ctx, r, &parts, &d_fault)) return rc2;
    TICK_CLOCK(2);
    if (!direct) {
        SDR_HIP(hipMemcpyAsync(p_head, d_fault, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        SDR_HIP(hipMemcpyAsync(p_head + 4, r.d_done, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (nav_bits) SDR_HIP(hipMemcpyAsync(p_head + 4 + n_ch, r.d_nbits, (size_t)n_ch * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (states_out) {
            // contiguous runs of channel indices come back in one copy each
            int c = 0;
            while (c < n_ch) {
                int run = 1;
                while (c + run < n_ch && channels[c + run] == channels[c] + run) ++run;
                SDR_HIP(hipMemcpyAsync(p_states + c, b->d_states + channels[c], (size_t)run * sizeof(sdr_track_state),
                                       hipMemcpyDeviceToHost, ctx->stream));
                c += run;
            }
        }
        if (records) SDR_HIP(hipMemcpyAsync(p_rec, ctx->traj.ptr, rec_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (nav_bits) SDR_HIP(hipMemcpyAsync(p_bits, r.d_bits, bits_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    BankPending local;
    BankPending& P = defer ? *defer : local;
    P.active = true, P.ctx = ctx, P.n_ch = n_ch, P.n_epochs = n_epochs, P.parts = parts;
    P.p_head = p_head, P.p_states = p_states, P.p_rec = p_rec, P.p_bits = p_bits;
    P.rec_bytes = rec_bytes, P.st_bytes = st_bytes, P.bits_bytes = bits_bytes;
    P.done_words = r.done_words, P.done_seq = r.done_seq;
    if (defer) return SDR_OK;
    return bank_collect(P, records, states_out, epochs_done, nav_bits, n_bits);
}

int sdr_bank_step(sdr_engine* e, sdr_bank* b, const int32_t* channels, int n_ch, int n_epochs,
                  sdr_track_epoch* records, sdr_track_state* states_out, int32_t* epochs_done,
                  int8_t* nav_bits, int max_bits, int32_t* n_bits, int stream_id) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (b->tick_open) return sdr_fail(SDR_ERR_STATE, "a tick of this bank is in flight: sdr_bank_tick_mirrored_end first");
    StreamCtx* ctx = sdr_stream_ctx(e, stream_id);
    if (!ctx) return sdr_fail(SDR_ERR_INVALID, "stream id %d does not exist", stream_id);
    if (int rc = sdr_iq_order_reader(e, ctx)) return rc;      // (behind the uploads queued on the engine's stream so far)
    return bank_run(e, b, ctx, channels, n_ch, n_epochs, records, states_out, epochs_done, nav_bits, max_bits, n_bits);
}

// sdr_bank_step in two halves.  _begin queues the launch and the copies of its results into page-locked memory on the
// engine's stream and returns; _end waits for them and hands them out.  One step may be in flight per bank; whatever is
// queued on the stream in between (a tick, an upload) runs after the step, as the stream orders it.
int sdr_bank_step_begin(sdr_engine* e, sdr_bank* b, const int32_t* channels, int n_ch, int n_epochs) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (b->tick_open) return sdr_fail(SDR_ERR_STATE, "a tick of this bank is in flight: sdr_bank_tick_mirrored_end first");
    if (!b->pending) b->pending = new BankPending();
    if (b->pending->active) return sdr_fail(SDR_ERR_STATE, "a step of this bank is already in flight: sdr_bank_step_end first");
    b->async_ctx.stream = e->ctx0.stream;
    return bank_run(e, b, &b->async_ctx, channels, n_ch, n_epochs, nullptr, nullptr, nullptr, nullptr, 0, nullptr, b->pending);
}

int sdr_bank_step_end(sdr_engine* e, sdr_bank* b, sdr_track_epoch* records, sdr_track_state* states_out, int32_t* epochs_done) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (!b->pending || !b->pending->active) return sdr_fail(SDR_ERR_STATE, "no step of this bank is in flight");
    return bank_collect(*b->pending, records, states_out, epochs_done, nullptr, nullptr);
}


}  // extern "C"

// ------------------------------------------------------------------------------------------- the tick server, host side
struct TickServerState {
    sdr_bank* bank = nullptr;
    std::vector<int32_t> channels;          // the channels it serves (every tracking channel of the bank when it started), ascending
    int parts = 0, n_taps = 0;
    StreamCtx ctx;                          // the trackers' own stream and exchange lines
    hipStream_t door_stream = nullptr;      // the doorman's
    TickServerHost* host = nullptr;         // page-locked: control words, then [ran][states][records]
    size_t host_bytes = 0;
    int* h_ran = nullptr;
    sdr_track_state* h_st = nullptr;
    sdr_track_epoch* h_rec = nullptr;
    unsigned* h_done = nullptr;             // per served channel: the last request it has answered
    DevBuf dev;                             // TickServerDev, then [release words][records][channel map]
    unsigned seq = 0;                       // requests posted to the running server
    unsigned pull_seq = 0;                  // slabs announced to it
    unsigned stamps_seq = 0;                // the last request whose stamps went into the sums below
    bool disabled = false;                  // a launch was refused, or a server died at work: plain ticks from then on
    int64_t served_total = 0, starts = 0;   // requests answered / servers started, over the engine's life
    double phase_us[4] = {0, 0, 0, 0};      // summed over the answered requests: slab pull, release, channels' answers, gather
    double tracker_us[6] = {0, 0, 0, 0, 0, 0};
    double seen_us[4] = {0, 0, 0, 0};       // (trace build) release -> the doorman sees 1, n/2, n - 1, n answers
    double fence_add_us = 0;                // channel 0: its release + count, as the NEXT tick's stamps tell (t[6] of the tick before)
    unsigned long long prev_t5 = 0;   // channel 0: release -> seen, -> samples visible, -> correlated, -> exchanged, -> updated, -> answered
    int64_t code_generation = -1;
    void* ring = nullptr;
};

static bool server_wait(volatile unsigned* word, unsigned want, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return true;
        if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    }
}

// Every served channel's done word at `want` (bounded; gives up at once when the doorman has left).
#ifdef SDR_SRV_HOSTTRACE
static double g_ht[8]; static long g_htn;
static inline double ht_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HT(k, t0) do { const double t1_ = ht_now(); g_ht[k] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define HT(k, t0) ((void)0)
#endif
static bool server_wait_all(volatile unsigned* words, int n, unsigned want, volatile unsigned* alive, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    int k = 0;
    for (long spins = 0;; ++spins) {
        while (k < n && __atomic_load_n(&words[k], __ATOMIC_ACQUIRE) == want) ++k;
        if (k == n) return true;
        if ((spins & 1023) == 1023) {
            if (!__atomic_load_n(alive, __ATOMIC_ACQUIRE)) return false;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        }
    }
}

// The doorman's stamps of the last request it has closed, into the sums sdr_tick_server_phases reports (once per request;
// the doorman closes a request a microsecond or two after the host has its answers: read when the next one is posted).
static void server_absorb_stamps(TickServerState* s) {
    TickServerHost* h = s->host;
    const unsigned closed = __atomic_load_n(&h->done_seq, __ATOMIC_ACQUIRE);
    if (closed == s->stamps_seq || closed != s->seq) return;
    s->stamps_seq = closed;
    for (int k = 0; k < 4; ++k) s->phase_us[k] += (double)(h->stamps[k + 1] - h->stamps[k]) * 0.01;
    if (h->tracker[5] > h->stamps[2]) {     // (channel 0 ran in this tick)
        s->tracker_us[0] += (double)((long long)(h->tracker[0] - h->stamps[2])) * 0.01;
        for (int k = 1; k < 6; ++k) s->tracker_us[k] += (double)((long long)(h->tracker[k] - h->tracker[k - 1])) * 0.01;
#ifdef SDR_SRV_TRACE
        for (int k = 0; k < 4; ++k) s->seen_us[k] += (double)h->tracker[8 + k] * 0.01;
#endif
        if (h->tracker[6] > h->tracker[5]) s->fence_add_us += (double)(h->tracker[6] - h->tracker[5]) * 0.01, s->prev_t5 += 1;
    }
}

// The request line (see TickServerHost): words 1-5, then the numbers' copy, then the numbers.
static void server_post(TickServerState* s, const unsigned long long* words5) {
    TickServerHost* h = s->host;
    for (int k = 0; k < 5; ++k) h->line[1 + k] = words5[k];
    const unsigned long long seqs = (unsigned long long)s->seq | ((unsigned long long)s->pull_seq << 32);
    __atomic_store_n(&h->line[kReqSeqsCopy], seqs, __ATOMIC_RELEASE);
    __atomic_store_n(&h->line[kReqSeqs], seqs, __ATOMIC_RELEASE);
}
static void server_post_stop(TickServerState* s) {
    __atomic_store_n(&s->host->line[kReqSeqsCopy], (unsigned long long)kServerStop, __ATOMIC_RELEASE);
    __atomic_store_n(&s->host->line[kReqSeqs], (unsigned long long)kServerStop, __ATOMIC_RELEASE);
}

int sdr_tick_server_stop(sdr_engine* e) {
    if (e) e->srv_steady_ticks = 0;
    if (!e || !e->srv_running) return SDR_OK;
    TickServerState* s = e->srv;
    e->srv_running = false;
    server_post_stop(s);
    // (the doorman looks at the word every few microseconds; the trackers follow its release word.  Should the words never be
    // seen -- they always are -- every workgroup still leaves by its own clock: the stream synchronisation below ends either way)
    (void)server_wait(&s->host->alive, 0u, 1.0);
    SDR_HIP(hipStreamSynchronize(s->door_stream));
    SDR_HIP(hipStreamSynchronize(s->ctx.stream));
    server_absorb_stamps(s);
    s->seq = s->pull_seq = s->stamps_seq = 0;
    // a slab the server had not pulled yet: into the ring the ordinary way
    if (e->srv_slab_pending) return sdr_iq_flush_server_slab(e);
    return SDR_OK;
}

void sdr_tick_server_free(sdr_engine* e) {
    if (!e || !e->srv) return;
    (void)sdr_tick_server_stop(e);
    TickServerState* s = e->srv;
    for (DevBuf* d : {&s->ctx.traj, &s->ctx.bits, &s->ctx.xchg, &s->dev})
        if (d->ptr) (void)hipFree(d->ptr);
    if (s->ctx.pinned) (void)hipHostFree(s->ctx.pinned);
    if (s->host) (void)hipHostFree(s->host);
    if (s->ctx.stream) (void)hipStreamDestroy(s->ctx.stream);
    if (s->door_stream) (void)hipStreamDestroy(s->door_stream);
    delete s;
    e->srv = nullptr;
}

// Start a server for `channels` (ascending, one tap count, n <= 64 so that the cluster form applies).  On return the
// kernel is queued on the server's own stream; it raises `alive` when it runs.
static int tick_server_start(sdr_engine* e, sdr_bank* b, const int32_t* channels, int n, int nt) {
    if (!e->srv) e->srv = new TickServerState();
    TickServerState* s = e->srv;
    if (!s->ctx.stream) SDR_HIP(hipStreamCreateWithFlags(&s->ctx.stream, hipStreamNonBlocking));
    if (!s->door_stream) SDR_HIP(hipStreamCreateWithFlags(&s->door_stream, hipStreamNonBlocking));
    int parts = 1;
    while (parts < kMaxParts && (long)n * parts * 2 <= (long)e->n_cus) parts *= 2;     // (the cluster a tick of these channels takes)
    if (parts < 2) return sdr_fail(SDR_ERR_UNSUPPORTED, "tick server: %d channels leave no cluster", n);
    const size_t res_bytes = (size_t)n * (sizeof(int) + sizeof(unsigned) + sizeof(sdr_track_state) + sizeof(sdr_track_epoch));
    const size_t host_bytes = ((sizeof(TickServerHost) + 63) & ~(size_t)63) + res_bytes + 64;
    if (host_bytes > s->host_bytes) {
        if (s->host) SDR_HIP(hipHostFree(s->host));
        s->host = nullptr;
        s->host_bytes = 0;
        hipError_t err = hipHostMalloc((void**)&s->host, host_bytes, hipHostMallocDefault);
        if (err != hipSuccess) {
            s->host = nullptr;
            return sdr_fail(SDR_ERR_NOMEM, "hipHostMalloc(%zu) for the tick server failed: %s", host_bytes, hipGetErrorString(err));
        }
        s->host_bytes = host_bytes;
    }
    memset(s->host, 0, host_bytes);
    char* hp = (char*)s->host + ((sizeof(TickServerHost) + 63) & ~(size_t)63);
    s->h_st = (sdr_track_state*)hp;                                   // (8-byte fields first: the channels write 8 bytes per lane)
    s->h_rec = (sdr_track_epoch*)(hp + (size_t)n * sizeof(sdr_track_state));
    s->h_ran = (int*)(hp + (size_t)n * (sizeof(sdr_track_state) + sizeof(sdr_track_epoch)));
    s->h_done = (unsigned*)(s->h_ran + n);
    const size_t dev_head = (sizeof(TickServerDev) + 127) & ~(size_t)127;
    const size_t go_bytes = (size_t)n * kGoStride * sizeof(unsigned);
    const size_t dev_bytes = dev_head + go_bytes + (size_t)n * (sizeof(sdr_track_epoch) + sizeof(int32_t)) + 64;
    if (int rc = sdr_devbuf_reserve_on(e, s->ctx.stream, &s->dev, dev_bytes)) return rc;
    SDR_HIP(hipMemsetAsync(s->dev.ptr, 0, dev_bytes, s->ctx.stream));
    char* dp = (char*)s->dev.ptr + dev_head;
    TickServer a = {};
    a.host = s->host;
    a.dev = (TickServerDev*)s->dev.ptr;
    a.go = (unsigned*)dp;
    a.ring16 = (uint4*)e->iq;
    a.ring_flip = e->iq_fmt == SDR_FMT_CI8 ? sdr::kCi8Flip : 0u;
    a.ring_n16 = (unsigned long long)((size_t)e->iq_capacity * sdr_fmt_bytes(e->iq_fmt) / 16);
    a.rec_out = (sdr_track_epoch*)(dp + go_bytes);
    int32_t* d_map = (int32_t*)(dp + go_bytes + (size_t)n * sizeof(sdr_track_epoch));
    a.h_ran = s->h_ran, a.h_st = s->h_st, a.h_rec = s->h_rec, a.h_done = s->h_done;
    a.idle_ticks = 20000000ull;      // 0.2 s of the 100 MHz wall clock without a request: leave (the next tick starts a new server)
    a.busy_ticks = 5000000ull;       // 50 ms for the channels' answers to one request: something is wrong, leave
    SDR_HIP(hipMemcpyAsync(d_map, channels, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s->ctx.stream));
    if (int rc = sdr_devbuf_reserve_on(e, s->ctx.stream, &s->ctx.traj, sizeof(sdr_track_epoch))) return rc;
    TrackRun r;
    r.d_states = b->d_states;
    r.d_cfgs = b->d_cfgs;
    r.cfg_stride = 1;
    r.d_map = d_map;
    r.n_ch = n, r.n_epochs = 0x7fffffff, r.n_taps = nt;
    r.keep = 0;
    r.d_traj = (sdr_track_epoch*)s->ctx.traj.ptr;
    r.fault_word = &a.dev->fault;
    r.force_parts = parts;
    r.server = &a;
    int used = 0;
    int* d_fault = nullptr;
    // the trackers first (a cooperative launch: it goes through or is refused as a whole), then the doorman; trackers without
    // a doorman leave by their own clock
    SDR_HIP(hipStreamSynchronize(s->ctx.stream));      // (the control block is zero, the channel list in place)
    if (int rc = launch_track(e, &s->ctx, r, &used, &d_fault)) {
        (void)hipGetLastError();                       // (a refused launch must not surface at somebody else's check)
        return rc;
    }
    hipLaunchKernelGGL(tick_doorman_kernel, dim3(kDoorGroups), dim3(kDoorThreads), 0, s->door_stream, a, n);
    const hipError_t door_err = hipGetLastError();
    // the doorman has to be RESIDENT beside the trackers (they fill the device): it says so itself
    if (door_err != hipSuccess || !server_wait(&s->host->alive, 1u, 0.05)) {
        // no doorman: tell it (should it still arrive) and the trackers (they poll the device word) to leave, wait for them
        server_post_stop(s);
        (void)hipMemsetAsync(a.go, 0xFF, go_bytes, e->ctx0.stream);
        (void)hipStreamSynchronize(e->ctx0.stream);
        (void)hipStreamSynchronize(s->ctx.stream);
        (void)hipStreamSynchronize(s->door_stream);
        return sdr_fail(SDR_ERR_HIP, "tick server: the doorman did not become resident beside the trackers (%s)",
                        door_err != hipSuccess ? hipGetErrorString(door_err) : "no room on a compute unit");
    }
    s->bank = b;
    s->channels.assign(channels, channels + n);
    s->parts = parts, s->n_taps = nt;
    s->seq = s->pull_seq = s->stamps_seq = 0;
    s->code_generation = e->code_generation;
    s->ring = e->iq;
    s->starts += 1;
    e->srv_running = true;
    return SDR_OK;
}

extern "C" {

int sdr_bank_tick(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                  const int32_t* channels, int n_ch, sdr_track_epoch* records, sdr_track_state* states_out,
                  int32_t* epochs_done) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (b->tick_open) return sdr_fail(SDR_ERR_STATE, "a tick of this bank is in flight: sdr_bank_tick_mirrored_end first");
    if (n_samples > 0)
        if (int rc = sdr_iq_upload_async(e, iq, n_samples, ring_offset)) return rc;
    if (n_ch == 0) {
        SDR_HIP(hipStreamSynchronize(e->ctx0.stream));
        return SDR_OK;
    }
    return bank_run(e, b, &e->ctx0, channels, n_ch, 1, records, states_out, epochs_done, nullptr, 0, nullptr);
}


// The tick with its bookkeeping here instead of in the caller's language: which channels are ready (channel.py:137-146
// -- the ring holds their next epoch completely), their epoch, and the caller's mirrors brought up to date in place.
// In two halves so that ONE host thread can drive several devices (channelManager.py:149-188 starts every channel
// before it waits for any): _begin decides who is ready and queues their epoch -- nothing is waited for -- and _end waits,
// absorbs the results into the mirrors and writes the tick's update rows.  sdr_bank_tick_mirrored is the two in a row.
int sdr_bank_tick_mirrored_begin(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                                 int64_t write_index, sdr_tick_mirror* m) {
#ifdef SDR_SRV_HOSTTRACE
    double ht0 = ht_now();
#endif
    if (int rc = sdr_set_device_keep(e)) return rc;        // (a resident tick server stays: this may be its next request)
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (b->tick_open) return sdr_fail(SDR_ERR_STATE, "a tick of this bank is already in flight: sdr_bank_tick_mirrored_end first");
    if (!m || !m->states || !m->last || !m->tracking || !m->lost || !m->ran || !m->records || !m->updates)
        return sdr_fail(SDR_ERR_INVALID, "tick mirror: a required array is NULL");
    if (m->max_channels != b->max_channels)
        return sdr_fail(SDR_ERR_INVALID, "tick mirror has %d rows, the bank %d channels", m->max_channels, b->max_channels);
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    const int64_t cap = e->iq_capacity;
    if (write_index < 0 || write_index >= cap) return sdr_fail(SDR_ERR_RANGE, "write index outside the ring");
    if (n_samples > 0)
        if (int rc = sdr_iq_upload_async(e, iq, n_samples, ring_offset)) return rc;
    auto unread_of = [&](int ch) {
        int64_t cur = m->states[ch].current_sample % cap;
        if (cur < 0) cur += cap;
        return cur <= write_index ? write_index - cur : cap - cur + write_index;   // circularbuffer.py:139-148
    };
    // ready channels, grouped by tap count (one launch per group: the kernels are compiled per tap count)
    std::vector<int32_t>& list = b->tick_list;
    m->n_ran = m->n_nav_bits = m->n_lost = 0;
    int taps_seen[2] = {0, 0};
    list.clear();
    for (int ch = 0; ch < b->max_channels; ++ch) {
        if (!m->tracking[ch] || m->lost[ch] || !b->n_taps[ch]) continue;
        if (unread_of(ch) < m->states[ch].n_samples) continue;
        list.push_back(ch);
        const int nt = b->n_taps[ch];
        if (taps_seen[0] == 0 || taps_seen[0] == nt) taps_seen[0] = nt;
        else taps_seen[1] = nt;
    }
    b->tick_write_index = write_index;
    b->tick_slab_queued = n_samples > 0;
    b->tick_rc = SDR_OK;
    b->tick_groups = 0;
    b->tick_two = taps_seen[1] != 0;
    b->tick_served = false;
    // ---- the resident tick server ("tick_server"): the request goes to the kernel that is already there
    if (e->tick_server_opt && !(e->srv && e->srv->disabled) && !b->tick_two && !e->track_force_parts && !e->track_one_launch_tick &&
        !e->prof && !(b->pending && b->pending->active)) {
        // the server serves EVERY tracking channel of the bank (who is ready is decided there, as it is above)
        std::vector<int32_t>& cand = b->tick_candidates;
        cand.clear();
        int nt = 0;
        bool one_tap_count = true;
        for (int ch = 0; ch < b->max_channels; ++ch) {
            if (!m->tracking[ch] || m->lost[ch] || !b->n_taps[ch]) continue;
            cand.push_back(ch);
            if (nt && b->n_taps[ch] != nt) one_tap_count = false;
            nt = b->n_taps[ch];
        }
        TickServerState* s = e->srv;
        if (e->srv_running && !__atomic_load_n(&s->host->alive, __ATOMIC_ACQUIRE)) {
            // the server has left by its own clock (0.2 s without a request: a host that paused) -- or given up: tidy up after
            // it (its streams, the channels' states it wrote on the way out); this tick and the next few are plain ones
            const unsigned why = __atomic_load_n(&s->host->fault, __ATOMIC_ACQUIRE);
            if (int rc = sdr_tick_server_stop(e)) return rc;
            if (why >= 2) s->disabled = true;
        }
        const bool same = e->srv_running && s->bank == b && s->channels == cand && s->code_generation == e->code_generation &&
                          s->ring == e->iq;
        bool use = one_tap_count && !cand.empty() && (long)cand.size() * 4 <= (long)e->n_cus;
        // A server costs ~25 ms to start (a cooperative launch, the doormen becoming resident): it is started for a receiver
        // that HAS settled into steady ticks -- eight in a row with nothing else on the engine in between -- not for the
        // ticks between two other calls (a manager that replays read-ahead blocks uploads and steps in between).
        if (use && !same && ++e->srv_steady_ticks < 8) use = false;
        if (use && !same) {
            if (e->srv_running)
                if (int rc = sdr_tick_server_stop(e)) return rc;
            // (what is queued on the engine's stream -- an ingest, the states a put uploaded -- is done before the server reads it)
            SDR_HIP(hipStreamSynchronize(e->ctx0.stream));
            if (int rc = tick_server_start(e, b, cand.data(), (int)cand.size(), nt)) {
                (void)rc;                       // refused (no room for the cooperative launch, ...): plain ticks from now on
                if (e->srv) e->srv->disabled = true;
                use = false;
            }
        }
        if (!use && e->srv_running)
            if (int rc = sdr_tick_server_stop(e)) return rc;
        if (use) {
            s = e->srv;
            // a slab that went into the ring the ordinary way while the server was resident (a second sdr_iq_upload_begin before
            // this tick: sdr_iq_flush_server_slab) is an ingest kernel on the engine's stream: it has to be IN the ring before
            // the trackers are released on it
            if (e->slab_busy[0] || e->slab_busy[1]) {
                SDR_HIP(hipStreamSynchronize(e->ctx0.stream));
                e->slab_busy[0] = e->slab_busy[1] = false;
            }
            HT(0, ht0);                       // begin: checks, ready list
            server_absorb_stamps(s);          // (the request before this one: the doorman has closed it long since)
            HT(1, ht0);                       // absorb
            // The slab this tick brought (staged by sdr_iq_upload_begin while the server was resident) is announced WITH the
            // request, not when it was staged: pulled earlier, its 800 cache lines are taken out of the host core's cache while
            // the caller is still at work between the two calls -- measured from Python: 43.1 us per tick against 36.4.
            unsigned long long words[5] = {(unsigned long long)write_index, 0, 0, 0, 0};
            if (e->srv_slab_pending) {
                const size_t sb = sdr_fmt_bytes(e->iq_fmt);
                words[1] = (unsigned long long)((uintptr_t)e->srv_slab_src / 16);
                words[2] = (unsigned long long)((size_t)e->srv_slab_n * sb / 16);
                words[3] = (unsigned long long)((size_t)e->srv_slab_off * sb / 16);
                words[4] = ++s->pull_seq;
            }
            ++s->seq;
            server_post(s, words);
            HT(2, ht0);                       // post
            b->tick_served = true;
            b->tick_open = true;
            return SDR_OK;
        }
    } else if (e->srv_running) {
        if (int rc = sdr_tick_server_stop(e)) return rc;
    } else {
        e->srv_steady_ticks = 0;
    }
    if (list.empty() && e->srv_slab_pending)        // (nobody to take it along: the slab goes into the ring the ordinary way)
        if (int rc = sdr_iq_flush_server_slab(e)) return rc;
    if (!list.empty()) {
        int n = (int)list.size();
        b->tick_states.resize((size_t)n);
        b->tick_done.resize((size_t)n);
        if (!b->tick_pending[0]) b->tick_pending[0] = new BankPending();
        if (b->tick_two && !b->tick_pending[1]) b->tick_pending[1] = new BankPending();
        b->tick_ctx[0].stream = b->tick_ctx[1].stream = e->ctx0.stream;      // (the engine's stream, the bank's scratch)
        int at = 0;
        for (int g = 0; g < 2 && taps_seen[g]; ++g) {
            // (stable partition: the group's channels to the front of the remaining range, ascending)
            int n_g = n - at;
            if (taps_seen[1]) {
                n_g = (int)(std::stable_partition(list.begin() + at, list.end(),
                                                  [&](int32_t ch) { return b->n_taps[ch] == taps_seen[g]; }) -
                            (list.begin() + at));
            }
            if (int rc = bank_run(e, b, &b->tick_ctx[g], list.data() + at, n_g, 1, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                  b->tick_pending[g])) {
                // a group queued before this one WILL advance its channels on the device: its results are absorbed by _end
                // (the caller's mirrors must keep matching the bank), which then returns the error
                if (at == 0) return rc;
                b->tick_rc = rc;
                list.resize((size_t)at);
                break;
            }
            b->tick_group_n[g] = n_g;
            b->tick_groups = g + 1;
            at += n_g;
        }
    }
    b->tick_open = true;
    return SDR_OK;
}

int sdr_bank_tick_mirrored_end(sdr_engine* e, sdr_bank* b, sdr_tick_mirror* m) {
    if (int rc = sdr_set_device_keep(e)) return rc;
    if (!b) return sdr_fail(SDR_ERR_INVALID, "bank is NULL");
    if (!b->tick_open) return sdr_fail(SDR_ERR_STATE, "no tick of this bank is in flight");
    if (!m || !m->states || !m->last || !m->tracking || !m->lost || !m->ran || !m->records || !m->updates || m->max_channels != b->max_channels)
        return sdr_fail(SDR_ERR_INVALID, "tick mirror: not the one the tick was begun with");
    b->tick_open = false;
    const int64_t cap = e->iq_capacity, write_index = b->tick_write_index;
    auto unread_of = [&](int ch) {
        int64_t cur = m->states[ch].current_sample % cap;
        if (cur < 0) cur += cap;
        return cur <= write_index ? write_index - cur : cap - cur + write_index;
    };
    std::vector<int32_t>& list = b->tick_list;
    int run_rc = b->tick_rc;
    if (b->tick_served) {
        // ---- the answer of the resident server: wait for its done word (bounded), take the listed channels' rows
        TickServerState* s = e->srv;
        TickServerHost* h = s->host;
#ifdef SDR_SRV_HOSTTRACE
        double ht0 = ht_now();
#endif
        const bool answered = server_wait_all(s->h_done, (int)s->channels.size(), s->seq, &h->alive, 0.25);
        HT(3, ht0);                           // wait
        if (!answered) {
            const unsigned alive = __atomic_load_n(&h->alive, __ATOMIC_ACQUIRE), why = __atomic_load_n(&h->fault, __ATOMIC_ACQUIRE);
            if (!alive && why <= 1) {
                // the doorman left by its idle clock just as this request was posted: it never saw it (a doorman that sees a
                // request answers it), no channel has moved.  The server is tidied up and the tick done over the plain way.
                if (int rc = sdr_tick_server_stop(e)) return rc;
                b->tick_served = false;
                if (int rc = sdr_bank_tick_mirrored_begin(e, b, nullptr, 0, 0, b->tick_write_index, m)) return rc;
                return sdr_bank_tick_mirrored_end(e, b, m);
            }
            (void)sdr_tick_server_stop(e);
            s->disabled = true;
            return sdr_fail(SDR_ERR_HIP, "the resident tick server did not answer request %u (alive %u, fault %u): stopped; plain "
                                         "ticks from now on -- the bank's channels may have advanced, read them back (sdr_bank_get)",
                            s->seq, alive, why);
        }
        s->served_total += 1;
        if (e->srv_slab_pending) {              // (the doorman has pulled it: its staging half is free again)
            e->srv_slab_pending = false;
            if (e->srv_slab_half >= 0) e->slab_busy[e->srv_slab_half] = false;
        }
        bool xchg_fault = __atomic_load_n(&h->fault, __ATOMIC_ACQUIRE) != 0;
        for (size_t c = 0; c < s->channels.size(); ++c) xchg_fault = xchg_fault || s->h_ran[c] == -2;
        if (xchg_fault) {
            (void)sdr_tick_server_stop(e);
            s->disabled = true;
            return sdr_fail(SDR_ERR_HIP, "closed-loop tracking: a workgroup of a %d-part cluster never published its sums (tick server)", s->parts);
        }
        const int n = (int)list.size();
        b->tick_states.resize((size_t)n);
        b->tick_done.resize((size_t)n);
        size_t k = 0;
        int i = 0;
        for (size_t c = 0; c < s->channels.size(); ++c) {
            const int ch = s->channels[c];
            const bool listed = i < n && list[(size_t)i] == ch;
            const int ran = s->h_ran[c];
            if (listed != (ran != 0)) {
                (void)sdr_tick_server_stop(e);
                s->disabled = true;
                return sdr_fail(SDR_ERR_STATE, "tick server: channel %d %s on the device but the mirror says otherwise -- the mirror and "
                                               "the bank disagree", ch, ran ? "ran" : "did not run");
            }
            if (!listed) continue;
            b->tick_states[(size_t)i] = s->h_st[c];
            b->tick_done[(size_t)i] = ran == 1 ? 1 : 0;
            m->records[i] = s->h_rec[c];
            if (ran != 1) b->tick_states[(size_t)i] = m->states[ch];        // (stopped before its epoch: the state it had)
            ++i, ++k;
        }
        if (i != n) {
            (void)sdr_tick_server_stop(e);
            s->disabled = true;
            return sdr_fail(SDR_ERR_STATE, "tick server: a listed channel is not among the channels it serves");
        }
        b->tick_two = false;
        b->tick_groups = 0;
        HT(4, ht0);                           // answers copied
#ifdef SDR_SRV_HOSTTRACE
        if (++g_htn % 500 == 0) {
            fprintf(stderr, "host trace (us per tick): begin %.2f absorb %.2f post %.2f wait %.2f copy %.2f\n", g_ht[0] / 500, g_ht[1] / 500,
                    g_ht[2] / 500, g_ht[3] / 500, g_ht[4] / 500);
            for (double& v : g_ht) v = 0;
        }
#endif
    } else if (list.empty()) {
        // (no channel ready: nothing of this tick waits for the stream -- but a slab queued with the tick, or one read IN PLACE
        // out of the caller's page-locked block by an ingest kernel, must be in the ring before the caller has its buffer back)
        if (b->tick_slab_queued || e->inplace_slab_in_flight) SDR_HIP(hipStreamSynchronize(e->ctx0.stream));
    }
    e->inplace_slab_in_flight = false;   // (ready channels: their kernel ran behind the ingest kernel on the same stream)
    if (!list.empty()) {
        int n = (int)list.size();
        int at = 0;
        for (int g = 0; g < b->tick_groups; ++g) {
            const int n_g = b->tick_group_n[g];
            if (int rc = bank_collect(*b->tick_pending[g], m->records + at, b->tick_states.data() + at, b->tick_done.data() + at, nullptr,
                                      nullptr)) {
                if (b->tick_pending[1]) b->tick_pending[1]->active = false;
                return rc;      // (a cluster that never published: the device's state is unknown, nothing to absorb)
            }
            at += n_g;
        }
        if (b->tick_two) {   // back to ascending channel order, records and states with them
            std::vector<int> order((size_t)n);
            for (int i = 0; i < n; ++i) order[(size_t)i] = i;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return list[(size_t)x] < list[(size_t)y]; });
            std::vector<int32_t> l2((size_t)n), d2((size_t)n);
            std::vector<sdr_track_state> s2((size_t)n);
            std::vector<sdr_track_epoch> r2((size_t)n);
            for (int i = 0; i < n; ++i) {
                const size_t o = (size_t)order[(size_t)i];
                l2[(size_t)i] = list[o], d2[(size_t)i] = b->tick_done[o], s2[(size_t)i] = b->tick_states[o], r2[(size_t)i] = m->records[o];
            }
            list = l2, b->tick_done = d2, b->tick_states = s2;
            memcpy(m->records, r2.data(), (size_t)n * sizeof(sdr_track_epoch));
        }
        // mirrors; a channel the device parked (its NCO left the replica / the ring) reports no epoch
        int w = 0;
        for (int i = 0; i < n; ++i) {
            const int ch = list[(size_t)i];
            m->states[ch] = b->tick_states[(size_t)i];
            if (b->tick_done[(size_t)i] < 1) {
                m->lost[ch] = 1;
                ++m->n_lost;
                continue;
            }
            if (w != i) m->records[w] = m->records[i];
            m->last[ch] = m->records[w];
            if (m->epochs_since_tow) m->epochs_since_tow[ch] += 1;
            if (m->records[w].nav_bit >= 0) ++m->n_nav_bits;
            m->ran[w++] = ch;
        }
        m->n_ran = w;
    }
    // what each tracking channel's CHANNEL_UPDATE reports after this tick (channel.py:205-228)
    int nu = 0;
    int64_t max_unread = 0;
    for (int ch = 0; ch < b->max_channels; ++ch) {
        if (!m->tracking[ch]) continue;
        sdr_tick_update& u = m->updates[nu++];
        u.channel = ch;
        u.track_flags = m->states[ch].track_flags | (m->host_flags ? (int32_t)m->host_flags[ch] : 0);
        u.unread = unread_of(ch);
        u.epochs_since_tow = m->epochs_since_tow ? m->epochs_since_tow[ch] : 0;
        if (!m->lost[ch] && u.unread > max_unread) max_unread = u.unread;
    }
    m->n_updates = nu;
    m->max_unread = max_unread;
    return run_rc;
}

int sdr_bank_tick_mirrored(sdr_engine* e, sdr_bank* b, const void* iq, int64_t n_samples, int64_t ring_offset,
                           int64_t write_index, sdr_tick_mirror* m) {
    if (int rc = sdr_bank_tick_mirrored_begin(e, b, iq, n_samples, ring_offset, write_index, m)) return rc;
    return sdr_bank_tick_mirrored_end(e, b, m);
}

int sdr_tick_server_stats(sdr_engine* e, int64_t* out4) {
    if (!e || !out4) return sdr_fail(SDR_ERR_INVALID, "null engine or output");
    out4[0] = e->srv_running ? 1 : 0;
    out4[1] = e->srv ? e->srv->served_total : 0;
    out4[2] = e->srv ? e->srv->starts : 0;
    out4[3] = e->srv && e->srv->disabled ? 1 : 0;
    return SDR_OK;
}

// Where the served requests' time went on the device, in microseconds summed over them (the doorman's wall-clock stamps):
// {the slab's shares in the ring (pulled since it was staged: what is left of that when the request arrives), channels released,
// every channel has answered (the host has the answers by then: the channels write them themselves), the request closed}.
int sdr_tick_server_phases(sdr_engine* e, double* out4) {
    if (!e || !out4) return sdr_fail(SDR_ERR_INVALID, "null engine or output");
    if (e->srv && e->srv_running) server_absorb_stamps(e->srv);
    for (int k = 0; k < 4; ++k) out4[k] = e->srv ? e->srv->phase_us[k] : 0.0;
    return SDR_OK;
}

// ... and channel 0's own tick (lane 0 of its first part), the same way: {release seen, samples visible (L2 invalidated),
// correlated, sums exchanged, loops updated, answer written}.
int sdr_tick_server_tracker_phases(sdr_engine* e, double* out6) {
    if (!e || !out6) return sdr_fail(SDR_ERR_INVALID, "null engine or output");
    if (e->srv && e->srv_running) server_absorb_stamps(e->srv);
    for (int k = 0; k < 6; ++k) out6[k] = e->srv ? e->srv->tracker_us[k] : 0.0;
#ifdef SDR_SRV_TRACE
    if (e->srv && e->srv->served_total)
        fprintf(stderr, "tick server: the doorman sees 1 / half / all but one / all answers %.2f / %.2f / %.2f / %.2f us after the release\n",
                e->srv->seen_us[0] / e->srv->served_total, e->srv->seen_us[1] / e->srv->served_total, e->srv->seen_us[2] / e->srv->served_total,
                e->srv->seen_us[3] / e->srv->served_total);
#endif
#ifdef SDR_SRV_TRACE
    if (e->srv && e->srv->dev.ptr) {
        TickServerDev d;
        if (hipMemcpy(&d, e->srv->dev.ptr, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "tick server: per channel, us after the request was seen: release seen / answered  [xcc:hw_id of the recording part]\n");
            fprintf(stderr, "  doormen at");
            for (int g = 0; g < 8; ++g) fprintf(stderr, " [%u:%04x]", d.door_where[g] >> 16, d.door_where[g] & 0xffff);
            fprintf(stderr, "\n");
            for (int c = 0; c < 64; ++c)
                if (d.ch_n[c])
                    fprintf(stderr, "  ch %2d: %6.2f %6.2f  [%u:%04x]\n", c, (double)d.ch_gate[c] / d.ch_n[c] * 0.01, (double)d.ch_done[c] / d.ch_n[c] * 0.01,
                            d.ch_where[c] >> 16, d.ch_where[c] & 0xffff);
        }
    }
    if (e->srv) fprintf(stderr, "tick server: channel 0 release fence + count: %.2f us (mean of %llu)\n", e->srv->fence_add_us / (double)(e->srv->prev_t5 ? e->srv->prev_t5 : 1), e->srv->prev_t5);
#endif
    return SDR_OK;
}

int sdr_iq_upload_begin(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset) {
    if (!e) return sdr_fail(SDR_ERR_INVALID, "null engine");
    return sdr_iq_upload_async(e, iq, n_samples, ring_offset);
}

}  // extern "C"
#endif  // SDR_TRACK_DENSE_TU
