/* The reference receiver's outer loop (sydr/receiver/receiver.py:120-131: addNewRFData of one millisecond, then run)
 * from plain C, one call each per tick:
 *     sdr_iq_upload_begin     -- the slab is copied out of the caller's buffer and queued for the ring (CircularBuffer.shift)
 *     sdr_bank_tick_mirrored  -- who is ready (channel.py:137-146), one epoch for them on the device, the caller's mirrors of
 *                                the channel bank and the rows the packets report (channelManager.py:149-188) updated in place
 * 32 channels @ 25 MHz: acquisition of the first millisecond, then `ticks` milliseconds tracked tick by tick with the host as
 * IQ source; prints where the channels ended and the time per tick.  A second argument `server` switches the resident
 * tick server on (sdr_set_option "tick_server"): the same two calls per tick, answered by a kernel that is already there.
 * The thread is kept on the CPUs next to the GPU (sdr_set_option "bind_thread_to_device"; `nobind` leaves it where the
 * scheduler puts it: ~6 us per tick more from the other socket of a two-socket host).  `pinned`: the recording in page-locked
 * memory (sdr_host_alloc): its slabs are read in place, ~2 us per tick less.
 * Build:  gcc -std=c99 -O2 -Iinclude examples/receiver_loop.c -Lsydr_amd -lsydr_amd -lm -Wl,-rpath,'$ORIGIN/../sydr_amd' -o examples/receiver_loop
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sydr_amd.h"

#define CK(call)                                                              \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != SDR_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sdr_last_error());  \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}

enum { N_CH = 32 };

int main(int argc, char** argv) {
    const double fs = 25e6, code_rate = 1.023e6;
    const int ticks = argc > 1 ? atoi(argv[1]) : 400;
    int server = 0, nobind = 0, pinned = 0;              /* any of `server`, `nobind`, `pinned` after the tick count */
    for (int k = 2; k < argc; ++k) {
        server |= !strcmp(argv[k], "server");
        nobind |= !strcmp(argv[k], "nobind");
        pinned |= !strcmp(argv[k], "pinned");
    }
    const int spms = (int)(fs * 1e-3);                   /* samples per millisecond */
    const int64_t ring = 100 * (int64_t)spms;            /* the reference's 100 ms ring (channelManager.py:57) */
    const int64_t total = (int64_t)(ticks + 2) * spms;
    sdr_engine* e = NULL;
    CK(sdr_engine_create(0, &e));
    if (!nobind) (void)sdr_set_option(e, "bind_thread_to_device", 1);       /* (best effort: no sysfs, no binding) */
    CK(sdr_code_slots(e, N_CH, 1023));
    sdr_synth_sat sats[N_CH];
    int32_t slots[N_CH];
    for (int c = 0; c < N_CH; ++c) {
        slots[c] = c;
        CK(sdr_code_gps_l1ca(e, c, c + 1));
        sats[c].prn = c + 1, sats[c].flags = 0;
        sats[c].doppler_hz = -4000.0 + 250.0 * c + 20.0, sats[c].code_phase = 31.7 * c + 3.25;
        sats[c].carrier_phase = 0.03 * c, sats[c].amplitude = 5.0;
    }
    /* the stream: synthesised on the device into a ring large enough for all of it, brought to the host (the IQ source) */
    CK(sdr_iq_alloc(e, total, SDR_FMT_CI8));
    CK(sdr_iq_synth(e, sats, N_CH, fs, 12.0, 20260004ull, 0, total));
    /* `pinned`: the samples in page-locked memory of the engine's (as a front end's DMA buffer would be): the slabs are then read
     * in place by the tick's launch instead of being copied into a staging buffer first */
    int8_t* stream = NULL;
    if (pinned) CK(sdr_host_alloc(e, (size_t)total * 2, (void**)&stream));
    else stream = (int8_t*)malloc((size_t)total * 2);
    CK(sdr_iq_download(e, stream, total, 0));
    CK(sdr_iq_alloc(e, ring, SDR_FMT_CI8));              /* the receiver's ring */
    CK(sdr_code_slots(e, N_CH, 1023));                   /* (the tables go with the ring's engine state: staged again) */
    for (int c = 0; c < N_CH; ++c) CK(sdr_code_gps_l1ca(e, c, c + 1));

    /* tick 0: the first millisecond enters the ring and is searched */
    int64_t write_index = 0;
    CK(sdr_iq_upload(e, stream, spms, write_index));
    write_index = (write_index + spms) % ring;
    int64_t bin[N_CH], code[N_CH];
    double ratio[N_CH];
    CK(sdr_pcps(e, slots, N_CH, 0, fs, 0.0, 5000.0, 250.0, 1, 1, bin, code, ratio, NULL, NULL));

    sdr_loop_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.loop_kind = 1, cfg.n_taps = 3, cfg.fs = fs;
    const double sp[3] = {-0.5, 0.0, 0.5};
    for (int t = 0; t < 3; ++t) cfg.spacing_wide[t] = cfg.spacing_narrow[t] = sp[t];
    const double zeta = 0.7, wn = 2.0 * 8.0 * zeta / (4.0 * zeta * zeta + 1.0);
    cfg.dll_tau1 = 1.0 / (wn * wn), cfg.dll_tau2 = 2.0 * zeta / wn, cfg.dll_pdi = 1e-3, cfg.dll_threshold = 10.0;
    cfg.fll_bw_pullin = 100.0, cfg.fll_bw_wide = 50.0, cfg.fll_bw_narrow = 15.0, cfg.fll_thr_wide = 0.5, cfg.fll_thr_narrow = 0.8;
    cfg.pll_bw_wide = 25.0, cfg.pll_bw_narrow = 15.0, cfg.pll_thr_wide = 0.5, cfg.pll_thr_narrow = 0.8;

    /* the mirrors: plain arrays of the caller, one row per channel of the bank */
    static sdr_track_state states[N_CH];
    static sdr_track_epoch last[N_CH], records[N_CH];
    static sdr_tick_update updates[N_CH];
    static int64_t since_tow[N_CH], host_flags[N_CH];
    static uint8_t tracking[N_CH], lost[N_CH];
    static int32_t ran[N_CH];
    sdr_bank* bank = NULL;
    CK(sdr_bank_create(e, N_CH, &bank));
    for (int c = 0; c < N_CH; ++c) {
        sdr_track_state* s = &states[c];
        s->code_slot = c, s->code_hz = code_rate, s->code_step = code_rate / fs;
        s->n_samples = (int)ceil(1023.0 / s->code_step);
        s->carrier_hz = 0.0 - (-5000.0 + 250.0 * (double)bin[c]);            /* postAcquisitionUpdate, channel_l1ca_kaplan.py:217-235 */
        s->current_sample = (int64_t)spms - s->n_samples + code[c] + 1;      /*  */
        s->fll_bw = cfg.fll_bw_pullin, s->pll_bw = cfg.pll_bw_wide, s->lock_state = 1;
        CK(sdr_bank_put(e, bank, c, s, &cfg));
        tracking[c] = 1;
    }
    sdr_tick_mirror m;
    memset(&m, 0, sizeof m);
    m.max_channels = N_CH;
    m.states = states, m.last = last, m.epochs_since_tow = since_tow, m.tracking = tracking, m.lost = lost, m.host_flags = host_flags;
    m.ran = ran, m.records = records, m.updates = updates;

    if (server) CK(sdr_set_option(e, "tick_server", 1));
    long epochs = 0, bits = 0;
    double t_ticks = 0.0;
    for (int k = 1; k <= ticks; ++k) {
        const double t0 = now_us();
        /* addNewRFData(rfSignal.getMilliseconds(1)) */
        CK(sdr_iq_upload_begin(e, stream + (size_t)k * spms * 2, spms, write_index));
        write_index = (write_index + spms) % ring;
        /* run(): records[0 .. n_ran) / ran[] are this tick's TRACKING_UPDATE packets, updates[0 .. n_updates) its CHANNEL_UPDATEs */
        CK(sdr_bank_tick_mirrored(e, bank, NULL, 0, 0, write_index, &m));
        if (k > 100) t_ticks += now_us() - t0;           /* (the first ticks load the kernels' code objects) */
        epochs += m.n_ran, bits += m.n_nav_bits;
        if (m.n_lost) fprintf(stderr, "tick %d: %d channel(s) parked\n", k, m.n_lost);
    }
    int locked = 0;
    for (int c = 0; c < N_CH; ++c) {
        /* (a data-bit edge inside the searched millisecond can put the search one bin off, where a Costas loop with 1 ms
         * epochs locks 500 Hz from the carrier just as well -- the reference does the same: counted as tracking) */
        const double err = fabs(states[c].carrier_hz - sats[c].doppler_hz);
        const int ok = (err < 25.0 || fabs(err - 500.0) < 25.0) && !lost[c];
        locked += ok;
        if (c < 4 || !ok)
            printf("PRN %2d: carrier %+9.2f Hz (true %+8.1f)  lock state %d  flags %d  unread %lld\n", c + 1, states[c].carrier_hz,
                   sats[c].doppler_hz, states[c].lock_state, states[c].track_flags, (long long)updates[c].unread);
    }
    printf("%d ticks, %ld epochs, %ld navigation bits, %d of %d channels on their Doppler (or its 500 Hz alias)\n", ticks, epochs, bits, locked, N_CH);
    if (ticks > 100) printf("%.1f us per tick = %.1f x real time\n", t_ticks / (ticks - 100), 1000.0 / (t_ticks / (ticks - 100)));
    if (server) {
        int64_t st[4];
        CK(sdr_tick_server_stats(e, st));
        printf("tick server: %lld requests answered, %lld server(s) started%s\n", (long long)st[1], (long long)st[2], st[3] ? ", gave up" : "");
        if (st[3] || st[1] < ticks - 10) return 4;        /* (a server starts once eight steady ticks have passed) */
    }
    sdr_bank_destroy(e, bank);
    if (pinned) CK(sdr_host_free(e, stream));
    else free(stream);
    sdr_engine_destroy(e);
    return locked == N_CH ? 0 : 3;
}
