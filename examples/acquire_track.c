/* Plain-C client of libsydr_amd.so: the C-ABI alone (include/sydr_amd.h), no Python.
 *   synthesise 1.2 s of ci8 IQ on the GPU -> PCPS acquisition of 8 PRNs -> 1000 ms of closed-loop
 *   Kaplan tracking for the satellites found -> print Doppler, C/N0 proxy and the navigation bits.
 * Build:  gcc -O2 -Iinclude examples/acquire_track.c -Lsydr_amd -lsydr_amd -lm -Wl,-rpath,'$ORIGIN/../sydr_amd' -o examples/acquire_track
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sydr_amd.h"

#define CK(call)                                                              \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != SDR_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sdr_last_error());  \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(void) {
    const double fs = 25e6, code_rate = 1.023e6;
    const int n_sat = 4, n_search = 8, epochs = 1000;
    const int64_t total = (int64_t)(1.2 * fs) / 8 * 8;
    sdr_engine* e = NULL;
    CK(sdr_engine_create(0, &e));
    CK(sdr_iq_alloc(e, total, SDR_FMT_CI8));
    CK(sdr_code_slots(e, n_search, 1023));
    int32_t slots[8];
    for (int s = 0; s < n_search; ++s) {
        slots[s] = s;
        CK(sdr_code_gps_l1ca(e, s, s + 1)); /* PRN 1..8, generated on the device */
    }
    sdr_synth_sat sats[4] = {{1, 0, 1750.0, 300.25, 0.1, 6.0}, {3, 0, -2500.0, 17.5, 0.6, 6.0},
                             {6, 0, 4000.0, 901.0, 0.3, 6.0},  {8, 0, -750.0, 512.5, 0.9, 6.0}};
    CK(sdr_iq_synth(e, sats, n_sat, fs, 12.0, 20260001ull, 0, total));

    int64_t bin[8], code[8];
    double ratio[8];
    int nbins = 0;
    CK(sdr_pcps(e, slots, n_search, 0, fs, 0.0, 5000.0, 250.0, 1, 1, bin, code, ratio, NULL, &nbins));

    sdr_loop_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.loop_kind = 1; /* Kaplan: FLL-assisted PLL + DLL, config/channels/channel_GPS_L1CA_kaplan.ini values */
    cfg.n_taps = 3;
    cfg.fs = fs;
    const double sp[3] = {-0.5, 0.0, 0.5};
    for (int t = 0; t < 3; ++t) cfg.spacing_wide[t] = cfg.spacing_narrow[t] = sp[t];
    const double zeta = 0.7, wn = 2.0 * 8.0 * zeta / (4.0 * zeta * zeta + 1.0);
    cfg.dll_tau1 = 1.0 / (wn * wn);
    cfg.dll_tau2 = 2.0 * zeta / wn;
    cfg.dll_pdi = 1e-3;
    cfg.dll_threshold = 10.0;
    cfg.fll_bw_pullin = 100.0, cfg.fll_bw_wide = 50.0, cfg.fll_bw_narrow = 15.0;
    cfg.fll_thr_wide = 0.5, cfg.fll_thr_narrow = 0.8;
    cfg.pll_bw_wide = 25.0, cfg.pll_bw_narrow = 15.0, cfg.pll_thr_wide = 0.5, cfg.pll_thr_narrow = 0.8;

    sdr_track_state st[8];
    int prn_of[8], n_ch = 0;
    const int n_code = (int)(fs * 1e-3);
    for (int s = 0; s < n_search; ++s) {
        printf("PRN %d: bin %2lld code %5lld ratio %.2f %s\n", s + 1, (long long)bin[s], (long long)code[s], ratio[s],
               ratio[s] > 2.0 ? "-> track" : "");
        if (ratio[s] <= 2.0) continue;
        sdr_track_state* c = &st[n_ch];
        memset(c, 0, sizeof *c);
        c->code_slot = s;
        c->code_hz = code_rate;
        c->code_step = code_rate / fs;
        c->n_samples = (int)ceil(1023.0 / c->code_step);
        c->carrier_hz = 0.0 + -(-5000.0 + 250.0 * (double)bin[s]); /* postAcquisitionUpdate, channel_l1ca_kaplan.py:217-235 */
        c->current_sample = n_code - c->n_samples + code[s] + 1;
        if (c->current_sample < 0) c->current_sample += n_code;
        c->fll_bw = cfg.fll_bw_pullin;
        c->pll_bw = cfg.pll_bw_wide;
        c->lock_state = 1;
        prn_of[n_ch++] = s + 1;
    }
    if (n_ch == 0) return 2;

    int8_t* bits = (int8_t*)malloc((size_t)n_ch * 64);
    int32_t nbits[8];
    CK(sdr_track_closed_loop_bits(e, n_ch, st, &cfg, epochs, NULL, bits, 64, nbits));
    for (int c = 0; c < n_ch; ++c) {
        printf("PRN %d: carrier %+9.2f Hz  lock state %d  flags %d  %d bits:", prn_of[c], st[c].carrier_hz, st[c].lock_state,
               st[c].track_flags, nbits[c]);
        for (int k = 0; k < nbits[c] && k < 40; ++k) printf("%d", bits[c * 64 + k]);
        printf("\n");
    }
    free(bits);
    sdr_engine_destroy(e);
    return 0;
}
