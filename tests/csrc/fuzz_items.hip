// CPU sanitizer build (`make -C sydr_amd/csrc check-sanitize`: -fsanitize=address,undefined, no recovery) of the host half of
// sdr_epl_plan_create: the per-item check every untrusted list goes through (epl_items.h check_item) and, for the items it
// lets pass, the per-item setups of the straight-line correlators (correlator_chip.h chip_setup, correlator_chip2.h
// chipn_setup) -- fed hostile items: NaN / Inf / denormal / negative NCO fields, 0 / INT_MAX / INT_MIN sample counts,
// negative and huge slots and starts.  A conversion of a non-finite value to an integer, a shift of a negative count or a
// read outside code_len[] ends the run with a sanitizer report and a non-zero exit.  Built with `hipcc --cuda-host-only`:
// no device code, no GPU.   usage: fuzz_items <n_items> <seed>
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../sydr_amd/csrc/engine_internal.h"
#include "../../sydr_amd/csrc/correlator.h"
#include "../../sydr_amd/csrc/correlator_chip2.h"
#include "../../sydr_amd/csrc/epl_items.h"

using namespace sdr;

int main(int argc, char** argv) {
    const int n_items = argc > 1 ? atoi(argv[1]) : 200000;
    uint64_t state = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 1) * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return state >> 11;
    };
    auto uni = [&]() { return (double)rnd() / 9007199254740992.0; };
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double hostile[] = {0.0, -0.0, nan, inf, -inf, 4.9e-324, -4.9e-324, 1e-300, 1e300, -1e300, 1.0, -1.0, 0.0625, 0.04092, 0.1023,
                              0.25575, 1.0 / 16.0, 1.0 / 25.9, 2.0, 1024.0, 1e9, 9.3e18, -9.3e18};
    const int hostile_n[] = {0, 1, -1, 2, 24, 25, 4000, 4001, 25000, 25001, 100000, INT_MAX, INT_MIN, INT_MAX - 1, 1 << 20, (1 << 20) + 1};
    const int64_t hostile_start[] = {0, -1, 1, 1 << 20, (1ll << 20) - 25000, INT64_MAX, INT64_MIN, 1ll << 40, 123456};
    const int n_slots = 4;
    const int32_t code_len[n_slots] = {1023, 1023, 0, 4092};
    const double spacing3[3] = {-0.5, 0.0, 0.5}, spacing5[5] = {-1.0, -0.5, 0.0, 0.5, 1.0};
    long passed = 0, rejected[8] = {0};
    for (int i = 0; i < n_items; ++i) {
        sdr_epl_item it = {};
        const double fs = (double[]){4e6, 10e6, 12e6, 16.368e6, 20e6, 25e6, 50e6}[rnd() % 7];
        const bool wild = rnd() % 4 == 0;                // a quarter of the items: anything at all; the rest: one hostile field
        const double step = 1.023e6 * (1.0 + (uni() - 0.5) * 1e-5) / fs;
        it.code_slot = (int)(rnd() % 2);
        it.code_step = step;
        it.rem_code = uni() * step;
        it.n_samples = (int)std::ceil((1023.0 - it.rem_code) / step) + (int)(rnd() % 3) - 1;
        it.start_sample = (int64_t)(rnd() % ((1 << 20) - 60000));
        it.carrier_hz = (uni() - 0.5) * 1e4;
        it.rem_carrier = uni() * 6.28;
        const int fields = wild ? 7 : 1;
        for (int f = 0; f < fields; ++f) {
            switch (wild ? f : (int)(rnd() % 8)) {
                case 0: it.code_slot = (int)(rnd() % 9) - 2; break;
                case 1: it.n_samples = hostile_n[rnd() % (sizeof(hostile_n) / sizeof(int))]; break;
                case 2: it.start_sample = hostile_start[rnd() % (sizeof(hostile_start) / sizeof(int64_t))]; break;
                case 3: it.code_step = hostile[rnd() % (sizeof(hostile) / sizeof(double))]; break;
                case 4: it.rem_code = hostile[rnd() % (sizeof(hostile) / sizeof(double))]; break;
                case 5: it.rem_carrier = hostile[rnd() % (sizeof(hostile) / sizeof(double))]; break;
                case 6: it.carrier_hz = hostile[rnd() % (sizeof(hostile) / sizeof(double))]; break;
                default: break;                          // (an honest item)
            }
        }
        for (int taps = 3; taps <= 5; taps += 2) {
            const double* spacing = taps == 3 ? spacing3 : spacing5;
            bool ok_all = true;
            for (double scale = 1.0; scale <= 2.0; scale += 1.0) {
                ItemRules r = {};
                r.n_slots = n_slots, r.lut_stride = 1023 + 2 * SDR_LUT_PAD + 8, r.n_taps = taps, r.iq_capacity = 1 << 20, r.scale = scale;
                r.smin = spacing[0], r.smax = spacing[taps - 1], r.s_anchor = scale * spacing[taps / 2 < 2 ? taps / 2 : 2];
                r.sp0 = spacing[0], r.sp2 = spacing[2], r.want_s12 = taps == 3;
                int maxlen = 0;
                double st = 0.0, lo = 0.0, hi = 0.0;
                int m_chip = 0;
                bool split = true;
                const int bad = check_item(it, r, code_len, maxlen, st, m_chip, split, lo, hi);
                ++rejected[bad];
                ok_all = ok_all && bad == ITEM_OK;
                if (bad != ITEM_OK && scale == 1.0) break;
            }
            if (!ok_all) continue;
            ++passed;
            if (getenv("FUZZ_VERBOSE")) fprintf(stderr, "item %d taps %d: slot %d n %d start %lld f %.17g remc %.17g remcode %.17g step %.17g fs %g\n", i, taps, it.code_slot, it.n_samples, (long long)it.start_sample, it.carrier_hz, it.rem_carrier, it.rem_code, it.code_step, fs);
            // what plan creation does next for an accepted item (the same functions its setup kernels run per thread)
            volatile double sink = 0.0;
            if (taps == 3) {
                ChipSetup<3> a, b, b9;
                chip_setup<3, 19, 9, 0>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, 64, b9);
                chip_setup<3, 24, 12, 0>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, 64, a);
                chip_setup<3, 24, 0, 1>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, 64, b);
                ChipNSetup<4, 9, 14, 19> c;
                ChipNSetup<5, 11, 17, 23> d;
                const bool c_ok = chipn_setup<4, 9, 14, 19>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, c);
                const bool d_ok = chipn_setup<5, 11, 17, 23>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, d);
                sink = a.dphi + b.dphi + (double)(c_ok + d_ok);
            } else {
                ChipSetup<5> a;
                chip_setup<5, 24, 0, 1>(it.n_samples, it.start_sample, 1 << 20, it.carrier_hz, it.rem_code, it.code_step, spacing, fs, 64, a);
                sink = a.dphi;
            }
            (void)sink;
        }
    }
    printf("fuzz_items: %d items, %ld (item, taps) pairs accepted and set up; rejected by reason (ok slot samples start nco replica): "
           "%ld %ld %ld %ld %ld %ld\n", n_items, passed, rejected[0], rejected[1], rejected[2], rejected[3], rejected[4], rejected[5]);
    return 0;
}
