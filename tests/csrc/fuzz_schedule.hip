// CPU sanitizer build (`make -C sydr_amd/csrc check-sanitize`: -fsanitize=address,undefined, no recovery) of sdr_block_schedule
// (sydr_amd/csrc/schedule.hip: host code of the library) -- fed blocks of random and hostile shape: epoch lengths of 0, negative,
// INT_MAX, channels that ran nothing, everything already complete, unread counts far beyond a tick, more epochs than ticks.
// Every output array is allocated at exactly the size the header promises: a write beyond it, a signed overflow or an
// out-of-range conversion ends the run with a sanitizer report.  Built with `hipcc --cuda-host-only`: no device code, no GPU.
//   usage: fuzz_schedule <n_blocks> <seed>
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sydr_amd/csrc/engine_internal.h"

// (the library's error reporter lives in engine.hip: this program links schedule.hip alone)
int sdr_fail(int code, const char* fmt, ...) {
    (void)fmt;
    return code;
}
#include "../../sydr_amd/csrc/schedule.hip"

int main(int argc, char** argv) {
    const int n_blocks = argc > 1 ? atoi(argv[1]) : 20000;
    uint64_t state = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 1) * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return state >> 11;
    };
    const int hostile_n[] = {0, 1, -1, -25000, 24999, 25000, 25001, 4000, 100000, INT_MAX, INT_MIN, INT_MAX - 1, 1 << 20};
    long ok = 0, refused = 0;
    for (int t = 0; t < n_blocks; ++t) {
        const int n_ch = 1 + (int)(rnd() % 40), n_cols = 1 + (int)(rnd() % 60);
        const int64_t spt = (int64_t[]){1, 4000, 10000, 25000, 50000}[rnd() % 5];
        const bool hostile = rnd() % 3 == 0;
        std::vector<sdr_track_epoch> rec((size_t)n_ch * n_cols);
        for (auto& r : rec) {
            r = sdr_track_epoch{};
            r.n_samples = hostile && rnd() % 4 == 0 ? hostile_n[rnd() % 13] : (int)spt + (int)(rnd() % 5) - 2;
            r.track_flags = (int)(rnd() % 8);
            r.nav_bit = rnd() % 16 == 0 ? (int)(rnd() % 2) : -1;
        }
        std::vector<int32_t> done(n_ch);
        std::vector<int64_t> unread(n_ch), flags0(n_ch), since0(n_ch);
        int total = 0;
        for (int c = 0; c < n_ch; ++c) {
            done[c] = rnd() % 5 == 0 ? (int)(rnd() % (n_cols + 1)) : n_cols;
            if (hostile && rnd() % 50 == 0) done[c] = rnd() % 2 ? n_cols + 1 : -1;        // must be refused, not indexed with
            unread[c] = hostile && rnd() % 8 == 0 ? (int64_t)(rnd() % 3 ? 1ll << 40 : -(1ll << 40)) : (int64_t)(rnd() % (3 * spt));
            flags0[c] = (int64_t)(rnd() % 8), since0[c] = (int64_t)(rnd() % 100000);
            if (done[c] > 0 && done[c] <= n_cols) total += done[c];
        }
        const int max_ticks = hostile && rnd() % 4 == 0 ? 1 + (int)(rnd() % 4) : n_cols + 8;
        std::vector<int32_t> first((size_t)n_ch * n_cols), rs(total ? total : 1), cs(total ? total : 1), starts(max_ticks + 1), last(n_ch),
            br(total ? total : 1), bc(total ? total : 1), bv(total ? total : 1);
        std::vector<sdr_track_epoch> sorted(total ? total : 1), lastr(n_ch);
        std::vector<int64_t> un((size_t)max_ticks * n_ch), df((size_t)max_ticks * n_ch), cc((size_t)max_ticks * n_ch);
        int32_t nt = -1, nb = -1;
        const int rc = sdr_block_schedule(rec.data(), n_ch, n_cols, done.data(), unread.data(), spt, flags0.data(), since0.data(), max_ticks,
                                          first.data(), &nt, rs.data(), cs.data(), starts.data(), sorted.data(), lastr.data(), un.data(),
                                          df.data(), cc.data(), last.data(), br.data(), bc.data(), bv.data(), &nb);
        if (rc == 0) {
            if (nt < 0 || nt > max_ticks || nb < 0 || nb > total || (nt && starts[nt] != total)) {
                fprintf(stderr, "block %d: inconsistent result (%d ticks of %d, %d bits of %d epochs)\n", t, nt, max_ticks, nb, total);
                return 1;
            }
            ++ok;
        } else {
            ++refused;
        }
    }
    printf("%ld blocks scheduled, %ld refused\n", ok, refused);
    return 0;
}
