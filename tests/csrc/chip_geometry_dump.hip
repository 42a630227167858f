// Host-side dump of what sdr_epl_plan_create works out per item for the straight-line correlators (correlator_chip.h:
// chip_geometry; correlator_chip2.h: chipn_setup), for tests/test_plan_geometry.py to hold against the reference's own
// chip-index expression.  Built with `hipcc --cuda-host-only`: no device code, no GPU.
//   usage: chip_geometry_dump <fs> <n_items> <seed>   -> one line per item
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../sydr_amd/csrc/engine_internal.h"
#include "../../sydr_amd/csrc/correlator.h"
#include "../../sydr_amd/csrc/correlator_chip2.h"

using namespace sdr;

int main(int argc, char** argv) {
    const double fs = atof(argv[1]);
    const int n_items = atoi(argv[2]);
    uint64_t state = strtoull(argv[3], nullptr, 10) * 6364136223846793005ull + 1442695040888963407ull;
    auto uni = [&]() {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(state >> 11) / 9007199254740992.0;
    };
    const double spacing[3] = {-0.5, 0.0, 0.5};
    for (int i = 0; i < n_items; ++i) {
        const double code_step = (1.023e6 + (uni() * 12.0 - 6.0)) / fs;
        const double rem_code = i == 0 ? 0.0 : uni() * code_step;
        const int n = (int)std::ceil((1023.0 - rem_code) / code_step) + (int)(uni() * 3.0) - 1;
        double shift[3], step[3], inv[3];
        for (int t = 0; t < 3; ++t) {
            shift[t] = rem_code + spacing[t];
            double stop = code_step * (double)n;
            stop = stop + shift[t];
            step[t] = (stop - shift[t]) / (double)n;
            inv[t] = 1.0 / step[t];
        }
        ChipGeom<3> g;
        chip_geometry<3, 0, 0, 0>(n, shift, step, inv, g);
        printf("item n=%d rem_code=%.17g code_step=%.17g q0=%d F=%d head_end=%d tail_start=%d Tfx=%lld Ufx=%lld dE=%llu dL=%llu mE=%d mL=%d JE=%d JL=%d bad=%d",
               n, rem_code, code_step, g.q0, g.F, g.head_end, g.tail_start, (long long)g.Tfx, (long long)g.Ufx,
               (unsigned long long)g.delta[0], (unsigned long long)g.delta[2], g.m[0], g.m[2], g.J[0], g.J[2], g.bad);
        ChipNSetup<4, 9, 14, 19> s2;
        const bool ok2 = chipn_setup<4, 9, 14, 19>(n, 1000 + 7 * i, (int64_t)1 << 24, 1500.0, rem_code, code_step, spacing, fs, s2);
        printf(" c2=%d F2=%d c2_tail=%d c2_Dmin=%d", ok2 ? 1 : 0, s2.FB, s2.tail_start, s2.Dmin);
        ChipRot r;
        const double dphi = carrier_step(1500.0 + 100.0 * i, fs);
        const int Dmin = (int)((64 * g.Tfx) >> 32);
        chip_rotations(dphi, Dmin, r);
        printf(" dphi=%.17g Dmin=%d urc5=%.17g urs5=%.17g urc13=%.17g urs13=%.17g rd1c=%.17g rd1s=%.17g biasc2=%.17g biass2=%.17g biasc0=%.17g\n",
               dphi, Dmin, r.urc[5], r.urs[5], r.urc[13], r.urs[13], r.rd1c, r.rd1s, r.biasc[2], r.biass[2], r.biasc[0]);
    }
    return 0;
}
