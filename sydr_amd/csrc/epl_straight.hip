// K1, the straight-line forms by block length: `epl_kernel<ci8, 3 taps, one chip per lane, KM, .., KS | KI>` for the block
// lengths KM = 16 .. 25 that epl.hip does not instantiate itself (it keeps 24 / 12, 19 / 9 and the whole-chip-tap forms of 24
// and 15: the headline geometries).  A list whose epochs all hold KM.x samples per (half-)chip gets the kernel compiled for
// that length -- 16.4 .. 26.6 MHz for GPS L1 C/A, 32.7 .. 53 MHz through the half-chip view -- instead of the
// run-time-position kernel.  One translation unit of its own: these are ~20 large kernels, and they compile beside epl.hip.
#ifdef SDR_TRACE_WG
#undef SDR_TRACE_WG   // (the per-workgroup clocks of the debug build live in epl.hip's kernels)
#endif
#include "engine_internal.h"
#include "epl_kernel.h"

// the outer taps switching floor(KM / 2).x samples into the prompt tap's chip (+-0.5 chip spacing)
const void* sdr_epl_ks_kernel(int km) {
    switch (km) {
#define SDR_KS_CASE(K) case K: return (const void*)epl_kernel<SDR_FMT_CI8, 3, sdr::kChipMax, K, 1, K / 2, 0>;
        SDR_KS_CASE(16) SDR_KS_CASE(17) SDR_KS_CASE(18) SDR_KS_CASE(20) SDR_KS_CASE(21) SDR_KS_CASE(22) SDR_KS_CASE(23) SDR_KS_CASE(25)
#undef SDR_KS_CASE
        default: return nullptr;
    }
}

// taps whole (half-)chips apart: they switch with the block
const void* sdr_epl_ki_kernel(int km) {
    switch (km) {
#define SDR_KI_CASE(K) case K: return (const void*)epl_kernel<SDR_FMT_CI8, 3, sdr::kChipMax, K, 1, 0, 1>;
        SDR_KI_CASE(16) SDR_KI_CASE(17) SDR_KI_CASE(18) SDR_KI_CASE(19) SDR_KI_CASE(20) SDR_KI_CASE(21) SDR_KI_CASE(22) SDR_KI_CASE(23) SDR_KI_CASE(25)
#undef SDR_KI_CASE
        default: return nullptr;
    }
}
