// K1, the straight-line forms by block length: `epl_kernel<ci8, 3 taps, one chip per lane, KM, .., KS | KI>` for the block
// lengths KM = 16 .. 25 that epl.hip does not instantiate itself (it keeps 24 / 12, 19 / 9 and the whole-chip-tap forms of 24
// and 15: the headline geometries).  A list whose epochs all hold KM.x samples per (half-)chip gets the kernel compiled for
// that length -- 16.4 .. 26.6 MHz for GPS L1 C/A, 32.7 .. 53 MHz through the half-chip view -- instead of the
// run-time-position kernel: three taps half a chip apart (KS), three or five taps whole (half-)chips apart (KI), three taps at
// any other spacing (the block length alone).  One translation unit of its own: ~35 large kernels, compiled beside epl.hip.
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

// five taps whole (half-)chips apart (VE / E / P / L / VL half a chip apart on the half-chip view)
const void* sdr_epl_ki5_kernel(int km) {
    switch (km) {
#define SDR_KI5_CASE(K) case K: return (const void*)epl_kernel<SDR_FMT_CI8, 5, sdr::kChipMax, K, 1, 0, 1>;
        SDR_KI5_CASE(16) SDR_KI5_CASE(17) SDR_KI5_CASE(18) SDR_KI5_CASE(19) SDR_KI5_CASE(20) SDR_KI5_CASE(21) SDR_KI5_CASE(22) SDR_KI5_CASE(23) SDR_KI5_CASE(25)
#undef SDR_KI5_CASE
        default: return nullptr;
    }
}

// the block length alone compiled in, tap positions at run time: three taps at any spacing (e.g. the +-0.25 chip of a narrow
// correlator), no per-item setups
const void* sdr_epl_km_kernel(int km) {
    switch (km) {
#define SDR_KM_CASE(K) case K: return (const void*)epl_kernel<SDR_FMT_CI8, 3, sdr::kChipMax, K, 1, 0, 0>;
        SDR_KM_CASE(17) SDR_KM_CASE(18) SDR_KM_CASE(19) SDR_KM_CASE(20) SDR_KM_CASE(21) SDR_KM_CASE(22) SDR_KM_CASE(23) SDR_KM_CASE(25)
#undef SDR_KM_CASE
        default: return nullptr;
    }
}
