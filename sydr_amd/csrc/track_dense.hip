// The closed-loop kernel for more channels than compute units: see the SDR_TRACK_DENSE_TU block of track.hip.
#define SDR_TRACK_DENSE_TU 1
#ifdef SDR_TRACE_DENSE   // (diagnostics: the per-phase clocks of THIS unit's kernels, under their own names)
#define SDR_TRACE_TRACK 1
#define g_track_phase g_track_phase_dense
#define sdr_debug_track_phases sdr_debug_track_phases_dense
#else
#undef SDR_TRACE_TRACK  // (the per-phase clocks of the debug build live in track.hip's own translation unit)
#endif
#include "track.hip"
