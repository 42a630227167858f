// The closed-loop kernel for more channels than compute units: see the SDR_TRACK_DENSE_TU block of track.hip.
#define SDR_TRACK_DENSE_TU 1
#undef SDR_TRACE_TRACK  // (the per-phase clocks of the debug build live in track.hip's own translation unit)
#include "track.hip"
