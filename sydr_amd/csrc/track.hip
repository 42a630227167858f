// On-device loop closure (persistent workgroup per channel).  Filled in below.
#include "engine_internal.h"

extern "C" {

int sdr_track_closed_loop(sdr_engine* e, int n_ch, sdr_track_state* st, const sdr_loop_cfg* cfg, int n_epochs,
                          sdr_track_epoch* traj) {
    (void)e; (void)n_ch; (void)st; (void)cfg; (void)n_epochs; (void)traj;
    return sdr_fail(SDR_ERR_UNSUPPORTED, "closed-loop tracking kernel not built yet");
}

}  // extern "C"
