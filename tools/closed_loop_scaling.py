"""Closed-loop tracking throughput vs number of channels (channels beyond 32 re-track the same 32 satellites)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sydr_amd._lib import LoopCfg, TrackState
from sydr_amd.engine import FMT_CI8, Engine
eng = Engine(0)
FS = bench.FS
total = int(3.0 * FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites(0)
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, FS, total)
cfg = LoopCfg()
cfg.loop_kind, cfg.n_taps, cfg.fs = 1, 3, FS
for t, s in enumerate(bench.SPACING):
    cfg.spacing_wide[t] = cfg.spacing_narrow[t] = s
wn = 2.0 * 8.0 * 0.7 / (4.0 * 0.7**2 + 1)
cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi, cfg.dll_threshold = 1.0 / wn**2, 2.0 * 0.7 / wn, 0.001, 10.0
cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow, cfg.fll_thr_wide, cfg.fll_thr_narrow = 100.0, 50.0, 15.0, 0.5, 0.8
cfg.pll_bw_wide, cfg.pll_bw_narrow, cfg.pll_thr_wide, cfg.pll_thr_narrow = 25.0, 15.0, 0.5, 0.8


def states(n_ch):
    out = []
    for c in range(n_ch):
        it = items[c % bench.N_CH]
        st = TrackState()
        st.code_slot, st.n_samples, st.current_sample = int(it["code_slot"]), int(it["n_samples"]), int(it["start_sample"])
        st.carrier_hz, st.code_hz = float(it["carrier_hz"]), bench.CODE_RATE
        st.rem_carrier, st.rem_code, st.code_step = float(it["rem_carrier"]), float(it["rem_code"]), bench.CODE_RATE / FS
        st.fll_bw, st.pll_bw, st.lock_state = 100.0, 25.0, 1
        out.append(st)
    return out


epochs = 1000
cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(32, 0), (32, 1), (64, 0), (128, 0), (256, 0), (512, 0), (1024, 0)]
for n_ch, parts in cases:
    eng.track_cluster(parts)
    eng.track_closed_loop(states(n_ch), cfg, 20, want_traj=False)
    eng.prof_reset(); eng.prof_enable(True)
    end, _ = eng.track_closed_loop(states(n_ch), cfg, epochs, want_traj=False)
    eng.prof_enable(False)
    ms, _ = eng.prof_read("track_kernel")
    lost = sum(abs(e.carrier_hz - s.carrier_hz) > 100.0 for e, s in zip(end, states(n_ch)))
    print(f"channels {n_ch:5d} parts {parts}: {ms * 1e3 / epochs:7.2f} us/epoch  {epochs * 1e-3 / (ms * 1e-3):7.1f}x real time  "
          f"{n_ch * epochs * 1e-3 / (ms * 1e-3):9.0f} channel-real-times  lost {lost}")
eng.track_cluster(0)
