"""The one-workgroup-per-transform inverse sweep of the map-free search (`pcps_fused`, sydr_amd/csrc/pcps_fused.h) against the
two-kernel register-resident path, same process, same box: identical peaks / ratios to rounding on a stream with real
satellites and on noise, then wall time and in-stream kernel time (one HIP-event pair per call) of both, alternating.
    python tools/pcps_fused_check.py [reps] [fs_mhz: 25 | 50]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
e = Engine(0)
fs = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 25e6
n = int(fs / 1000)
e.iq_alloc(4 * n, FMT_CI8)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
slots = np.arange(32)
sats = [dict(prn=p, doppler=float(-4500 + 281.25 * p), code_phase=31.7 * p + 0.25, phase=0.1 * p, amp=8.0) for p in range(1, 33, 2)]
e.iq_synth(sats, fs, 20.0, 20260002, 0, 4 * n)
out = {}
for start in (0, 12345):
    res = {}
    for fused in (0, 1):
        e.set_option("pcps_fused", fused)
        res[fused] = e.pcps(slots, start, fs, 0.0, 5000.0, 250.0)
    pb0, pc0, pr0, _ = res[0]
    pb1, pc1, pr1, _ = res[1]
    out[f"start_{start}"] = {"bins_equal": bool(np.array_equal(pb0, pb1)), "codes_equal": bool(np.array_equal(pc0, pc1)),
                             "ratio_max_rel": float(np.max(np.abs(pr0 - pr1) / pr0)), "ratio_max": float(np.max(pr0))}
e.set_option("pcps_one_stream", 1)
timing = {}
for rnd in range(2):
    for fused in (0, 1):
        e.set_option("pcps_fused", fused)
        for _ in range(30):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
        wall = (time.perf_counter() - t0) / reps * 1e3
        e.prof_reset(); e.prof_enable(True, calls_only=True)
        for _ in range(reps):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
        e.prof_enable(False)
        ms, _ = e.prof_read("call_pcps")
        timing[f"fused{fused}_round{rnd}"] = {"wall_ms": wall, "kernel_ms": ms / reps}
out["timing"] = timing
print(json.dumps(out, indent=1))
