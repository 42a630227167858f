"""The reference's shipped search (10 MHz, +-5 kHz @ 300 Hz = 34 bins, 1 ms x 10 non-coherent; 32 PRNs, indices + ratio) through the
fused one-workgroup-per-(PRN, bin) kernel (pcps_fused10k.h) and through the path that accumulates the map in memory, same
process, alternating: identical peaks, wall and in-stream kernel time of both.   python tools/pcps_10mhz_check.py [reps]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = Engine(0)
fs, n, noncoh = 10e6, 10000, 10
e.iq_alloc(noncoh * n + 64, FMT_CI8)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
slots = np.arange(32)
sats = [dict(prn=p, doppler=float(-4500 + 281.25 * p), code_phase=31.7 * p + 0.25, phase=0.1 * p, amp=6.0) for p in range(1, 33, 2)]
e.iq_synth(sats, fs, 14.0, 20260010, 0, noncoh * n + 64)
out = {}
res = {}
for fused in (0, 1):
    e.set_option("pcps_fused", fused)
    res[fused] = e.pcps(slots, 0, fs, 0.0, 5000.0, 300.0, 1, noncoh)
out["equal"] = bool(np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]))
out["ratio_max_rel"] = float(np.max(np.abs(res[0][2] - res[1][2]) / res[0][2]))
timing = {}
for rnd in range(2):
    for fused in (0, 1):
        e.set_option("pcps_fused", fused)
        for _ in range(10):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 300.0, 1, noncoh)
        t0 = time.perf_counter()
        for _ in range(reps):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 300.0, 1, noncoh)
        wall = (time.perf_counter() - t0) / reps * 1e3
        e.prof_reset(); e.prof_enable(True, calls_only=True)
        for _ in range(reps):
            e.pcps(slots, 0, fs, 0.0, 5000.0, 300.0, 1, noncoh)
        e.prof_enable(False)
        ms, _ = e.prof_read("call_pcps")
        timing[f"fused{fused}_round{rnd}"] = {"wall_ms": wall, "kernel_ms": ms / reps}
out["timing"] = timing
print(json.dumps(out, indent=1))
