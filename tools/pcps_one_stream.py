"""One-stream acquisitions at BASELINE config 2 (32 PRNs, 25 MHz, 41 bins, map-free) for `rocprofv3 --kernel-trace --stats`:
every sdr_pcps call of this process runs on ONE HIP stream (`pcps_one_stream`), so the per-kernel totals of the profile
divided by the number of calls ARE the kernel time of one call -- the figure bench.py reports as `kernel_ms_32_prn` from
its own HIP events, printed here for the same process.  python tools/pcps_one_stream.py [fs_mhz: 25 | 50]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8
e = Engine(0)
fs = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 25e6
n = int(fs / 1000)
e.iq_alloc(n, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * n).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
slots = np.arange(32)
e.set_option("pcps_one_stream", 1)
warm, reps = 60, 40
for _ in range(warm):
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
print(json.dumps({"calls": warm + 2 * reps, "wall_ms_per_call_one_stream": wall, "hip_event_kernel_ms_per_call": ms / reps}))
