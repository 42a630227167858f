"""Per-stage hipEvent timing of sdr_pcps at BASELINE config 2 (32 PRNs, 25 MHz, 41 bins)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8
e = Engine(0)
fs, n = 25e6, 25000
e.iq_alloc(n, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * n).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
slots = np.arange(32)
if len(sys.argv) > 1:
    e.set_option('pcps_prn_chunk', int(sys.argv[1]))
for _ in range(60):   # allocations, twiddles, and the ~35 ms the clocks take to settle
    e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
e.prof_reset(); e.prof_enable(True)
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
wall = (time.perf_counter() - t0) / reps * 1e3
for name in ("pcps_upsample", "pcps_code_fft", "pcps_fwd_fft", "pcps_inv_fft", "pcps_peak"):
    ms, cnt = e.prof_read(name)
    print(f"{name:16s} {ms / reps:8.4f} ms per call ({cnt // reps} launches)")
print(os.environ.get("SYDR_AMD_LIB", "default").split("/")[-1], f"wall per sdr_pcps call: {wall:.4f} ms = {wall / 32:.5f} ms/PRN")
