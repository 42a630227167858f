"""Debug: where an epoch of the closed-loop kernel spends its time (needs a -DSDR_TRACE_TRACK build)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import sydr_amd._lib as L
from sydr_amd.engine import FMT_CI8, Engine
eng = Engine(0)
total = int(3.0 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites()
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
print(bench.closed_loop_leg(eng, items, 2000))
buf = np.zeros(64, dtype=np.uint64)
lib = L.load()
lib.sdr_debug_track_phases.argtypes = [ctypes.c_void_p]
assert lib.sdr_debug_track_phases(buf.ctypes.data) == 0
names = ["first barrier", "constants", "correlate", "reduce", "update (wave 0 role)", "loop top", "exchange", "corr hand-over barrier"]
tot = float(buf[:8].sum())
for n, v in zip(names, buf[:8]):
    print(f"{n:16s} {float(v)*10/1e3/2000:8.2f} us/epoch  {100.0*float(v)/tot:5.1f} %")
print("per-wave arrival at the reduction (us after the epoch's first barrier):", " ".join(f"{float(v)*10/1e3/2000:.2f}" for v in buf[8:24]))
print("reduction barrier -> end of the role's update (us): carrier %.2f  code %.2f  lock %.2f  carrier-phase/publisher %.2f" % tuple(float(v) * 10 / 1e3 / 2000 for v in buf[48:52]))
if buf[63]:
    print(f"shader clock during the kernel: {float(buf[62]) / (float(buf[63]) * 10e-9) / 1e6:.0f} MHz")
