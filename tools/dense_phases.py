"""Per-phase clocks of channel 0 in the dense closed-loop form (variant built with -DSDR_TRACE_DENSE)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import sydr_amd._lib as L
from sydr_amd.engine import FMT_CI8, Engine
eng = Engine(0)
total = int(1.5 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites()
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
N = 1000
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 768
SYM = "sdr_debug_track_phases_dense" if NCH > 256 else "sdr_debug_track_phases"
print(bench.closed_loop_leg(eng, items, N, n_ch=NCH))
buf = np.zeros(64, dtype=np.uint64)
lib = L.load()
fn = getattr(lib, SYM)
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
names = ["first barrier", "constants", "correlate", "reduce", "update (wave 0 role)", "loop top", "exchange", "corr hand-over barrier"]
tot = float(buf[:8].sum())
for n, v in zip(names, buf[:8]):
    print(f"{n:24s} {float(v)*10/1e3/N:8.2f} us/epoch  {100.0*float(v)/tot:5.1f} %")
print("reduction barrier -> end of the role's update (us): carrier %.2f  code %.2f  lock %.2f  carrier-phase/publisher %.2f" % tuple(float(v) * 10 / 1e3 / N for v in buf[48:52]))
