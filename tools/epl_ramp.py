"""Launch-time trajectory of the headline E/P/L launch from a cold start: the chip settles its clocks under this kernel
over the first ~100 launches (~35 ms), whatever ran before (a copy kernel does not do it).  Prints kernel ms / wall ms per
launch for consecutive groups of 20 launches.      python tools/epl_ramp.py [seconds idle before the first launch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sydr_amd.engine import FMT_CI8, Engine
eng = Engine(0)
total = int(3.0 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8); eng.code_slots(32)
sats = bench.satellites()
for s, sat in enumerate(sats): eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
per = 1000 * 32
plan = eng.epl_plan(items, bench.SPACING, bench.FS)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
res = []
for rnd in range(40):
    eng.prof_reset(); eng.prof_enable(True)
    t0 = time.perf_counter()
    for k in range(20): plan.run((k % 2) * per, per)
    eng.sync(); t1 = time.perf_counter(); eng.prof_enable(False)
    ms, n = eng.prof_read("epl_kernel")
    res.append((ms / n, (t1 - t0) * 1e3 / 20))
print(" ".join(f"{a:.3f}/{b:.3f}" for a, b in res))
