import os, sys, time
sys.path.insert(0, os.getcwd())
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
time.sleep(2.0)
print("copy", [round(eng.hbm_copy_rate(1 << 30, r)) for r in (10, 10, 50, 100, 100)])
res = []
for rnd in range(12):
    eng.prof_reset(); eng.prof_enable(True)
    for k in range(5): plan.run((k % 2) * per, per)
    eng.sync(); eng.prof_enable(False)
    ms, n = eng.prof_read("epl_kernel")
    res.append(ms / n)
print(" ".join(f"{a:.3f}" for a in res))
