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
plan = eng.epl_plan(items, bench.SPACING, bench.FS)
per = 1000 * 32
for k in range(3): plan.run(0, per)
eng.sync(); eng.prof_reset(); eng.prof_enable(True)
for k in range(20): plan.run((k % 2) * per, per)
eng.sync(); eng.prof_enable(False)
ms, n = eng.prof_read("epl_kernel")
print(os.environ.get("SYDR_AMD_LIB", "default").split("/")[-1], f"{ms / n:.4f} ms per launch")
