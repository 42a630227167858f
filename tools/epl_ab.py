"""Same-process A/B of the two headline E/P/L kernels (tap switch positions compiled in / found at run time) on one box:
alternating groups of 20 launches, and the difference of their outputs.      python tools/epl_ab.py"""
import os, sys
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
plans = {}
for name, opt in (("split12", 0), ("dynamic", 1)):
    eng.set_option("epl_no_split_variant", opt)
    plans[name] = eng.epl_plan(items, bench.SPACING, bench.FS)
outs = {}
for rnd in range(3):
    for name, plan in plans.items():
        for k in range(3): plan.run(0, per)
        eng.sync(); eng.prof_reset(); eng.prof_enable(True)
        for k in range(20): plan.run((k % 2) * per, per)
        eng.sync(); eng.prof_enable(False)
        ms, n = eng.prof_read("epl_kernel")
        print(name, f"{ms / n:.4f} ms per launch")
        outs[name] = plan.fetch()
a, b = outs["split12"], outs["dynamic"]
print("max abs diff", np.abs(a - b).max(), "max rel", (np.abs(a - b) / np.maximum(np.abs(b), 1.0)).max())
