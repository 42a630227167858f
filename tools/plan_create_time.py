"""What sdr_epl_plan_create costs for the benchmark's list (60 s x 32 channels = 1.92 M items): wall time of the call, its
stages with SDR_PLAN_TIMING=1 in the environment (printed by the library), and of the first pass behind it.
    SDR_PLAN_TIMING=1 python tools/plan_create_time.py [seconds]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sydr_amd.engine import Engine, FMT_CI8

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
eng = Engine(0)
total = int(secs * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites(bench.N_CH)
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
out = {"items": int(len(items))}
for rep in range(3):
    t0 = time.perf_counter()
    plan = eng.epl_plan(items, bench.SPACING, bench.FS)
    t1 = time.perf_counter()
    plan.run(0, n_epochs * bench.N_CH)
    eng.sync()
    t2 = time.perf_counter()
    out[f"rep{rep}"] = {"create_ms": (t1 - t0) * 1e3, "first_pass_ms": (t2 - t1) * 1e3}
    plan.close()
print(json.dumps(out))
