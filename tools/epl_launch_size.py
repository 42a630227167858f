"""Cost per second of stream of the headline E/P/L correlators against the launch size (1, 2, 4 s and the whole stream per
launch) once the clocks have settled: every launch ends with a partial round of workgroups.      python tools/epl_launch_size.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sydr_amd.engine import FMT_CI8, Engine
eng = Engine(0)
total = int(20.0 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8); eng.code_slots(32)
sats = bench.satellites()
for s, sat in enumerate(sats): eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
per = 1000 * 32
nl = n_epochs // 1000
plan = eng.epl_plan(items, bench.SPACING, bench.FS)
for k in range(150): plan.run((k % nl) * per, per)
eng.sync()
for rnd in range(3):
    for chunk in (1, 2, 4, nl):
        t0 = time.perf_counter()
        for rep in range(4):
            for j in range(0, nl - chunk + 1, chunk): plan.run(j * per, chunk * per)
        eng.sync(); t1 = time.perf_counter()
        done = 4 * (nl // chunk) * chunk
        print(f"chunk {chunk:3d} s per launch: {(t1 - t0) * 1e3 / done:.4f} ms per second of stream")
