"""Fixed-vs-proportional cost of epl_kernel: time per workgroup as a function of samples per item."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, make_items, FMT_CI8
e = Engine(0)
if '--no-chip' in sys.argv:
    e.set_option('epl_no_chip_variant', 1)
cap = 8 * 400000
e.iq_alloc(cap, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * cap).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
rng = np.random.default_rng(1)
n_items = 32000
only = int(sys.argv[sys.argv.index('--only') + 1]) if '--only' in sys.argv else None
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
for n, step in ((3125, 0.32), (6250, 0.16), (10000, 0.1023), (12500, 0.0818), (25000, 0.04092), (50000, 0.02046), (100000, 0.01023)):
    if only and n != only:
        continue
    items = make_items(np.arange(n_items) % 32, n, rng.integers(0, cap, n_items), 1000.0, 0.3, 0.01, step)
    plan = e.epl_plan(items, (-0.5, 0.0, 0.5), 25e6 * 0.04092 / step if step != 0.1023 else 10e6)
    plan.run(); e.sync()
    e.prof_reset(); e.prof_enable(True)
    for _ in range(reps):
        plan.run()
    ms, cnt = e.prof_read("epl_kernel"); e.prof_enable(False)
    t = ms / cnt
    print(f"n={n:6d} step={step:.5f} variant={16 if step<=0.06 else (8 if step<=0.125 else 0)}  {t:.4f} ms/launch  {t*1e6/n_items:.1f} ns/WG-slot  {t*1e6/(n_items*n)*1e3:.3f} ps/ch-sample")
    plan.close()
