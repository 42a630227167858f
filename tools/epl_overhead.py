"""Per-workgroup fixed cost of epl_kernel: tiny epochs, different item counts."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, make_items, FMT_CI8
e = Engine(0)
cap = 8 * 400000
e.iq_alloc(cap, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * cap).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
rng = np.random.default_rng(1)
for n_items in (1024, 8000, 32000, 128000):
    for n, step in ((16, 0.04092), (4096, 0.04092)):
        items = make_items(np.arange(n_items) % 32, n, rng.integers(0, cap, n_items), 1000.0, 0.3, 0.01, step)
        plan = e.epl_plan(items, (-0.5, 0.0, 0.5), 25e6)
        plan.run(); e.sync()
        e.prof_reset(); e.prof_enable(True)
        for _ in range(5):
            plan.run()
        ms, cnt = e.prof_read("epl_kernel"); e.prof_enable(False)
        t = ms / cnt
        print(f"items={n_items:6d} n={n:5d}  {t*1e3:9.1f} us/launch  {t*1e6/n_items:7.2f} ns/WG")
        plan.close()
