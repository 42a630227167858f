"""Per-epoch (fixed) against per-sample cost of epl_kernel at one code step: launches of 32 000 items of n samples,
n varied at a fixed step; t(n) = a + b * n fitted per variant.  `--step 0.1023` (10 MHz, default) / 0.25575 / 0.04092;
`--only-n N`: one length only (under `rocprofv3 --pmc`: tools/pmc_fixed_cost.sh)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, make_items, FMT_CI8

step = float(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else 0.1023
e = Engine(0)
cap = 8 * 400000
e.iq_alloc(cap, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * cap).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
rng = np.random.default_rng(1)
n_items = 32000
full = int(1022.5 / step)
options = [("default", {})]
if 2 / 20 < step <= 2 / 19 or 2 / 24 < step <= 2 / 23:
    options.append(("no_two_chip_variant", {"epl_no_two_chip_variant": 1}))   # 8-sample groups instead of two chips per lane
if step <= 1 / 16:
    options.append(("no_split_variant", {"epl_no_split_variant": 1}))      # run-time switch positions, LDS strip
    options.append(("no_chip_variant", {"epl_no_chip_variant": 1}))
for name, opts in options:
    for k, v in opts.items():
        e.set_option(k, v)
    pts = []
    for n in ([int(sys.argv[sys.argv.index('--only-n') + 1])] if '--only-n' in sys.argv else (full // 4, full // 2, full)):
        items = make_items(np.arange(n_items) % 32, n, rng.integers(0, cap - n - 64, n_items), 1000.0, 0.3, 0.01, step)
        plan = e.epl_plan(items, (-0.5, 0.0, 0.5), 1.023e6 / step)
        plan.run(); e.sync()
        e.prof_reset(); e.prof_enable(True)
        for _ in range(8):
            plan.run()
        ms, cnt = e.prof_read("epl_kernel"); e.prof_enable(False)
        pts.append((n, ms / cnt, plan.variant))
        plan.close()
    for k in opts:
        e.set_option(k, 0)
    (n0, t0, _), (n2, t2, v) = pts[0], pts[-1]
    b = (t2 - t0) / (n2 - n0)
    a = t2 - b * n2
    # 1024 SIMDs, one wave-instruction per 4 cycles at 2.4 GHz
    slots = lambda ms: ms * 1e-3 * 2.4e9 / 4 * 1024 / n_items
    print(f"step {step} {name:16s} variant {v & 255}+{v >> 8 << 8}: " + "  ".join(f"n={n}: {t:.3f} ms" for n, t, _ in pts) +
          f"  | fixed {a:.3f} ms = {slots(a):.0f} issue slots per epoch, {slots(b) * 64:.1f} per 64 samples = {slots(b):.2f} per lane-sample"
          f"  | frac of 8 TB/s at full length {2.0 * n2 * n_items / (t2 * 1e-3) / 8e12:.3f}")
