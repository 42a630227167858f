"""Closed-loop latency only: us per epoch of the device loop closure for 32 channels on clusters of 8 workgroups
(and other channel counts on request).  SYDR_AMD_LIB selects the build (A/B of kernel variants on one box).
    python tools/closed_loop_latency.py [n_ch ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sydr_amd.engine import FMT_CI8, Engine

counts = [int(a) for a in sys.argv[1:]] or [32]
eng = Engine(0)
total = int(3.0 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites()
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
out = []
for n_ch in counts:
    best = min(bench.closed_loop_leg(eng, items, 2000, n_ch=n_ch)["us_per_epoch"] for _ in range(3))
    out.append(f"{n_ch} ch: {best:.3f} us/epoch")
print(os.environ.get("SYDR_AMD_LIB", "default"), "|", "; ".join(out))
