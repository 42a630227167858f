"""A warm stream of map-free 32-PRN acquisitions at 25 MHz (profiling target):  python tools/pcps_stream.py [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sydr_amd.engine import Engine, FMT_CI8
e = Engine(0)
fs, n = 25e6, 25000
e.iq_alloc(n, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * n).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    e.pcps(np.arange(32), 0, fs, 0.0, 5000.0, 250.0)
