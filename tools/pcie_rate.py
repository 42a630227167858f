"""Host -> HBM ingest rate of sdr_iq_upload (pageable NumPy buffer), the PCIe-inclusive bound of the streaming path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8
eng = Engine(0)
fs = 25e6
for ms in (1, 10, 100, 1000):
    n = int(fs * ms * 1e-3)
    eng.iq_alloc(max(n, 8), FMT_CI8)
    raw = np.random.default_rng(0).integers(-100, 100, 2 * n).astype(np.int8)
    eng.iq_upload(raw, 0); eng.sync()
    reps = max(3, 200 // ms)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.iq_upload(raw, 0)
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{ms:5d} ms of ci8 @25 MHz ({2 * n / 1e6:8.2f} MB): {dt * 1e3:8.3f} ms per upload = {2 * n / dt / 1e9:6.2f} GB/s = "
          f"{n / dt / 1e6:9.0f} Msamples/s = {ms * 1e-3 / dt:7.1f}x real time")
