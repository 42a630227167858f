"""Map-free acquisition (32 PRN x 41 Doppler bins, 1 ms) at the four usual rates: the register-resident kernels
(pcps_fast.h / pcps_fastn.h) against the general four-step kernels on the same inputs.  Wall clock around blocking
calls, median of `reps`; `frac` is the algorithmic 32 N bytes per (PRN, bin) over 8 TB/s."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sydr_amd.engine import Engine, FMT_CI8


def measure(eng, fs, reps=30, n_prn=32):
    n = int(round(fs * 1e-3))
    eng.iq_alloc((n + 7) // 8 * 8, FMT_CI8)
    eng.code_slots(32)
    for s in range(32):
        eng.load_gps_code(s, s + 1)
    eng.iq_synth(bench.satellites(8), fs, 12.0, 5, 0, (n + 7) // 8 * 8)
    slots = np.arange(n_prn)
    out = {}
    for name, general in (("register_resident", 0), ("general", 1)):
        eng.set_option("pcps_general_kernels", general)
        for _ in range(12):
            res = eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, 1, 1)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, 1, 1)
            t.append(time.perf_counter() - t0)
        ms = float(np.median(t)) * 1e3
        out[name] = dict(ms_per_search=ms, frac_of_8TBs=32.0 * n * n_prn * 41 / (ms * 1e-3) / 8e12,
                         peaks=[int(v) for v in res[1][:8]])
    eng.set_option("pcps_general_kernels", 0)
    out["speedup"] = out["general"]["ms_per_search"] / out["register_resident"]["ms_per_search"]
    out["same_peaks"] = out["general"]["peaks"] == out["register_resident"]["peaks"]
    return out


if __name__ == "__main__":
    eng = Engine(0)
    print(json.dumps({f"{int(fs / 1e6)}MHz": measure(eng, fs) for fs in (4e6, 10e6, 25e6, 50e6)}, indent=1))
