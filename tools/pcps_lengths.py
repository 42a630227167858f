"""Acquisition (32 PRN x 41 Doppler bins) at the four usual rates -- map-free, with the map written on the device,
ten non-coherent and 2 x 2 coherent / non-coherent milliseconds: the register-resident kernels (pcps_fast.h /
pcps_fastn.h) against the general four-step kernels on the same inputs.  Wall clock around blocking calls, median of
`reps`; `frac` is the algorithmic 32 N (map-free) or 40 N bytes per (PRN, bin, millisecond) over 8 TB/s."""
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
    for mode, (coh, noncoh, materialise) in (("map_free", (1, 1, 0)), ("map_written", (1, 1, 1)), ("noncoh10", (1, 10, 0)),
                                             ("coh2_noncoh2", (2, 2, 0))):
        need = n * coh * noncoh
        if need > n and eng.iq_capacity < need:
            eng.iq_alloc((need + 7) // 8 * 8, FMT_CI8)
            eng.iq_synth(bench.satellites(8), fs, 12.0, 5, 0, (need + 7) // 8 * 8)
        eng.set_option("pcps_materialise_map", materialise)
        o = {}
        for name, general in (("register_resident", 0), ("general", 1)):
            eng.set_option("pcps_general_kernels", general)
            for _ in range(12 if noncoh == 1 else 3):
                res = eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, coh, noncoh)
            t = []
            for _ in range(reps if noncoh == 1 else max(5, reps // 4)):
                t0 = time.perf_counter()
                eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, coh, noncoh)
                t.append(time.perf_counter() - t0)
            ms = float(np.median(t)) * 1e3
            per_point = 32.0 if mode == "map_free" else 40.0
            o[name] = dict(ms_per_search=ms, frac_of_8TBs=per_point * n * n_prn * 41 * coh * noncoh / (ms * 1e-3) / 8e12,
                           peaks=[int(v) for v in res[1][:8]])
        eng.set_option("pcps_general_kernels", 0)
        eng.set_option("pcps_materialise_map", 0)
        o["speedup"] = o["general"]["ms_per_search"] / o["register_resident"]["ms_per_search"]
        o["same_peaks"] = o["general"]["peaks"] == o["register_resident"]["peaks"]
        out[mode] = o
    return out


if __name__ == "__main__":
    eng = Engine(0)
    print(json.dumps({f"{int(fs / 1e6)}MHz": measure(eng, fs) for fs in (4e6, 10e6, 25e6, 50e6)}, indent=1))
