"""Where a transform's time goes inside the one-workgroup-per-transform PCPS kernel: shader cycles between phase boundaries,
summed by wave 0 of every workgroup (diagnostic build:  tools/build_variant.sh stamps pcps_fused -DSDR_FUSED_STAMPS, then
SYDR_AMD_LIB=tools/scratch/var/lib_stamps.so python tools/pcps_fused_phases.py [fs_mhz: 25 | 50])."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sydr_amd
from sydr_amd.engine import Engine, FMT_CI8
e = Engine(0)
lib = sydr_amd.load()
fs = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 25e6
n = int(fs / 1000)
whole = 5 if n == 25000 else 10          # whole transforms (units) per workgroup and call at 32 PRNs x 41 bins
e.iq_alloc(n, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * n).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
slots = np.arange(32)
e.set_option("pcps_fused", 1)
for _ in range(30):
    e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
buf = np.zeros((256, 8), dtype=np.uint64)
lib.sdr_debug_fused_stamps(None, 1)
reps = 20
for _ in range(reps):
    e.pcps(slots, 0, fs, 0.0, 5000.0, 250.0)
lib.sdr_debug_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
names = ["top barrier (tail of previous, table)", "column item 0", "column item 1", "barrier after columns", "round: Y in place + barrier",
         "round: row stage 1 + barrier", "round: exchange write + barrier", "round: row stage 2 + max"]
per_call = buf.astype(np.float64) / reps
tot = per_call.sum(axis=1)
five = per_call[tot < np.median(tot) * 1.1]      # workgroups with the usual number of units
print(json.dumps({"workgroups_counted": int(len(five)), "units_per_workgroup_assumed": whole,
                  "cycles_per_transform": {nm: float(five[:, i].mean() / whole) for i, nm in enumerate(names)},
                  "total_cycles_per_transform": float(five.sum(axis=1).mean() / whole),
                  "max_workgroup_cycles_per_call": float(tot.max()), "median_workgroup_cycles_per_call": float(np.median(tot))}, indent=1))
