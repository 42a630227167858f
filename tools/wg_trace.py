"""Debug: per-workgroup timeline of epl_kernel (needs a build with -DSDR_TRACE_WG loaded from SYDR_TRACE_LIB)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sydr_amd._lib as L
from sydr_amd.engine import Engine, make_items, FMT_CI8
e = Engine(0)
cap = 8 * 400000
e.iq_alloc(cap, FMT_CI8)
e.iq_upload(np.random.default_rng(0).integers(-60, 60, 2 * cap).astype(np.int8), 0)
e.code_slots(32)
for s in range(32):
    e.load_gps_code(s, s + 1)
rng = np.random.default_rng(1)
n_items, n, step = 32000, 25000, 0.04092
items = make_items(np.arange(n_items) % 32, n, rng.integers(0, cap, n_items), 1000.0, 0.3, 0.01, step)
plan = e.epl_plan(items, (-0.5, 0.0, 0.5), 25e6)
for _ in range(3):
    plan.run()
e.sync()
lib = L.load()
buf = np.zeros(3 * n_items, dtype=np.uint64)
rc = lib.sdr_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), n_items)
assert rc == 0, rc
t0, t1, hw = buf[0::3].astype(np.int64), buf[1::3].astype(np.int64), buf[2::3]
tick = 1e-8  # wall_clock64: 100 MHz
print("kernel span %.3f ms; WG duration mean %.2f us (min %.2f max %.2f)" % ((t1.max() - t0.min()) * tick * 1e3, (t1 - t0).mean() * tick * 1e6, (t1 - t0).min() * tick * 1e6, (t1 - t0).max() * tick * 1e6))
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print("time-weighted mean concurrency %.0f WGs, max %d" % ((conc[:-1] * dt).sum() / dt.sum(), conc.max()))
hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 3; wave = hwid & 0xf
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs", len(np.unique(key)), "WGs per CU min/max", np.bincount(key)[np.bincount(key) > 0].min(), np.bincount(key).max())
# peak residency per CU: sample at mid-kernel
tm = (t0.min() + t1.max()) // 2
live = (t0 <= tm) & (t1 > tm)
per_cu = np.bincount(key[live])
print("mid-kernel live WGs", live.sum(), "per CU max", per_cu.max(), "hist", np.bincount(per_cu[per_cu > 0]))
print("simd use", np.bincount(simd[live]), "wave slots", np.bincount(wave[live]))
# dispatch: how long after a WG ends does the next one start on the same CU?
order = np.argsort(t0)
print("start times: first 10 WG starts (ticks)", (np.sort(t0)[:3000:300] - t0.min()))
