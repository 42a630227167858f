"""Where a segment-by-segment E/P/L pass (bench.py's headline since round 6) loses time against one resident plan: range launches of
one plan, resident segment plans, create + run + destroy in turn, creation alone.   python tools/epl_segments.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd._lib import EPL_ITEM_DTYPE
eng = Engine(0)
FS, N_CH, SP = bench.FS, bench.N_CH, bench.SPACING
total = int(60.0 * FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8); eng.code_slots(N_CH)
sats = bench.satellites(N_CH)
for s, sat in enumerate(sats): eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, FS, total)
n_run = n_epochs * N_CH
st = eng.stream_create()
seg = 10000 * N_CH
starts = list(range(0, n_run, seg))
pin = eng.host_alloc(n_run * EPL_ITEM_DTYPE.itemsize, np.uint8).view(EPL_ITEM_DTYPE); pin[:] = items[:n_run]
def t(fn, reps=8):
    fn(); eng.stream_sync(st); eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.stream_sync(st); eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3
whole = eng.epl_plan(items, SP, FS)
print("one plan, one launch        %.3f ms" % t(lambda: whole.run(0, n_run, stream=st)))
print("one plan, %d range launches  %.3f ms" % (len(starts), t(lambda: [whole.run(j, min(seg, n_run - j), stream=st) for j in starts])))
plans = [eng.epl_plan(pin[j:min(j + seg, n_run)], SP, FS) for j in starts]
print("resident segment plans      %.3f ms" % t(lambda: [p.run(stream=st) for p in plans]))
for p in plans: p.close()
def seq():
    for j in starts:
        p = eng.epl_plan(pin[j:min(j + seg, n_run)], SP, FS); p.run(stream=st); p.close()
print("create, run, destroy in turn %.3f ms" % t(seq))
def create_only():
    for j in starts:
        p = eng.epl_plan(pin[j:min(j + seg, n_run)], SP, FS); p.close()
print("create + destroy only       %.3f ms" % t(create_only))
