import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sydr_amd.engine import Engine, FMT_CI8
eng = Engine(0)
total = int(1.2 * bench.FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(bench.N_CH)
sats = bench.satellites(bench.N_CH)
for s, sat in enumerate(sats):
    eng.load_gps_code(s, sat["prn"])
eng.iq_synth(sats, bench.FS, 12.0, 20260003, 0, total)
items, n_epochs = bench.truth_items(sats, bench.FS, total)
for n_ch in [int(a) for a in sys.argv[1:]] or [768]:
    r = bench.closed_loop_leg(eng, items, 1000, n_ch=n_ch)
    print(n_ch, r["us_per_epoch"], r["channels_lost"])
eng.close()
