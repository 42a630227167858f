import sys, os, time, configparser
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd import engine as engmod
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
from sydr_amd.channel.manager import ChannelManager
from sydr_amd.signal.iqsource import RFSignal
FS = bench.FS
n_ms, n_ch = 400, 32
eng = Engine(0)
total = int(n_ms * 1e-3 * FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8); eng.code_slots(32)
sats = bench.satellites()[:n_ch]
eng.iq_synth(sats, FS, 12.0, 20260003, 0, total)
raw = eng.iq_download(total, 0)
rf = RFSignal(dict(filepath="none", sampling_frequency=FS, is_complex="true", intermediate_frequency=0.0, data_size=8))
cfg = configparser.ConfigParser(); cfg.read(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "channel_GPS_L1CA_kaplan.ini"))
mgr = ChannelManager(rf, engine=eng, keepCorrelationMap=False)
mgr.addChannel(ChannelL1CA_Kaplan, cfg, n_ch)
for s in sats: mgr.requestTracking(s["prn"])
spms = int(FS * 1e-3)
lib = eng._lib
acc, acc_up = [], []
def timed_of(orig, into):
    def timed(*a):
        t0 = time.perf_counter(); r = orig(*a); into.append(time.perf_counter() - t0); return r
    return timed
timed = timed_of(lib.sdr_bank_tick_mirrored, acc)
timed_up = timed_of(lib.sdr_iq_upload_begin, acc_up)
class Wrap:
    def __getattr__(self, n):
        return timed if n == "sdr_bank_tick_mirrored" else timed_up if n == "sdr_iq_upload_begin" else getattr(lib, n)
tt = []
for k in range(n_ms):
    if k == 50:
        mgr.bank.device._lib = Wrap()
        eng._lib = Wrap()
    t0 = time.perf_counter()
    mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
    t1 = time.perf_counter()
    pk = mgr.run()
    t2 = time.perf_counter()
    tt.append((t1 - t0, t2 - t1))
tt = np.array(tt[100:]); acc_a = np.array(acc[50:])
print("addNewRFData us", np.median(tt[:,0])*1e6, "of which sdr_iq_upload_begin", np.median(acc_up[50:])*1e6, "run us", np.median(tt[:,1])*1e6,
      "of which sdr_bank_tick_mirrored us", np.median(acc_a)*1e6)
# kernel-only: prof
eng.prof_enable(True); eng.prof_reset()
rec, st, done, _ = mgr.bank.device.step(np.arange(32, dtype=np.int32), 1)
print("step done", done[:4], "kernel ms", eng.prof_read("track_kernel"))
