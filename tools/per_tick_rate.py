"""The literal drop-in mode: ChannelManager.addNewRFData(1 ms) + run() per millisecond, 32 channels @ 25 MHz.
Host-driven (one PCIe upload + one batched launch + Python bookkeeping per tick)."""
import configparser, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from test_host_layer import KAPLAN_INI
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
from sydr_amd.channel.manager import ChannelManager
from sydr_amd.signal.iqsource import RFSignal
from sydr_amd.utils.enumerations import ChannelMessage, ChannelState

FS = bench.FS
n_ms = int(sys.argv[1]) if len(sys.argv) > 1 else 600
eng = Engine(0)
# synthesise the stream on the device, then bring it to the host: the host is the IQ source in this mode
total = int(n_ms * 1e-3 * FS) // 8 * 8
eng.iq_alloc(total, FMT_CI8)
eng.code_slots(32)
sats = bench.satellites(0)
eng.iq_synth(sats, FS, 12.0, 20260003, 0, total)
raw = eng.iq_download(total, 0)
rf = RFSignal(dict(filepath="none", sampling_frequency=FS, is_complex="true", intermediate_frequency=0.0, data_size=8))
cfg = configparser.ConfigParser(); cfg.read_string(KAPLAN_INI)
mgr = ChannelManager(rf, engine=eng, keepCorrelationMap=False)
mgr.addChannel(ChannelL1CA_Kaplan, cfg, 32)
for s in sats:
    mgr.requestTracking(s["prn"])
spms = int(FS * 1e-3)
t_acq = t_trk = 0.0
n_trk_ticks = 0
n_pkts = 0
import cProfile, pstats
pr = cProfile.Profile()
for k in range(n_ms):
    if k == n_ms - 150 and "--profile" in sys.argv:
        pr.enable()
    t0 = time.perf_counter()
    mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
    pk = mgr.run()
    dt = time.perf_counter() - t0
    tracking = sum(p["type"] is ChannelMessage.TRACKING_UPDATE for p in pk)
    n_pkts += tracking
    if tracking == 32:
        t_trk += dt; n_trk_ticks += 1
    else:
        t_acq += dt
states = [ch.channelState for ch in mgr.channels.values()]
print(f"{n_ms} ms fed; channels tracking at the end: {sum(s is ChannelState.TRACKING for s in states)}/32")
print(f"ticks with all 32 channels tracking: {n_trk_ticks}, {t_trk / max(1, n_trk_ticks) * 1e3:.3f} ms per tick "
      f"= {1e-3 / (t_trk / max(1, n_trk_ticks)):.2f}x real time; other ticks (buffering/acquisition) {t_acq * 1e3:.1f} ms total")
if "--profile" in sys.argv:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
