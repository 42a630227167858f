"""The literal drop-in mode: ChannelManager.addNewRFData(1 ms) + run() per millisecond, 32 channels @ 25 MHz,
the host as IQ source.  One tick = the slab queued for the ring (sdr_iq_upload_begin) + ONE library call
(sdr_bank_tick_mirrored: who is ready, one epoch for them from the device-resident bank, the host's mirrors updated
in place).  Reported twice: with the packets left unread (what a consumer that only wants some of them pays) and
with the "type" of every packet read (what the reference's Receiver loop does to route them, receiver.py:291-299;
the packets are dicts that fill themselves when more is asked of them)."""
import configparser, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
from sydr_amd.channel.manager import ChannelManager
from sydr_amd.signal.iqsource import RFSignal
from sydr_amd.utils.enumerations import ChannelMessage, ChannelState

FS = bench.FS


WARMUP_MS = 100     # ticks not counted: acquisition, the first launches of each kernel (code objects load on first use), the first block


def measure(n_ms=600, n_ch=32, profile=False, engine=None, read_ahead=0, tick_server=False, bind=True, pinned_source=False):
    """read_ahead > 0: the same loop with ChannelManager.enableReadAhead(read_ahead) and the stream served from a file
    through this package's RFSignal (what lets the manager look ahead); the calls per tick are the reference's.
    The first WARMUP_MS ticks are fed and run but not counted (their time is reported as warmup_ms_total).
    tick_server: the steady ticks answered by the resident kernel (sdr_set_option "tick_server")."""
    eng = engine or Engine(0)
    if bind:        # (the thread onto the CPUs next to the GPU: a tick is a few round trips through page-locked words)
        try:
            eng.set_option("bind_thread_to_device", 1)
        except Exception:
            pass
    eng.set_option("tick_server", 1 if tick_server else 0)
    served0 = eng.tick_server_stats()["served"]
    # synthesise the stream on the device, then bring it to the host: the host is the IQ source in this mode
    total = int(n_ms * 1e-3 * FS) // 8 * 8
    eng.iq_alloc(total, FMT_CI8)
    eng.code_slots(max(32, n_ch))
    sats = bench.satellites(n_ch)
    eng.iq_synth(sats, FS, 12.0, 20260003, 0, total)
    raw = eng.iq_download(total, 0)
    pinned_block = None
    if pinned_source:     # the recording in page-locked memory of the engine's (a front end's DMA buffer): slabs are read in place
        pinned_block = eng.host_alloc(raw.size, raw.dtype)
        pinned_block[:] = raw
        raw = pinned_block
    tmp = None
    if read_ahead:
        import tempfile
        tmp = tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, suffix=".iq")
        raw.tofile(tmp.name)
    rf = RFSignal(dict(filepath=tmp.name if tmp else "none", sampling_frequency=FS, is_complex="true", intermediate_frequency=0.0, data_size=8))
    cfg = configparser.ConfigParser(); cfg.read(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "channel_GPS_L1CA_kaplan.ini"))
    mgr = ChannelManager(rf, engine=eng, keepCorrelationMap=False)
    mgr.addChannel(ChannelL1CA_Kaplan, cfg, n_ch)
    for s in sats:
        mgr.requestTracking(s["prn"])
    if read_ahead:
        mgr.enableReadAhead(read_ahead)
    spms = int(FS * 1e-3)
    lazy, eager, other, warm = [], [], 0.0, 0.0
    pr = None
    if profile:
        import cProfile
        pr = cProfile.Profile()
    for k in range(n_ms):
        if pr and k == n_ms - 150:
            pr.enable()
        t0 = time.perf_counter()
        mgr.addNewRFData(rf.getMilliseconds(1) if read_ahead else raw[2 * k * spms:2 * (k + 1) * spms])
        pk = mgr.run()
        t1 = time.perf_counter()
        tracking = sum(1 for p in pk if p["type"] is ChannelMessage.TRACKING_UPDATE)   # materialises every packet
        t2 = time.perf_counter()
        if k < WARMUP_MS:
            warm += t2 - t0
        elif tracking == n_ch:
            lazy.append(t1 - t0)
            eager.append(t2 - t0)
        else:
            other += t2 - t0
    if pr:
        import pstats
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    tracking_now = sum(ch.channelState is ChannelState.TRACKING for ch in mgr.channels.values())
    lost = sum(getattr(ch, "lostLock", False) for ch in mgr.channels.values())
    srv = eng.tick_server_stats()
    mgr.close()
    eng.set_option("tick_server", 0)
    if pinned_block is not None:
        eng.host_free(pinned_block)
    # read-ahead: a tick in ~50 pays for the block, so the MEAN per tick is the honest figure there (the median is the
    # price of a tick that only hands packets out); the plain loop keeps its median (every tick is alike)
    avg = (lambda v: float(np.mean(v))) if read_ahead else (lambda v: float(np.median(v)))
    res = dict(channels=n_ch, fs_hz=FS, ms_fed=n_ms, ticks_all_tracking=len(lazy), channels_tracking_at_end=tracking_now,
               channels_lost=lost, read_ahead_ms=int(read_ahead), statistic="mean" if read_ahead else "median",
               ms_per_tick=avg(lazy) * 1e3 if lazy else None,
               x_realtime=1e-3 / avg(lazy) if lazy else None,
               ms_per_tick_all_packets_read=avg(eager) * 1e3 if eager else None,
               x_realtime_all_packets_read=1e-3 / avg(eager) if eager else None,
               mean_ms_per_tick=float(np.mean(lazy)) * 1e3 if lazy else None, p99_ms_per_tick=float(np.percentile(lazy, 99)) * 1e3 if lazy else None,
               max_ms_per_tick=float(np.max(lazy)) * 1e3 if lazy else None,
               other_ticks_ms_total=other * 1e3, warmup_ms_excluded=WARMUP_MS, warmup_ms_total=warm * 1e3)
    if tick_server:
        res.update(tick_server=True, requests_answered=srv["served"] - served0, servers_started=srv["starts"], gave_up=srv["disabled"],
                   device_us_per_request={k: v / max(1, srv["served"]) for k, v in srv["device_us_total"].items()},
                   channel0_us_per_request={k: v / max(1, srv["served"]) for k, v in srv["channel0_us_total"].items()})
    if tmp is not None:
        tmp.close()
    return res


if __name__ == "__main__":
    n_ms = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
    ra = int(sys.argv[sys.argv.index("--read-ahead") + 1]) if "--read-ahead" in sys.argv else 0
    print(json.dumps(measure(n_ms, profile="--profile" in sys.argv, read_ahead=ra, tick_server="--tick-server" in sys.argv, bind="--no-bind" not in sys.argv, pinned_source="--pinned-source" in sys.argv)))
