"""Soak of the literal per-millisecond loop with read-ahead against the plain loop on the real engine: 32 channels @ 25 MHz
(the headline geometry), a few seconds of stream, read-ahead blocks of random lengths re-drawn now and then (the next
block queued on the device while the current one is handed out), late joiners -- every packet of every tick equal to the
plain loop's, bit for bit (both run an epoch on the cluster of 8 workgroups).  --general: the plain loop's library-side tick
against the manager's general tick instead.  --tick-server: the plain loop answered by the resident tick server
(sdr_set_option "tick_server": late joiners stop and restart it) against the plain loop on launches.  --steady: every channel
requested at the start (no late joiners: a block is queued ahead all the time).
Usage: python tests/stress_readahead.py [ms] [seed] [--general | --tick-server] [--steady]"""
import configparser, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
from sydr_amd.channel.manager import ChannelManager
from sydr_amd.signal.iqsource import RFSignal

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n_ms=2000, seed=1, fs=25e6, n_ch=32, general=False, steady=False, tick_server=False):
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    spms = int(fs * 1e-3)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(n_ch)]
    total = n_ms * spms
    eng.iq_alloc(total, FMT_CI8)
    eng.code_slots(n_ch)
    eng.iq_synth(sats, fs, 10.0, int(rng.integers(1, 1 << 30)), 0, total)
    tmp = tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, suffix=".iq")
    eng.iq_download(total, 0).tofile(tmp.name)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))
    late = [] if steady else sorted(int(t) for t in rng.integers(150, n_ms // 2, 6))   # ticks at which one more satellite is requested
    redraw = {int(t): int(b) for t, b in zip(rng.integers(100, n_ms - 100, 8), rng.choice([7, 16, 25, 40, 50], 8))}

    def receiver(read_ahead, steady=True, server=False):
        rf = RFSignal(dict(filepath=tmp.name, sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        eng.set_option("tick_server", 1 if server else 0)
        mgr = ChannelManager(rf, engine=eng, keepCorrelationMap=False)
        mgr.STEADY_TICK = steady
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, n_ch)
        for s in sats[:n_ch - len(late)]:
            mgr.requestTracking(s["prn"])
        if read_ahead:
            mgr.enableReadAhead(read_ahead)
        ticks, joined, queued = [], 0, 0
        t0 = time.perf_counter()
        for k in range(n_ms):
            if joined < len(late) and k == late[joined]:
                mgr.requestTracking(sats[n_ch - len(late) + joined]["prn"])
                joined += 1
            if read_ahead and k in redraw:
                mgr.enableReadAhead(redraw[k])
            mgr.addNewRFData(rf.getMilliseconds(1))
            ticks.append([dict(p) for p in mgr.run()])
            queued += mgr._ahead is not None
        dt = time.perf_counter() - t0
        if server:
            st = eng.tick_server_stats()
            assert not st["disabled"], st
            queued = st["served"]
        mgr.close()
        eng.set_option("tick_server", 0)
        return ticks, dt, queued

    plain, t_plain, _ = receiver(0)
    if tick_server:  # the same ticks answered by the resident server
        ahead, t_ahead, queued = receiver(0, server=True)
    elif general:      # the library-side tick (sdr_bank_tick_mirrored) against the manager's general tick instead
        ahead, t_ahead, queued = receiver(0, steady=False)
    else:
        ahead, t_ahead, queued = receiver(50)
    n_pk = 0
    for k, (a, b) in enumerate(zip(plain, ahead)):
        key = lambda p: (p["cid"], p["type"].value)
        a, b = sorted(a, key=key), sorted(b, key=key)
        assert [key(p) for p in a] == [key(p) for p in b], k
        for p, q in zip(a, b):
            for name in p:
                same = p[name] == q[name] or (isinstance(p[name], float) and np.isnan(p[name]) and np.isnan(q[name]))
                assert same or name == "peak_ratio", (k, key(p), name, p[name], q[name])
            n_pk += 1
    tmp.close()
    return n_pk, t_plain, t_ahead, queued


if __name__ == "__main__":
    n_ms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    general = "--general" in sys.argv
    n_pk, t_plain, t_ahead, queued = run(n_ms, seed, general=general, steady="--steady" in sys.argv, tick_server="--tick-server" in sys.argv)
    if "--tick-server" in sys.argv:
        print(f"{n_ms} ticks x 32 channels: {n_pk} packets equal bit for bit between the plain loop on launches ({t_plain:.2f} s incl. "
              f"materialising) and the plain loop answered by the resident tick server ({t_ahead:.2f} s; {queued} requests answered)")
        sys.exit(0)
    if general:
        print(f"{n_ms} ticks x 32 channels: {n_pk} packets equal bit for bit between the library-side tick ({t_plain:.2f} s) and the "
              f"manager's general tick ({t_ahead:.2f} s)")
        sys.exit(0)
    print(f"{n_ms} ticks x 32 channels: {n_pk} packets equal bit for bit between the plain loop ({t_plain:.2f} s incl. materialising) and "
          f"read-ahead ({t_ahead:.2f} s; a block queued ahead during {queued} ticks)")
