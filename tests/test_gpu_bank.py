"""GPU tests of the device-resident channel bank (sdr_bank_*), the per-batch streams and the multi-GPU
partition of north_star exercised on ONE GPU: the same stream in two engines, channels split 16/16, outputs
bitwise equal to the 32-in-one launch (per-channel outputs must not depend on the sharding)."""
import numpy as np
import pytest

from oracle import sydr_oracle as orc
from test_gpu_tracking import initial_state, loop_cfg
from test_oracle_golden import BORRE_CFG, KAPLAN_CFG, trajectory_iq

from sydr_amd._lib import LOOP_CFG_DTYPE, TRACK_STATE_DTYPE
from sydr_amd.channel.manager import shard_channels
from sydr_amd.engine import FMT_CI8, Engine, make_items

pytestmark = pytest.mark.gpu


def as_row(struct, dtype):
    return np.frombuffer(bytes(struct), dtype=dtype)[0].copy()


def test_bank_steps_equal_the_closed_loop_run(engine):
    """One epoch per call from the bank == the same epochs inside one closed-loop launch (one workgroup per channel
    either way: the summation order, hence every bit, is the same); states stay on the device in between."""
    g, fs, raw = trajectory_iq()
    n = raw.size // 2 // 8 * 8
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw[:2 * n], 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    acq = g["kaplan_acq"]
    st_k = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=1)
    st_b = initial_state(0, fs, acq[3], int(acq[5]), BORRE_CFG, slot=1)
    engine.track_cluster(1)
    try:
        _bank_vs_closed_loop(engine, fs, st_k, st_b)
    finally:
        engine.track_cluster(0)


def test_one_epoch_steps_run_on_the_cluster_a_block_launch_uses(engine):
    """With the cluster size left to the library, a one-epoch step (a receiver tick) runs as two plain launches cut at
    the exchange of the partial sums, on the 8 workgroups per channel the 60-epoch cooperative launch spreads a channel
    over: 40 single epochs + a block of 20 == the 60 epochs of one launch, bit for bit; and the one-launch form of the
    tick (`track_one_launch_tick`) agrees with it to rounding."""
    g, fs, raw = trajectory_iq()
    n = raw.size // 2 // 8 * 8
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw[:2 * n], 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    acq = g["kaplan_acq"]
    st_k = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=1)
    st_b = initial_state(0, fs, acq[3], int(acq[5]), BORRE_CFG, slot=1)
    _bank_vs_closed_loop(engine, fs, st_k, st_b)
    recs = {}
    for one_launch in (0, 1):
        engine.set_option("track_one_launch_tick", one_launch)
        try:
            bank = engine.bank(4)
            bank.put(0, as_row(st_k, TRACK_STATE_DTYPE), as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE))
            recs[one_launch] = np.concatenate([bank.step([0], 1)[0][0] for _ in range(30)])
            bank.close()
        finally:
            engine.set_option("track_one_launch_tick", 0)
    a, b = recs[0], recs[1]
    assert a.tobytes() != b.tobytes()                       # (another order of additions: the forms do differ in the last bits)
    for name in ("start_sample", "n_samples", "lock_state", "track_flags", "nav_bit"):
        assert np.array_equal(a[name], b[name]), name
    scale = np.hypot(a["corr"][:, 2], a["corr"][:, 3])[:, None]
    assert np.max(np.abs(a["corr"] - b["corr"]) / scale) < 1e-9
    for name in ("carrier_hz", "code_hz", "dll", "pll", "fll", "cn0"):
        assert np.allclose(a[name], b[name], rtol=1e-9, atol=1e-9), name


def _bank_vs_closed_loop(engine, fs, st_k, st_b):
    ref_k, traj_k = engine.track_closed_loop([st_k], loop_cfg(1, fs, KAPLAN_CFG), 60)
    ref_b, traj_b = engine.track_closed_loop([st_b], loop_cfg(0, fs, BORRE_CFG), 60)
    bank = engine.bank(8)
    bank.put(5, as_row(st_k, TRACK_STATE_DTYPE), as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE))
    bank.put(2, as_row(st_b, TRACK_STATE_DTYPE), as_row(loop_cfg(0, fs, BORRE_CFG), LOOP_CFG_DTYPE))
    recs = []
    for k in range(40):
        rec, states, done, _ = bank.step([5, 2], 1)
        assert list(done) == [1, 1]
        recs.append(rec[:, 0])
    rec, states, done, _ = bank.step([2, 5], 20)            # a block, other order: continues from the device state
    assert list(done) == [20, 20]
    got_k = np.concatenate([np.array([r[0] for r in recs]), rec[1]])
    got_b = np.concatenate([np.array([r[1] for r in recs]), rec[0]])
    assert got_k.tobytes() == traj_k[0].tobytes() and got_b.tobytes() == traj_b[0].tobytes()
    assert states[1].tobytes() == bytes(ref_k[0]) and bank.get(2).tobytes() == bytes(ref_b[0])
    with pytest.raises(Exception, match="not in the bank"):
        bank.step([1], 1)
    with pytest.raises(Exception, match="listed twice"):
        bank.step([5, 5], 1)
    bank.close()


def test_bank_tick_ingests_and_tracks_in_one_call(engine):
    """sdr_bank_tick == CircularBuffer.shift of one slab followed by one epoch of the ready channels."""
    g, fs, raw = trajectory_iq()
    spms = int(fs * 1e-3)
    ring = 100 * spms
    acq = g["kaplan_acq"]
    st = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=0)
    row_s, row_c = as_row(st, TRACK_STATE_DTYPE), as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE)

    def run(fused):
        engine.iq_alloc(ring, FMT_CI8)
        engine.code_slots(1)
        engine.load_gps_code(0, 7)
        bank = engine.bank(1)
        bank.put(0, row_s, row_c)
        out, cur, n_next = [], int(acq[5]), int(st.n_samples)
        for ms in range(120):
            slab, off = raw[2 * ms * spms:2 * (ms + 1) * spms], (ms * spms) % ring
            written = (ms + 1) * spms
            ready = [0] if written - cur >= n_next else []
            if fused:
                rec, states, done = bank.tick(slab, off, ready)
            else:
                engine.iq_upload(slab, off)
                rec, states, done = bank.tick(None, 0, ready)
            if ready:
                out.append(rec[0].copy())
                cur, n_next = int(states[0]["current_sample"]), int(states[0]["n_samples"])
        bank.close()
        return np.array(out)

    a, b = run(True), run(False)
    assert len(a) > 100 and a.tobytes() == b.tobytes()
    ref = g["kaplan_epochs"][:len(a)]
    assert np.array_equal(a["n_samples"], ref[:, 1].astype(np.int32))
    assert np.allclose(a["carrier_hz"], ref[:, 15], rtol=1e-9, atol=0)


def _stream_and_items(fs, n_ch, n_ms, seed=77):
    """A synthetic multi-satellite stream description + open-loop items for every channel-epoch."""
    rng = np.random.default_rng(seed)
    sats = [dict(prn=1 + c, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.uniform(0, 1)), amp=6.0) for c in range(n_ch)]
    spms = int(fs * 1e-3)
    items = []
    for c, s in enumerate(sats):
        step = orc.CODE_RATE * (1.0 + s["doppler"] / 1575.42e6) / fs
        start0 = int(np.ceil((1023.0 - s["code_phase"]) % 1023.0 / step))
        for k in range(n_ms - 2):
            start = start0 + int(round(k * 1023.0 / step))
            items.append((c, spms, start, s["doppler"], 0.1 * k, 0.0, step))
    cols = list(zip(*items))
    return sats, make_items(*[np.array(col) for col in cols])


def test_two_engines_on_one_gpu_split_channels_bitwise():
    """north_star's partition on one GPU: the SAME stream replicated in two engines, 32 channels split 16/16 with
    shard_channels -- open-loop correlators and closed-loop trajectories of every channel bitwise equal to the
    32-in-one launch."""
    fs, n_ch, n_ms = 25e6, 32, 12
    spms = int(fs * 1e-3)
    sats, items = _stream_and_items(fs, n_ch, n_ms)
    engines = [Engine(0) for _ in range(3)]
    try:
        for e in engines:                                    # stream replicated: same seed, same satellites
            e.iq_alloc(n_ms * spms, FMT_CI8)
            e.code_slots(n_ch)
            for c in range(n_ch):
                e.load_gps_code(c, sats[c]["prn"])
            e.iq_synth(sats, fs, 12.0, 4242, 0, n_ms * spms)
        assert engines[0].iq_download(4096, 12345).tobytes() == engines[2].iq_download(4096, 12345).tobytes()
        whole = engines[0].epl_batch(items, (-0.5, 0.0, 0.5), fs)
        cfg = loop_cfg(1, fs, KAPLAN_CFG)
        mk = lambda c: initial_state(1, fs, sats[c]["doppler"] + 40.0, int(items["start_sample"][items["code_slot"] == c][0]),
                                     KAPLAN_CFG, slot=c)
        engines[0].track_cluster(4)
        all_states, all_traj = engines[0].track_closed_loop([mk(c) for c in range(n_ch)], cfg, 8)
        for rank, e in enumerate(engines[1:]):
            mine = shard_channels(n_ch, rank, 2)
            assert len(mine) == 16
            sel = np.isin(items["code_slot"], mine)
            part = e.epl_batch(items[sel], (-0.5, 0.0, 0.5), fs)
            assert part.tobytes() == whole[sel].tobytes()
            e.track_cluster(4)
            st, traj = e.track_closed_loop([mk(c) for c in mine], cfg, 8)
            assert traj.tobytes() == all_traj[mine].tobytes()
            assert b"".join(bytes(s) for s in st) == b"".join(bytes(all_states[c]) for c in mine)
    finally:
        for e in engines:
            e.close()


def test_one_stream_per_channel_batch(engine):
    """Ranges of a plan launched on two streams of one engine (one per channel batch) == the single-stream run."""
    fs, n_ch, n_ms = 25e6, 8, 10
    spms = int(fs * 1e-3)
    sats, items = _stream_and_items(fs, n_ch, n_ms, seed=5)
    engine.iq_alloc(n_ms * spms, FMT_CI8)
    engine.code_slots(n_ch)
    for c in range(n_ch):
        engine.load_gps_code(c, sats[c]["prn"])
    engine.iq_synth(sats, fs, 12.0, 99, 0, n_ms * spms)
    plan = engine.epl_plan(items, (-0.5, 0.0, 0.5), fs)
    plan.run()
    ref = plan.fetch()
    s1, s2 = engine.stream_create(), engine.stream_create()
    assert (s1, s2) == (1, 2) or s2 == s1 + 1
    plan2 = engine.epl_plan(items, (-0.5, 0.0, 0.5), fs)
    half = len(items) // 2
    plan2.run(0, half, stream=s1)
    plan2.run(half, len(items) - half, stream=s2)
    engine.stream_sync(s1)
    engine.stream_sync(s2)
    assert plan2.fetch().tobytes() == ref.tobytes()
    with pytest.raises(Exception, match="stream id"):
        plan2.run(0, 1, stream=99)
    # a plan does not survive a re-allocation of the code tables (ADVICE: stale lut_words / slots)
    engine.code_slots(n_ch + 1)
    with pytest.raises(Exception, match="stale"):
        plan.run()
    plan.close()
    plan2.close()


def test_manager_with_more_than_32_channels(engine):
    """40 channels through the drop-in ChannelManager on the real engine: the code tables are re-allocated and the
    device bank re-created when the 33rd channel arrives (after the first channels already track); every requested
    satellite is acquired, tracked per tick and ends on its Doppler."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    from sydr_amd.utils.enumerations import ChannelMessage, ChannelState
    fs, n_ms = 10e6, 260
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(4040)
    # Dopplers close to the 250 Hz acquisition grid: a search that starts half a bin off can end on the Costas loop's
    # +-500 Hz alias, which is receiver physics and not what this test is about
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.uniform(0, 1)), amp=4.0) for c in range(36)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(36)
    engine.iq_synth(sats, fs, 10.0, 4041, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))
    rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
    mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
    mgr.addChannel(ChannelL1CA_Kaplan, cfg, 8)
    for s in sats[:8]:
        mgr.requestTracking(s["prn"])
    tracked = 0
    for k in range(n_ms):
        if k == 60:                                   # the first eight already track: now the bank has to grow
            old = mgr.bank
            mgr.addChannel(ChannelL1CA_Kaplan, cfg, 32)
            for s in sats[8:]:
                mgr.requestTracking(s["prn"])
            assert mgr.bank is not old and mgr.nbChannels == 40
        mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
        tracked += sum(1 for p in mgr.run() if p["type"] is ChannelMessage.TRACKING_UPDATE)
    chans = [ch for ch in mgr.channels.values() if ch.channelState is not ChannelState.IDLE]
    assert len(chans) == 36 and all(ch.channelState is ChannelState.TRACKING for ch in chans)
    for ch in chans:
        dop = sats[ch.satelliteID - 1]["doppler"]
        err = abs(ch.carrierFrequency - dop)
        # (a data-bit edge inside the acquired millisecond can put the search 500 Hz off, where a Costas loop with 1 ms
        # epochs locks just as well -- the reference does the same; not what this test is about)
        assert min(err, abs(err - 500.0)) < 40.0, (ch.channelID, ch.carrierFrequency, dop)
    assert sum(abs(ch.carrierFrequency - sats[ch.satelliteID - 1]["doppler"]) < 40.0 for ch in chans) >= 30
    assert tracked > 8 * 250 + 28 * 180
    mgr.close()


def test_read_ahead_on_the_device_equals_plain_ticks(engine, tmp_path):
    """The per-millisecond loop with ChannelManager.enableReadAhead (blocks of epochs tracked ahead in one launch and
    handed out tick by tick) against the plain one-device-call-per-tick loop: 12 channels at 10 MHz with late joiners
    (acquisition in the middle of a replay), every packet of every tick bitwise equal."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    fs, n_ms = 10e6, 400
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(5050)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 5051, 0, total)
    path = tmp_path / "iq.bin"
    engine.iq_download(total, 0).tofile(path)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(read_ahead):
        rf = RFSignal(dict(filepath=str(path), sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats[:8]:
            mgr.requestTracking(s["prn"])
        if read_ahead:
            mgr.enableReadAhead(read_ahead)
        ticks = []
        for k in range(n_ms):
            if k == 150:                                # four more satellites while blocks are being replayed
                for s in sats[8:]:
                    mgr.requestTracking(s["prn"])
            mgr.addNewRFData(rf.getMilliseconds(1))
            ticks.append([dict(p) for p in mgr.run()])
        opened = 0 if mgr._readahead is None else mgr._readahead.tick
        mgr.close()
        return ticks, opened

    plain, _ = receiver(0)
    ahead, _ = receiver(40)
    # (floating fields compared relative to the prompt magnitude; everything else exactly)
    n_trk, worst = 0, 0.0
    for k, (a, b) in enumerate(zip(plain, ahead)):
        key = lambda p: (p["cid"], p["type"].value)
        a, b = sorted(a, key=key), sorted(b, key=key)
        assert [key(p) for p in a] == [key(p) for p in b], k
        for p, q in zip(a, b):
            assert p.keys() == q.keys()
            scale = max(1.0, float(np.hypot(p.get("i_prompt", 0.0), p.get("q_prompt", 0.0))))
            for name in p:
                if isinstance(p[name], float) and name not in ("peak_ratio",):
                    ref = scale if name[:2] in ("i_", "q_") else max(1.0, abs(p[name]))
                    if not (np.isnan(p[name]) and np.isnan(q[name])):
                        worst = max(worst, abs(p[name] - q[name]) / ref)
                else:
                    assert p[name] == q[name] or name == "peak_ratio", (k, name, p[name], q[name])
        n_trk += sum(1 for p in b if "i_prompt" in p)
    # a tick runs its epoch on the cluster a block launch would use (two plain launches cut at the exchange of the partial
    # sums: track.hip, launch_track): up to 32 channels both spread a channel over 8 workgroups -- same partition, same
    # order of additions, same bits
    assert worst == 0.0, worst
    assert n_trk > 8 * 350 + 4 * 200


def test_library_side_tick_equals_the_general_tick(engine):
    """The steady tick (sdr_bank_tick_mirrored: readiness, the epoch and the mirror updates in ONE library call, the
    slab queued by addNewRFData) against the manager's general tick (readiness and mirrors in NumPy around
    sdr_bank_tick): 12 channels at 10 MHz from acquisition on, every packet of every tick equal bit for bit, the
    channel objects' attributes too -- and the packets are dicts that fill themselves when read."""
    import configparser
    import os
    import pickle
    from conftest import REPO
    from sydr_amd.channel.bank import LazyPacket
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    from sydr_amd.utils.enumerations import ChannelMessage
    fs, n_ms = 10e6, 300
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(6060)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 6061, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(steady):
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
        mgr.STEADY_TICK = steady
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats:
            mgr.requestTracking(s["prn"])
        ticks, lazy = [], 0
        for k in range(n_ms):
            mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
            pk = mgr.run()
            if k == 200:
                # born with two keys; anything else fills them; what the consumer wrote stays; they travel as dicts
                p = [q for q in pk if q["type"] is ChannelMessage.TRACKING_UPDATE][0]
                assert isinstance(p, LazyPacket) and dict.__len__(p) == 2 and not (p == None)   # noqa: E711 (receiver.py:292)
                p["channel_id"] = 77
                assert dict.__len__(p) == 3 and p["i_prompt"] == p["i_prompt"] and len(p) == 20 and p["channel_id"] == 77
                assert type(pickle.loads(pickle.dumps(p))) is dict and pickle.loads(pickle.dumps(p)) == p
                del p["channel_id"]
            lazy += sum(isinstance(q, LazyPacket) for q in pk)
            ticks.append([dict(q) for q in pk])
        attrs = [(ch.carrierFrequency, ch.codeFrequency, ch.currentSample, ch.codeSinceTOW, int(ch.trackFlags), len(ch.navBits))
                 for ch in mgr.channels.values()]
        mgr.close()
        return ticks, attrs, lazy

    general, attrs_g, _ = receiver(False)
    steady, attrs_s, lazy = receiver(True)
    assert lazy > 2 * 12 * 200
    assert sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in steady for p in t) > 12 * 250
    for k, (a, b) in enumerate(zip(general, steady)):
        assert a == b, k
    assert attrs_g == attrs_s


def test_mirrored_tick_with_three_and_five_tap_channels(engine):
    """sdr_bank_tick_mirrored when the ready channels do not all run the same number of taps (the kernels are compiled per
    tap count: one launch per group inside the call): channels 0, 2, 4, 5 track E/P/L, channels 1 and 3 VE/E/P/L/VL on the
    same satellite; `ran`, the records and the mirror rows come back in ascending channel order, each channel's record and
    state bit for bit what stepping its own group alone gives; a channel without a complete epoch is left out."""
    import ctypes as C
    from sydr_amd import _lib
    from sydr_amd._lib import TICK_UPDATE_DTYPE, TRACK_EPOCH_DTYPE
    g, fs, raw = trajectory_iq()
    n = raw.size // 2 // 8 * 8
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw[:2 * n], 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    acq = g["kaplan_acq"]
    taps_of = [3, 5, 3, 5, 3, 3]
    cfg3 = loop_cfg(1, fs, KAPLAN_CFG)
    cfg5 = loop_cfg(1, fs, KAPLAN_CFG)
    cfg5.n_taps = 5
    for t, (w, nar) in enumerate(zip((-1.0, -0.5, 0.0, 0.5, 1.0), (-0.5, -0.25, 0.0, 0.25, 0.5))):
        cfg5.spacing_wide[t], cfg5.spacing_narrow[t] = w, nar
    st0 = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=1)
    spms = int(fs * 1e-3)

    def fresh_bank():
        bank = engine.bank(6)
        for ch, nt in enumerate(taps_of):
            bank.put(ch, as_row(st0, TRACK_STATE_DTYPE), as_row(cfg5 if nt == 5 else cfg3, LOOP_CFG_DTYPE))
        return bank
    # reference: each tap group stepped on its own, epoch by epoch
    ref_bank = fresh_bank()
    n_ticks = 25
    ref_rec = {ch: [] for ch in range(6)}
    for k in range(n_ticks):
        for group in ([0, 2, 4, 5], [1, 3]):
            rec, states, done, _ = ref_bank.step(group, 1)
            assert list(done) == [1] * len(group)
            for i, ch in enumerate(group):
                ref_rec[ch].append(rec[i, 0].copy())
    ref_bank.close()
    # the mirrored tick over all six
    bank = fresh_bank()
    states = np.zeros(6, dtype=TRACK_STATE_DTYPE)
    for ch in range(6):
        states[ch] = as_row(st0, TRACK_STATE_DTYPE)
    last = np.zeros(6, dtype=TRACK_EPOCH_DTYPE)
    since, host_flags = np.zeros(6, dtype=np.int64), np.zeros(6, dtype=np.int64)
    tracking, lost = np.ones(6, dtype=bool), np.zeros(6, dtype=bool)
    tracking[5] = False                                     # (not tracking: never runs, no update row)
    bank.bind_mirror(states, last, since, tracking, lost, host_flags)
    got = {ch: [] for ch in range(6)}
    for k in range(n_ticks + 3):
        # the write index as a receiver's would stand: far enough ahead for every channel's next epoch -- except in
        # tick 3, when nothing new has arrived and nobody has a complete epoch
        write = (int(states["current_sample"].max()) + int(states["n_samples"].max()) + 8) % n if k != 3 else int(states["current_sample"].max()) % n
        m = bank.tick_mirrored(None, 0, write)       # (the tick's own mirror: the per-tick outputs alternate between two sets)
        ran = bank.ran[:m.n_ran].tolist()
        if k == 3:
            assert ran == [] and m.n_updates == 5
            continue
        assert ran == [0, 1, 2, 3, 4] and m.n_updates == 5 and m.n_lost == 0
        assert bank.updates["channel"][:5].tolist() == [0, 1, 2, 3, 4]
        for i, ch in enumerate(ran):
            got[ch].append(bank.records[i].copy())
            assert last[ch].tobytes() == bank.records[i].tobytes()
        assert (bank.updates["epochs_since_tow"][:5] == since[:5]).all() and since[5] == 0
    for ch in range(5):
        a = np.array(got[ch][:n_ticks]).tobytes()
        b = np.array(ref_rec[ch]).tobytes()
        assert len(got[ch]) == n_ticks + 2 and a == b, ch
    bank.close()


def test_mirrored_tick_refuses_what_it_cannot_trust(engine):
    """sdr_bank_tick_mirrored validates its mirror before anything reaches the device: missing arrays, a mirror of another
    size than the bank, a write index outside the ring -- an error code and a message, no launch."""
    import ctypes as C
    from sydr_amd import _lib
    from sydr_amd._lib import SdrError, TRACK_EPOCH_DTYPE
    engine.iq_alloc(80000, FMT_CI8)
    engine.code_slots(2)
    engine.load_gps_code(0, 3)
    bank = engine.bank(4)
    states, last = np.zeros(4, dtype=TRACK_STATE_DTYPE), np.zeros(4, dtype=TRACK_EPOCH_DTYPE)
    since, host_flags = np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64)
    tracking, lost = np.zeros(4, dtype=bool), np.zeros(4, dtype=bool)
    with pytest.raises(ValueError):
        bank.bind_mirror(states[:3], last, since, tracking, lost, host_flags)         # rows != bank channels
    with pytest.raises(ValueError):
        bank.bind_mirror(states, last, since.astype(np.int32), tracking, lost, host_flags)
    bank.bind_mirror(states, last, since, tracking, lost, host_flags)
    m = bank.tick_mirrored(None, 0, 0)                        # nothing tracking: nothing runs, no update rows
    assert (m.n_ran, m.n_updates, m.n_lost) == (0, 0, 0)
    with pytest.raises(SdrError, match="write index"):
        bank.tick_mirrored(None, 0, 80000)
    with pytest.raises(SdrError, match="write index"):
        bank.tick_mirrored(None, 0, -1)
    lib = _lib.load()
    broken = _lib.TickMirror.from_buffer_copy(m)
    broken.records = None
    assert lib.sdr_bank_tick_mirrored(engine._h, bank._h, None, 0, 0, 0, C.byref(broken)) != 0
    assert b"NULL" in lib.sdr_last_error()
    broken = _lib.TickMirror.from_buffer_copy(m)
    broken.max_channels = 5
    assert lib.sdr_bank_tick_mirrored(engine._h, bank._h, None, 0, 0, 0, C.byref(broken)) != 0
    assert b"rows" in lib.sdr_last_error()
    # a tracking flag on a channel that was never put into the bank: it has no epoch to run, but reports its row
    tracking[2] = True
    m = bank.tick_mirrored(None, 0, 100)
    assert (m.n_ran, m.n_updates) == (0, 1) and bank.updates["channel"][0] == 2
    # a slab queued without waiting is in the ring when the tick returns
    slab = np.arange(-100, 100, dtype=np.int8)
    engine.iq_upload_begin(slab, 16)
    bank.tick_mirrored(None, 0, 116)
    assert np.array_equal(engine.iq_download(100, 16), slab)
    # ... also when it does not fit the ingest kernel's 16-byte granules, or wraps the ring
    odd = np.arange(-30, 30, dtype=np.int8)
    engine.iq_upload_begin(odd, 7)
    engine.sync()
    assert np.array_equal(engine.iq_download(30, 7), odd)
    wrap = (np.arange(64) % 100).astype(np.int8)
    engine.iq_upload_begin(wrap, 80000 - 16)
    engine.sync()
    assert np.array_equal(engine.iq_download(32, 80000 - 16), wrap)
    bank.close()


def test_step_in_two_halves_equals_the_step(engine):
    """sdr_bank_step_begin + sdr_bank_step_end = sdr_bank_step: the same records, states and epoch counts bit for bit; a tick
    queued between the halves runs after the step (the stream orders them) and leaves its results alone; a second begin
    while one is in flight and an end without a begin are refused."""
    from sydr_amd._lib import SdrError
    g, fs, raw = trajectory_iq()
    n = raw.size // 2 // 8 * 8
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw[:2 * n], 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    acq = g["kaplan_acq"]
    st_k = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=1)
    st_b = initial_state(0, fs, acq[3], int(acq[5]), BORRE_CFG, slot=1)

    def fresh():
        bank = engine.bank(4)
        bank.put(0, as_row(st_k, TRACK_STATE_DTYPE), as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE))
        bank.put(2, as_row(st_b, TRACK_STATE_DTYPE), as_row(loop_cfg(0, fs, BORRE_CFG), LOOP_CFG_DTYPE))
        bank.put(3, as_row(st_k, TRACK_STATE_DTYPE), as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE))
        return bank
    a = fresh()
    rec_a, st_a, done_a, _ = a.step([0, 2], 30)
    tick_a = a.tick(None, 0, np.array([3], dtype=np.int32))
    a.close()
    b = fresh()
    with pytest.raises(RuntimeError, match="no step"):
        b.step_end()
    b.step_begin([0, 2], 30)
    with pytest.raises(SdrError, match="already in flight"):
        b.step_begin([3], 1)
    tick_b = b.tick(None, 0, np.array([3], dtype=np.int32))     # (another channel, queued behind the step, its own scratch)
    rec_b, st_b2, done_b = b.step_end()
    assert rec_a.tobytes() == rec_b.tobytes() and st_a.tobytes() == st_b2.tobytes() and list(done_a) == list(done_b) == [30, 30]
    assert all(x.tobytes() == y.tobytes() for x, y in zip(tick_a, tick_b))
    b.step_begin([0, 2], 5)                                      # (and again: the halves can be used over and over)
    rec_c, _, done_c = b.step_end()
    assert list(done_c) == [5, 5] and rec_c["start_sample"][0, 0] == rec_b["start_sample"][0, -1] + rec_b["n_samples"][0, -1]
    b.step_begin([0], 3)
    b.close()                                                    # (a bank destroyed with a step in flight waits for it)


def test_mirrored_tick_parks_a_runaway_channel_alone(engine):
    """A channel whose NCO has run away (carrier beyond any sampling rate) is stopped by the device before its epoch: the
    mirrored tick reports it in n_lost, sets its `lost` flag, hands out no record for it and never lists it again; the other
    channels' records are what they are without it; its CHANNEL_UPDATE row stays."""
    from sydr_amd._lib import TRACK_EPOCH_DTYPE
    g, fs, raw = trajectory_iq()
    n = raw.size // 2 // 8 * 8
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw[:2 * n], 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    acq = g["kaplan_acq"]
    st0 = initial_state(1, fs, acq[3], int(acq[5]), KAPLAN_CFG, slot=1)
    cfg = as_row(loop_cfg(1, fs, KAPLAN_CFG), LOOP_CFG_DTYPE)
    rows = [as_row(st0, TRACK_STATE_DTYPE) for _ in range(3)]
    rows[1]["carrier_hz"] = 2e9
    ref = engine.bank(3)
    ref.put(0, rows[0], cfg)
    want = [ref.step([0], 1)[0][0, 0].copy() for _ in range(6)]
    ref.close()
    bank = engine.bank(3)
    states = np.zeros(3, dtype=TRACK_STATE_DTYPE)
    for ch in range(3):
        bank.put(ch, rows[ch], cfg)
        states[ch] = rows[ch]
    last = np.zeros(3, dtype=TRACK_EPOCH_DTYPE)
    since, host_flags = np.zeros(3, dtype=np.int64), np.zeros(3, dtype=np.int64)
    tracking, lost = np.ones(3, dtype=bool), np.zeros(3, dtype=bool)
    bank.bind_mirror(states, last, since, tracking, lost, host_flags)
    for k in range(6):
        write = (int(states["current_sample"][[0, 2]].max()) + int(states["n_samples"].max()) + 8) % n
        m = bank.tick_mirrored(None, 0, write)
        assert bank.ran[:m.n_ran].tolist() == [0, 2] and m.n_updates == 3 and m.n_lost == (1 if k == 0 else 0)
        assert lost.tolist() == [False, True, False] and since.tolist() == [k + 1, 0, k + 1]
        assert bank.records[0].tobytes() == want[k].tobytes() == bank.records[1].tobytes()
    assert states["carrier_hz"][1] == 2e9 and last["n_samples"][1] == 0       # (its state as it was put; no record ever)
    bank.close()


@pytest.mark.parametrize("n_ch, forced_cluster", [(64, 1), (32, 0)])
def test_one_manager_over_two_devices_equals_the_single_device_manager(engine, n_ch, forced_cluster):
    """ChannelManager(rfSignal, engines=[a, b]) -- ONE manager over two devices in one process (two engines on this
    box's one card) -- against the single-device manager of the same channels on the same stream, 300 ticks at 10 MHz
    from acquisition on: every packet of every tick equal BIT FOR BIT and in the same order, channel attributes too.
    64 channels: a channel's partial sums are added in the order its cluster size gives, and a launch of 64 channels
    picks clusters of 4 where one of 32 picks 8, so the cluster size is pinned (1) on all three engines; 32 channels
    (16 + 16 against 32) take clusters of 8 either way and need no pinning.  Every device's tick is begun before any
    is ended."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.channel.multidevice import MultiDeviceChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    from sydr_amd.utils.enumerations import ChannelMessage, ChannelState
    fs, n_ms = 10e6, 300
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(8080 + n_ch)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=3.0) for c in range(n_ch)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(n_ch)
    engine.iq_synth(sats, fs, 10.0, 8081, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))
    half = n_ch // 2

    def receiver(engines):
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        for e in engines:
            e.track_cluster(forced_cluster)
        if len(engines) == 1:
            mgr = ChannelManager(rf, engine=engines[0], keepCorrelationMap=False)
            order = list(range(n_ch))
        else:
            mgr = ChannelManager(rf, engines=engines, keepCorrelationMap=False)
            assert isinstance(mgr, MultiDeviceChannelManager)
            # devices fill round-robin (request k lands on device k % 2, channel k // 2 + half * (k % 2)): asking for the
            # satellites in that order gives every satellite the channel number the single manager gives it
            order = [k // 2 + half * (k % 2) for k in range(n_ch)]
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, n_ch)
        for c in order:
            ch = mgr.requestTracking(sats[c]["prn"])
            assert ch.channelID == c
        ticks = []
        try:
            for k in range(n_ms):
                mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
                ticks.append([dict(q) for q in mgr.run()])
            attrs = [(c, ch.channelState, ch.carrierFrequency, ch.codeFrequency, ch.currentSample, ch.codeSinceTOW, int(ch.trackFlags),
                      len(ch.navBits)) for c, ch in sorted(mgr.channels.items())]
            if len(engines) > 1:
                assert [mgr.deviceOf(c) for c in range(n_ch)] == [0] * half + [1] * half
                assert [sum(ch.channelState is ChannelState.TRACKING for ch in p.channels.values()) for p in mgr.parts] == [half, half]
        finally:
            mgr.close()
            for e in engines:
                e.track_cluster(0)
        return ticks, attrs

    one, attrs_1 = receiver([engine])
    a, b = Engine(0), Engine(0)
    try:
        two, attrs_2 = receiver([a, b])
    finally:
        a.close()
        b.close()
    assert sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in two for p in t) > n_ch * 250
    for k, (x, y) in enumerate(zip(one, two)):
        assert x == y, k
    assert attrs_1 == attrs_2
    assert all(st is ChannelState.TRACKING for _, st, *_ in attrs_2)


def test_one_launch_tick_equals_the_two_launch_tick_and_the_separate_ingest(engine):
    """The plain receiver tick is ONE launch: the cluster's two halves in one plain launch (the part that draws its channel's
    last ticket collects the sums -- in part order, whoever it is) and the tick's slab pulled into the ring by workgroups of the
    same launch.  Against the tick as two launches behind an ingest launch of its own ("track_two_launch_tick",
    "ingest_with_tick" = 0): every packet of every tick equal BIT FOR BIT, from acquisition on, with ticks in which a channel
    is not ready (10 MHz epochs are 10 000 +- a sample long against slabs of exactly 10 000), a call in between that needs the
    ring (the staged slab goes in the ordinary way then) and two slabs before one tick."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    from sydr_amd.utils.enumerations import ChannelMessage
    fs, n_ms = 10e6, 260
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(8181)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 8182, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(two_launch, with_tick):
        engine.set_option("track_two_launch_tick", 1 if two_launch else 0)
        engine.set_option("ingest_with_tick", 1 if with_tick else 0)
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats:
            mgr.requestTracking(s["prn"])
        ticks, k = [], 0
        try:
            while k < n_ms:
                if k == 150:
                    engine.iq_download(64, 0)                # a call that reads the ring: a staged slab goes in first
                mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
                k += 1
                if k == 200:                                 # two slabs before one tick
                    mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
                    k += 1
                ticks.append([dict(q) for q in mgr.run()])
        finally:
            mgr.close()
            engine.set_option("track_two_launch_tick", 0)
            engine.set_option("ingest_with_tick", 1)
        return ticks

    reference = receiver(True, False)
    assert sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in reference for p in t) > 12 * 200
    for form in ((False, False), (False, True), (True, True)):
        got = receiver(*form)
        assert len(got) == len(reference)
        for k, (a, b) in enumerate(zip(reference, got)):
            assert a == b, (form, k)


def test_slabs_in_page_locked_memory_are_read_in_place_with_the_same_bits(engine):
    """A slab that lies in memory from sdr_host_alloc is read in place by whoever pulls it into the ring (the tick's ingest
    workgroups, the tick server's doormen, an ingest kernel) instead of being staged first: the same packets, bit for bit, as
    from pageable memory -- plain ticks and served ones, incl. a slab at an address that is NOT on a 16-byte boundary (staged
    like any other) and a ring-reading call in between."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    fs, n_ms = 10e6, 220
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(9191)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 9192, 0, total)
    raw = engine.iq_download(total, 0)
    block = engine.host_alloc(raw.size + 16, raw.dtype)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(source, server):
        engine.set_option("tick_server", 1 if server else 0)
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats:
            mgr.requestTracking(s["prn"])
        ticks = []
        try:
            for k in range(n_ms):
                if k == 120:
                    engine.iq_download(64, 0)
                mgr.addNewRFData(source[2 * k * spms:2 * (k + 1) * spms])
                ticks.append([dict(q) for q in mgr.run()])
        finally:
            mgr.close()
            engine.set_option("tick_server", 0)
        return ticks

    try:
        reference = receiver(raw, False)
        block[:raw.size] = raw
        aligned = block[:raw.size]
        assert aligned.ctypes.data % 16 == 0
        for server in (False, True):
            got = receiver(aligned, server)
            for k, (a, b) in enumerate(zip(reference, got)):
                assert a == b, (server, k)
        block[2:2 + raw.size] = raw                       # the same samples two bytes further on: no 16-byte boundary, staged
        got = receiver(block[2:2 + raw.size], False)
        for k, (a, b) in enumerate(zip(reference, got)):
            assert a == b, k
    finally:
        engine.host_free(block)


def test_packets_read_long_after_their_tick_are_the_ticks_own(engine):
    """The tick's rows are views of arrays the device writes into alternately (two sets: `Bank.bind_mirror`,
    `ChannelBank.hold`): a packet list read many ticks later -- when both sets have been rewritten many times -- still holds
    its own tick's values, as does one read a tick later or at once."""
    import configparser
    import os
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    fs, n_ms = 10e6, 160
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(4242)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 4243, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(read):
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=engine, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats:
            mgr.requestTracking(s["prn"])
        kept, out = [], []
        try:
            for k in range(n_ms):
                mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
                pk = mgr.run()
                if read == "at once":
                    out.append([dict(q) for q in pk])
                elif read == "a tick later":
                    if kept:
                        out.append([dict(q) for q in kept.pop()])
                    kept.append(pk)
                else:
                    kept.append(pk)
            out.extend([dict(q) for q in t] for t in kept)
        finally:
            mgr.close()
        return out

    at_once = receiver("at once")
    assert len(at_once) == n_ms and sum(len(t) for t in at_once) > 12 * 100
    for mode in ("a tick later", "at the end"):
        assert receiver(mode) == at_once, mode


def test_bind_thread_to_device_restricts_the_calling_thread_and_gives_the_mask_back(engine):
    """sdr_set_option "bind_thread_to_device": the calling thread onto the CPUs sysfs lists for the GPU's PCI function -- a
    non-empty subset of the mask it had -- and back (a served tick is round trips through page-locked words: from the other
    socket of a two-socket host each takes the sockets' interconnect as well)."""
    import os
    from sydr_amd._lib import SdrError
    before = os.sched_getaffinity(0)
    try:
        engine.set_option("bind_thread_to_device", 1)
    except SdrError as exc:           # (no sysfs entry for the device, or a cpuset without its CPUs)
        pytest.skip(f"no binding here: {exc}")
    try:
        during = os.sched_getaffinity(0)
        assert during and during <= before
        engine.set_option("bind_thread_to_device", 1)          # (again: the mask to go back to is still the first one)
        assert os.sched_getaffinity(0) == during
    finally:
        engine.set_option("bind_thread_to_device", 0)
    assert os.sched_getaffinity(0) == before
    with pytest.raises(SdrError):
        engine.set_option("no_such_option", 1)


def test_resident_tick_server_equals_plain_ticks(engine):
    """sdr_set_option("tick_server", 1): the steady receiver tick is answered by a RESIDENT kernel (the cluster form of the
    tracking kernel + a doorman workgroup that polls a request word in page-locked memory, pulls the slab, releases the
    trackers and gathers their answers) instead of two launches and a synchronisation.  12 channels at 10 MHz from
    acquisition on, against the same receiver on plain ticks: every packet of every tick equal BIT FOR BIT (same cluster,
    same order of additions), channel attributes too; the server is started when the channels have all reached tracking,
    survives ticks in which a channel is not ready, is stopped by any other call on the engine (here: an acquisition-sized
    download) and started again by the next ticks, leaves by itself when the host pauses for 0.3 s (the ticks after the pause
    are plain ones, then a server again); an engine closed with a server resident returns at once."""
    import configparser
    import os
    import time
    from conftest import REPO
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.signal.iqsource import RFSignal
    from sydr_amd.utils.enumerations import ChannelMessage
    fs, n_ms = 10e6, 400
    spms = int(fs * 1e-3)
    rng = np.random.default_rng(7070)
    sats = [dict(prn=1 + c, doppler=float(250.0 * rng.integers(-15, 16) + rng.uniform(-40, 40)),
                 code_phase=float(rng.uniform(0, 1023)), phase=float(rng.uniform(0, 1)), amp=5.0) for c in range(12)]
    total = n_ms * spms
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    engine.iq_synth(sats, fs, 10.0, 7071, 0, total)
    raw = engine.iq_download(total, 0)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REPO, "examples", "channel_GPS_L1CA_kaplan.ini"))

    def receiver(eng, server):
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        eng.set_option("tick_server", 1 if server else 0)
        mgr = ChannelManager(rf, engine=eng, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 12)
        for s in sats:
            mgr.requestTracking(s["prn"])
        ticks = []
        try:
            for k in range(n_ms):
                if k == 250:
                    eng.iq_download(64, 0)                   # any other call on the engine: the server leaves, the next tick starts one
                if k == 320 and server:
                    time.sleep(0.3)                          # a host that pauses: the server leaves by its own clock (0.2 s), the
                                                             # next ticks are plain ones, the ninth starts a server again
                mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
                ticks.append([dict(q) for q in mgr.run()])
            attrs = [(ch.carrierFrequency, ch.codeFrequency, ch.currentSample, ch.codeSinceTOW, int(ch.trackFlags), len(ch.navBits))
                     for ch in mgr.channels.values()]
            stats = eng.tick_server_stats()
        finally:
            mgr.close()
            eng.set_option("tick_server", 0)
        return ticks, attrs, stats

    plain, attrs_p, stats_p = receiver(engine, False)
    before = engine.tick_server_stats()
    served, attrs_s, stats_s = receiver(engine, True)
    assert stats_p["served"] == before["served"] and not stats_p["running"]
    # the server answered the steady ticks (all but acquisition and the ticks around it), was resident at the end, was
    # started three times (the download at tick 250 sent the first one away, the pause at tick 320 the second) and never gave up
    assert stats_s["served"] - before["served"] > 315 and stats_s["running"] and stats_s["starts"] - before["starts"] == 3 and not stats_s["disabled"]
    assert sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in served for p in t) > 12 * 330
    for k, (a, b) in enumerate(zip(plain, served)):
        assert a == b, k
    assert attrs_p == attrs_s
    # ticks with nothing ready: the server answers "nobody ran"; a channel whose epoch is not complete waits a tick
    # (both happen above: 10 MHz epochs are 10 000 +- a sample long against slabs of exactly 10 000)
    # an engine that goes away with a server resident: the server is told to leave, the destroy returns at once
    e2 = Engine(0)
    try:
        e2.set_option("tick_server", 1)
        rf = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        mgr = ChannelManager(rf, engine=e2, keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 4)
        for s in sats[:4]:
            mgr.requestTracking(s["prn"])
        for k in range(60):
            mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
            mgr.run()
        t0 = time.perf_counter()
    finally:
        e2.close()
    assert time.perf_counter() - t0 < 1.0
