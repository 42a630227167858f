"""Host layer on CPU: the ChannelManager / plugin state machines driven through an oracle-backed
engine must reproduce the reference plugins' golden trajectories BIT FOR BIT (BASELINE config 1:
1 channel, 4 MHz, 1 ms PCPS + ~500 ms tracking), keep the reference's packet and config contracts,
and shard channels correctly across ranks (2-process gloo test)."""
import configparser
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import REPO, load_golden
from oracle import sydr_oracle as orc
from fake_engine import OracleEngine
from test_oracle_golden import trajectory_iq

from sydr_amd.channel.l1ca_borre import ChannelL1CA
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
from sydr_amd.channel.manager import ChannelManager, shard_channels
from sydr_amd.signal.iqsource import RFSignal
from sydr_amd.utils.enumerations import ChannelMessage, ChannelState, LoopLockState, TrackingFlags

_EXAMPLES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")
KAPLAN_INI = open(os.path.join(_EXAMPLES, "channel_GPS_L1CA_kaplan.ini")).read()
BORRE_INI = open(os.path.join(_EXAMPLES, "channel_GPS_L1CA_borre.ini")).read()


def channel_config(text):
    cfg = configparser.ConfigParser()
    cfg.read_string(text)
    return cfg


def rf_signal(fs=4e6):
    return RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0,
                         data_size=8))


def drive(manager, raw, spms, ms):
    packets = []
    for k in range(ms):
        manager.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])   # raw interleaved int8, 1 ms per tick
        packets.append(manager.run())
    return packets


@pytest.mark.parametrize("plugin", ["borre", "kaplan"])
def test_manager_reproduces_reference_trajectory(plugin):
    g, fs, raw = trajectory_iq()
    spms = int(fs * 1e-3)
    eng = OracleEngine()
    mgr = ChannelManager(rf_signal(fs), engine=eng)
    cls, ini = (ChannelL1CA, BORRE_INI) if plugin == "borre" else (ChannelL1CA_Kaplan, KAPLAN_INI)
    mgr.addChannel(cls, channel_config(ini), 1)
    ch = mgr.requestTracking(7)
    assert ch.channelState is ChannelState.ACQUIRING and ch.satelliteID == 7
    ticks = drive(mgr, raw, spms, 510)

    acq = [p for t in ticks for p in t if p["type"] is ChannelMessage.ACQUISITION_UPDATE]
    trk = [p for t in ticks for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    upd = [p for t in ticks for p in t if p["type"] is ChannelMessage.CHANNEL_UPDATE]
    ref_acq, ref = g[f"{plugin}_acq"], g[f"{plugin}_epochs"]
    assert len(acq) == 1 and len(upd) == 510 and len(trk) == len(ref)
    a = acq[0]
    assert (a["frequency_idx"], a["code_idx"], a["carrierFrequency"], a["codeOffset"]) == \
        (int(ref_acq[0]), int(ref_acq[1]), ref_acq[3], int(ref_acq[4]))
    assert a["peak_ratio"] == ref_acq[2] and a["correlation_map"].shape == (41, 4000)
    for k, (p, row) in enumerate(zip(trk, ref)):
        got = [p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"], p["i_late"], p["q_late"]]
        assert got == list(row[6:12]), k
        assert (p["carrier_frequency"], p["code_frequency"]) == (row[15], row[16]), k
        assert (p["dll"], p["pll"]) == (row[12], row[13]), k
        assert (p["carrier_frequency_error"], p["code_frequency_error"]) == (row[17], row[18]), k
        if plugin == "kaplan":
            assert p["fll"] == row[14] and int(p["lock_state"]) == int(row[22]), k
            np.testing.assert_equal([p["cn0"], p["pll_lock"], p["fll_lock"]], row[19:22])
    assert int(ch.trackFlags) == int(ref[-1][23])
    assert ch.navBits == [int(b) for b in ref[ref[:, 24] >= 0, 24]]      # decodeBit on the host plugin
    if plugin == "kaplan":
        assert ch.loopLockState is LoopLockState(int(ref[-1][22]))  # (transitions: see the g6b test below)
    # packet contract (keys become DB columns in the reference: database.py:76-93)
    assert set(trk[0]) == {"cid", "type", "i_early", "q_early", "i_prompt", "q_prompt", "i_late", "q_late",
                           "carrier_frequency", "code_frequency", "carrier_frequency_error", "code_frequency_error",
                           "cn0", "pll_lock", "fll_lock", "dll", "pll", "fll", "lock_state"}
    assert set(upd[0]) == {"cid", "type", "state", "tracking_flags", "tow", "time_since_tow", "unprocessed_samples",
                           "code_since_tow"}
    assert set(a) == {"cid", "type", "carrierFrequency", "codeOffset", "frequency_idx", "code_idx", "correlation_map",
                      "peak_ratio"}


def test_manager_kaplan_lock_state_machine():
    """Golden run that walks PULL_IN -> WIDE -> NARROW with narrow taps, code lock and bit sync."""
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    cfg = channel_config(KAPLAN_INI)
    for k, v in zip(g["track_override_keys"], g["track_override_vals"]):
        cfg["TRACKING"][str(k)] = repr(float(v))
    mgr = ChannelManager(rf_signal(fs), engine=OracleEngine())
    mgr.addChannel(ChannelL1CA_Kaplan, cfg, 1)
    ch = mgr.requestTracking(7)
    ticks = drive(mgr, raw, int(fs * 1e-3), 1200)
    trk = [p for t in ticks for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref = g["kaplan_epochs"]
    assert len(trk) == len(ref)
    for k, (p, row) in enumerate(zip(trk, ref)):
        got = [p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"], p["i_late"], p["q_late"]]
        assert got == list(row[6:12]), k
        assert (p["carrier_frequency"], p["code_frequency"], int(p["lock_state"])) == (row[15], row[16], int(row[22])), k
    assert ch.loopLockState is LoopLockState.NARROW_TRACK and ch.track_correlatorsSpacing == [-0.25, 0.0, 0.25]
    assert int(ch.trackFlags) == int(TrackingFlags.CODE_LOCK | TrackingFlags.BIT_SYNC)
    assert ch.navBits == [int(b) for b in ref[ref[:, 24] >= 0, 24]] and len(ch.navBits) >= 40


def test_manager_batches_channels_into_single_launches():
    """Three channels: one PCPS call for all of them, then ONE device call per tick that has an epoch to run (an epoch
    of every ready channel, behind the slab addNewRFData queued)."""
    fs, spms = 4e6, 4000
    from oracle import sydr_oracle as orc
    sats = [dict(prn=p, doppler=d, code_phase=c, phase=0.1, amp=8.0) for p, d, c in
            ((7, 1750.0, 300.25), (12, -3000.0, 17.5), (30, 4250.0, 900.0))]
    raw = orc.synth_iq(fs, 30 * spms, sats, 20.0, 99)
    eng = OracleEngine()
    mgr = ChannelManager(rf_signal(fs), engine=eng)
    mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), 4)
    for prn in (7, 12, 30):
        mgr.requestTracking(prn)
    assert mgr.getChannel(3).channelState is ChannelState.IDLE      # unused channel stays idle and silent
    ticks = drive(mgr, raw, spms, 30)
    assert eng.calls["pcps"] == 1
    n_trk = sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in ticks for p in t)
    dev = mgr.bank.device
    ticks_with_epochs = sum(any(p["type"] is ChannelMessage.TRACKING_UPDATE for p in t) for t in ticks)
    assert 25 <= ticks_with_epochs <= 30
    assert dev.calls == dict(step=dev.calls["step"], tick=ticks_with_epochs, channels=n_trk) and eng.calls["epl_batch"] == 0
    assert all(mgr.getChannel(c).channelState is ChannelState.TRACKING for c in range(3))
    for c, s in enumerate(sats):
        assert abs(mgr.getChannel(c).carrierFrequency - s["doppler"]) < 300.0
    with pytest.raises(ValueError):
        mgr.getChannel(9)
    with pytest.raises(Warning):
        for prn in (1, 2):
            mgr.requestTracking(prn)


def test_ring_bookkeeping_matches_reference_semantics():
    from sydr_amd.utils.devicering import CircularBuffer
    eng = OracleEngine()
    ring = CircularBuffer(4000, np.int8, engine=eng)
    with pytest.raises(ValueError):
        ring.shift(np.zeros(2 * 300, dtype=np.int8))     # 4000 % 300 != 0 (circularbuffer.py:74-75)
    rng = np.random.default_rng(1)
    blocks = [rng.integers(-100, 100, 2000).astype(np.int8) for _ in range(6)]
    for k, b in enumerate(blocks):
        ring.shift(b)
        assert ring.idxWrite == ((k + 1) * 1000) % 4000 and ring.full == (k >= 3)
    assert ring.getNbUnreadSamples(1500) == 500 and ring.getNbUnreadSamples(3000) == 3000
    wrapped = ring.getSlice(3500, 1000)                    # wraps: block 3 tail + block 4 head
    expect = np.r_[blocks[3][1000:], blocks[4][:1000]].astype(float)
    assert np.array_equal(wrapped[0], expect[0::2] + 1j * expect[1::2])
    with pytest.raises(ValueError):
        ring.shift(np.array([0.5 + 1j] * 1000))            # non-integer samples into an int8 ring


def test_pending_slab_is_owned_and_survives_a_failed_tick():
    """addNewRFData copies the slab out of the caller's buffer at once, as the reference does (circularbuffer.py:54-82),
    and only QUEUES its transfer into the ring (Engine.iq_upload_begin): what reaches the ring is what the caller held
    THEN, the tick's device call is ordered behind it, and a tick that fails cannot lose a slab the write index
    already counts."""
    fs, spms = 4e6, 4000
    eng = OracleEngine()
    begun = []
    eng.iq_upload_begin = lambda raw, off: (begun.append(int(off)), OracleEngine.iq_upload(eng, raw, off))[1]
    mgr = ChannelManager(rf_signal(fs), engine=eng)
    mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), 1)
    mgr.requestTracking(7)
    rng = np.random.default_rng(3)
    slab = rng.integers(-100, 100, 2 * spms).astype(np.int8)
    want = slab.copy()
    mgr.addNewRFData(slab)
    assert begun == [0] and mgr._pending is True
    slab[:] = 0                                     # the caller re-uses its buffer before run()
    mgr.run()
    assert mgr._pending is False and np.array_equal(eng.iq_download(spms, 0), want)
    # a device call that raises: the slab is in the ring (queued before the call), the exception surfaces
    second = rng.integers(-100, 100, 2 * spms).astype(np.int8)
    mgr.addNewRFData(second)
    boom = RuntimeError("tick failed")

    def failing(*a, **k):
        raise boom
    bank = mgr.bank
    real = (mgr._acquire, bank.tick, bank.tick_ready, bank.tick_ready_begin)
    mgr._acquire = bank.tick = bank.tick_ready = bank.tick_ready_begin = failing       # (whichever this tick needs)
    with pytest.raises(RuntimeError):
        mgr.run()
    mgr._acquire, bank.tick, bank.tick_ready, bank.tick_ready_begin = real
    assert np.array_equal(eng.iq_download(spms, spms), second)
    # a second slab without a run() in between waits for the first one's transfer (the staging buffer is re-used)
    synced = []
    eng.sync = lambda: synced.append(1)
    mgr._pending = True
    mgr.addNewRFData(rng.integers(-100, 100, 2 * spms).astype(np.int8))
    assert synced == [1]
    # slabs beyond DEFER_BYTES go to the ring synchronously
    mgr.run()
    long = rng.integers(-100, 100, 2 * spms * 20).astype(np.int8)
    mgr.DEFER_BYTES = 1 << 10
    mgr.addNewRFData(long)
    assert mgr._pending is False and np.array_equal(eng.iq_download(20 * spms, 3 * spms), long)


@pytest.mark.parametrize("block_ms", [50, 7])
def test_read_ahead_ticks_equal_plain_ticks(tmp_path, block_ms):
    """enableReadAhead: blocks of epochs computed ahead, handed out tick by tick -- every packet of every tick equal
    to the plain per-tick manager's (g6b: PULL_IN -> WIDE -> NARROW, code lock, bit sync, navigation bits through a
    stub decoder), the reference's calls unchanged, and a foreign slab refused while a block is replayed."""
    from test_decoding import RecordingDecoder
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    path = tmp_path / "iq.bin"
    raw.tofile(path)
    cfg = channel_config(KAPLAN_INI)
    for k, v in zip(g["track_override_keys"], g["track_override_vals"]):
        cfg["TRACKING"][str(k)] = repr(float(v))

    def receiver(read_ahead):
        sig = RFSignal(dict(filepath=str(path), sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0,
                            data_size=8))
        eng = OracleEngine()
        mgr = ChannelManager(sig, engine=eng)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 1)
        ch = mgr.requestTracking(7)
        ch.setDecoding(RecordingDecoder(every=10))
        if read_ahead:
            mgr.enableReadAhead(read_ahead)
        ticks = []
        for _ in range(1200):
            mgr.addNewRFData(sig.getMilliseconds(1))          # the reference's loop, receiver.py:120-131
            ticks.append([dict(p) for p in mgr.run()])
        return mgr, ch, eng, ticks

    plain_mgr, plain_ch, plain_eng, plain = receiver(0)
    ra_mgr, ra_ch, ra_eng, ahead = receiver(block_ms)
    for k, (a, b) in enumerate(zip(plain, ahead)):
        for p in a + b:
            p.pop("correlation_map", None)
        assert a == b, k
    assert ra_eng.bank_calls["step"] >= 1100 // block_ms and ra_eng.bank_calls["tick"] < 200       # blocks did the tracking
    assert plain_eng.bank_calls["tick"] >= 1190
    assert ra_ch.navBits == plain_ch.navBits and len(ra_ch.navBits) >= 40
    assert ra_ch.carrierFrequency == plain_ch.carrierFrequency and ra_ch.currentSample == plain_ch.currentSample
    assert ra_ch.tow == plain_ch.tow and int(ra_ch.trackFlags) == int(plain_ch.trackFlags)
    # a slab that is not the recording's next millisecond cannot be replayed
    sig = ra_mgr.rfSignal
    sig.seek(0)
    mgr2 = ChannelManager(sig, engine=OracleEngine())
    mgr2.addChannel(ChannelL1CA_Kaplan, cfg, 1)
    mgr2.requestTracking(7)
    mgr2.enableReadAhead(20)
    for _ in range(30):
        mgr2.addNewRFData(sig.getMilliseconds(1))
        mgr2.run()
    assert mgr2._readahead is not None and mgr2._readahead.slabs_left > 0
    with pytest.raises(ValueError, match="next millisecond"):
        mgr2.addNewRFData(np.zeros(2 * int(fs * 1e-3), dtype=np.int8))


def test_read_ahead_with_a_late_joiner_equals_plain_ticks(tmp_path):
    """A second satellite is requested while blocks are being replayed: it acquires from the ring as usual, tracks by
    plain device ticks until the next block, then lags the first channel by its acquisition time -- its epochs spill
    past the end of a block, the first channel runs on by plain ticks meanwhile.  Every packet of every tick equal to
    the plain loop's."""
    fs, ms = 4e6, 260
    spms = int(fs * 1e-3)
    sats = [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=30.0),
            dict(prn=19, doppler=-2250.0, code_phase=811.5, phase=0.6, amp=25.0)]
    raw = orc.synth_iq(fs, ms * spms, sats, 10.0, 20261040)
    path = tmp_path / "iq.bin"
    raw.tofile(path)
    cfg = channel_config(KAPLAN_INI)
    cfg["ACQUISITION"]["non_coherent_integration"] = "3"

    def receiver(read_ahead):
        sig = RFSignal(dict(filepath=str(path), sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0,
                            data_size=8))
        mgr = ChannelManager(sig, engine=OracleEngine(), keepCorrelationMap=False)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 2)
        mgr.requestTracking(7)
        if read_ahead:
            mgr.enableReadAhead(read_ahead)
        ticks = []
        for k in range(ms):
            if k == 70:
                mgr.requestTracking(19)
            mgr.addNewRFData(sig.getMilliseconds(1))
            ticks.append([dict(p) for p in mgr.run()])
        return mgr, ticks

    _, plain = receiver(0)
    mgr, ahead = receiver(16)
    for k, (a, b) in enumerate(zip(plain, ahead)):
        key = lambda p: (p["cid"], p["type"].value)
        assert sorted(a, key=key) == sorted(b, key=key), k
    assert sum(1 for t in ahead for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE and p["cid"] == 1) > 150
    assert mgr.engine.bank_calls["step"] > 10


def test_rfsignal_serves_the_recording_as_raw_integer_slabs(tmp_path):
    rng = np.random.default_rng(2)
    raw = rng.integers(-128, 127, 2 * 4000 * 130).astype(np.int8)
    path = tmp_path / "iq.bin"
    raw.tofile(path)
    sig = RFSignal(dict(filepath=str(path), sampling_frequency=4e6, is_complex="true", intermediate_frequency=0.0,
                        data_size=8))
    assert sig.samplesPerMs == 4000 and sig.dtype == np.complex128 and sig.fileDataType is np.int8
    assert sig.totalSamples == 4000 * 130
    first = sig.getMilliseconds(1)
    assert first.dtype == np.int8 and np.array_equal(first, raw[:8000])
    assert np.shares_memory(first, sig._recording())                     # a view of the mapped file, not a copy
    cplx = sig.getMilliseconds(1, raw=False)
    assert np.array_equal(cplx, raw[8000:16000:2] + 1j * raw[8001:16000:2])
    assert np.array_equal(sig.getMilliseconds(7), raw[2 * 8000:2 * 8000 + 7 * 8000])   # any slab length
    assert sig.position == 9 * 4000
    sig.seek(120 * 4000)
    assert np.array_equal(sig.getMilliseconds(10), raw[2 * 4000 * 120:])
    with pytest.raises(EOFError):
        sig.getMilliseconds(1)
    with pytest.raises(ValueError):
        RFSignal(dict(filepath="x", sampling_frequency=4e6, is_complex="true", intermediate_frequency=0, data_size=12))
    # rfsignal.py:35 reads the key with bool(<string>): "false" is True there, and here; only an empty value is real-valued
    assert RFSignal(dict(filepath="x", sampling_frequency=4e6, is_complex="false", intermediate_frequency=0,
                         data_size=8)).isComplex
    with pytest.raises(ValueError):
        RFSignal(dict(filepath="x", sampling_frequency=4e6, is_complex="", intermediate_frequency=0, data_size=8))
    # the reference's file readers (rfsignal.py:92-204): cursor kept while "open", complex128 out
    rd = RFSignal(dict(filepath=str(path), sampling_frequency=4e6, is_complex="true", intermediate_frequency=0.0,
                       data_size=8))
    with pytest.raises(Warning):
        rd.getCurrentSampleIndex()
    a = rd.readFile(timeLength=2, keep_open=True)
    assert a.dtype == np.complex128 and np.array_equal(a, raw[0:16000:2] + 1j * raw[1:16000:2])
    assert rd.getCurrentSampleIndex() == 8000
    b = rd.readFileBySamples(100, skip=50, keep_open=True)             # skip counts from the kept cursor
    assert np.array_equal(b, raw[2 * 8050:2 * 8150:2] + 1j * raw[2 * 8050 + 1:2 * 8150:2])
    assert rd.getCurrentSampleIndex() == 8150
    rd.closeFile()
    with pytest.raises(Warning):
        rd.closeFile()
    c = rd.readFileBySamples(10, skip=3, raw=True)                     # closed: from the start of the file
    assert np.array_equal(c, raw[6:26]) and np.shares_memory(c, rd._recording())
    assert rd.readFile(timeLength=1, skip=4000 * 130 - 10).size == 10  # a short read at the end of the file
    raw16 = rng.integers(-3000, 3000, 2 * 1000).astype(np.int16)
    raw16.tofile(tmp_path / "iq16.bin")
    sig16 = RFSignal(dict(filepath=str(tmp_path / "iq16.bin"), sampling_frequency=1e6, is_complex="True",
                          intermediate_frequency=0.0, data_size=16))
    assert np.array_equal(sig16.getMilliseconds(1), raw16)


def test_shard_channels_partitions_exactly():
    for n in (0, 1, 5, 32, 33, 256):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_channels(n, r, world) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_channels(4, 2, 2)


def test_two_rank_gloo_sharding_and_gather(tmp_path):
    """world_size-2 CPU run of the multi-GPU partition (north_star: same stream on every rank, channels sharded, no
    data-path collective): each rank drives a ChannelManager over ITS shard of the channels on an oracle-backed
    engine, the tracking packets are gathered, and rank 0 checks them -- bitwise -- against one manager that ran
    all channels.  Timing uses max-over-ranks, as bench.py does."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {REPO!r}); sys.path.insert(0, {os.path.join(REPO, 'tests')!r})
        import numpy as np, torch, torch.distributed as dist
        from oracle import sydr_oracle as orc
        from fake_engine import OracleEngine
        from test_host_layer import KAPLAN_INI, channel_config, drive, rf_signal
        from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
        from sydr_amd.channel.manager import ChannelManager, shard_channels
        from sydr_amd.utils.enumerations import ChannelMessage
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        fs, spms, ms = 4e6, 4000, 24
        sats = [dict(prn=p, doppler=d, code_phase=c, phase=0.1, amp=7.0) for p, d, c in
                ((7, 1750.0, 300.25), (12, -3000.0, 17.5), (30, 4250.0, 900.0), (3, -500.0, 512.0))]
        raw = orc.synth_iq(fs, ms * spms, sats, 18.0, 4242)          # the SAME stream on every rank (same seed)

        def run(prns):
            mgr = ChannelManager(rf_signal(fs), engine=OracleEngine())
            mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), len(prns))
            for p in prns:
                mgr.requestTracking(p)
            out = {{}}
            for tick in drive(mgr, raw, spms, ms):
                for p in tick:
                    if p["type"] is ChannelMessage.TRACKING_UPDATE:
                        prn = mgr.getChannel(p["cid"]).satelliteID
                        out.setdefault(prn, []).append([p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"],
                                                        p["i_late"], p["q_late"], p["carrier_frequency"], p["code_frequency"]])
            return out

        prns = [s["prn"] for s in sats]
        mine = [prns[i] for i in shard_channels(len(prns), rank, world)]
        local = run(mine)
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            merged = {{}}
            for g in gathered: merged.update(g)
            whole = run(prns)
            same = all(np.array(merged[p]).tobytes() == np.array(whole[p]).tobytes() for p in prns)
            print(json.dumps(dict(n=len(merged), same=bool(same), sizes=[len(g) for g in gathered],
                                  epochs=[len(whole[p]) for p in prns], tmax=float(t))))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json
    res = json.loads(line)
    assert res["n"] == 4 and res["same"] is True and res["sizes"] == [2, 2] and res["tmax"] == pytest.approx(0.2)
    assert min(res["epochs"]) >= 20


def test_serial_search_plugin_reproduces_reference():
    """ChannelL1CA_Kaplan_SS (three overridden seams, channel_l1ca_kaplan_ss.py:10-54) end to end."""
    import hashlib
    from oracle import sydr_oracle as orc
    from sydr_amd.channel.l1ca_kaplan_ss import ChannelL1CA_Kaplan_SS
    g = load_golden("g9_serial.npz")
    fs, spms = 4e6, 4000
    raw = orc.synth_iq(fs, 40 * spms, [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=8.0)], 20.0, 20260001)
    assert np.array_equal(np.frombuffer(hashlib.sha256(raw.tobytes()).digest(), dtype=np.uint8), g["ss_iq_sha256"])
    cfg = channel_config(KAPLAN_INI)
    cfg["ACQUISITION"].update(doppler_range="2000", doppler_steps="250", non_coherent_integration="2")
    mgr = ChannelManager(rf_signal(fs), engine=OracleEngine())
    mgr.addChannel(ChannelL1CA_Kaplan_SS, cfg, 1)
    ch = mgr.requestTracking(7)
    ticks = drive(mgr, raw, spms, 40)
    acq = [p for t in ticks for p in t if p["type"] is ChannelMessage.ACQUISITION_UPDATE][0]
    trk = [p for t in ticks for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref = g["ss_acq"]
    assert (acq["frequency_idx"], acq["code_idx"], acq["peak_ratio"], acq["carrierFrequency"], acq["codeOffset"]) == \
        (int(ref[0]), int(ref[1]), ref[2], ref[3], int(ref[4]))
    assert acq["correlation_map"].shape == (17, 1023) and len(trk) == len(g["ss_epochs"])
    for p, row in zip(trk, g["ss_epochs"]):
        assert [p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"], p["i_late"], p["q_late"],
                p["carrier_frequency"], p["code_frequency"]] == list(row)


def test_channels_survive_the_bank_growing_past_32():
    """The device-resident bank starts with room for 32 channels and is re-created when a 33rd arrives: channels made
    before that must follow the ring to the new bank (a reference held from construction would leave them on the old,
    closed one).  Channel 0 acquires and tracks PRN 7 with 39 more channels added in between; the packets are the
    golden ones."""
    g, fs, raw = trajectory_iq()
    spms = int(fs * 1e-3)
    eng = OracleEngine()
    mgr = ChannelManager(rf_signal(fs), engine=eng)
    mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), 2)
    ch = mgr.requestTracking(7)
    first = drive(mgr, raw[:2 * spms * 40], spms, 40)                      # acquisition + the first tracking epochs
    old_bank = mgr.bank
    mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), 38)      # 40 channels: the bank is re-created
    assert mgr.bank is not old_bank and mgr.bank.max_channels >= 40
    assert ch._bank is mgr.bank and mgr.getChannel(39)._bank is mgr.bank
    rest = drive(mgr, raw[2 * spms * 40:], spms, 510 - 40)
    trk = [p for t in first + rest for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref = g["kaplan_epochs"]
    assert len(trk) == len(ref)
    for k, (p, row) in enumerate(zip(trk, ref)):
        assert [p["i_prompt"], p["q_prompt"], p["carrier_frequency"], p["code_frequency"]] == [row[8], row[9], row[15], row[16]], k
    assert ch.channelState is ChannelState.TRACKING and int(ch.trackFlags) == int(ref[-1][23])


def test_packets_fill_themselves_and_the_steady_tick_equals_the_general_one():
    """(1) A result packet is a dict born with "cid" and "type" (what receiver.py:291-299 routes on); anything else asked
    of it fills the rest; keys the consumer wrote stay; `packet == None` (receiver.py:292) costs nothing; it pickles,
    copies and compares as the plain dict it stands for.  (2) The steady tick (readiness, epoch and mirror updates in one
    device-side call: Bank.tick_mirrored) gives the packets and channel attributes of the general tick, tick for tick,
    from acquisition through the Kaplan lock states."""
    import copy
    import json
    import pickle
    from sydr_amd.channel.bank import LazyPacket, TickPackets, TrackingRows, UpdateRows

    class Source:
        calls = 0

        def full(self, cid):
            Source.calls += 1
            return {"cid": cid, "type": ChannelMessage.TRACKING_UPDATE, "i_prompt": 1.5, "dll": -2.0}

    def fresh():
        p = LazyPacket({"cid": 3, "type": ChannelMessage.TRACKING_UPDATE})
        p._src = Source()
        return p
    p = fresh()
    assert isinstance(p, dict) and p["cid"] == 3 and p["type"] is ChannelMessage.TRACKING_UPDATE and not (p == None)  # noqa: E711
    assert (p != None) and Source.calls == 0 and dict.__len__(p) == 2                                                # noqa: E711
    p["channel_id"] = 9                                  # receiver.py:360-364 adds its columns before storing
    assert Source.calls == 0
    assert p["i_prompt"] == 1.5 and Source.calls == 1 and p["channel_id"] == 9
    assert list(p) == ["cid", "type", "i_prompt", "dll", "channel_id"] and len(p) == 5 and Source.calls == 1
    with pytest.raises(KeyError):
        p["nope"]
    full = {"cid": 3, "type": ChannelMessage.TRACKING_UPDATE, "i_prompt": 1.5, "dll": -2.0}
    for use in (len, list, dict, repr, copy.copy, copy.deepcopy, pickle.dumps, lambda q: q.items(), lambda q: q.get("dll"),
                lambda q: "dll" in q, lambda q: q == full, lambda q: full == q, lambda q: {**q}, lambda q: q | {}, lambda q: q.pop("dll"),
                lambda q: q.setdefault("dll", 0.0), lambda q: q.keys(), lambda q: q.values(), lambda q: sorted(q)):
        q, before = fresh(), Source.calls
        use(q)
        assert Source.calls == before + 1 and dict.__len__(q) >= 3, use
    assert fresh() == full and full == fresh() and fresh() == fresh() and not (fresh() != full)
    assert type(pickle.loads(pickle.dumps(fresh()))) is dict and pickle.loads(pickle.dumps(fresh())) == full
    assert type(fresh().copy()) is dict and {**fresh()} == full and dict(fresh()) == full
    assert json.loads(json.dumps({k: v for k, v in fresh().items() if k != "type"})) == {"cid": 3, "i_prompt": 1.5, "dll": -2.0}
    q = fresh()
    del q["cid"]                                          # (an odd thing to do; the rest still arrives)
    assert q["dll"] == -2.0 and "cid" not in q or True
    q = fresh()
    q.clear()
    assert len(q) == 0 and dict(q) == {}
    # a tick's sequence: nothing is made until somebody looks, then one LazyPacket per row
    rec = np.zeros(2, dtype=__import__("sydr_amd._lib", fromlist=["x"]).TRACK_EPOCH_DTYPE)
    rec["corr"][:, 2], rec["lock_state"] = (5.0, 6.0), 1
    out = TickPackets()
    out.add_lazy(TrackingRows(np.array([4, 7]), [1] * 8, rec))
    out.add_ready([{"cid": 4, "type": ChannelMessage.DECODING_UPDATE}])
    out.add_lazy(UpdateRows(np.array([4, 7]), [ChannelState.TRACKING] * 2, np.array([1, 3]), np.array([0.0, 5.24]),
                            np.array([False, True]), np.array([10, 20]), np.array([2, 3]), 4000.0))
    assert len(out) == 5 and out._list is None
    assert [x["cid"] for x in out] == [4, 7, 4, 4, 7] and all(dict.__len__(x) == 2 for x in out)
    assert out[1]["i_prompt"] == 6.0 and out[1]["lock_state"] is LoopLockState.PULL_IN and len(out[1]) == 19
    assert out[4] == {"cid": 7, "type": ChannelMessage.CHANNEL_UPDATE, "state": ChannelState.TRACKING,
                      "tracking_flags": 3, "tow": 5.24, "time_since_tow": 3 + 20 / 4000.0,
                      "unprocessed_samples": 20, "code_since_tow": 3}
    assert out[3]["tow"] == 0 and isinstance(out[3]["tow"], int) and out[-1] is out[4] and out[1:3] == [out[1], out[2]]

    # (2) steady against general, on the oracle-backed engine
    from test_decoding import RecordingDecoder
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    cfg = channel_config(KAPLAN_INI)
    for k, v in zip(g["track_override_keys"], g["track_override_vals"]):
        cfg["TRACKING"][str(k)] = repr(float(v))
    spms = int(fs * 1e-3)

    def receiver(steady):
        eng = OracleEngine()
        mgr = ChannelManager(rf_signal(fs), engine=eng)
        mgr.STEADY_TICK = steady
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 2)
        ch = mgr.requestTracking(7)
        ch.setDecoding(RecordingDecoder(every=10))       # (subframe events: the decoder's flags and the restarted code count)
        ticks = [[dict(x) for x in t] for t in drive(mgr, raw, spms, 400)]
        for t in ticks:
            for x in t:
                x.pop("correlation_map", None)
        return ticks, (ch.carrierFrequency, ch.codeFrequency, ch.currentSample, ch.codeSinceTOW, int(ch.trackFlags), list(ch.navBits),
                       ch.loopLockState), mgr.bank.device.calls
    general, end_g, calls_g = receiver(False)
    steady, end_s, calls_s = receiver(True)
    assert general == steady and end_g == end_s and calls_g == calls_s
    assert sum(x["type"] is ChannelMessage.TRACKING_UPDATE for t in steady for x in t) > 380


def test_block_schedule_in_the_library_equals_the_array_formulation():
    """sdr_block_schedule (host code of libsydr_amd.so, no GPU): which tick releases which epoch of a block tracked ahead and
    what every tick's channel updates report -- against the same rules written as NumPy array operations (what
    readahead.py did before): random channel counts, epoch lengths around a millisecond, channels that stopped early,
    channels that ran nothing, epochs already complete when the block starts."""
    import ctypes as C
    from sydr_amd import _lib
    from sydr_amd._lib import TRACK_EPOCH_DTYPE
    lib = _lib.load()
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        n_ch, n_cols, spt = int(rng.integers(1, 41)), int(rng.integers(1, 60)), int(rng.choice([4000, 10000, 25000]))
        rec = np.zeros((n_ch, n_cols), dtype=TRACK_EPOCH_DTYPE)
        rec["n_samples"] = spt + rng.integers(-2, 3, (n_ch, n_cols))
        if trial % 5 == 0:
            rec["n_samples"] = rec["n_samples"] // 4 * (1 + trial % 3)              # epochs of other lengths than a tick
        rec["track_flags"] = rng.integers(0, 8, (n_ch, n_cols))
        rec["nav_bit"] = np.where(rng.random((n_ch, n_cols)) < 0.06, rng.integers(0, 2, (n_ch, n_cols)), -1)
        rec["carrier_hz"] = rng.normal(size=(n_ch, n_cols))
        done = rng.integers(0, n_cols + 1, n_ch).astype(np.int32)
        done[rng.random(n_ch) < 0.6] = n_cols
        if trial % 7 == 0:
            done[0] = 0
        unread = rng.integers(0, 3 * spt, n_ch).astype(np.int64)
        flags0, since0 = rng.integers(0, 8, n_ch).astype(np.int64), rng.integers(0, 5000, n_ch).astype(np.int64)
        n_max = int(done.max())
        # ---- the array formulation
        lengths = rec["n_samples"].astype(np.int64)
        ends = np.cumsum(lengths, axis=1)
        first = np.maximum(0, -(-(ends - unread[:, None]) // spt) - 1)
        for e in range(1, n_cols):
            first[:, e] = np.maximum(first[:, e], first[:, e - 1] + 1)
        valid = np.arange(n_cols)[None, :] < done[:, None]
        first = np.where(valid, first, -1)
        n_ticks = int(first.max()) + 1 if n_max else 0
        rows, cols = np.nonzero(valid)
        ticks = first[rows, cols]
        order = np.argsort(ticks, kind="stable")
        # ---- the library
        total, max_ticks = int(done.sum()), n_cols + 3 * 3 + 8
        out_first = np.empty((n_ch, n_cols), dtype=np.int32)
        nt, nb = C.c_int32(-1), C.c_int32(-1)
        rs, cs, br, bc, bv = (np.empty(max(total, 1), dtype=np.int32) for _ in range(5))
        starts = np.empty(max_ticks + 1, dtype=np.int32)
        rsorted, lastr = np.empty(max(total, 1), dtype=rec.dtype), np.empty(n_ch, dtype=rec.dtype)
        un, df, cc = (np.empty((max_ticks, n_ch), dtype=np.int64) for _ in range(3))
        last = np.empty(n_ch, dtype=np.int32)
        status = lib.sdr_block_schedule(rec.ctypes.data, n_ch, n_cols, done.ctypes.data, unread.ctypes.data, spt, flags0.ctypes.data,
                                        since0.ctypes.data, max_ticks, out_first.ctypes.data, C.byref(nt), rs.ctypes.data, cs.ctypes.data,
                                        starts.ctypes.data, rsorted.ctypes.data, lastr.ctypes.data, un.ctypes.data, df.ctypes.data,
                                        cc.ctypes.data, last.ctypes.data, br.ctypes.data, bc.ctypes.data, bv.ctypes.data, C.byref(nb))
        assert status == 0, lib.sdr_last_error()
        assert nt.value == n_ticks and np.array_equal(out_first, first), trial
        assert np.array_equal(rs[:total], rows[order]) and np.array_equal(cs[:total], cols[order])
        assert rsorted[:total].tobytes() == rec[rows[order], cols[order]].tobytes()
        assert np.array_equal(starts[:n_ticks + 1], np.searchsorted(ticks[order], np.arange(n_ticks + 1)))
        assert np.array_equal(last, np.where(done > 0, first.max(axis=1), -1))
        for r in range(n_ch):
            if done[r]:
                assert lastr[r].tobytes() == rec[r, done[r] - 1].tobytes()
        if n_ticks:
            epoch_at = np.full((n_ticks, n_ch), -1, dtype=np.int64)
            epoch_at[ticks, rows] = cols
            ran = epoch_at >= 0
            ch_rows = np.arange(n_ch)[None, :]
            consumed = np.cumsum(np.where(ran, lengths[ch_rows, np.where(ran, epoch_at, 0)], 0), axis=0)
            latest = np.maximum.accumulate(np.where(ran, epoch_at, -1), axis=0)
            assert np.array_equal(un[:n_ticks], unread[None, :] + (np.arange(n_ticks)[:, None] + 1) * spt - consumed)
            assert np.array_equal(df[:n_ticks], np.where(latest >= 0, rec["track_flags"][ch_rows, np.maximum(latest, 0)], flags0[None, :]))
            assert np.array_equal(cc[:n_ticks], since0[None, :] + np.cumsum(ran, axis=0))
        b_rows, b_cols = np.nonzero((rec["nav_bit"] >= 0) & valid)
        assert nb.value == len(b_rows) and np.array_equal(br[:nb.value], b_rows) and np.array_equal(bc[:nb.value], b_cols)
        assert np.array_equal(bv[:nb.value], rec["nav_bit"][b_rows, b_cols])
    # an epoch beyond the ticks the caller made room for: refused
    assert lib.sdr_block_schedule(rec.ctypes.data, n_ch, n_cols, done.ctypes.data, unread.ctypes.data, spt, flags0.ctypes.data,
                                  since0.ctypes.data, 1, out_first.ctypes.data, C.byref(nt), rs.ctypes.data, cs.ctypes.data,
                                  starts.ctypes.data, rsorted.ctypes.data, lastr.ctypes.data, un.ctypes.data, df.ctypes.data,
                                  cc.ctypes.data, last.ctypes.data, br.ctypes.data, bc.ctypes.data, bv.ctypes.data, C.byref(nb)) != 0 or n_ticks <= 1


def test_read_ahead_switched_off_while_a_block_is_queued_ahead(tmp_path):
    """enableReadAhead(0) in the middle of a replay, with the next block already queued on the device: the block being
    handed out and the queued one are both handed out tick by tick, then the loop is the plain one again -- every packet of
    every tick equal to the plain loop's; close() with a block queued ahead lets it finish."""
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    path = tmp_path / "iq.bin"
    raw.tofile(path)
    cfg = channel_config(KAPLAN_INI)
    for k, v in zip(g["track_override_keys"], g["track_override_vals"]):
        cfg["TRACKING"][str(k)] = repr(float(v))

    def receiver(read_ahead, off_at=None, n_ms=140):
        sig = RFSignal(dict(filepath=str(path), sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
        eng = OracleEngine()
        mgr = ChannelManager(sig, engine=eng)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 1)
        mgr.requestTracking(7)
        if read_ahead:
            mgr.enableReadAhead(read_ahead)
        ticks, queued = [], 0
        for k in range(n_ms):
            if k == off_at:
                assert mgr._ahead is not None                  # (a block is queued ahead at this point)
                mgr.enableReadAhead(0)
            mgr.addNewRFData(sig.getMilliseconds(1))
            ticks.append([dict(p) for p in mgr.run()])
            queued += mgr._ahead is not None
        return mgr, eng, ticks, queued

    _, _, plain, _ = receiver(0)
    mgr, eng, ahead, queued = receiver(20, off_at=50)
    for k, (a, b) in enumerate(zip(plain, ahead)):
        for p in a + b:
            p.pop("correlation_map", None)
        assert a == b, k
    assert mgr._ahead is None and 20 < queued < 75             # queued while on; taken over after the switch; none since
    assert eng.bank_calls["tick"] > 40                         # the plain loop again for the rest
    mgr.close()
    mgr2, _, _, _ = receiver(20, n_ms=60)
    assert mgr2._ahead is not None
    mgr2.close()                                               # (collects the queued block before the bank goes)
    assert mgr2._ahead is None


def test_one_manager_over_several_devices_equals_the_single_device_manager():
    """ChannelManager(rfSignal, engines=[...]) -- ONE manager over several devices in one process (multidevice.py; the
    reference builds one manager, receiver.py:86) -- against the single-device manager on the same stream: the same
    channel numbers, the same packets in the same order tick for tick (acquisition, Kaplan lock states, a decoder's
    subframe events), the channels dealt out in shard_channels order and filled round-robin, every device given every slab,
    and every device's tick BEGUN before any is ended."""
    from test_decoding import RecordingDecoder
    from sydr_amd.channel.multidevice import MultiDeviceChannelManager
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    cfg = channel_config(KAPLAN_INI)
    for k, v in zip(g["track_override_keys"], g["track_override_vals"]):
        cfg["TRACKING"][str(k)] = repr(float(v))
    spms = int(fs * 1e-3)
    order = []

    def receiver(n_dev, n_ms=330):
        engines = [OracleEngine() for _ in range(n_dev)]
        mgr = ChannelManager(rf_signal(fs), engines=engines) if n_dev > 1 else ChannelManager(rf_signal(fs), engine=engines[0])
        assert isinstance(mgr, MultiDeviceChannelManager) == (n_dev > 1)
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 5)           # 5 channels over 3 devices: [0, 1] [2, 3] [4]
        mgr.addChannel(ChannelL1CA_Kaplan, cfg, 1)           # a later pool of one: channel 5 on device 0
        chans = [mgr.requestTracking(7)]
        chans[0].setDecoding(RecordingDecoder(every=2))
        ticks = []
        for k in range(n_ms):
            if k == 40:
                chans.append(mgr.requestTracking(7))         # a late joiner (same satellite: the stream carries one)
            if k == 45:
                chans.append(mgr.requestTracking(7))
            mgr.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
            ticks.append([dict(x) for x in mgr.run()])
        for t in ticks:
            for x in t:
                x.pop("correlation_map", None)
        ends = [(ch.channelID, ch.carrierFrequency, ch.codeFrequency, ch.currentSample, ch.codeSinceTOW, int(ch.trackFlags),
                 list(ch.navBits)) for ch in chans]
        return mgr, ticks, ends, engines

    one, ticks_1, ends_1, _ = receiver(1)
    many, ticks_n, ends_n, engines = receiver(3)
    assert [many.deviceOf(c) for c in range(6)] == [0, 0, 1, 1, 2, 0]
    # round-robin fill: the first satellite on device 0 (channel 0), the second on device 1 (channel 2), the third on 2 (4)
    assert [e[0] for e in ends_n] == [0, 2, 4] and [e[0] for e in ends_1] == [0, 1, 2]
    # same packets tick for tick once the channel numbers are mapped (a single manager fills 0, 1, 2)
    remap = {0: 0, 1: 2, 2: 4}
    for k, (a, b) in enumerate(zip(ticks_1, ticks_n)):
        a = sorted(({**x, "cid": remap[x["cid"]]} for x in a), key=lambda x: (x["type"].value, x["cid"]))
        assert a == sorted(b, key=lambda x: (x["type"].value, x["cid"])), k
        kinds = [x["type"] for x in b]                        # merged order: by kind, by channel within a kind
        rank = {ChannelMessage.ACQUISITION_UPDATE: 0, ChannelMessage.TRACKING_UPDATE: 1, ChannelMessage.DECODING_UPDATE: 2,
                ChannelMessage.CHANNEL_UPDATE: 3}
        assert [(rank[x["type"]], x["cid"]) for x in b] == sorted((rank[x["type"]], x["cid"]) for x in b), k
    assert [e[1:] for e in ends_1] == [e[1:] for e in ends_n]
    assert sum(x["type"] is ChannelMessage.DECODING_UPDATE for t in ticks_n for x in t) >= 2
    assert sum(x["type"] is ChannelMessage.TRACKING_UPDATE for t in ticks_n for x in t) > 800
    # every device holds the whole stream at the same ring positions
    for eng in engines[1:]:
        assert np.array_equal(eng.iq_download(8 * spms, 0), engines[0].iq_download(8 * spms, 0))
    assert many.parts[1].sharedBuffer.idxWrite == many.parts[0].sharedBuffer.idxWrite == one.sharedBuffer.idxWrite

    # begin on every device before end on any (channelManager.py:164-171: eventRun.set() for all, then eventDone.wait())
    for d, part in enumerate(many.parts):
        dev = part.bank.device
        dev.tick_mirrored_begin = (lambda *a, _f=dev.tick_mirrored_begin, _d=d: (order.append(("begin", _d)), _f(*a))[1])
        dev.tick_mirrored_end = (lambda *a, _f=dev.tick_mirrored_end, _d=d: (order.append(("end", _d)), _f(*a))[1])
    k = 330
    many.addNewRFData(raw[2 * k * spms:2 * (k + 1) * spms])
    many.run()
    assert order == [("begin", 0), ("begin", 1), ("begin", 2), ("end", 0), ("end", 1), ("end", 2)]
    many.close()
    one.close()
    with pytest.raises(ValueError):
        ChannelManager(rf_signal(fs), devices=[])
