"""BASELINE.json's full-size configurations through size-independent properties (the oracle would need hours there):
what a node of 8 GPUs would compute -- every rank the same stream, its own 32 channels -- must be, bit for bit, what
one launch over all channels computes.  Run here as 8 shards one after the other on the one GPU of the box."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import sydr_oracle as orc
from sydr_amd.channel.manager import shard_channels
from sydr_amd.engine import FMT_CI8, make_items

pytestmark = pytest.mark.gpu


def _bench():
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import bench
    return bench


def test_config3_sixty_seconds_sharded_like_eight_gpus(engine):
    """Config 3 at full size: 32 channels, 25 MHz, 60 s (1.92 M channel-epochs, 3 GB of ci8).  The 32 channels split
    4 per shard (8 'GPUs') give bitwise the single launch; the prompt tap carries the satellites' power over the whole
    minute; a sample of epochs agrees with the oracle."""
    bench = _bench()
    fs = bench.FS
    total = int(60.0 * fs) // 8 * 8
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    sats = bench.satellites()
    for s, sat in enumerate(sats):
        engine.load_gps_code(s, sat["prn"])
    engine.iq_synth(sats, fs, 12.0, 20260003, 0, total)
    items, n_epochs = bench.truth_items(sats, fs, total)
    assert n_epochs >= 59990 and len(items) == n_epochs * 32
    plan = engine.epl_plan(items, bench.SPACING, fs)
    plan.run()
    whole = plan.fetch()
    plan.close()
    per_channel = items.reshape(n_epochs, 32)
    out_ch = whole.reshape(n_epochs, 32, 6)
    for rank in range(8):
        mine = shard_channels(32, rank, 8)
        sub = np.ascontiguousarray(per_channel[:, mine]).reshape(-1)
        p = engine.epl_plan(sub, bench.SPACING, fs)
        p.run()
        got = p.fetch().reshape(n_epochs, len(mine), 6)
        p.close()
        assert np.array_equal(got, out_ch[:, mine]), rank
    # every channel stays on its satellite for the whole minute: prompt power >> early/late imbalance, no dropouts
    prompt = np.hypot(out_ch[:, :, 2], out_ch[:, :, 3])
    expect = 3.0 * 25000                                           # amplitude x samples per epoch
    # (31 equally strong C/A interferers: cross-correlation makes a few epochs in a thousand dip by a third)
    assert np.median(prompt) == pytest.approx(expect, rel=0.1) and np.quantile(prompt, 0.01) > 0.7 * expect
    assert prompt.min() > 0.3 * expect
    early, late = np.hypot(out_ch[:, :, 0], out_ch[:, :, 1]), np.hypot(out_ch[:, :, 4], out_ch[:, :, 5])
    assert abs(np.median(early / prompt) - 0.5) < 0.05 and abs(np.median(late / prompt) - 0.5) < 0.05
    # a handful of epochs from the last second against the oracle
    lo = (n_epochs - 3) * 32
    first = int(items["start_sample"][lo:].min())
    last = int((items["start_sample"][lo:] + items["n_samples"][lo:]).max())
    rf = orc.iq_to_complex(engine.iq_download(last - first, first))
    for k in (lo, lo + 17, lo + 63, lo + 95):
        it = items[k]
        s0 = int(it["start_sample"]) - first
        ref = np.array(orc.epl(rf[s0:s0 + int(it["n_samples"])], orc.pad_code(orc.gold_code(sats[int(it["code_slot"])]["prn"])),
                               fs, float(it["carrier_hz"]), float(it["rem_carrier"]), float(it["rem_code"]),
                               float(it["code_step"]), bench.SPACING))
        scale = np.repeat(np.hypot(ref[0::2], ref[1::2]), 2)
        assert np.max(np.abs(whole[k] - ref) / scale) < 1e-9


def test_config5_256_channels_sharded_32_per_gpu(engine):
    """Config 5's geometry: 256 channels (128 GPS with 4-period epochs + 128 BOC(1,1) half-chip codes), 5 taps, 50 MHz,
    4 ms epochs, ONE stream; 32 channels per 'GPU'.  Shards == the single 256-channel launch, bit for bit."""
    fs, n_gps, n_e1, n_ep = 50e6, 128, 128, 6
    taps = (-1.0, -0.5, 0.0, 0.5, 1.0)
    total = int((n_ep + 2) * 4e-3 * fs) // 8 * 8
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(n_gps + 2 * n_e1, 8184)
    rng = np.random.default_rng(20260005)
    sats = []
    for i in range(n_gps):
        engine.load_gps_code(i, 1 + i % 210)
        sats.append(dict(prn=1 + i % 210, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 1023)),
                         phase=float(rng.random()), amp=1.0, slot=None, chips=1023.0, half=1, corr_slot=i))
    for i in range(n_e1):
        code = np.where(rng.random(4092) < 0.5, -1, 1).astype(np.int8)
        engine.set_code(n_gps + i, code)
        half = np.empty(8184, dtype=np.int8)
        half[0::2], half[1::2] = code, -code
        engine.set_code(n_gps + n_e1 + i, half)
        sats.append(dict(slot=n_gps + i, boc=True, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 4092)),
                         phase=float(rng.random()), amp=1.0, chips=4092.0, half=2, corr_slot=n_gps + n_e1 + i))
    engine.iq_synth([{k: v for k, v in s.items() if v is not None and k in ("prn", "slot", "boc", "doppler", "code_phase", "phase", "amp")}
                     for s in sats], fs, 10.0, 20260005, 0, total)

    def items_for(group):
        """Epoch-major items along the true trajectories (as bench.py's multignss workload builds them)."""
        dop = np.array([s["doppler"] for s in group])
        chips, half = group[0]["chips"], group[0]["half"]
        span = chips * (4 if chips == 1023.0 else 1)
        cstep = 1.023e6 * (1.0 + dop / 1575.42e6) / fs
        cp0, ph0 = np.array([s["code_phase"] for s in group]), np.array([s["phase"] for s in group])
        start = np.ceil((chips - cp0) / cstep).astype(np.int64)
        rem = cp0 + start * cstep - chips
        rows = []
        for _ in range(n_ep):
            n = np.ceil((span - rem) / cstep).astype(np.int64)
            cyc = dop / fs * start + ph0
            rows.append((n.copy(), start.copy(), (-2.0 * np.pi * (cyc - np.floor(cyc))) % (2.0 * np.pi), rem.copy()))
            rem = rem + n * cstep - span
            start = start + n
        slots = np.tile([s["corr_slot"] for s in group], n_ep)
        return make_items(slots, np.stack([r[0] for r in rows]).reshape(-1), np.stack([r[1] for r in rows]).reshape(-1),
                          np.tile(dop, n_ep), np.stack([r[2] for r in rows]).reshape(-1),
                          half * np.stack([r[3] for r in rows]).reshape(-1), np.tile(half * cstep, n_ep))

    for group, spacing in ((sats[:n_gps], taps), (sats[n_gps:], tuple(2 * t for t in taps))):
        items = items_for(group)
        n_ch = len(group)
        whole = engine.epl_batch(items, spacing, fs).reshape(n_ep, n_ch, 10)
        prompt = np.hypot(whole[:, :, 4], whole[:, :, 5])
        # every channel on its satellite (a 4 ms GPS epoch that straddles a data-bit edge loses part of its sum)
        assert np.median(prompt) > 0.8 * 1.0 * 200000 and prompt.min() > 0.1 * 200000
        grid = items.reshape(n_ep, n_ch)
        for rank in range(8):                                      # 16 of this constellation's channels per 'GPU'
            mine = shard_channels(n_ch, rank, 8)
            got = engine.epl_batch(np.ascontiguousarray(grid[:, mine]).reshape(-1), spacing, fs).reshape(n_ep, len(mine), 10)
            assert np.array_equal(got, whole[:, mine]), rank


def _mix64(z):
    """mix64 of the device generator (sydr_amd/csrc/codes.hip), vectorised over uint64 arrays."""
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return z ^ (z >> np.uint64(31))


def test_config3_closed_loop_sixty_seconds_recovers_every_data_bit(engine):
    """Config 3 with the loops closed on the device for the whole minute: 32 channels x ~60 000 epochs in ONE launch.
    Every channel keeps lock, and the navigation bits the device decides (20-prompt sums after bit sync) are the data
    the generator modulated onto that satellite -- ~3000 bits per channel, none wrong (up to the Costas loop's sign)."""
    bench = _bench()
    from sydr_amd._lib import LoopCfg, TrackState
    fs, seed = bench.FS, 20260003
    total = int(60.0 * fs) // 8 * 8
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(32)
    sats = bench.satellites()
    for s, sat in enumerate(sats):
        engine.load_gps_code(s, sat["prn"])
    engine.iq_synth(sats, fs, 12.0, seed, 0, total)
    items, n_epochs = bench.truth_items(sats, fs, total)
    n_run = n_epochs - 5
    cfg = LoopCfg()
    cfg.loop_kind, cfg.n_taps, cfg.fs = 1, 3, fs
    for t, sp in enumerate(bench.SPACING):
        cfg.spacing_wide[t] = cfg.spacing_narrow[t] = sp
    cfg.dll_tau1, cfg.dll_tau2 = orc.loop_coefficients(2.0, 0.7, 1.0)      # channel_GPS_L1CA_kaplan.ini
    cfg.dll_pdi, cfg.dll_threshold = 0.001, 10.0
    cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow, cfg.fll_thr_wide, cfg.fll_thr_narrow = 100.0, 50.0, 15.0, 0.5, 0.8
    cfg.pll_bw_wide, cfg.pll_bw_narrow, cfg.pll_thr_wide, cfg.pll_thr_narrow = 25.0, 15.0, 0.5, 0.8
    states = []
    for c in range(32):
        it = items[c]
        st = TrackState()
        st.code_slot, st.n_samples, st.current_sample = int(it["code_slot"]), int(it["n_samples"]), int(it["start_sample"])
        st.carrier_hz, st.code_hz = float(it["carrier_hz"]), 1.023e6
        st.rem_carrier, st.rem_code, st.code_step = float(it["rem_carrier"]), float(it["rem_code"]), 1.023e6 / fs
        st.fll_bw, st.pll_bw, st.lock_state = 100.0, 25.0, 1
        states.append(st)
    end, _, bits = engine.track_closed_loop(states, cfg, n_run, want_traj=False, want_bits=True)
    j = np.arange(n_run // 20 + 8, dtype=np.uint64)
    for c, sat in enumerate(sats):
        assert abs(end[c].carrier_hz - sat["doppler"]) < 20.0, c                       # still on its Doppler
        assert end[c].current_sample > total - 10 * 25000 and end[c].lock_state == 3    # ran to the end, NARROW lock
        got = np.asarray(bits[c], dtype=np.int64)
        assert len(got) > 2900, (c, len(got))                                           # synced within the first 2 s
        sent = (_mix64(np.uint64(seed) ^ _mix64(np.uint64(sat["prn"]) * np.uint64(0x100000001B3) + j)) & np.uint64(1)).astype(np.int64)
        # the first decided bit is data bit j0 for some j0 (the epoch of bit sync is not returned): find it
        matches = [(j0, pol) for j0 in range(0, len(sent) - len(got) + 1) for pol in (0, 1)
                   if np.array_equal(got[:64] ^ pol, sent[j0:j0 + 64])]
        assert len(matches) == 1, (c, matches)
        j0, pol = matches[0]
        assert np.array_equal(got ^ pol, sent[j0:j0 + len(got)]), c
