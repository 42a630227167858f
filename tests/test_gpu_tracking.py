"""GPU parity of the closed-loop tracking kernel, the host plugins on the real engine and the
function-level drop-ins, against the reference's golden trajectories and the CPU oracle.

Integers (sample indices, epoch lengths, lock states, flags) must match exactly; floating-point
loop quantities to 1e-9 relative -- the feedback loop is contractive, so the ~1e-15 reduction-order
differences of the correlators do not grow."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import sydr_oracle as orc
from test_host_layer import BORRE_INI, KAPLAN_INI, channel_config, drive, rf_signal
from test_oracle_golden import BORRE_CFG, KAPLAN_CFG, kaplan_strong_cfg, trajectory_iq

from sydr_amd._lib import LoopCfg, TrackState
from sydr_amd.engine import FMT_CI8

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def loop_cfg(kind, fs, c):
    cfg = LoopCfg()
    cfg.loop_kind, cfg.n_taps, cfg.fs = kind, 3, fs
    cfg.dll_tau1, cfg.dll_tau2 = orc.loop_coefficients(c["dll_noise_bandwidth"], c["dll_damping_ratio"], c["dll_loop_gain"])
    cfg.dll_pdi = c["dll_pdi"]
    if kind == 0:
        sp = [c["correlator_early"], c["correlator_prompt"], c["correlator_late"]]
        for t in range(3):
            cfg.spacing_wide[t] = cfg.spacing_narrow[t] = sp[t]
        cfg.pll_tau1, cfg.pll_tau2 = orc.loop_coefficients(c["pll_noise_bandwidth"], c["pll_damping_ratio"], c["pll_loop_gain"])
        cfg.pll_pdi = c["pll_pdi"]
    else:
        for t, s in enumerate((-1.0, 0.0, 1.0)):
            cfg.spacing_wide[t] = s * c["correlator_epl_wide"]
            cfg.spacing_narrow[t] = s * c["correlator_epl_narrow"]
        cfg.dll_threshold = c["dll_threshold"]
        cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow = c["fll_bandwidth_pullin"], c["fll_bandwidth_wide"], c["fll_bandwidth_narrow"]
        cfg.fll_thr_wide, cfg.fll_thr_narrow = c["fll_threshold_wide"], c["fll_threshold_narrow"]
        cfg.pll_bw_wide, cfg.pll_bw_narrow = c["pll_bandwidth_wide"], c["pll_bandwidth_narrow"]
        cfg.pll_thr_wide, cfg.pll_thr_narrow = c["pll_threshold_wide"], c["pll_threshold_narrow"]
    return cfg


def initial_state(kind, fs, carrier, current_sample, c, slot=0):
    st = TrackState()
    st.code_slot = slot
    st.code_step = orc.CODE_RATE / fs
    st.n_samples = orc.required_samples(0.0, st.code_step)
    st.current_sample = current_sample
    st.carrier_hz, st.code_hz = carrier, orc.CODE_RATE
    if kind == 1:
        st.fll_bw, st.pll_bw = c["fll_bandwidth_pullin"], c["pll_bandwidth_wide"]
        st.lock_state = orc.LOCK_PULL_IN
    return st


def close(a, b, scale=None):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    s = np.maximum(np.abs(b), 1e-300) if scale is None else scale
    return np.all(np.abs(a - b) <= RTOL * s)


CASES = {"borre": ("g6_trajectories.npz", 0), "kaplan": ("g6_trajectories.npz", 1),
         "kaplan_strong": ("g6b_kaplan_strong.npz", 1),
         # the headline sampling rate: the reference's own plugins at 25 MHz (boundary variant of the correlator)
         "borre_25mhz": ("g6c_25mhz.npz", 0), "kaplan_25mhz": ("g6c_25mhz.npz", 1)}


@pytest.mark.parametrize("parts", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("case", list(CASES))
def test_closed_loop_kernel_matches_reference_trajectory(engine, case, parts):
    """parts = workgroups cooperating on the channel (0: the library fills the GPU, i.e. 8 for one channel)."""
    engine.track_cluster(parts)
    try:
        _check_closed_loop_against_reference(engine, case)
    finally:
        engine.track_cluster(0)


def _check_closed_loop_against_reference(engine, case):
    fname, kind = CASES[case]
    g, fs, raw = trajectory_iq(fname)
    plugin = "borre" if kind == 0 else "kaplan"
    c = BORRE_CFG if kind == 0 else (kaplan_strong_cfg(g) if case == "kaplan_strong" else KAPLAN_CFG)
    ref, acq = g[f"{plugin}_epochs"], g[f"{plugin}_acq"]
    n = raw.size // 2
    engine.iq_alloc((n + 7) // 8 * 8, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    st = initial_state(kind, fs, acq[3], int(acq[5]), c, slot=1)
    states, traj, bits = engine.track_closed_loop([st], loop_cfg(kind, fs, c), len(ref), want_bits=True)
    tr = traj[0]
    # navigation bits (20-prompt sums after bit sync) decided on the device == the reference's
    assert np.array_equal(tr["nav_bit"], ref[:, 24].astype(np.int32))
    assert np.array_equal(bits[0], ref[ref[:, 24] >= 0, 24].astype(np.int8))
    assert states[0].nav_bits_emitted == len(bits[0])
    ring = 100 * int(fs * 1e-3)
    # integers: exact
    assert np.array_equal(tr["start_sample"] % ring, ref[:, 0].astype(np.int64))
    assert np.array_equal(tr["n_samples"], ref[:, 1].astype(np.int32))
    # NCO inputs and correlators of every epoch
    assert close(tr["carrier_hz_in"], ref[:, 2]) and close(tr["code_step_in"], ref[:, 5])
    assert close(tr["rem_carrier_in"], ref[:, 3], scale=2 * np.pi) and close(tr["rem_code_in"], ref[:, 4], scale=1.0)
    corr = tr["corr"][:, :6]
    for t in range(3):
        mag = np.hypot(ref[:, 6 + 2 * t], ref[:, 7 + 2 * t])
        err = np.hypot(corr[:, 2 * t] - ref[:, 6 + 2 * t], corr[:, 2 * t + 1] - ref[:, 7 + 2 * t])
        assert np.all(err <= RTOL * np.maximum(mag, 1.0)), (t, (err / mag).max())
    assert close(tr["carrier_hz"], ref[:, 15]) and close(tr["code_hz"], ref[:, 16])
    assert close(tr["dll"], ref[:, 12], scale=1.0) and close(tr["pll"], ref[:, 13], scale=1.0)
    assert close(tr["carrier_err"], ref[:, 17], scale=1.0) and close(tr["code_err"], ref[:, 18], scale=1.0)
    if kind == 1:
        assert close(tr["fll"], ref[:, 14], scale=1.0)
        assert close(tr["cn0"], ref[:, 19], scale=1.0)
        assert close(tr["pll_lock"], ref[:, 20], scale=1.0) and close(tr["fll_lock"], ref[:, 21], scale=1.0)
        assert np.array_equal(tr["lock_state"], ref[:, 22].astype(np.int32))
        assert np.array_equal(tr["track_flags"], ref[:, 23].astype(np.int32))
    # end state continues where the trajectory stopped
    assert states[0].current_sample == int(tr["start_sample"][-1]) + int(tr["n_samples"][-1])
    assert states[0].code_counter == len(ref)


@pytest.mark.parametrize("kind,parts", [(1, 0), (1, 1), (1, 4), (0, 0), (0, 2)])
def test_closed_loop_at_25_mhz_matches_the_oracle_loops(engine, kind, parts):
    """The golden trajectories are 4 MHz runs (per-sample correlator variant).  At 25 MHz the tracking kernel
    uses the boundary variant; the same loops (oracle restatement of the plugins, pinned by the goldens) run on
    the CPU over the same synthetic stream must give the same integers and the same loop quantities."""
    fs, ms, prn = 25e6, 130, 11
    n = int(ms * fs * 1e-3)
    sat = dict(prn=prn, doppler=2250.0, code_phase=417.3, phase=0.2, amp=9.0)
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(1)
    engine.load_gps_code(0, prn)
    engine.iq_synth([sat], fs, 14.0, 991, 0, n)
    rf = orc.iq_to_complex(engine.iq_download(n, 0))
    pb, pc, _, _ = engine.pcps([0], 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    n_code = orc.samples_per_code(fs)
    n0 = orc.required_samples(0.0, orc.CODE_RATE / fs)
    carrier, _, cur = orc.post_acquisition(0.0, 5000.0, 250.0, [int(pb[0]), int(pc[0])], 0, n_code, n0)
    c = KAPLAN_CFG if kind == 1 else BORRE_CFG
    loop = (orc.KaplanLoop if kind == 1 else orc.BorreLoop)(fs, orc.gold_code(prn), c, carrier, cur)
    epochs = 120
    ref = [loop.step(rf[loop.current_sample:loop.current_sample + loop.n]) for _ in range(epochs)]
    engine.track_cluster(parts)
    try:
        states, traj = engine.track_closed_loop([initial_state(kind, fs, carrier, cur, c)], loop_cfg(kind, fs, c), epochs)
    finally:
        engine.track_cluster(0)
    tr = traj[0]
    assert np.array_equal(tr["start_sample"], [r["start"] for r in ref])
    assert np.array_equal(tr["n_samples"], [r["n"] for r in ref])
    corr_ref = np.array([r["corr"] for r in ref])
    for t in range(3):
        mag = np.hypot(corr_ref[:, 2 * t], corr_ref[:, 2 * t + 1])
        err = np.hypot(tr["corr"][:, 2 * t] - corr_ref[:, 2 * t], tr["corr"][:, 2 * t + 1] - corr_ref[:, 2 * t + 1])
        assert np.all(err <= RTOL * np.maximum(mag, 1.0)), (t, (err / mag).max())
    assert close(tr["carrier_hz"], [r["carrier_hz"] for r in ref]) and close(tr["code_hz"], [r["code_hz"] for r in ref])
    assert close(tr["carrier_err"], [r["carrier_err"] for r in ref], scale=1.0)
    assert close(tr["code_err"], [r["code_err"] for r in ref], scale=1.0)
    if kind == 1:
        assert np.array_equal(tr["lock_state"], [r["lock_state"] for r in ref])
        assert np.array_equal(tr["track_flags"], [r["flags"] for r in ref])
        assert close(tr["fll_lock"], [r["fll_lock"] for r in ref], scale=1.0)
    # the loop is on the signal: prompt dominates and the carrier sits on the Doppler
    assert abs(tr["carrier_hz"][-1] - sat["doppler"]) < 40.0
    assert np.hypot(tr["corr"][-1, 2], tr["corr"][-1, 3]) > np.hypot(tr["corr"][-1, 0], tr["corr"][-1, 1])


@pytest.mark.parametrize("fmt_name", ["ci16", "cf32", "cf64"])
def test_closed_loop_other_ring_formats(engine, fmt_name):
    """The tracking kernel is instantiated per ring format: the golden stream stored as int16 / float32 / float64
    pairs gives the trajectory of the int8 run, bit for bit (same values, same arithmetic)."""
    from sydr_amd.engine import FMT_CF32, FMT_CF64, FMT_CI16
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    c = kaplan_strong_cfg(g)
    acq = g["kaplan_acq"]
    n = raw.size // 2
    cap = (n + 7) // 8 * 8

    def run(fmt, data):
        engine.iq_alloc(cap, fmt)
        engine.iq_upload(data, 0)
        engine.code_slots(2)
        engine.load_gps_code(1, 7)
        st = initial_state(1, fs, acq[3], int(acq[5]), c, slot=1)
        return engine.track_closed_loop([st], loop_cfg(1, fs, c), 300)[1][0]

    base = run(FMT_CI8, raw)
    fmt, data = {"ci16": (FMT_CI16, raw.astype(np.int16)), "cf32": (FMT_CF32, raw.astype(np.float32)),
                 "cf64": (FMT_CF64, raw.astype(np.float64))}[fmt_name]
    other = run(fmt, data)
    assert other.tobytes() == base.tobytes()


def test_closed_loop_more_channels_than_compute_units(engine):
    """Beyond one channel per CU the launcher switches to the 256-thread kernel built for three workgroups per CU.
    260 channels tracking the golden stream from the same state: every one reproduces the reference trajectory."""
    g, fs, raw = trajectory_iq("g6b_kaplan_strong.npz")
    c = kaplan_strong_cfg(g)
    ref, acq = g["kaplan_epochs"], g["kaplan_acq"]
    n = raw.size // 2
    engine.iq_alloc((n + 7) // 8 * 8, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 7)
    n_ch, epochs = 260, 400
    states = [initial_state(1, fs, acq[3], int(acq[5]), c, slot=1) for _ in range(n_ch)]
    end, traj = engine.track_closed_loop(states, loop_cfg(1, fs, c), epochs)
    ring = 100 * int(fs * 1e-3)
    for k in (0, 137, 259):
        tr = traj[k]
        assert np.array_equal(tr["start_sample"] % ring, ref[:epochs, 0].astype(np.int64))
        assert np.array_equal(tr["n_samples"], ref[:epochs, 1].astype(np.int32))
        assert np.array_equal(tr["lock_state"], ref[:epochs, 22].astype(np.int32))
        assert np.array_equal(tr["track_flags"], ref[:epochs, 23].astype(np.int32))
        assert np.array_equal(tr["nav_bit"], ref[:epochs, 24].astype(np.int32))
        assert close(tr["carrier_hz"], ref[:epochs, 15]) and close(tr["code_hz"], ref[:epochs, 16])
    assert traj[5].tobytes() == traj[200].tobytes()      # same inputs, same kernel: bitwise the same


@pytest.mark.parametrize("kind", [1, 0])
def test_dense_form_at_25_mhz_matches_the_oracle_loops(engine, kind):
    """More channels than compute units at the headline rate: the 256-thread kernel (three workgroups per CU) correlates
    with the chip-aligned core there (correlator_chip.h: block length compiled in, tap positions at run time) instead of
    the 16-sample boundary groups.  260 channels on one synthetic stream from the same state: the oracle's loops (pinned
    by the goldens) over the same stream give the same integers and the same loop quantities, for both plugins."""
    fs, ms, prn = 25e6, 130, 11
    n = int(ms * fs * 1e-3)
    sat = dict(prn=prn, doppler=-1750.0, code_phase=893.6, phase=0.4, amp=9.0)
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(1)
    engine.load_gps_code(0, prn)
    engine.iq_synth([sat], fs, 14.0, 4711, 0, n)
    rf = orc.iq_to_complex(engine.iq_download(n, 0))
    pb, pc, _, _ = engine.pcps([0], 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    n_code = orc.samples_per_code(fs)
    n0 = orc.required_samples(0.0, orc.CODE_RATE / fs)
    carrier, _, cur = orc.post_acquisition(0.0, 5000.0, 250.0, [int(pb[0]), int(pc[0])], 0, n_code, n0)
    c = KAPLAN_CFG if kind == 1 else BORRE_CFG
    loop = (orc.KaplanLoop if kind == 1 else orc.BorreLoop)(fs, orc.gold_code(prn), c, carrier, cur)
    epochs = 100
    ref = [loop.step(rf[loop.current_sample:loop.current_sample + loop.n]) for _ in range(epochs)]
    n_ch = 260
    states, traj = engine.track_closed_loop([initial_state(kind, fs, carrier, cur, c) for _ in range(n_ch)], loop_cfg(kind, fs, c), epochs)
    corr_ref = np.array([r["corr"] for r in ref])
    for k in (0, 131, 259):
        tr = traj[k]
        assert np.array_equal(tr["start_sample"], [r["start"] for r in ref])
        assert np.array_equal(tr["n_samples"], [r["n"] for r in ref])
        for t in range(3):
            mag = np.hypot(corr_ref[:, 2 * t], corr_ref[:, 2 * t + 1])
            err = np.hypot(tr["corr"][:, 2 * t] - corr_ref[:, 2 * t], tr["corr"][:, 2 * t + 1] - corr_ref[:, 2 * t + 1])
            assert np.all(err <= RTOL * np.maximum(mag, 1.0)), (k, t, (err / mag).max())
        assert close(tr["carrier_hz"], [r["carrier_hz"] for r in ref]) and close(tr["code_hz"], [r["code_hz"] for r in ref])
        assert close(tr["carrier_err"], [r["carrier_err"] for r in ref], scale=1.0)
        assert close(tr["code_err"], [r["code_err"] for r in ref], scale=1.0)
        if kind == 1:
            assert np.array_equal(tr["lock_state"], [r["lock_state"] for r in ref])
            assert np.array_equal(tr["track_flags"], [r["flags"] for r in ref])
    assert traj[7].tobytes() == traj[201].tobytes()      # same inputs, same kernel: bitwise the same


def test_closed_loop_across_the_ring_seam_at_25_mhz(engine):
    """A 100 ms ring (the reference's size) fed block by block: epochs that straddle the end of the ring take the
    per-sample correlator, the others the boundary variant -- the trajectory is that of the uninterrupted stream."""
    fs, prn, ms_total = 25e6, 19, 260
    spms = int(fs * 1e-3)
    n = ms_total * spms
    sat = dict(prn=prn, doppler=-3100.0, code_phase=88.8, phase=0.7, amp=9.0)
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(1)
    engine.load_gps_code(0, prn)
    engine.iq_synth([sat], fs, 14.0, 4242, 0, n)
    raw = engine.iq_download(n, 0).copy()
    rf = orc.iq_to_complex(raw)
    pb, pc, _, _ = engine.pcps([0], 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    n_code = orc.samples_per_code(fs)
    n0 = orc.required_samples(0.0, orc.CODE_RATE / fs)
    carrier, _, cur = orc.post_acquisition(0.0, 5000.0, 250.0, [int(pb[0]), int(pc[0])], 0, n_code, n0)
    loop = orc.KaplanLoop(fs, orc.gold_code(prn), KAPLAN_CFG, carrier, cur)
    ref = [loop.step(rf[loop.current_sample:loop.current_sample + loop.n]) for _ in range(240)]
    # now the same through a 100 ms ring, 80 epochs at a time
    ring = 100 * spms
    engine.iq_alloc(ring, FMT_CI8)
    engine.code_slots(1)
    engine.load_gps_code(0, prn)
    engine.iq_upload(raw[:2 * ring], 0)
    written = ring                                   # samples of the stream that have entered the ring
    st = initial_state(1, fs, carrier, cur, KAPLAN_CFG)
    cfg = loop_cfg(1, fs, KAPLAN_CFG)
    got = []
    for block in range(3):
        (st,), traj = engine.track_closed_loop([st], cfg, 80)
        got.append(traj[0])
        consumed = int(st.current_sample)            # absolute stream position (the kernel wraps ring addresses itself)
        # refill everything behind the channel: the next 80 epochs need ~80.1 ms ahead of it
        new_written = min(n, consumed + ring - spms)
        for a in range(written, new_written, spms):  # 1 ms pieces, each placed at its ring position
            b = min(a + spms, new_written)
            engine.iq_upload(raw[2 * a:2 * b], a % ring)
        written = new_written
    tr = np.concatenate(got)
    assert np.array_equal(tr["start_sample"], [r["start"] for r in ref])
    assert np.array_equal(tr["n_samples"], [r["n"] for r in ref])
    corr_ref = np.array([r["corr"] for r in ref])
    mag = np.maximum(np.hypot(corr_ref[:, 2], corr_ref[:, 3]), 1.0)
    assert np.all(np.hypot(tr["corr"][:, 2] - corr_ref[:, 2], tr["corr"][:, 3] - corr_ref[:, 3]) <= RTOL * mag)
    assert close(tr["carrier_hz"], [r["carrier_hz"] for r in ref]) and close(tr["code_hz"], [r["code_hz"] for r in ref])
    assert np.array_equal(tr["lock_state"], [r["lock_state"] for r in ref])
    wraps = sum((int(s) % ring) + int(m) > ring for s, m in zip(tr["start_sample"], tr["n_samples"]))
    assert wraps >= 2                                # the seam was crossed (twice in 240 ms)


def test_closed_loop_many_channels_and_resume(engine):
    """8 channels in one launch == each channel alone; 2 x 100 epochs == 200 epochs (state round trip)."""
    fs, ms = 4e6, 260
    rng = np.random.default_rng(20261010)
    prns = [2, 5, 9, 13, 17, 21, 26, 31]
    sats = [dict(prn=p, doppler=float(rng.integers(-16, 17) * 250.0), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=7.0) for p in prns]
    n = int(ms * fs * 1e-3)
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(8)
    for s, p in enumerate(prns):
        engine.load_gps_code(s, p)
    engine.iq_synth(sats, fs, 15.0, 4711, 0, n)
    pb, pc, pr, _ = engine.pcps(np.arange(8), 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    bins = orc.doppler_bins(5000.0, 250.0)
    n0 = orc.required_samples(0.0, orc.CODE_RATE / fs)
    cfg = loop_cfg(1, fs, KAPLAN_CFG)

    def fresh():
        out = []
        for s in range(8):
            carrier, _, cur = orc.post_acquisition(0.0, 5000.0, 250.0, [int(pb[s]), int(pc[s])], 0, 4000, n0)
            out.append(initial_state(1, fs, carrier, cur, KAPLAN_CFG, slot=s))
        return out

    all_states, all_traj = engine.track_closed_loop(fresh(), cfg, 200)
    for s in (0, 3, 7):
        one_state, one_traj = engine.track_closed_loop([fresh()[s]], cfg, 200)
        assert one_traj[0].tobytes() == all_traj[s].tobytes()          # bitwise: no cross-channel coupling
    half, t1 = engine.track_closed_loop(fresh(), cfg, 100)
    rest, t2 = engine.track_closed_loop(half, cfg, 100)
    assert np.concatenate([t1, t2], axis=1).tobytes() == all_traj.tobytes()
    assert bytes(rest[5]) == bytes(all_states[5])
    # and the loops did lock on: prompt power dominates, Doppler within a bin of the truth
    for s, sat in enumerate(sats):
        tail = all_traj[s][-50:]
        assert abs(np.mean(tail["carrier_hz"]) - sat["doppler"]) < 30.0
        p = np.hypot(tail["corr"][:, 2], tail["corr"][:, 3]).mean()
        assert p > np.hypot(tail["corr"][:, 0], tail["corr"][:, 1]).mean()   # prompt above early (0.5 chip off)


def test_closed_loop_stops_instead_of_reading_out_of_range(engine):
    from sydr_amd import SdrError
    fs = 4e6
    engine.iq_alloc(40000, FMT_CI8)
    engine.iq_upload(np.random.default_rng(1).integers(-50, 50, 80000).astype(np.int8), 0)
    engine.code_slots(1)
    engine.load_gps_code(0, 3)
    st = initial_state(0, fs, 100.0, 0, BORRE_CFG)
    st.code_step = 0.4                     # 4001 * 0.4 chips: far outside the staged replica
    with pytest.raises(SdrError, match="stopped after 0 epochs"):
        engine.track_closed_loop([st], loop_cfg(0, fs, BORRE_CFG), 10)
    cfg = loop_cfg(0, fs, BORRE_CFG)
    cfg.n_taps = 4
    with pytest.raises(SdrError):
        engine.track_closed_loop([initial_state(0, fs, 0.0, 0, BORRE_CFG)], cfg, 1)
    # per-channel outcome (sdr_track_closed_loop_ex): the runaway channel stops, its neighbour is unaffected and
    # both states come back valid
    good = initial_state(0, fs, 100.0, 0, BORRE_CFG)
    alone, traj_alone = engine.track_closed_loop([good], loop_cfg(0, fs, BORRE_CFG), 5)
    states, traj, _, done = engine.track_closed_loop_ex([st, good], loop_cfg(0, fs, BORRE_CFG), 5)
    assert list(done) == [0, 5]
    assert bytes(states[1]) == bytes(alone[0]) and traj[1].tobytes() == traj_alone[0].tobytes()
    assert states[0].n_samples == st.n_samples and states[0].code_step == 0.4 and np.all(traj[0]["n_samples"] == 0)


def test_closed_loop_with_one_configuration_per_channel(engine):
    """channelManager.addChannel takes a configuration per call: channels of one launch may run different loop
    parameters (and loop kinds).  Each must evolve exactly as it does alone with its own configuration."""
    g, fs, raw = trajectory_iq()
    engine.iq_alloc(len(raw) // 2 // 8 * 8, FMT_CI8)
    engine.iq_upload(raw[:len(raw) // 16 * 16], 0)
    engine.code_slots(1)
    engine.load_gps_code(0, 7)
    start = int(g["kaplan_acq"][5])
    kap2 = dict(KAPLAN_CFG, dll_noise_bandwidth=4.0, fll_bandwidth_pullin=60.0)
    cfgs = [loop_cfg(1, fs, KAPLAN_CFG), loop_cfg(0, fs, BORRE_CFG), loop_cfg(1, fs, kap2)]
    mk = lambda: [initial_state(1, fs, 1750.0, start, KAPLAN_CFG), initial_state(0, fs, 1750.0, start, BORRE_CFG),
                  initial_state(1, fs, 1750.0, start, kap2)]
    states, traj, _, done = engine.track_closed_loop_ex(mk(), cfgs, 120)
    assert list(done) == [120, 120, 120]
    for k in range(3):
        alone, t_alone = engine.track_closed_loop([mk()[k]], cfgs[k], 120)
        assert traj[k].tobytes() == t_alone[0].tobytes() and bytes(states[k]) == bytes(alone[0])
    assert traj[0].tobytes() != traj[2].tobytes()


# ------------------------------------------------------------------------------------------------ host plugins on the GPU
@pytest.mark.parametrize("fname", ["g6_trajectories.npz", "g6c_25mhz.npz"])
@pytest.mark.parametrize("plugin", ["borre", "kaplan"])
def test_manager_on_gpu_matches_reference_trajectory(engine, plugin, fname):
    """The drop-in ChannelManager, one tick per millisecond (acquisition, then one device step per tick from the
    channel bank), against the reference plugins' own packets -- at 4 MHz and at the headline 25 MHz."""
    from sydr_amd.channel.l1ca_borre import ChannelL1CA
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.utils.enumerations import ChannelMessage
    g, fs, raw = trajectory_iq(fname)
    mgr = ChannelManager(rf_signal(fs), engine=engine)
    cls, ini = (ChannelL1CA, BORRE_INI) if plugin == "borre" else (ChannelL1CA_Kaplan, KAPLAN_INI)
    mgr.addChannel(cls, channel_config(ini), 1)
    mgr.requestTracking(7)
    ticks = drive(mgr, raw, int(fs * 1e-3), raw.size // 2 // int(fs * 1e-3))
    acq = [p for t in ticks for p in t if p["type"] is ChannelMessage.ACQUISITION_UPDATE][0]
    trk = [p for t in ticks for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref_acq, ref = g[f"{plugin}_acq"], g[f"{plugin}_epochs"]
    assert (acq["frequency_idx"], acq["code_idx"], acq["codeOffset"]) == (int(ref_acq[0]), int(ref_acq[1]), int(ref_acq[4]))
    assert acq["peak_ratio"] == pytest.approx(ref_acq[2], rel=RTOL) and acq["correlation_map"].shape == (41, int(fs * 1e-3))
    assert len(trk) == len(ref)
    got = np.array([[p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"], p["i_late"], p["q_late"],
                     p["carrier_frequency"], p["code_frequency"]] for p in trk])
    scale = np.maximum(np.abs(ref[:, [6, 7, 8, 9, 10, 11, 15, 16]]), 1.0)
    mag = np.repeat(np.hypot(ref[:, 6:12:2], ref[:, 7:12:2]), 2, axis=1)
    scale[:, :6] = np.maximum(mag, 1.0)
    assert np.all(np.abs(got - ref[:, [6, 7, 8, 9, 10, 11, 15, 16]]) <= RTOL * scale)


def test_run_block_continues_a_per_tick_run(engine):
    """Per-tick host loop for 150 ms, then 300 epochs closed-loop on the device: same trajectory as
    the reference plugin ran per millisecond."""
    from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.utils.enumerations import ChannelMessage
    g, fs, raw = trajectory_iq()
    spms = int(fs * 1e-3)
    # a ring long enough to hold the whole record, so the block run has its samples resident
    mgr = ChannelManager(rf_signal(fs), engine=engine, ring_ms=520)
    mgr.addChannel(ChannelL1CA_Kaplan, channel_config(KAPLAN_INI), 1)
    ch = mgr.requestTracking(7)
    ticks = drive(mgr, raw, spms, 150)
    done = sum(p["type"] is ChannelMessage.TRACKING_UPDATE for t in ticks for p in t)
    mgr.addNewRFData(raw[2 * 150 * spms:2 * 410 * spms])  # 260 more ms, resident before the block run
    packets = mgr.runBlock(250)
    trk = [p for p in packets if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref = g["kaplan_epochs"][done:done + 250]
    got = np.array([[p["i_prompt"], p["q_prompt"], p["carrier_frequency"], p["code_frequency"]] for p in trk])
    want = ref[:, [8, 9, 15, 16]]
    scale = np.maximum(np.abs(want), 1.0)
    scale[:, :2] = np.maximum(np.hypot(ref[:, 8], ref[:, 9]), 1.0)[:, None]
    assert np.all(np.abs(got - want) <= RTOL * scale)
    assert ch.codeCounter == done + 250 and packets[-1]["type"] is ChannelMessage.CHANNEL_UPDATE


# ------------------------------------------------------------------------------------------------ function-level drop-ins
def test_function_level_dropins_read_like_the_reference():
    """`from sydr_amd.dsp.acquisition import PCPS ...` used exactly as sydr.dsp.* is used by the plugins."""
    from sydr_amd.dsp.acquisition import PCPS, TwoCorrelationPeakComparison
    from sydr_amd.dsp.tracking import EPL
    from sydr_amd.signal.replica import GenerateGPSGoldCode, UpsampleCode, getSamplesPerCode
    g = load_golden("g3_pcps.npz")
    fs, if_hz, rng_hz, step, coh, noncoh, n, spc = g["d_params"]
    rf = orc.iq_to_complex(g["d_iq"]).reshape(1, -1)
    code = GenerateGPSGoldCode(7)
    assert np.array_equal(code, orc.gold_code(7)) and getSamplesPerCode(fs) == int(n)
    up = UpsampleCode(code, fs)
    assert np.array_equal(up, orc.upsample_code(orc.gold_code(7), fs))
    ramp = np.arange(1023, dtype=np.float64)
    assert np.array_equal(UpsampleCode(ramp, fs), orc.upsample_index(fs).astype(float))
    codeFFT = np.conj(np.fft.fft(up))
    cmap = PCPS(rfData=rf, interFrequency=if_hz, samplingFrequency=fs, codeFFT=codeFFT, dopplerRange=rng_hz,
                dopplerStep=step, samplesPerCode=int(n), coherentIntegration=int(coh),
                nonCoherentIntegration=int(noncoh))
    idx, ratio = TwoCorrelationPeakComparison(cmap, int(n), int(spc))
    assert idx == list(g["d_peak"][0]) and ratio == pytest.approx(float(g["d_ratio"][0]), rel=RTOL)
    np.testing.assert_allclose(cmap[idx[0]], g["d_row"][0], rtol=0, atol=RTOL * g["d_row"][0].max())

    e = load_golden("g5_epl.npz")
    prn, fs, f, rc, rk, cstep, n = e["r20_params"]
    rfd = orc.iq_to_complex(e["r20_iq"]).reshape(1, -1)
    padded = np.r_[orc.gold_code(int(prn))[-1], orc.gold_code(int(prn)), orc.gold_code(int(prn))[0]]
    out = EPL(rfData=rfd, code=padded, samplingFrequency=fs, carrierFrequency=f, remainingCarrier=rc,
              remainingCode=rk, codeStep=cstep, correlatorsSpacing=tuple(e["r20_spacing"]))
    assert isinstance(out, list) and len(out) == 6
    ref = e["r20_out"]
    scale = np.repeat(np.hypot(ref[0::2], ref[1::2]), 2)
    assert np.all(np.abs(np.array(out) - ref) <= RTOL * scale)


def test_file_driven_example_with_the_reference_ini_layout(engine, tmp_path, capsys):
    """examples/run_file.py: the reference's receiver.ini / channel ini layout, an int8 I/Q file, per-tick acquisition,
    then closed-loop blocks -- every requested PRN ends up tracking on its Doppler."""
    import importlib.util
    import os
    fs, ms = 10e6, 2700      # well past the 100 ms ring's capacity times twenty: the block mode must not build a backlog
    n = int(fs * 1e-3) * ms
    sats = [dict(prn=4, doppler=2100.0, code_phase=200.5, phase=0.1, amp=9.0),
            dict(prn=9, doppler=-3300.0, code_phase=777.0, phase=0.5, amp=9.0)]
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(2)
    engine.iq_synth(sats, fs, 14.0, 555, 0, n)
    engine.iq_download(n, 0).tofile(tmp_path / "iq.bin")
    (tmp_path / "channel.ini").write_text(KAPLAN_INI)
    (tmp_path / "receiver.ini").write_text(f"""
[DEFAULT]
name = test
nb_channels = 2
ms_to_process = 2600
[RFSIGNAL]
filepath = {tmp_path / 'iq.bin'}
sampling_frequency = {fs}
intermediate_frequency = 0.0
data_size = 8
is_complex = true
[SATELLITES]
include_prn = 4,9
[CHANNELS]
gps_l1ca = ./channel.ini
""")
    spec = importlib.util.spec_from_file_location("run_file", os.path.join(os.path.dirname(__file__), "..", "examples", "run_file.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main([str(tmp_path / "receiver.ini"), "--block", "80", "--csv", str(tmp_path / "out.csv")])
    out = capsys.readouterr().out
    import re
    carriers = {int(p): float(f) for p, f in re.findall(r"G(\d+): state TRACKING, carrier ([-+0-9.]+) Hz", out)}
    assert set(carriers) == {4, 9}, out
    assert abs(carriers[4] - 2100.0) < 30.0 and abs(carriers[9] + 3300.0) < 30.0, out
    rows = np.loadtxt(tmp_path / "out.csv", delimiter=",", skiprows=1)
    assert rows.shape[0] > 2 * 2400                 # both channels, nearly all of the 2600 ms
    # the same recording through the reference's own loop (one millisecond per iteration) with the manager tracking ahead
    mod.main([str(tmp_path / "receiver.ini"), "--read-ahead", "40", "--csv", str(tmp_path / "out_ra.csv")])
    out = capsys.readouterr().out
    carriers = {int(p): float(f) for p, f in re.findall(r"G(\d+): state TRACKING, carrier ([-+0-9.]+) Hz", out)}
    assert set(carriers) == {4, 9} and abs(carriers[4] - 2100.0) < 30.0 and abs(carriers[9] + 3300.0) < 30.0, out
    rows_ra = np.loadtxt(tmp_path / "out_ra.csv", delimiter=",", skiprows=1)
    assert rows_ra.shape[0] > 2 * 2400


@pytest.mark.parametrize("kind", [0, 1])
def test_discriminator_corner_inputs_on_the_device(engine, kind):
    """The corner rows of the scalar loop math (g7: iPrompt = 0 -> atan(+-inf); 0/0 -> the NaN branch of FLL_ATAN) met
    by the DEVICE's discriminators.  A stream whose in-phase component is identically zero and whose quadrature
    component is the code itself correlates, with a zero carrier NCO, to iPrompt = 0 exactly and qPrompt = A*n; after two
    milliseconds of that the stream falls silent, so the third epoch has every correlator at exactly 0.  Epochs 1-2
    against the oracle's loops (pinned bit for bit by g6 / g7), epoch 3 against the oracle's scalar functions; the code
    loop's 0/0 leaves no next epoch length, and the device parks the channel instead of reading anywhere."""
    fs, prn, amp = 4e6, 13, 9
    n_code = orc.samples_per_code(fs)
    n = 8 * n_code
    code = orc.gold_code(prn)
    raw = np.zeros(2 * n, dtype=np.int8)
    live = 2 * n_code + 2
    idx = np.ceil(np.arange(live) * (orc.CODE_RATE / fs)).astype(int) % 1023 + 0     # chip per sample at zero code phase
    raw[1:2 * live:2] = (amp * orc.pad_code(code)[np.where(idx == 0, 1023, idx)]).astype(np.int8)   # Q = A * code, I = 0
    c = KAPLAN_CFG if kind == 1 else BORRE_CFG

    def two_epochs():
        loop = (orc.KaplanLoop if kind == 1 else orc.BorreLoop)(fs, code, c, 0.0, 0)
        rf = orc.iq_to_complex(raw)
        with np.errstate(all="ignore"):
            return [loop.step(rf[loop.current_sample:loop.current_sample + loop.n]) for _ in range(2)]
    ref = two_epochs()
    raw[2 * (ref[1]["start"] + ref[1]["n"]):] = 0       # silence from the first sample of the third epoch on
    ref = two_epochs()
    engine.iq_alloc(n, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(1)
    engine.load_gps_code(0, prn)
    assert ref[0]["corr"][2] == 0.0 and ref[0]["corr"][3] > 0.9 * amp * ref[0]["n"]     # iPrompt = 0 exactly: atan(+inf)
    if kind == 0:
        assert ref[0]["carrier_err"] == orc.pll_costas(0.0, ref[0]["corr"][3]) and abs(ref[0]["carrier_err"] - 0.25) < 1e-12
    states, traj, _, done = engine.track_closed_loop_ex([initial_state(kind, fs, 0.0, 0, c)], [loop_cfg(kind, fs, c)], 5)
    tr = traj[0]

    def same(got, want):
        got, want = float(got), float(want)
        return (np.isnan(got) and np.isnan(want)) or abs(got - want) <= RTOL * max(1.0, abs(want))
    for k, r in enumerate(ref):
        assert tr["start_sample"][k] == r["start"] and tr["n_samples"][k] == r["n"]
        assert np.all(np.abs(tr["corr"][k][:6] - np.array(r["corr"])) <= RTOL * amp * r["n"]), k
        if k == 0:
            assert tr["corr"][k][2] == 0.0                                             # exactly zero on the device too
        for field in ("dll", "pll", "carrier_err", "code_err", "carrier_hz", "code_hz"):
            assert same(tr[field][k], r[field]), (k, field, tr[field][k], r[field])
        if kind == 1:
            assert same(tr["fll"][k], r["fll"])
    # epoch 3: silence.  Every correlator is exactly 0; the carrier loop meets 0/0
    e3 = tr[2]
    assert np.all(e3["corr"][:6] == 0.0) and e3["n_samples"] > 0
    ip2, qp2 = ref[1]["corr"][2], ref[1]["corr"][3]
    with np.errstate(all="ignore"):
        z = np.float64(0.0)                                   # (NumPy scalars, as the plugins hand them over: 0/0 is NaN, not an exception)
        assert np.isnan(orc.dll_nneml(z, z, z, z)) and np.isnan(e3["dll"])
        if kind == 1:
            # Kaplan, PULL_IN, third epoch: the FLL discriminator runs; atan(0/0) - atan(q'/i') is NaN and reads as no error
            assert orc.fll_atan(z, z, ip2, qp2, 1e-3) == 0.0 and e3["fll"] == 0.0 and e3["pll"] == 0.0
            assert e3["carrier_hz"] == ref[1]["carrier_hz"] + (e3["carrier_err"]) and np.isfinite(e3["carrier_hz"])
        else:
            assert np.isnan(orc.pll_costas(z, z)) and np.isnan(e3["carrier_err"])
    # the code NCO is NaN now: the device stops this channel after the third epoch instead of reading anywhere
    assert int(done[0]) == 3
