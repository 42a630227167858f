"""GPU parity: on-device PRN replicas and the E/P/L correlator kernel vs the golden vectors
captured from the reference and vs the CPU oracle on the same seeded inputs.

Bar: chip indices bit-exact (a single wrong chip moves an accumulator by >= 1e-5 relative,
so the 1e-9 accumulator tolerance used here also proves the indices); complex accumulators
within 1e-6 relative per BASELINE.json -- the kernel is held to 1e-9 of |I+jQ| scale."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import sydr_oracle as orc
from sydr_amd.engine import FMT_CF32, FMT_CF64, FMT_CI16, FMT_CI8, make_items

pytestmark = pytest.mark.gpu

EPL_RTOL = 1e-9  # relative to the accumulator norm; north_star allows 1e-6


def assert_corr_close(got, ref, rtol=EPL_RTOL):
    got = np.asarray(got, dtype=np.float64).reshape(-1, 2)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1, 2)
    scale = np.maximum(np.hypot(ref[:, 0], ref[:, 1]), 1.0)
    err = np.hypot(got[:, 0] - ref[:, 0], got[:, 1] - ref[:, 1]) / scale
    assert err.max() <= rtol, f"max relative accumulator error {err.max():.3e}"


# ------------------------------------------------------------------------------------------------ codes
def test_gold_codes_on_device(engine):
    g = load_golden("g1_codes.npz")
    engine.code_slots(len(g["prns"]))
    for slot, (prn, chips) in enumerate(zip(g["prns"], g["chips"])):
        engine.load_gps_code(slot, int(prn))
        assert np.array_equal(engine.read_code(slot), chips), f"PRN {prn}"


@pytest.mark.parametrize("fs", [4e6, 10e6, 12e6, 25e6, 50e6])
def test_upsample_on_device(engine, fs):
    g = load_golden("g1_codes.npz")
    idx = g[f"upsample_idx_{int(fs)}"]
    engine.code_slots(2)
    engine.load_gps_code(1, 19)
    got = engine.upsample(1, fs, len(idx))
    assert np.array_equal(got, orc.gold_code(19).astype(np.int8)[idx])


def test_custom_code_roundtrip_and_validation(engine):
    from sydr_amd import SdrError
    engine.code_slots(1, max_chips=4092)
    chips = np.where(np.random.default_rng(5).random(4092) < 0.5, -1, 1).astype(np.int8)
    engine.set_code(0, chips)
    assert np.array_equal(engine.read_code(0), chips)
    bad = chips.copy()
    bad[7] = 0
    with pytest.raises(SdrError):
        engine.set_code(0, bad)
    with pytest.raises(SdrError):
        engine.load_gps_code(0, 211)


# ------------------------------------------------------------------------------------------------ EPL golden
def _run_case(engine, raw, fmt, prn, fs, f, rc, rk, step, spacing, start=0, capacity=None):
    n = raw.size if np.iscomplexobj(raw) else raw.size // 2
    cap = capacity or ((n + start + 15) // 8) * 8
    engine.iq_alloc(cap, fmt)
    engine.iq_upload(raw, start)
    engine.code_slots(4)
    engine.load_gps_code(2, int(prn))
    items = make_items(2, n, start, f, rc, rk, step)
    return engine.epl_batch(items, spacing, fs)[0]


def test_epl_reference_fixture(engine):
    g = load_golden("g5_epl.npz")
    prn, fs, f, rc, rk, step = g["fixture_params"]
    got = _run_case(engine, g["fixture_iq"], FMT_CI8, prn, fs, f, rc, rk, step, (-0.5, 0.0, 0.5))
    assert_corr_close(got, g["fixture_out"])


def test_epl_golden_cases(engine):
    g = load_golden("g5_epl.npz")
    for tag in g["cases"]:
        prn, fs, f, rc, rk, step, n = g[f"{tag}_params"]
        raw = g[f"{tag}_iq"]
        fmt = FMT_CI16 if raw.dtype == np.int16 else FMT_CI8
        got = _run_case(engine, raw, fmt, prn, fs, f, rc, rk, step, tuple(g[f"{tag}_spacing"]))
        assert_corr_close(got, g[f"{tag}_out"])


@pytest.mark.parametrize("fmt", [FMT_CI8, FMT_CI16, FMT_CF32, FMT_CF64])
def test_epl_all_ring_formats(engine, fmt):
    g = load_golden("g5_epl.npz")
    prn, fs, f, rc, rk, step, n = g["r10_params"]
    raw = g["r10_iq"]
    got = _run_case(engine, raw.astype({FMT_CI8: np.int8, FMT_CI16: np.int16, FMT_CF32: np.float32,
                                        FMT_CF64: np.float64}[fmt]), fmt, prn, fs, f, rc, rk, step,
                    tuple(g["r10_spacing"]))
    assert_corr_close(got, g["r10_out"])


def test_epl_complex128_non_integer_samples(engine):
    """Function-level drop-in takes complex128 rfData like the reference; values need not be integers."""
    rng = np.random.default_rng(77)
    n, fs = 4001, 4e6
    rf = rng.normal(0, 11.3, n) + 1j * rng.normal(0, 11.3, n)
    code = orc.pad_code(orc.gold_code(5))
    ref = orc.epl(rf, code, fs, 812.5, 2.2, 0.11, 0.25574, (-0.5, 0.0, 0.5))
    got = _run_case(engine, rf, FMT_CF64, 5, fs, 812.5, 2.2, 0.11, 0.25574, (-0.5, 0.0, 0.5))
    assert_corr_close(got, ref)


# ------------------------------------------------------------------------------------------------ EPL vs oracle
def test_epl_batch_random_items_vs_oracle(engine):
    """Many channel-epochs in one launch: misaligned starts, ring wrap, n = N-1/N/N+1, 32 PRNs."""
    rng = np.random.default_rng(20261001)
    fs, cap = 25e6, 8 * 40000
    sats = [dict(prn=p, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=6.0) for p in (3, 11, 27)]
    raw = orc.synth_iq(fs, cap, sats, 20.0, 20261002)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(32)
    for s in range(32):
        engine.load_gps_code(s, s + 1)
    rf = orc.iq_to_complex(raw)
    n_items = 96
    slot = rng.integers(0, 32, n_items)
    step = (1.023e6 + rng.uniform(-4, 4, n_items)) / fs
    rem_code = rng.uniform(0, step)
    n = np.ceil((1023 - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
    start = rng.integers(0, cap, n_items)
    start[:8] = cap - rng.integers(1, 25000, 8)  # force ring wrap
    start[8:16] = (start[8:16] // 8) * 8 + np.arange(8)  # every misalignment 0..7
    f = rng.uniform(-5000, 5000, n_items)
    rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
    items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
    got = engine.epl_batch(items, (-0.5, 0.0, 0.5), fs)
    for k in range(n_items):
        x = orc.ring_slice(rf, int(start[k]), int(n[k]))
        ref = orc.epl(x, orc.pad_code(orc.gold_code(int(slot[k]) + 1)), fs, f[k], rem_carrier[k], rem_code[k],
                      step[k], (-0.5, 0.0, 0.5))
        assert_corr_close(got[k], ref)


@pytest.mark.parametrize("spacing", [(0.0,), (-0.1, 0.1), (-0.5, 0.0, 0.5), (-1.0, -0.5, 0.0, 0.5),
                                     (-1.0, -0.5, 0.0, 0.5, 1.0), (-1.0, -0.5, -0.1, 0.0, 0.1, 0.5, 1.0),
                                     (-1.5, -1.0, -0.5, -0.1, 0.0, 0.1, 0.5, 1.0)])
def test_epl_any_number_of_taps(engine, spacing):
    """1..8 taps incl. the 5-tap VE/E/P/L/VL of BASELINE config 4 (oracle = generalised restatement)."""
    g = load_golden("g5_epl.npz")
    prn, fs, f, rc, rk, step, n = g["r11_params"]
    raw = g["r11_iq"]
    got = _run_case(engine, raw, FMT_CI8, prn, fs, f, rc, rk, step, spacing)
    ref = orc.epl(orc.iq_to_complex(raw), orc.pad_code(orc.gold_code(int(prn))), fs, f, rc, rk, step, spacing)
    assert_corr_close(got, ref)


def test_epl_tiny_and_ragged_epochs(engine):
    rng = np.random.default_rng(9)
    fs, cap = 4e6, 8192
    raw = rng.integers(-60, 60, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(1)
    engine.load_gps_code(0, 9)
    rf = orc.iq_to_complex(raw)
    ns = np.array([3, 2, 7, 8, 9, 63, 64, 65, 255, 2047, 2049, 4000])  # (n=1 crashes the reference: np.squeeze)
    starts = np.array([5, 0, 3, 8, 1, 17, 6, 2, 4095, 100, 6000, 4200])
    items = make_items(0, ns, starts, 1000.0, 0.3, 0.01, 0.25575)
    got = engine.epl_batch(items, (-0.5, 0.0, 0.5), fs)
    for k in range(len(ns)):
        x = orc.ring_slice(rf, int(starts[k]), int(ns[k]))
        ref = orc.epl(x, orc.pad_code(orc.gold_code(9)), fs, 1000.0, 0.3, 0.01, 0.25575, (-0.5, 0.0, 0.5))
        assert_corr_close(got[k], ref)


def test_epl_chip_crossings_exactly_on_samples(engine):
    """The boundary variant predicts where a lane's 16 samples change chip and re-checks predictions that
    fall on (or within 2^-16 of) a sample.  Steps that are exact binary fractions with integer / half-integer
    shifts put EVERY crossing exactly on a sample -- the tie ceil() resolves downwards; a single wrong chip
    moves an accumulator by far more than the tolerance."""
    rng = np.random.default_rng(21)
    cap = 1 << 17
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(1)
    engine.load_gps_code(0, 14)
    rf = orc.iq_to_complex(raw)
    code = orc.pad_code(orc.gold_code(14))
    for step, fs in [(1 / 32, 32 * 1.023e6), (1 / 64, 64 * 1.023e6), (3 / 64, 1.023e6 * 64 / 3), (1 / 17, 17 * 1.023e6),
                     # 8-sample groups (0.06 < step <= 0.125)
                     (1 / 16, 16 * 1.023e6), (3 / 32, 1.023e6 * 32 / 3), (1 / 8, 8 * 1.023e6), (0.1023, 10e6)]:
        n = int(1000 / step)
        for rem, start in [(0.0, 0), (0.5, 3), (0.25, 16), (1 / 32, 37), (0.999999999999, 8), (1e-12, 5)]:
            items = make_items(0, n, start, 1234.5, 0.1, rem, step)
            for spacing in [(-0.5, 0.0, 0.5), (-0.25, 0.0, 0.25), (-0.0625, 0.0, 0.0625)]:
                got = engine.epl_batch(items, spacing, fs)[0]
                ref = orc.epl(orc.ring_slice(rf, start, n), code, fs, 1234.5, 0.1, rem, step, spacing)
                assert_corr_close(got, ref)


def test_epl_high_rate_random_sweep(engine):
    """Random code steps across the boundary variant's whole range (fs 17 MHz .. 200 MHz), random phases."""
    rng = np.random.default_rng(22)
    cap = 1 << 18
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(2)
    engine.load_gps_code(1, 30)
    rf = orc.iq_to_complex(raw)
    code = orc.pad_code(orc.gold_code(30))
    for _ in range(24):
        step = float(rng.uniform(0.005, 0.06))
        n = int(rng.integers(2000, min(cap - 100, int(1022.0 / step))))
        start = int(rng.integers(0, cap))
        f, rc, rk = float(rng.uniform(-6e3, 6e3)), float(rng.uniform(0, 6.28)), float(rng.uniform(0, 1))
        fs = 1.023e6 / step
        got = engine.epl_batch(make_items(1, n, start, f, rc, rk, step), (-0.5, 0.0, 0.5), fs)[0]
        ref = orc.epl(orc.ring_slice(rf, start, n), code, fs, f, rc, rk, step, (-0.5, 0.0, 0.5))
        assert_corr_close(got, ref)


def test_epl_linearity_and_sign(engine):
    """Size-independent property: correlators are linear in the IQ (x -> -x flips every sign exactly)."""
    rng = np.random.default_rng(10)
    fs, n = 25e6, 25000
    raw = rng.integers(-100, 100, 2 * n).astype(np.int8)
    a = _run_case(engine, raw, FMT_CI8, 21, fs, -3333.0, 1.0, 0.02, 0.04092, (-0.5, 0.0, 0.5))
    b = _run_case(engine, (-raw).astype(np.int8), FMT_CI8, 21, fs, -3333.0, 1.0, 0.02, 0.04092, (-0.5, 0.0, 0.5))
    assert np.array_equal(a, -b)


def test_epl_rejects_bad_items(engine):
    from sydr_amd import SdrError
    engine.iq_alloc(8192, FMT_CI8)
    engine.code_slots(2)
    engine.load_gps_code(0, 1)
    ok = dict(code_slot=0, n_samples=4000, start_sample=0, carrier_hz=0.0, rem_carrier=0.0, rem_code=0.0,
              code_step=0.25575)
    for bad in (dict(code_slot=1), dict(code_slot=5), dict(n_samples=0), dict(n_samples=9000),
                dict(start_sample=-1), dict(code_step=0.0), dict(rem_code=float("nan")), dict(rem_code=-30.0),
                dict(code_step=0.3)):
        with pytest.raises(SdrError):
            engine.epl_batch(make_items(**{**ok, **bad}), (-0.5, 0.0, 0.5), 4e6)
    with pytest.raises(SdrError):
        engine.epl_batch(make_items(**ok), tuple(np.linspace(-1, 1, 9)), 4e6)


def test_epl_plan_is_deterministic(engine):
    """Same plan run twice -> bit-identical outputs (fixed reduction order; needed for N-GPU == 1-GPU)."""
    rng = np.random.default_rng(12)
    fs, cap = 25e6, 8 * 12500
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(rng.integers(-100, 100, 2 * cap).astype(np.int8), 0)
    engine.code_slots(4)
    for s in range(4):
        engine.load_gps_code(s, 10 + s)
    items = make_items(np.arange(64) % 4, 25000, rng.integers(0, cap, 64), 100.0, 0.0, 0.0, 0.04092)
    plan = engine.epl_plan(items, (-0.5, 0.0, 0.5), fs)
    plan.run()
    a = plan.fetch()
    plan.run()
    b = plan.fetch()
    plan.close()
    assert np.array_equal(a, b)
    # and a permutation of the items permutes the outputs bit for bit
    perm = rng.permutation(64)
    c = engine.epl_batch(items[perm], (-0.5, 0.0, 0.5), fs)
    assert np.array_equal(c, a[perm])


def test_epl_randomised_stress(engine):
    """tests/stress_epl.py: random formats, ring sizes, 1-8 taps, code steps over both correlator variants (incl. binary
    fractions whose chip switches fall exactly on samples), ring wrap -- every channel-epoch against the oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_epl", os.path.join(os.path.dirname(__file__), "stress_epl.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checked, worst = mod.run(120, 20261003, engine)
    assert checked == 120 * 24 and worst <= EPL_RTOL


# ------------------------------------------------------------------------------------------------ chip-aligned variant
@pytest.mark.parametrize("spacing", [(-0.5, 0.0, 0.5), (-0.1, 0.0, 0.1), (-0.25, 0.0, 0.25), (-1.0, -0.5, 0.0, 0.5, 1.0),
                                     (0.0,), (-0.5, 0.5), (-0.7, -0.2, 0.3, 0.9)])
def test_chip_aligned_variant_over_its_whole_range(engine, spacing):
    """The chip-aligned correlator (lanes own whole chips; correlator_chip.h) over its range of code steps -- 16 to 26
    samples per chip, every block length M, with and without the compile-time M = 24 kernel -- including steps
    whose chip switches fall EXACTLY on samples (1/20, 1/24, 1/25: the exact-evaluation path and, where taps that
    are whole chips apart jitter, the per-sample fallback), odd and even starts, short and multi-period epochs.
    Against the oracle, and against the 16-sample boundary variant of the same library."""
    rng = np.random.default_rng(len(spacing) * 1000 + int(abs(spacing[0]) * 100))
    cap = 8 * 60000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 3)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    for group in range(3):
        n_items = 48
        if group == 0:            # the headline geometry: every item 24/25 samples per chip -> the M = 24 kernel
            step = (1.023e6 + rng.uniform(-4, 4, n_items)) / 25e6
        elif group == 1:          # the whole range, mixed block lengths -> the run-time-M kernel
            step = 1.0 / rng.uniform(16.05, 25.85, n_items)
        else:                     # switches exactly on samples
            step = np.array([1 / 20, 1 / 24, 1 / 25, 1 / 16.5, 1 / 18, 0.05, 0.0625, 1 / 25.5] * 6)
        rem_code = rng.uniform(0, step)
        rem_code[:6] = [0.0, 0.5, 0.25, 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
        periods = rng.integers(1, 3, n_items)
        n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
        n[6:10] = [3, 40, 70, 26]                      # shorter than a chip, than a wave of chips
        start = rng.integers(0, cap - 60000, n_items)
        start[10:14] = [0, 1, cap - int(n[12]) - 1, 2 * (int(start[13]) // 2) + 1]
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-6000, 6000, n_items)
        rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
        items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
        got = engine.epl_batch(items, spacing, 25e6)
        engine.set_option("epl_no_chip_variant", 1)
        try:
            other = engine.epl_batch(items, spacing, 25e6)
        finally:
            engine.set_option("epl_no_chip_variant", 0)
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), 25e6, f[k], rem_carrier[k],
                                   rem_code[k], step[k], spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (group, k, step[k], n[k])
            assert np.max(np.abs(other[k] - ref) / scale) < 1e-9, (group, k)


def test_compile_time_tap_switch_variant(engine):
    """At 25 MHz with the reference's default spacing (+-0.5 chip) both outer taps switch chips 12.2 samples into the
    prompt tap's chip in EVERY block: the plan then selects the kernel that has those positions compiled in (running
    sums kept in registers, rotations in scalar registers, two half-block sums).  Checked against the oracle and against
    the run-time-position kernel of the same library; a spacing whose switch falls within the host's margin of a
    sample (12.00002 samples) must NOT select it, and epochs the kernel cannot cover (too short, block length 25)
    fall back inside the launch."""
    rng = np.random.default_rng(20261005)
    cap = 8 * 60000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 3)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    n_items = 96
    step = (1.023e6 + rng.uniform(-4, 4, n_items)) / 25e6
    rem_code = rng.uniform(0, step)
    rem_code[:6] = [0.0, 0.5, 0.25, 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
    periods = rng.integers(1, 3, n_items)
    n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
    n[6:10] = [3, 40, 70, 26]
    start = rng.integers(0, cap - 60000, n_items)
    start[10:13] = [0, 1, cap - int(n[12]) - 1]
    slot = rng.integers(0, 8, n_items)
    f = rng.uniform(-6000, 6000, n_items)
    f[13:16] = [0.0, 4.092e6, -4.092e6]                  # no carrier at all; an intermediate frequency
    rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
    items = make_items(slot, n, start, f, rem_carrier, rem_code, step)

    def run(spacing, no_split):
        engine.set_option("epl_no_split_variant", int(no_split))
        try:
            plan = engine.epl_plan(items, spacing, 25e6)
            plan.run()
            return plan.variant, plan.fetch()
        finally:
            engine.set_option("epl_no_split_variant", 0)

    def check(got, spacing):
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), 25e6, f[k], rem_carrier[k],
                                   rem_code[k], step[k], spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (k, step[k], n[k])

    half = (-0.5, 0.0, 0.5)
    v_split, got = run(half, False)
    v_dyn, other = run(half, True)
    assert v_split == 26 + 24 + 256 * 12 and v_dyn == 26 + 24
    check(got, half)
    check(other, half)
    # a switch 12.00002 samples into the block is too close to a sample for a compile-time position
    near = 12.00002 * 1.023e6 / 25e6
    v_near, got_near = run((-near, 0.0, near), False)
    assert v_near == 26 + 24
    check(got_near, (-near, 0.0, near))
    # asymmetric spacings: one tap at 12.x, the other not
    v_asym, got_asym = run((-0.5, 0.0, 0.3), False)
    assert v_asym == 26 + 24
    check(got_asym, (-0.5, 0.0, 0.3))


def test_half_chip_view_for_32_to_52_samples_per_chip(engine):
    """A BPSK code at 32-52 samples per chip (GPS L1 C/A at 40 / 50 MHz) runs on the chip-aligned correlator through
    the half-chip view of its replica (every chip twice; 2*rem_code, 2*code_step, 2*spacing -- exact scalings):
    selected by the plan, equal to the oracle and to the boundary variant, for 3 and 5 taps, one and several
    periods per epoch, n = N +- 1, switches exactly on samples, and after a slot has been re-staged."""
    rng = np.random.default_rng(20261007)
    cap = 8 * 120000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(6, 1023, 4)
    prn_of = [3 * s + 2 for s in range(6)]
    for s in range(6):
        engine.load_gps_code(s, prn_of[s])
    rf = orc.iq_to_complex(raw)

    def check(fs, spacing, periods_hi, restage=False):
        n_items = 40
        step = (1.023e6 + rng.uniform(-5, 5, n_items)) / fs
        step[:3] = [1 / 40, 1 / 48, 1 / 50][: 3]                       # chip switches exactly on samples
        step = np.clip(step, 1 / 51.5, 1 / 32.5)
        rem_code = rng.uniform(0, step)
        rem_code[3:6] = [0.0, 0.5, 1e-9]
        periods = rng.integers(1, periods_hi + 1, n_items)
        n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
        n[6:8] = [5, 90]
        start = rng.integers(0, cap - 4 * 60000, n_items)
        slot = rng.integers(0, 6, n_items)
        f = rng.uniform(-6000, 6000, n_items)
        rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
        items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
        plan = engine.epl_plan(items, spacing, fs)
        assert plan.variant >= 65536 + 26, plan.variant
        if restage:                                                    # the doubled tables follow the staged codes
            prn_of[2] = 29
            engine.load_gps_code(2, 29)
        plan.run()
        got = plan.fetch()
        engine.set_option("epl_no_half_chip_view", 1)
        try:
            other_plan = engine.epl_plan(items, spacing, fs)
            assert other_plan.variant in (8, 16)
            other_plan.run()
            other = other_plan.fetch()
        finally:
            engine.set_option("epl_no_half_chip_view", 0)
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(prn_of[int(slot[k])])), fs, f[k], rem_carrier[k],
                                   rem_code[k], step[k], spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (fs, k, step[k], n[k])
            assert np.max(np.abs(other[k] - ref) / scale) < 1e-9, (fs, k)

    check(50e6, (-0.5, 0.0, 0.5), 1)
    check(50e6, (-1.0, -0.5, 0.0, 0.5, 1.0), 4)
    check(40e6, (-0.25, 0.0, 0.25), 2, restage=True)


def test_compile_time_tap_switch_variant_randomised(engine):
    """1200 random channel-epochs through the kernel with the tap switch positions compiled in: code Doppler of +-12 Hz,
    code phases at and next to zero, one and two periods, n = N - 2 .. N + 2, carriers from 0 to +-4 MHz, phases of
    +-10 rad, full-scale int8 samples -- against the oracle."""
    rng = np.random.default_rng(99)
    cap = 8 * 200000
    raw = rng.integers(-128, 128, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 2)
    for s in range(8):
        engine.load_gps_code(s, 4 * s + 1)
    rf = orc.iq_to_complex(raw)
    half = (-0.5, 0.0, 0.5)
    for rnd in range(3):
        n_items = 400
        step = (1.023e6 + rng.uniform(-12, 12, n_items)) / 25e6
        rem = rng.uniform(0, step) * rng.choice([1.0, 1.0, 1e-6, 0.999999], n_items)
        per = rng.integers(1, 3, n_items)
        n = np.ceil((1023 * per - rem) / step).astype(np.int64) + rng.integers(-2, 3, n_items)
        start = rng.integers(0, cap - 60000, n_items)
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-20000, 20000, n_items) * rng.choice([1.0, 0.0, 200.0], n_items)
        ph = rng.uniform(-10, 10, n_items)
        plan = engine.epl_plan(make_items(slot, n, start, f, ph, rem, step), half, 25e6)
        assert plan.variant == 26 + 24 + 256 * 12
        plan.run()
        got = plan.fetch()
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(4 * int(slot[k]) + 1)), 25e6, f[k], ph[k], rem[k], step[k], half))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 60.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (rnd, k, step[k], n[k], rem[k], f[k])


def test_whole_chip_tap_variant(engine):
    """Taps whole (half-)chips apart -- the VE/E/P/L/VL set of BASELINE configs 4-5 on the half-chip view of a replica
    at 50 MHz, or three taps one chip apart at 25 MHz: no tap switches inside the anchor's block, and the plan selects
    the kernel that has that compiled in (one running sum per block, turned once, two FMAs per tap).  Against the
    oracle and against the run-time-position kernel: random code Doppler, phases at and next to zero, one to four
    periods, n = N - 2 .. N + 2, chip switches exactly on samples, short epochs."""
    rng = np.random.default_rng(20261011)
    cap = 8 * 240000
    raw = rng.integers(-128, 128, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 5)
    for s in range(8):
        engine.load_gps_code(s, 4 * s + 1)
    rf = orc.iq_to_complex(raw)
    five, three = (-1.0, -0.5, 0.0, 0.5, 1.0), (-1.0, 0.0, 1.0)

    def run(items, spacing, fs, no_split):
        engine.set_option("epl_no_split_variant", int(no_split))
        try:
            plan = engine.epl_plan(items, spacing, fs)
            plan.run()
            return plan.variant, plan.fetch()
        finally:
            engine.set_option("epl_no_split_variant", 0)

    for fs, spacing, per_hi, want in ((50e6, five, 4, 65536 + 26 + 24 + 4096), (25e6, three, 2, 26 + 24 + 4096),
                                      (50e6, (-0.5, 0.0, 0.5), 1, 65536 + 26 + 24 + 4096)):
        n_items = 120
        step = (1.023e6 + rng.uniform(-12, 12, n_items)) / fs
        rem = rng.uniform(0, step) * rng.choice([1.0, 1.0, 1e-6, 0.999999], n_items)
        rem[:2] = [0.5 * step[0], 1e-12]
        per = rng.integers(1, per_hi + 1, n_items)
        n = np.ceil((1023 * per - rem) / step).astype(np.int64) + rng.integers(-2, 3, n_items)
        n[2:6] = [3, 60, 130, 26]
        start = rng.integers(0, cap - 4 * 60000, n_items)
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-20000, 20000, n_items) * rng.choice([1.0, 0.0, 200.0], n_items)
        ph = rng.uniform(-10, 10, n_items)
        items = make_items(slot, n, start, f, ph, rem, step)
        v_ki, got = run(items, spacing, fs, False)
        v_dyn, other = run(items, spacing, fs, True)
        assert v_ki == want and v_dyn == want - 4096, (v_ki, v_dyn)
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(4 * int(slot[k]) + 1)), fs, f[k], ph[k], rem[k], step[k],
                                   spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 60.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (fs, k, step[k], n[k], rem[k], f[k])
            assert np.max(np.abs(other[k] - ref) / scale) < 1e-9, (fs, k)
    # a spacing that is not a whole number of (half-)chips keeps the run-time-position kernel
    items = make_items(0, 50000, 100, 1000.0, 0.3, 0.01, 1.023e6 / 50e6)
    v, _ = run(items, (-1.0, -0.4, 0.0, 0.4, 1.0), 50e6, False)
    assert v == 65536 + 26 + 24


def test_randomised_stress_of_the_straight_line_kernels(engine):
    """tests/stress_static.py (shortened): the compile-time tap geometries at 25 / 50 MHz with random code Doppler, phases
    at and next to zero, epochs that wrap the ring, carriers up to an intermediate frequency, and the ring rewritten in
    pieces between launches (the flipped image the kernels read must follow) -- 640 channel-epochs against the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("stress_static", os.path.join(os.path.dirname(__file__), "stress_static.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checked, worst = mod.run(8, 2026, eng=engine, n_items=80)
    assert checked == 640 and worst < 1e-9


@pytest.mark.parametrize("fs,seg", [(10e6, 5), (12e6, 6), (9.8e6, 5), (4e6, 2), (3.9e6, 2)])
def test_two_chips_per_lane_variant(engine, fs, seg):
    """Chips of 9.5 .. 10 (11.5 .. 12) samples, taps half a chip apart -- a C/A code at the reference's shipped
    10 MHz (config/receiver.ini:18-20), or at 12 MHz: a lane owns two whole chips of the prompt tap (four of 3.75 .. 4 samples
    at the reference's 4 MHz, BASELINE configs[0]),
    every tap switch at a compile-time position up to + 1 (correlator_chip2.h; the plan holds a host-made setup per item).  Random Doppler on
    code and carrier, odd and even numbers of whole chips, odd starts, short and two-period epochs, zero code phase;
    lists with a stray item the scheme does not cover (redone per sample inside the kernel) -- against the oracle, and
    against the 8-sample boundary variant of the same library."""
    rng = np.random.default_rng(int(fs) // 1000 + seg)
    cap = 1 << 18
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 2)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    spacing = (-0.5, 0.0, 0.5)
    for group in range(3):
        n_items = 130
        step = (1.023e6 + rng.uniform(-6, 6, n_items)) / fs
        rem_code = rng.uniform(0, step)
        rem_code[:6] = [0.0, 0.5 * step[1], 0.25 * step[2], 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
        periods = np.where(rng.random(n_items) < 0.2, 2, 1)
        n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
        n[6:10] = [25, 3 * 4 * seg, 700, 64 * 4 * seg + 5]            # one block, a few, a round of blocks and a bit
        if group == 1:
            n[10:14] = n[10:14] - rng.integers(5, 12, 4)             # other parities of the number of whole chips
        start = rng.integers(0, cap - 30000, n_items)
        start[14:18] = [0, 1, cap - int(n[16]) - 40, 2 * (int(start[17]) // 2) + 1]
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-6000, 6000, n_items)
        rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
        if group == 2:                                               # a stray item: twice the code rate -> not this scheme's
            step[20] *= 0.93
        items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
        plan = engine.epl_plan(items, spacing, fs)
        try:
            # <4,9,14,19> / <5,11,17,23> / (4 MHz: four chips per lane) <1,3,5,7,9,11,13,15>
            assert (plan.variant >> 13) & 3 == {5: 1, 6: 2, 2: 3}[seg], (group, plan.variant)
            plan.run()
            got = plan.fetch()
        finally:
            plan.close()
        if group == 0:                                               # a range of a fresh plan: its items' setups, not the first ones'
            part = engine.epl_plan(items, spacing, fs)
            try:
                part.run(37, 50)
                assert np.array_equal(part.fetch()[37:87], got[37:87])
            finally:
                part.close()
        engine.set_option("epl_no_two_chip_variant", 1)
        try:
            other = engine.epl_batch(items, spacing, fs)
        finally:
            engine.set_option("epl_no_two_chip_variant", 0)
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k],
                                   rem_code[k], step[k], spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (group, k, step[k], n[k], got[k], ref)
            assert np.max(np.abs(other[k] - ref) / scale) < 1e-9, (group, k)
    # chips of exactly ten samples: every switch on a sample, 2 * T = 20 -- not this scheme's; the plan keeps the boundary variant
    items = make_items(0, 5000, 3, 100.0, 0.2, 0.01, 0.1)
    plan = engine.epl_plan(items, spacing, 10.23e6)
    assert not (plan.variant >> 13) & 3
    plan.close()


@pytest.mark.parametrize("fs", [25e6, 10e6])
def test_plan_setups_made_on_the_device_equal_the_host_made_ones(engine, fs):
    """The per-item setups of the straight-line kernels are made by one launch (a thread per item) for lists of 4096 items
    and more, and by the same functions on the host for shorter ones (sdr_epl_batch): one list of 5000 items as a whole
    and in two halves -- bitwise the same accumulators -- and a sample of them against the oracle."""
    rng = np.random.default_rng(int(fs) // 1000 + 9)
    cap = 1 << 19
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    n_items = 5000
    step = (1.023e6 + rng.uniform(-6, 6, n_items)) / fs
    rem_code = rng.uniform(0, step)
    n = np.ceil((1023 - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
    start = rng.integers(0, cap - 40000, n_items)
    slot = rng.integers(0, 8, n_items)
    f = rng.uniform(-6000, 6000, n_items)
    rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
    items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
    spacing = (-0.5, 0.0, 0.5)
    whole = engine.epl_plan(items, spacing, fs)
    try:
        assert whole.variant & (3072 if fs == 25e6 else 8192)           # the straight-line kernel of that rate
        whole.run()
        got = whole.fetch()
    finally:
        whole.close()
    halves = np.concatenate([engine.epl_batch(items[:2500], spacing, fs), engine.epl_batch(items[2500:], spacing, fs)])
    assert got.tobytes() == halves.tobytes()
    rf = orc.iq_to_complex(raw)
    for k in rng.choice(n_items, 12, replace=False):
        x = orc.ring_slice(rf, int(start[k]), int(n[k]))
        ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k], rem_code[k], step[k], spacing))
        scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
        assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (k,)


def _hip():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    return hip


def test_long_lists_are_checked_on_the_device_like_short_ones_on_the_host(engine):
    """sdr_epl_plan_create checks a list of 4096 items or more with one thread per item behind its upload (the host's walk
    of 1.92 M items was most of what a plan cost) and a shorter one on the host, with ONE function for both: the same
    variant for the same kind of list at 25 / 10 / 50 / 4 MHz, the same status and message for every kind of bad item
    wherever it sits (the lowest bad index is the one reported), and a list handed over in device memory
    (sdr_epl_plan_create_dev) gives the plan of the same list in host memory, bit for bit."""
    import ctypes
    from sydr_amd import SdrError
    rng = np.random.default_rng(77)
    cap = 1 << 20
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(rng.integers(-90, 90, 2 * cap).astype(np.int8), 0)
    engine.code_slots(4)
    for s in range(3):
        engine.load_gps_code(s, s + 5)                       # (slot 3 stays empty)
    spacing = (-0.5, 0.0, 0.5)

    def some_items(fs, count):
        step = (1.023e6 + rng.uniform(-4, 4, count)) / fs
        rem_code = rng.uniform(0, step)
        n = np.ceil((1023 - rem_code) / step).astype(np.int64)
        return make_items(rng.integers(0, 3, count), n, rng.integers(0, cap - 60000, count), rng.uniform(-5000, 5000, count),
                          rng.uniform(0, 6.28, count), rem_code, step)

    for fs in (25e6, 10e6, 50e6, 4e6):
        items = some_items(fs, 6000)
        long_plan, short_plan = engine.epl_plan(items, spacing, fs), engine.epl_plan(items[:3000], spacing, fs)
        try:
            assert long_plan.variant == short_plan.variant, fs
            long_plan.run()
            short_plan.run()
            assert long_plan.fetch()[:3000].tobytes() == short_plan.fetch().tobytes()
        finally:
            long_plan.close()
            short_plan.close()
    fs = 25e6
    good = some_items(fs, 5000)
    for field, value in (("code_slot", 3), ("code_slot", 9), ("code_slot", -1), ("n_samples", 0), ("n_samples", cap + 1),
                         ("start_sample", -5), ("code_step", 0.0), ("code_step", float("nan")), ("rem_code", float("inf")),
                         ("carrier_hz", float("nan")), ("rem_code", -30.0), ("code_step", 0.3)):
        for where in (0, 2500, 4999):
            bad = good.copy()
            bad[field][where] = value
            bad[field][min(4999, where + 7)] = value           # (a second one further on: the first is reported)
            with pytest.raises(SdrError) as long_err:
                engine.epl_plan(bad, spacing, fs)
            with pytest.raises(SdrError) as short_err:
                engine.epl_plan(bad[where:where + 1], spacing, fs)
            assert f"item {where}:" in str(long_err.value), (field, value, where, str(long_err.value))
            assert str(long_err.value).replace(f"item {where}:", "item 0:") == str(short_err.value)
    # the list in device memory
    hip = _hip()
    items = some_items(fs, 5000)
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), items.nbytes) == 0
    try:
        assert hip.hipMemcpy(d, items.ctypes.data_as(ctypes.c_void_p), items.nbytes, 1) == 0
        on_dev, on_host = engine.epl_plan_dev(d.value, len(items), spacing, fs), engine.epl_plan(items, spacing, fs)
        try:
            assert on_dev.variant == on_host.variant and on_dev.variant & 3072
            on_dev.run()
            on_host.run()
            assert on_dev.fetch().tobytes() == on_host.fetch().tobytes()
        finally:
            on_dev.close()
            on_host.close()
        bad = items.copy()
        bad["n_samples"][1234] = 0
        assert hip.hipMemcpy(d, bad.ctypes.data_as(ctypes.c_void_p), bad.nbytes, 1) == 0
        with pytest.raises(SdrError) as err:
            engine.epl_plan_dev(d.value, len(bad), spacing, fs)
        assert "item 1234:" in str(err.value)
        short = some_items(fs, 100)                                   # (a short list on the device goes the same way)
        assert hip.hipMemcpy(d, short.ctypes.data_as(ctypes.c_void_p), short.nbytes, 1) == 0
        on_dev = engine.epl_plan_dev(d.value, 100, spacing, fs)
        try:
            on_dev.run()
            assert on_dev.fetch().tobytes() == engine.epl_batch(short, spacing, fs).tobytes()
        finally:
            on_dev.close()
    finally:
        hip.hipFree(d)


def test_compile_time_tap_switch_variant_at_20_mhz(engine):
    """The straight-line kernel is written for a block length KM and a switch position KS = floor(KM / 2); besides 24 / 12
    (25 MHz) it is instantiated for 19 / 9: 20 MHz, 19.55 samples per chip, the outer taps switching 9.8 samples into the
    prompt tap's chip.  Long and short lists (setups by the launch / on the host) against the oracle and against the
    run-time-position kernel; odd epochs (short, n = N +- 1, an intermediate frequency, exact-integer phases) fall back
    inside the launch; a list that spans two block lengths (19 and 20 samples per chip) must not select it."""
    rng = np.random.default_rng(20261020)
    fs = 20e6
    cap = 8 * 50000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 3)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    half = (-0.5, 0.0, 0.5)
    for n_items in (96, 4200):
        step = (1.023e6 + rng.uniform(-4, 4, n_items)) / fs
        rem_code = rng.uniform(0, step)
        rem_code[:6] = [0.0, 0.5, 0.25, 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
        periods = rng.integers(1, 3, n_items)
        n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
        n[6:10] = [3, 40, 70, 21]
        start = rng.integers(0, cap - 50000, n_items)
        start[10:13] = [0, 1, cap - int(n[12]) - 1]
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-6000, 6000, n_items)
        f[13:16] = [0.0, 4.092e6, -4.092e6]
        rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
        items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
        got = {}
        for no_split in (0, 1):
            engine.set_option("epl_no_split_variant", no_split)
            try:
                plan = engine.epl_plan(items, half, fs)
                plan.run()
                got[no_split] = (plan.variant, plan.fetch())
                plan.close()
            finally:
                engine.set_option("epl_no_split_variant", 0)
        assert got[0][0] == 26 + 19 + 256 * 9 and got[1][0] == 26
        for k in (range(n_items) if n_items < 1000 else rng.choice(n_items, 60, replace=False)):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k], rem_code[k], step[k], half))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            for no_split in (0, 1):
                assert np.max(np.abs(got[no_split][1][k] - ref) / scale) < 1e-9, (n_items, k, no_split, step[k], n[k])
    # 20.46 MHz is exactly 20.0 samples per chip: Doppler decides between 19 and 20 -- not this kernel's list
    items2 = items[:200].copy()
    items2["code_step"] = (1.023e6 + rng.uniform(-4, 4, 200)) / 20.46e6
    items2["n_samples"] = np.ceil((1023 - items2["rem_code"]) / items2["code_step"]).astype(np.int64)
    plan = engine.epl_plan(items2, half, 20.46e6)
    assert plan.variant == 26
    plan.close()


@pytest.mark.parametrize("km", [16, 17, 18, 20, 21, 22, 23, 25])
def test_straight_line_kernels_of_every_block_length(engine, km):
    """The straight-line kernel exists for every block length KM = 16 .. 25 (epl_straight.hip carries the ones epl.hip does
    not): a list at KM.5 samples per chip gets `epl_kernel<.., KM, .., KM / 2>` at +-0.5 chip spacing (plan variant 26 + KM +
    256 * (KM / 2)), the whole-chip-tap form at +-1 chip (26 + KM + 4096) -- on the half-chip view at twice the rate too, with
    three taps and with five -- and the block length alone (26 + KM) at any other spacing.
    Long and short lists (setups by a launch / on the host) against the oracle and the run-time-position kernel, odd epochs
    included (they fall back inside the launch)."""
    rng = np.random.default_rng(7000 + km)
    fs = 1.023e6 * (km + 0.5)
    n_code = int(fs * 1e-3) + 1
    cap = 8 * 2 * n_code
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 3)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    # (+-0.25 chip -- a narrow correlator: neither switch position nor whole chips, the block length alone is compiled in;
    # at 16.x samples per chip that is the two-length kernel's list)
    for spacing, want in (((-0.5, 0.0, 0.5), 26 + km + 256 * (km // 2)), ((-1.0, 0.0, 1.0), 26 + km + 4096),
                          ((-0.25, 0.0, 0.25), 26 + km)):
        for n_items in (90, 4200):
            step = (1.023e6 + rng.uniform(-4, 4, n_items)) / fs
            rem_code = rng.uniform(0, step)
            rem_code[:6] = [0.0, 0.5, 0.25, 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
            periods = rng.integers(1, 3, n_items)
            n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
            n[6:10] = [3, 40, 70, 21]
            start = rng.integers(0, cap - 2 * n_code - 64, n_items)
            start[10:13] = [0, 1, cap - int(n[12]) - 1]
            slot = rng.integers(0, 8, n_items)
            f = rng.uniform(-6000, 6000, n_items)
            f[13:16] = [0.0, 4.092e6, -4.092e6]
            rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
            items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
            got = {}
            for no_split in (0, 1):
                engine.set_option("epl_no_split_variant", no_split)
                try:
                    plan = engine.epl_plan(items, spacing, fs)
                    plan.run()
                    got[no_split] = (plan.variant, plan.fetch())
                    plan.close()
                finally:
                    engine.set_option("epl_no_split_variant", 0)
            assert got[0][0] == want and got[1][0] == 26, (spacing, n_items, got[0][0], got[1][0])
            for k in (range(n_items) if n_items < 1000 else rng.choice(n_items, 50, replace=False)):
                x = orc.ring_slice(rf, int(start[k]), int(n[k]))
                ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k], rem_code[k], step[k], spacing))
                scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
                for no_split in (0, 1):
                    assert np.max(np.abs(got[no_split][1][k] - ref) / scale) < 1e-9, (spacing, n_items, k, no_split, step[k], n[k])
    # the half-chip view: twice the rate, +-0.5 chip = +-1 half chip -> the whole-chip-tap form of the same block length
    fs2 = 2 * fs
    n_items = 300
    step = (1.023e6 + rng.uniform(-4, 4, n_items)) / fs2
    rem_code = rng.uniform(0, step)
    n = np.ceil((1023 - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
    start = rng.integers(0, cap - 2 * n_code - 64, n_items)
    slot = rng.integers(0, 8, n_items)
    f = rng.uniform(-6000, 6000, n_items)
    rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
    items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
    for spacing in ((-0.5, 0.0, 0.5), (-1.0, -0.5, 0.0, 0.5, 1.0)):        # three taps, and five: VE / E / P / L / VL
        plan = engine.epl_plan(items, spacing, fs2)
        plan.run()
        got2 = plan.fetch()
        assert plan.variant == 65536 + 26 + km + 4096, (spacing, plan.variant)
        plan.close()
        for k in rng.choice(n_items, 40, replace=False):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs2, f[k], rem_carrier[k], rem_code[k], step[k], spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            assert np.max(np.abs(got2[k] - ref) / scale) < 1e-9, (spacing, k, step[k], n[k])


def test_two_block_lengths_in_one_kernel_at_16_368_mhz(engine):
    """16.368 MHz is exactly 16.0 samples per chip: an epoch's chips hold 15.x or 16.x samples by the sign of its code
    Doppler, so one list holds both.  The kernel with BOTH block lengths compiled in (`sdr_epl_plan_variant` = 26 + 16) takes
    such lists -- every epoch the body of its own length, tap positions at run time (at +-0.5 chip the taps switch ON a
    sample: nothing to compile in).  Long and short lists against the oracle and against the run-time-position kernel;
    all-positive and all-negative Doppler lists take it too; odd epochs fall back inside the launch; 17.x samples per chip
    in the list: not this kernel."""
    rng = np.random.default_rng(16368)
    fs = 16.368e6
    cap = 8 * 40000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 3)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    half = (-0.5, 0.0, 0.5)
    for n_items, sign in ((96, 0), (4200, 0), (300, 1), (300, -1)):
        dop = rng.uniform(-4, 4, n_items) if sign == 0 else sign * rng.uniform(0.01, 4, n_items)
        step = (1.023e6 + dop) / fs
        rem_code = rng.uniform(0, step)
        rem_code[:6] = [0.0, 0.5, 0.25, 1e-9, step[4] / 2, step[5] * (1 - 1e-12)]
        periods = rng.integers(1, 3, n_items)
        n = np.ceil((1023 * periods - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
        n[6:10] = [3, 40, 70, 21]
        start = rng.integers(0, cap - 40000, n_items)
        start[10:13] = [0, 1, cap - int(n[12]) - 1]
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-6000, 6000, n_items)
        f[13:16] = [0.0, 4.092e6, -4.092e6]
        rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
        items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
        got = {}
        for no_split in (0, 1):
            engine.set_option("epl_no_split_variant", no_split)
            try:
                plan = engine.epl_plan(items, half, fs)
                plan.run()
                got[no_split] = (plan.variant, plan.fetch())
                plan.close()
            finally:
                engine.set_option("epl_no_split_variant", 0)
        assert got[0][0] == 26 + 16 and got[1][0] == 26, (n_items, sign, got[0][0], got[1][0])
        if sign == 0:
            assert (step > 1 / 16).any() and (step < 1 / 16).any()        # both block lengths in the list
        for k in (range(n_items) if n_items < 1000 else rng.choice(n_items, 60, replace=False)):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k], rem_code[k], step[k], half))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
            for no_split in (0, 1):
                assert np.max(np.abs(got[no_split][1][k] - ref) / scale) < 1e-9, (n_items, k, no_split, step[k], n[k])
    # 17.4 MHz: 17.0 samples per chip -- 16.x and 17.x: outside the pair compiled in
    items2 = items[:200].copy()
    items2["code_step"] = (1.023e6 + rng.uniform(-4, 4, 200)) / 17.391e6
    items2["n_samples"] = np.ceil((1023 - items2["rem_code"]) / items2["code_step"]).astype(np.int64)
    plan = engine.epl_plan(items2, half, 17.391e6)
    assert plan.variant == 26
    plan.close()


@pytest.mark.parametrize("n_items", [150, 4300])
def test_whole_chip_tap_variant_on_the_half_chip_view_at_32_mhz(engine, n_items):
    """31-32.7 MHz: a chip holds 30.3-32 samples, a half chip 15.x -- the half-chip view with the whole-chip-tap kernel
    compiled for blocks of 15 / 16 samples (KM = 15; 16-sample boundary groups before: 0.47 of the roof against 0.68).
    Against the oracle and against the boundary variant, short and long lists, odd epochs included."""
    rng = np.random.default_rng(3200 + n_items)
    fs = 32e6
    cap = 8 * 70000
    raw = rng.integers(-100, 100, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(8, 1023, 2)
    for s in range(8):
        engine.load_gps_code(s, 3 * s + 2)
    rf = orc.iq_to_complex(raw)
    half = (-0.5, 0.0, 0.5)
    step = (1.023e6 + rng.uniform(-4, 4, n_items)) / fs
    rem_code = rng.uniform(0, step)
    rem_code[:4] = [0.0, 0.5, 0.25, 1e-9]
    n = np.ceil((1023 - rem_code) / step).astype(np.int64) + rng.integers(-1, 2, n_items)
    n[4:8] = [3, 40, 70, 33]
    start = rng.integers(0, cap - 70000, n_items)
    slot = rng.integers(0, 8, n_items)
    f = rng.uniform(-6000, 6000, n_items)
    rem_carrier = rng.uniform(0, 2 * np.pi, n_items)
    items = make_items(slot, n, start, f, rem_carrier, rem_code, step)
    got = {}
    for plain in (0, 1):
        engine.set_option("epl_no_half_chip_view", plain)
        try:
            plan = engine.epl_plan(items, half, fs)
            plan.run()
            got[plain] = (plan.variant, plan.fetch())
            plan.close()
        finally:
            engine.set_option("epl_no_half_chip_view", 0)
    assert got[0][0] == 65536 + 26 + 15 + 4096 and got[1][0] == 16
    for k in (range(n_items) if n_items < 1000 else rng.choice(n_items, 60, replace=False)):
        x = orc.ring_slice(rf, int(start[k]), int(n[k]))
        ref = np.array(orc.epl(x, orc.pad_code(orc.gold_code(3 * int(slot[k]) + 2)), fs, f[k], rem_carrier[k], rem_code[k], step[k], half))
        scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n[k])) * 50.0), 2)
        for plain in (0, 1):
            assert np.max(np.abs(got[plain][1][k] - ref) / scale) < 1e-9, (k, plain, step[k], n[k])
