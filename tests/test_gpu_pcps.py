"""GPU parity: PCPS acquisition + two-peak comparison vs golden vectors captured from the
reference (peak indices bit-exact, map values / ratio to 1e-9 relative) and vs the oracle."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import sydr_oracle as orc
from sydr_amd.engine import FMT_CF64, FMT_CI8

pytestmark = pytest.mark.gpu

MAP_RTOL = 1e-9


def _stage(engine, raw, prns, fmt=FMT_CI8, start=0, capacity=None):
    n = raw.size // 2
    cap = capacity or ((n + start + 7) // 8) * 8
    engine.iq_alloc(cap, fmt)
    engine.iq_upload(raw, start)
    engine.code_slots(len(prns))
    for s, p in enumerate(prns):
        engine.load_gps_code(s, int(p))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_pcps_golden(engine, tag):
    g = load_golden("g3_pcps.npz")
    fs, if_hz, rng_hz, step, coh, noncoh, n, spc = g[f"{tag}_params"]
    prns = g[f"{tag}_prns"]
    _stage(engine, g[f"{tag}_iq"], prns)
    pb, pc, pr, cmap = engine.pcps(np.arange(len(prns)), 0, fs, if_hz, rng_hz, step, int(coh), int(noncoh),
                                   want_map=True)
    assert cmap.shape == (len(prns), len(orc.doppler_bins(rng_hz, step)), int(n))
    for k in range(len(prns)):
        assert [int(pb[k]), int(pc[k])] == list(g[f"{tag}_peak"][k]), f"case {tag} PRN {prns[k]}"
        scale = g[f"{tag}_row"][k].max()
        np.testing.assert_allclose(cmap[k, pb[k]], g[f"{tag}_row"][k], rtol=0, atol=MAP_RTOL * scale)
        np.testing.assert_allclose(cmap[k, :, pc[k]], g[f"{tag}_col"][k], rtol=0, atol=MAP_RTOL * scale)
        np.testing.assert_allclose(cmap[k].sum(axis=1), g[f"{tag}_binsum"][k], rtol=MAP_RTOL)
        assert pr[k] == pytest.approx(float(g[f"{tag}_ratio"][k]), rel=1e-9)
    # indices-only call (no map transfer) returns the same answer
    pb2, pc2, pr2, none = engine.pcps(np.arange(len(prns)), 0, fs, if_hz, rng_hz, step, int(coh), int(noncoh))
    # (same indices; the ratio to rounding -- at 25 MHz the map-free sweep runs the register-resident 125 x 200 kernels,
    # whose transforms are ordered differently from the general ones behind the map)
    assert none is None and np.array_equal(pb, pb2) and np.array_equal(pc, pc2)
    np.testing.assert_allclose(pr2, pr, rtol=1e-12, atol=0)


def test_two_peak_compare_edge_cases(engine):
    g = load_golden("g4_peaks.npz")
    n, bins, spc = (int(v) for v in g["geometry"])
    for m, idx, ratio in zip(g["maps"], g["idx"], g["ratio"]):
        got_idx, got_ratio = engine.two_peak_compare(m, spc)
        assert got_idx == list(idx)
        assert got_ratio == ratio  # a single fp64 division of two map entries: exact


def test_pcps_full_map_vs_oracle_with_ring_offset(engine):
    """Whole map (every bin) against the oracle; slice starts mid-ring and wraps."""
    fs, n = 4e6, 4000
    sats = [dict(prn=14, doppler=-2250.0, code_phase=512.3, phase=0.2, amp=7.0)]
    raw = orc.synth_iq(fs, 2 * n, sats, 20.0, 4242)
    cap, start = 8000, 6104
    ring = np.zeros(2 * cap, dtype=np.int8)
    lin = np.arange(start, start + 2 * n) % cap
    ring[2 * lin] = raw[0::2]
    ring[2 * lin + 1] = raw[1::2]
    _stage(engine, ring, [14, 2], capacity=cap)
    pb, pc, pr, cmap = engine.pcps([0, 1], start, fs, 0.0, 5000.0, 250.0, 1, 2, want_map=True)
    rf = orc.iq_to_complex(raw).reshape(1, -1)
    for k, prn in enumerate((14, 2)):
        ref = orc.pcps_map(rf, 0.0, fs, orc.code_spectrum(orc.gold_code(prn), fs), 5000.0, 250.0, n, 1, 2)
        np.testing.assert_allclose(cmap[k], ref, rtol=0, atol=MAP_RTOL * ref.max())
        peak, ratio = orc.two_peak_compare(ref, n, 4)
        assert [int(pb[k]), int(pc[k])] == peak
        assert pr[k] == pytest.approx(ratio, rel=1e-9)


def test_pcps_complex128_ring(engine):
    rng = np.random.default_rng(3)
    fs, n = 4e6, 4000
    rf = rng.normal(0, 9.0, n) + 1j * rng.normal(0, 9.0, n)
    engine.iq_alloc(n, FMT_CF64)
    engine.iq_upload(rf, 0)
    engine.code_slots(1)
    engine.load_gps_code(0, 8)
    pb, pc, pr, cmap = engine.pcps([0], 0, fs, 2000.0, 3000.0, 500.0, 1, 1, want_map=True)
    ref = orc.pcps_map(rf.reshape(1, -1), 2000.0, fs, orc.code_spectrum(orc.gold_code(8), fs), 3000.0, 500.0, n)
    np.testing.assert_allclose(cmap[0], ref, rtol=0, atol=MAP_RTOL * ref.max())
    peak, _ = orc.two_peak_compare(ref, n, 4)
    assert [int(pb[0]), int(pc[0])] == peak


@pytest.mark.parametrize("fs", [2.046e6, 3e6, 6e6, 7e6, 12e6])
def test_pcps_other_transform_sizes(engine, fs):
    """N = 2046 (2*3*11*31), 3000, 6000, 7000 (7 as a generic radix), 12000 (radix 3)."""
    n = orc.samples_per_code(fs)
    sats = [dict(prn=6, doppler=750.0, code_phase=100.5, phase=0.0, amp=9.0)]
    raw = orc.synth_iq(fs, n, sats, 15.0, 99)
    _stage(engine, raw, [6])
    pb, pc, pr, cmap = engine.pcps([0], 0, fs, 0.0, 1000.0, 250.0, 1, 1, want_map=True)
    ref = orc.pcps_map(orc.iq_to_complex(raw).reshape(1, -1), 0.0, fs, orc.code_spectrum(orc.gold_code(6), fs),
                       1000.0, 250.0, n)
    np.testing.assert_allclose(cmap[0], ref, rtol=0, atol=MAP_RTOL * ref.max())
    peak, ratio = orc.two_peak_compare(ref, n, round(fs / orc.CODE_RATE))
    assert [int(pb[0]), int(pc[0])] == peak


def test_pcps_32_prns_25mhz_properties(engine):
    """BASELINE config 2 size (32 PRNs, 25 MHz, 41 bins) through size-independent properties:
    every present PRN is found at its true Doppler bin and code delay; the search is linear in
    IQ scale (x -> 2x doubles the map, same indices, same ratio)."""
    fs, n = 25e6, 25000
    rng = np.random.default_rng(20260002)
    prns = list(range(1, 33))
    sats = [dict(prn=p, doppler=float(rng.integers(-18, 19) * 250.0), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=3.0) for p in prns]
    engine.iq_alloc(n, FMT_CI8)
    engine.code_slots(32)
    for s, p in enumerate(prns):
        engine.load_gps_code(s, p)
    engine.iq_synth(sats, fs, 12.0, 777, 0, n)
    raw = engine.iq_download(n, 0)
    pb, pc, pr, _ = engine.pcps(np.arange(32), 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    bins = orc.doppler_bins(5000.0, 250.0)
    for k, s in enumerate(sats):
        # estimated Doppler = -bin (acquisition.py:42; kaplan:222-224)
        # (a neighbouring 250 Hz bin loses < 1 dB over 1 ms, so noise may pick it: allow one bin)
        assert abs(-bins[pb[k]] - s["doppler"]) <= 250.0, (k, pb[k])
        expect = ((1023.0 - s["code_phase"]) / (1.023e6 / fs)) % n
        assert min(abs(pc[k] - expect), n - abs(pc[k] - expect)) <= 1.5, (k, pc[k], expect)
        assert pr[k] > 1.5
    # three PRNs cross-checked in full against the oracle on the downloaded bytes
    rf = orc.iq_to_complex(raw).reshape(1, -1)
    for k in (0, 13, 31):
        ref = orc.pcps_map(rf, 0.0, fs, orc.code_spectrum(orc.gold_code(prns[k]), fs), 5000.0, 250.0, n)
        peak, ratio = orc.two_peak_compare(ref, n, 24)
        assert [int(pb[k]), int(pc[k])] == peak
        assert pr[k] == pytest.approx(ratio, rel=1e-9)
    half = (raw // 2).astype(np.int8)
    engine.iq_upload(half, 0)
    b1, c1, r1, _ = engine.pcps(np.arange(32), 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    engine.iq_upload((2 * half).astype(np.int8), 0)
    b2, c2, r2, _ = engine.pcps(np.arange(32), 0, fs, 0.0, 5000.0, 250.0, 1, 1)
    assert np.array_equal(b1, b2) and np.array_equal(c1, c2)
    np.testing.assert_allclose(r1, r2, rtol=1e-12)


def test_pcps_rejects_bad_requests(engine):
    from sydr_amd import SdrError
    engine.iq_alloc(4000, FMT_CI8)
    engine.code_slots(2)
    engine.load_gps_code(0, 1)
    with pytest.raises(SdrError):
        engine.pcps([1], 0, 4e6, 0.0, 5000.0, 250.0)       # slot not staged
    with pytest.raises(SdrError):
        engine.pcps([0], 0, 4e6, 0.0, 5000.0, 250.0, 2, 1)  # needs 8000 samples
    with pytest.raises(SdrError):
        engine.pcps([0], 0, 4e6, 0.0, 5000.0, 0.0)          # empty grid


@pytest.mark.parametrize("n_code,coh,noncoh", [(4079, 1, 1), (4079, 2, 2), (10007, 1, 1), (2 * 4099, 1, 2)])
def test_pcps_code_lengths_with_large_prime_factors(engine, n_code, coh, noncoh):
    """fs such that a code period is a prime (or 2 x prime) number of samples: no mixed-radix plan exists, NumPy uses
    Bluestein, and so does the library (chirp-z over the next 2^a 3^b 5^c length).  Same peaks, same map."""
    fs = n_code * 1000.0
    total = n_code * coh * noncoh + 8
    cap = (total + 7) // 8 * 8
    engine.iq_alloc(cap, FMT_CI8)
    engine.code_slots(2)
    engine.load_gps_code(0, 9)
    engine.load_gps_code(1, 23)      # absent
    engine.iq_synth([dict(prn=9, doppler=-1250.0, code_phase=611.7, phase=0.4, amp=8.0)], fs, 12.0, 77, 0, cap)
    rf = orc.iq_to_complex(engine.iq_download(cap, 0))
    pb, pc, pr, cmap = engine.pcps([0, 1], 3, fs, 0.0, 5000.0, 250.0, coh, noncoh, want_map=True)
    x = rf[3:3 + n_code * coh * noncoh].reshape(1, -1)
    for s, prn in enumerate((9, 23)):
        m = orc.pcps_map(x, 0.0, fs, orc.code_spectrum(orc.gold_code(prn), fs), 5000.0, 250.0, n_code, coh, noncoh)
        peak, ratio = orc.two_peak_compare(m, n_code, round(fs / orc.CODE_RATE))
        assert [int(pb[s]), int(pc[s])] == peak
        assert pr[s] == pytest.approx(ratio, rel=1e-9)
        assert np.max(np.abs(cmap[s] - m)) <= 1e-9 * m.max()
    assert pr[0] > 2.5 > pr[1]


def test_pcps_randomised_stress(engine):
    """tests/stress_pcps.py: random code lengths in samples (four-step, per-pass and generic-radix transforms), IF,
    Doppler grids, integrations, present / absent PRNs -- peak indices identical, maps within 1e-9 of the oracle's."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_pcps", os.path.join(os.path.dirname(__file__), "stress_pcps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checked, worst, refused = mod.run(25, 20261003, engine)
    assert checked >= 45 and worst <= 1e-9


def test_map_free_search_equals_the_materialised_one(engine):
    """Indices + ratio without writing the map (running maximum in the inverse row kernel, winning rows recomputed
    alone) == the same search with the map written and re-read -- ties, edge windows and absent PRNs included (every
    PRN x 4/10/25 MHz x two Doppler grids), with the register-resident and with the general kernels."""
    rng = np.random.default_rng(31)
    for fs, grid in ((4e6, (5000.0, 250.0)), (10e6, (5000.0, 300.0)), (25e6, (5000.0, 250.0)), (25e6, (2000.0, 100.0))):
        n = orc.samples_per_code(fs)
        sats = [dict(prn=int(p), doppler=float(rng.uniform(-4000, 4000)), code_phase=float(rng.uniform(0, 1023)),
                     phase=0.2, amp=7.0) for p in rng.choice(np.arange(1, 33), 9, replace=False)]
        engine.iq_alloc(n, FMT_CI8)
        engine.code_slots(32)
        for s in range(32):
            engine.load_gps_code(s, s + 1)
        engine.iq_synth(sats, fs, 15.0, int(fs) % 1000 + 17, 0, n)
        slots = np.arange(32)
        # N = N1 x 200: every search runs the register-resident kernels (pcps_fast.h, pcps_fastn.h), whose transforms
        # are ordered differently from the general ones -- same indices, map and ratio to rounding.  Within a family
        # of kernels the map-free search and the two that write the map agree bit for bit on the indices; on the
        # ratio too with the general kernels (the second peak of the map-free search is theirs in both families).
        found = {}
        for general in (0, 1):
            engine.set_option("pcps_general_kernels", general)
            try:
                pb, pc, pr, none = engine.pcps(slots, 0, fs, 0.0, grid[0], grid[1], 1, 1)
                assert none is None
                engine.set_option("pcps_materialise_map", 1)
                try:
                    qb, qc, qr, _ = engine.pcps(slots, 0, fs, 0.0, grid[0], grid[1], 1, 1)
                finally:
                    engine.set_option("pcps_materialise_map", 0)
                mb, mc, mr, cmap = engine.pcps(slots, 0, fs, 0.0, grid[0], grid[1], 1, 1, want_map=True)
            finally:
                engine.set_option("pcps_general_kernels", 0)
            assert np.array_equal(pb, qb) and np.array_equal(pc, qc) and np.array_equal(pb, mb) and np.array_equal(pc, mc)
            assert qr.tobytes() == mr.tobytes()
            if general:
                assert pr.tobytes() == qr.tobytes()
            else:
                np.testing.assert_allclose(pr, qr, rtol=1e-12, atol=0)
            found[general] = (pb, pc, pr, cmap)
        assert np.array_equal(found[0][0], found[1][0]) and np.array_equal(found[0][1], found[1][1])
        np.testing.assert_allclose(found[0][2], found[1][2], rtol=1e-12, atol=0)
        np.testing.assert_allclose(found[0][3], found[1][3], rtol=1e-12, atol=1e-12 * float(found[1][3].max()))
        for p in range(32):                                   # and against NumPy's own argmax of the returned map
            top = np.unravel_index(np.argmax(cmap[p]), cmap[p].shape)
            assert (int(pb[p]), int(pc[p])) == (int(top[0]), int(top[1]))
    # constant stream: every map value of a row ties -> first index must win in both paths
    engine.iq_alloc(4000, FMT_CI8)
    engine.iq_upload(np.zeros(8000, dtype=np.int8), 0)
    pb, pc, pr, _ = engine.pcps(np.arange(4), 0, 4e6, 0.0, 5000.0, 250.0, 1, 1)
    assert np.all(pb == 0) and np.all(pc == 0)


def test_cached_code_spectra_follow_the_staged_codes(engine):
    """conj(fft(code)) is kept between searches of the same staged codes (the reference rebuilds it every time,
    kaplan:184-185); re-staging a slot, another slot list or another rate must not reuse it."""
    fs = 4e6
    n = orc.samples_per_code(fs)
    sats = [dict(prn=5, doppler=1000.0, code_phase=100.5, phase=0.0, amp=9.0),
            dict(prn=9, doppler=-2000.0, code_phase=700.25, phase=0.3, amp=9.0)]
    raw = orc.synth_iq(fs, n, sats, 12.0, 77)
    _stage(engine, raw, [5, 9, 11])
    first = engine.pcps([0, 1, 2], 0, fs, 0.0, 5000.0, 250.0)
    again = engine.pcps([0, 1, 2], 0, fs, 0.0, 5000.0, 250.0)            # served from the cached spectra
    assert all(np.array_equal(a, b) for a, b in zip(first[:3], again[:3]))
    engine.load_gps_code(2, 9)                                            # slot 2 now carries PRN 9
    restaged = engine.pcps([0, 1, 2], 0, fs, 0.0, 5000.0, 250.0)
    assert (restaged[0][2], restaged[1][2]) == (first[0][1], first[1][1]) and restaged[2][2] == first[2][1]
    swapped = engine.pcps([1, 0], 0, fs, 0.0, 5000.0, 250.0)             # another slot list
    assert (swapped[0][0], swapped[1][0], swapped[2][0]) == (first[0][1], first[1][1], first[2][1])
    rf = orc.iq_to_complex(raw)
    cmap = orc.pcps_map(rf.reshape(1, -1), 0.0, fs, orc.code_spectrum(orc.gold_code(9), fs), 5000.0, 250.0, n)
    peak, ratio = orc.two_peak_compare(cmap, n, round(fs / orc.CODE_RATE))
    assert [int(restaged[0][2]), int(restaged[1][2])] == peak and restaged[2][2] == pytest.approx(ratio, rel=1e-9)


@pytest.mark.parametrize("n_prn,drange,dstep", [(32, 5000.0, 250.0), (8, 5000.0, 250.0), (16, 3750.0, 500.0), (32, 4000.0, 100.0),
                                                  (5, 5000.0, 50.0)])
def test_fused_sweep_vs_two_kernel_sweeps_and_oracle(engine, n_prn, drange, dstep):
    """A map-free search at 25 MHz of 256 transforms or more runs its inverse transforms in ONE launch of persistent
    workgroups, one (PRN, bin) transform per workgroup (pcps_fused.h): whole rounds of 256 and a tail cut into single
    rounds (32 x 41 and 8 x 41: both; 16 x 16 = 256: no tail; 32 x 81: ten rounds and a tail; 5 x 201: more than a
    round left over -- whole transforms throughout).  Peaks equal to the two-kernel sweeps' bit for bit, ratios to
    rounding, and both equal to the oracle's for the PRNs checked; a constant stream (every value of a row ties) must
    give the first index."""
    _fused_sweep_case(engine, 25e6, n_prn, drange, dstep)


@pytest.mark.parametrize("n_prn,drange,dstep", [(32, 5000.0, 250.0), (8, 5000.0, 250.0), (16, 1750.0, 500.0), (3, 5000.0, 100.0),
                                                  (2, 5000.0, 50.0)])
def test_fused_sweep_at_50_mhz_vs_two_kernel_sweeps_and_oracle(engine, n_prn, drange, dstep):
    """N = 50 000 (BASELINE configs 4-5's rate): 800 KB of state per transform do not fit a compute unit, so a radix-2
    decimation-in-frequency step in front splits every (PRN, bin) transform into its even and its odd output samples --
    two 25 000-point transforms with two operand terms per point (pcps_fused.h, TERMS = 2; the odd half's twiddle lives in
    a second image of the code spectrum).  Units = PRNs x bins x 2: ten rounds of 256 and a tail (32 x 41), two rounds and
    a tail (8 x 41), exactly one round (16 x 8), odd PRN counts.  Same checks as at 25 MHz: indices equal to the
    two-kernel sweeps' bit for bit (records carry 2m + parity), ratios to rounding, both equal to the oracle's; a
    constant stream gives the first index."""
    _fused_sweep_case(engine, 50e6, n_prn, drange, dstep)


def _fused_sweep_case(engine, fs, n_prn, drange, dstep):
    rng = np.random.default_rng(9000 + n_prn + int(dstep) + int(fs / 1e6))
    n = orc.samples_per_code(fs)
    prns = [int(p) for p in rng.choice(np.arange(1, 33), n_prn, replace=False)]
    sats = [dict(prn=p, doppler=float(rng.uniform(-3500, 3500)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=float(rng.uniform(5, 10))) for p in prns[::2]]
    start = int(rng.integers(0, 64))
    cap = (n + start + 7) // 8 * 8
    engine.iq_alloc(cap, FMT_CI8)
    engine.code_slots(n_prn)
    for s, p in enumerate(prns):
        engine.load_gps_code(s, p)
    engine.iq_synth(sats, fs, 12.0, 4242, 0, cap)
    nbins = len(np.arange(-drange, drange + 1, dstep))
    assert n_prn * nbins * (2 if fs == 50e6 else 1) >= 256
    res = {}
    for fused in (1, 0):
        engine.set_option("pcps_fused", fused)
        try:
            res[fused] = engine.pcps(np.arange(n_prn), start, fs, 0.0, drange, dstep, 1, 1)
        finally:
            engine.set_option("pcps_fused", 1)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-12, atol=0)
    rf = orc.iq_to_complex(engine.iq_download(cap, 0))
    x = rf[start:start + n].reshape(1, -1)
    for s in (0, 1, n_prn - 1):
        m = orc.pcps_map(x, 0.0, fs, orc.code_spectrum(orc.gold_code(prns[s]), fs), drange, dstep, n)
        peak, ratio = orc.two_peak_compare(m, n, round(fs / orc.CODE_RATE))
        assert peak == [int(res[1][0][s]), int(res[1][1][s])], prns[s]
        assert res[1][2][s] == pytest.approx(ratio, rel=1e-9)
    engine.iq_upload(np.zeros(2 * cap, dtype=np.int8), 0)
    pb, pc, _, _ = engine.pcps(np.arange(n_prn), 0, fs, 0.0, drange, dstep, 1, 1)
    assert np.all(pb == 0) and np.all(pc == 0)


@pytest.mark.parametrize("n_prn,noncoh,drange,dstep,if_hz", [(32, 10, 5000.0, 300.0, 0.0), (4, 10, 5000.0, 300.0, 0.0), (32, 1, 5000.0, 250.0, 0.0),
                                                                 (1, 3, 5000.0, 250.0, 1250.0), (7, 2, 2000.0, 100.0, 0.0)])
def test_fused_search_at_10_mhz_vs_the_map_accumulating_path_and_oracle(engine, n_prn, noncoh, drange, dstep, if_hz):
    """The reference's shipped search (10 MHz, 300 Hz grid, 1 ms x 10 non-coherent) when the caller wants indices and ratio:
    one workgroup per (PRN, bin) keeps the 10 000-point transform in its LDS, the non-coherent sum in registers, and finds
    the row's first AND second peak itself (pcps_fused10k.h) -- no map, no second sweep.  Against the path that accumulates
    the map in memory (`pcps_fused` = 0): same indices, ratio to rounding; against the oracle's map for the PRNs checked;
    a constant stream (every value ties) gives the first index."""
    fs = 10e6
    rng = np.random.default_rng(1000 + n_prn + noncoh)
    n = orc.samples_per_code(fs)
    prns = [int(p) for p in rng.choice(np.arange(1, 33), n_prn, replace=False)]
    sats = [dict(prn=p, doppler=float(rng.uniform(-3500, 3500)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=float(rng.uniform(4, 8))) for p in prns[::2]]
    start = int(rng.integers(0, 64))
    cap = (n * noncoh + start + 7) // 8 * 8
    engine.iq_alloc(cap, FMT_CI8)
    engine.code_slots(n_prn)
    for s, p in enumerate(prns):
        engine.load_gps_code(s, p)
    engine.iq_synth(sats, fs, 14.0, 777 + noncoh, 0, cap)
    res = {}
    for fused in (1, 0):
        engine.set_option("pcps_fused", fused)
        try:
            res[fused] = engine.pcps(np.arange(n_prn), start, fs, if_hz, drange, dstep, 1, noncoh)
        finally:
            engine.set_option("pcps_fused", 1)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-12, atol=0)
    rf = orc.iq_to_complex(engine.iq_download(cap, 0))
    x = rf[start:start + n * noncoh].reshape(1, -1)
    for s in sorted({0, n_prn - 1}):
        m = orc.pcps_map(x, if_hz, fs, orc.code_spectrum(orc.gold_code(prns[s]), fs), drange, dstep, n, 1, noncoh)
        peak, ratio = orc.two_peak_compare(m, n, round(fs / orc.CODE_RATE))
        assert peak == [int(res[1][0][s]), int(res[1][1][s])], prns[s]
        assert res[1][2][s] == pytest.approx(ratio, rel=1e-9)
    engine.iq_upload(np.zeros(2 * cap, dtype=np.int8), 0)
    pb, pc, _, _ = engine.pcps(np.arange(n_prn), 0, fs, if_hz, drange, dstep, 1, noncoh)
    assert np.all(pb == 0) and np.all(pc == 0)


@pytest.mark.parametrize("fs", [25e6, 4e6, 10e6, 50e6])
def test_register_resident_kernels_vs_oracle(engine, fs):
    """The map-free search at N = N1 x 200 (4 / 10 / 25 / 50 MHz) runs its inverse transforms through the
    register-resident kernels (pcps_fast.h: 125 x 200; pcps_fastn.h: 20 / 50 / 250 x 200): peaks and ratio against the
    oracle's map for present and absent satellites, an intermediate frequency, other Doppler grids (21 / 41 / 81 / 101
    bins: other workgroup-to-XCD mappings) and PRN counts that do not fill a group, from a ring offset -- and the
    general kernels on the same inputs."""
    rng = np.random.default_rng(250001 + int(fs) // 1000000)
    n = orc.samples_per_code(fs)
    cases = ((0.0, 5000.0, 250.0, 6), (1250.0, 5000.0, 500.0, 5), (0.0, 4000.0, 100.0, 3), (-2000.0, 5000.0, 100.0, 1))
    if fs == 50e6:
        cases = ((0.0, 5000.0, 250.0, 3), (1250.0, 2000.0, 100.0, 2), (-2000.0, 5000.0, 500.0, 1))     # (the oracle's maps take a while)
    for case, (if_hz, drange, dstep, n_prn) in enumerate(cases):
        prns = [int(p) for p in rng.choice(np.arange(1, 33), n_prn, replace=False)]
        present = prns[:max(1, n_prn // 2)]
        sats = [dict(prn=p, doppler=float(rng.uniform(-3500, 3500)), code_phase=float(rng.uniform(0, 1023)),
                     phase=float(rng.random()), amp=float(rng.uniform(5, 10))) for p in present]
        start = int(rng.integers(0, 64))
        cap = (n + start + 7) // 8 * 8
        engine.iq_alloc(cap, FMT_CI8)
        engine.code_slots(n_prn)
        for s, p in enumerate(prns):
            engine.load_gps_code(s, p)
        engine.iq_synth(sats, fs, 12.0, 1000 + case, 0, cap)
        rf = orc.iq_to_complex(engine.iq_download(cap, 0))
        pb, pc, pr, none = engine.pcps(np.arange(n_prn), start, fs, if_hz, drange, dstep, 1, 1)
        assert none is None
        engine.set_option("pcps_general_kernels", 1)
        try:
            gb, gc, gr, _ = engine.pcps(np.arange(n_prn), start, fs, if_hz, drange, dstep, 1, 1)
        finally:
            engine.set_option("pcps_general_kernels", 0)
        assert np.array_equal(pb, gb) and np.array_equal(pc, gc)
        np.testing.assert_allclose(pr, gr, rtol=1e-12, atol=0)
        x = rf[start:start + n].reshape(1, -1)
        for s, p in enumerate(prns):
            m = orc.pcps_map(x, if_hz, fs, orc.code_spectrum(orc.gold_code(p), fs), drange, dstep, n)
            peak, ratio = orc.two_peak_compare(m, n, round(fs / orc.CODE_RATE))
            assert peak == [int(pb[s]), int(pc[s])], (case, p)
            assert pr[s] == pytest.approx(ratio, rel=1e-9)
