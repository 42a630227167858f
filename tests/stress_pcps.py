"""Randomised GPU-vs-oracle stress of PCPS acquisition: arbitrary code lengths in samples (four-step, per-pass and
generic-radix transforms), IF, Doppler grids, coherent / non-coherent integrations, present and absent satellites.
Peak indices must be identical, maps within 1e-9 of the map maximum.  Usage: python tests/stress_pcps.py [rounds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sydr_oracle as orc
from sydr_amd.engine import Engine, FMT_CI8
from sydr_amd import SdrError


def run(rounds, seed, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or Engine(0)
    checked, worst, refused = 0, 0.0, 0
    for r in range(rounds):
        n_code = int(rng.choice([2046, 3000, 4000, 4092, 5000, 5115, 6138, 8184, 10000, 10230, 12000, 16368, 20460, 25000, 50000, 4000, 10000]))
        if r % 5 == 4:
            n_code = int(rng.integers(2046, 12000))          # anything, including sizes with large prime factors
        if r % 10 == 9:
            n_code = int(rng.choice([2053, 4099, 8191, 10007, 12289]))   # primes: always the chirp-z path
        fs = n_code * 1000.0
        coh, noncoh = (1, 1) if r % 3 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        total = n_code * coh * noncoh + 64
        cap = (total + 7) // 8 * 8
        prns = [int(p) for p in rng.choice(np.arange(1, 33), 3, replace=False)]
        sats = [dict(prn=prns[0], doppler=float(rng.uniform(-4000, 4000)), code_phase=float(rng.uniform(0, 1023)),
                     phase=float(rng.random()), amp=float(rng.uniform(4, 10)))]
        eng.iq_alloc(cap, FMT_CI8)
        eng.code_slots(3)
        for s, p in enumerate(prns):
            eng.load_gps_code(s, p)
        eng.iq_synth(sats, fs, 12.0, int(rng.integers(1, 1 << 30)), 0, cap)
        rf = orc.iq_to_complex(eng.iq_download(cap, 0))
        if_hz = float(rng.choice([0.0, 0.0, 1250.0, -2000.0]))
        drange = float(rng.choice([5000.0, 4000.0, 2500.0]))
        dstep = float(rng.choice([250.0, 500.0, 300.0, 125.0]))
        start = int(rng.integers(0, 32))
        try:
            pb, pc, pr, cmap = eng.pcps(np.arange(3), start, fs, if_hz, drange, dstep, coh, noncoh, want_map=True)
        except SdrError as e:     # (no size is refused any more: unfactorable lengths take the chirp-z path)
            raise AssertionError(f"n_code={n_code}: {e}")
        spc = round(fs / orc.CODE_RATE)
        x = rf[start:start + n_code * coh * noncoh].reshape(1, -1)
        for s, p in enumerate(prns):
            m = orc.pcps_map(x, if_hz, fs, orc.code_spectrum(orc.gold_code(p), fs), drange, dstep, n_code, coh, noncoh)
            peak, ratio = orc.two_peak_compare(m, n_code, spc)
            err = float(np.max(np.abs(cmap[s] - m)) / m.max())
            worst = max(worst, err)
            ok = peak == [int(pb[s]), int(pc[s])] and abs(ratio - pr[s]) <= 1e-9 * ratio and err <= 1e-9
            if not ok:
                raise AssertionError(str(dict(round=r, n_code=n_code, prn=p, coh=coh, noncoh=noncoh, if_hz=if_hz, drange=drange,
                                              dstep=dstep, start=start, got=(int(pb[s]), int(pc[s]), float(pr[s])),
                                              want=(peak, ratio), map_err=err)))
            checked += 1
    return checked, worst, refused


if __name__ == "__main__":
    t0 = time.time()
    checked, worst, refused = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{checked} random acquisitions checked in {time.time() - t0:.1f} s ({refused} sizes refused), worst map error {worst:.2e}")
