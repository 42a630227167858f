"""Randomised GPU-vs-oracle stress of the straight-line E/P/L kernels (tap switch positions compiled in at 25 MHz +-0.5
chip; whole-chip tap geometry at 25 / 50 MHz; two chips per lane at 10 MHz) and of the flipped ring image they read: code Doppler, phases at and next to
zero, one to four periods, n = N - 2 .. N + 2, epochs that wrap the ring (redone per sample inside the launch), carriers up
to an intermediate frequency, full-scale samples, and the ring rewritten in pieces between launches (the image follows its
dirty range).   Usage: python tests/stress_static.py [n_rounds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sydr_oracle as orc
from sydr_amd.engine import Engine, make_items, FMT_CI8

GEOMETRIES = (  # fs, spacing, most periods per epoch, variant bits the plan must report
    (25e6, (-0.5, 0.0, 0.5), 2, 26 + 24 + 256 * 12),
    (50e6, (-1.0, -0.5, 0.0, 0.5, 1.0), 4, 65536 + 26 + 24 + 4096),
    (25e6, (-1.0, 0.0, 1.0), 2, 26 + 24 + 4096),
    (50e6, (-0.5, 0.0, 0.5), 1, 65536 + 26 + 24 + 4096),
    (10e6, (-0.5, 0.0, 0.5), 2, 8 + 8192),                      # two chips per lane (correlator_chip2.h): the shipped rate
    (20e6, (-0.5, 0.0, 0.5), 2, 26 + 19 + 256 * 9),             # round 4: block length 19, switch at 9
    (32e6, (-0.5, 0.0, 0.5), 1, 65536 + 26 + 15 + 4096),        # round 4: half-chip view, block length 15, whole-chip taps
    (18e6, (-0.5, 0.0, 0.5), 2, 26 + 17 + 256 * 8),             # round 4: the other block lengths (epl_straight.hip) -- 17 / 8 ...
    (22e6, (-1.0, 0.0, 1.0), 2, 26 + 21 + 4096),                # ... 21 with whole-chip taps ...
    (40e6, (-0.5, 0.0, 0.5), 1, 65536 + 26 + 19 + 4096),        # ... 19 on the half-chip view
    (16.368e6, (-0.5, 0.0, 0.5), 2, 26 + 16),                   # round 4: exactly 16.0 per chip -- block lengths 15 and 16 in one kernel
)


# every block length of epl_straight.hip in every form (python tests/stress_static.py <rounds> <seed> --all-lengths)
ALL_LENGTHS = tuple(g for km in range(16, 26) for g in (
    (1.023e6 * (km + 0.5), (-0.5, 0.0, 0.5), 2, 26 + km + 256 * (km // 2)),
    (1.023e6 * (km + 0.5), (-1.0, 0.0, 1.0), 2, 26 + km + 4096),
    (1.023e6 * (km + 0.5), (-0.25, 0.0, 0.25), 2, 26 + km),
    (2.046e6 * (km + 0.5), (-0.5, 0.0, 0.5), 1, 65536 + 26 + km + 4096),
    (2.046e6 * (km + 0.5), (-1.0, -0.5, 0.0, 0.5, 1.0), 1, 65536 + 26 + km + 4096)))


def run(rounds, seed, eng=None, n_items=160, geometries=None):
    """Returns (channel-epochs checked, worst relative error); raises AssertionError on the first mismatch."""
    rng = np.random.default_rng(seed)
    eng = eng or Engine(0)
    worst, checked = 0.0, 0
    cap = 8 * 230000
    raw = rng.integers(-128, 128, 2 * cap).astype(np.int8)
    eng.iq_alloc(cap, FMT_CI8)
    eng.iq_upload(raw, 0)
    eng.code_slots(8, 1023, 5)
    prns = [int(p) for p in rng.choice(np.arange(1, 33), 8, replace=False)]
    for s, p in enumerate(prns):
        eng.load_gps_code(s, p)
    codes = [orc.pad_code(orc.gold_code(p)) for p in prns]
    for r in range(rounds):
        geometries = geometries or GEOMETRIES
        fs, spacing, per_hi, want = geometries[r % len(geometries)]
        # a piece of the ring is rewritten before every launch but the first: the flipped image must follow
        if r:
            lo = int(rng.integers(0, cap - 1000))
            n_new = int(rng.integers(1, min(cap - lo, 300000)))
            piece = rng.integers(-128, 128, 2 * n_new).astype(np.int8)
            raw[2 * lo:2 * (lo + n_new)] = piece
            eng.iq_upload(piece, lo)
        rf = raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)
        step = (1.023e6 + rng.uniform(-12, 12, n_items)) / fs
        rem = rng.uniform(0, step) * rng.choice([1.0, 1.0, 1.0, 1e-6, 0.999999], n_items)
        per = rng.integers(1, per_hi + 1, n_items)
        n = np.ceil((1023 * per - rem) / step).astype(np.int64) + rng.integers(-2, 3, n_items)
        n[:3] = [7, 61, 129][: 3]
        start = rng.integers(0, cap, n_items)                 # some epochs wrap the ring: redone per sample in the launch
        start[3] = cap - 5
        if want & 8192:
            # (a plan takes the two-chip kernel when at most 1 item in 64 falls outside its scheme: two strays here -- one
            # tiny epoch, one that wraps the ring -- both redone per sample inside the launch)
            n[1:3] = n[4:6]
            if n_items < 128:
                n[0] = n[6]                                   # (a short list: the wrapping epoch alone)
            start = rng.integers(0, cap - 30000, n_items)
            start[3] = cap - 5
        slot = rng.integers(0, 8, n_items)
        f = rng.uniform(-20000, 20000, n_items) * rng.choice([1.0, 1.0, 0.0, 200.0], n_items)
        ph = rng.uniform(-10, 10, n_items)
        plan = eng.epl_plan(make_items(slot, n, start, f, ph, rem, step), spacing, fs)
        if plan.variant != want:
            raise AssertionError(f"round {r}: variant {plan.variant} instead of {want}")
        plan.run()
        got = plan.fetch()
        plan.close()
        for k in range(n_items):
            x = orc.ring_slice(rf, int(start[k]), int(n[k]))
            ref = np.asarray(orc.epl(x, codes[int(slot[k])], fs, float(f[k]), float(ph[k]), float(rem[k]), float(step[k]), spacing))
            g, q = got[k].reshape(-1, 2), ref.reshape(-1, 2)
            scale = np.maximum(np.hypot(q[:, 0], q[:, 1]), np.sqrt(float(n[k])) * 60.0)
            err = float(np.max(np.hypot(g[:, 0] - q[:, 0], g[:, 1] - q[:, 1]) / scale))
            worst = max(worst, err)
            checked += 1
            if err > 1e-9:
                raise AssertionError(str(dict(round=r, item=k, fs=fs, spacing=spacing, n=int(n[k]), start=int(start[k]),
                                              step=step[k], rem=rem[k], f=f[k], ph=ph[k], err=err)))
    return checked, worst


if __name__ == "__main__":
    t0 = time.time()
    checked, worst = run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                         geometries=ALL_LENGTHS if "--all-lengths" in sys.argv else None)
    print(f"{checked} random channel-epochs checked in {time.time() - t0:.1f} s, worst relative error {worst:.2e}")
