"""Randomised stress of the map-free searches that keep a whole inverse transform inside one workgroup: N = 25 000
(pcps_fused.h: persistent workgroups, whole rounds + a tail of single-round units, the second sweep on five workgroups per
PRN), N = 50 000 (the same kernel behind a radix-2 decimation-in-frequency step: a unit is one parity of a transform, two
operand terms per point) and N = 10 000 (pcps_fused10k.h: transform in LDS, non-coherent sum in registers).  Every round draws PRN count,
Doppler grid, IF, start offset, noise level, non-coherent blocks (10 MHz) and which satellites are present; the result must
equal the path with the fused kernels switched off (`pcps_fused` = 0) -- indices bit for bit, ratio to 1e-12 -- and, for
three PRNs of every `oracle_every`-th round, the oracle's map (indices exactly, ratio to 1e-9).
Usage: python tests/stress_pcps_fused.py [rounds] [seed] [oracle_every]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sydr_oracle as orc
from sydr_amd.engine import Engine, FMT_CI8


def run(rounds, seed, oracle_every=8, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or Engine(0)
    searched, checked, worst_ratio, worst_vs_general = 0, 0, 0.0, 0.0
    t0 = time.time()
    for r in range(rounds):
        fs = (25e6, 10e6, 50e6)[r % 3]
        n = orc.samples_per_code(fs)
        noncoh = 1 if fs != 10e6 else int(rng.choice([1, 2, 3, 10]))
        drange = float(rng.choice([5000.0, 4000.0, 2500.0, 1000.0]))
        dstep = float(rng.choice([250.0, 500.0, 300.0, 125.0, 100.0]))
        nbins = len(np.arange(-drange, drange + 1, dstep))
        need = {25e6: 256, 50e6: 128, 10e6: 32}[fs]         # transforms from which the fused kernels take a search
        lo = max(1, -(-need // nbins))
        if lo > 32:
            continue
        n_prn = int(rng.integers(lo, 33))
        prns = [int(p) for p in rng.choice(np.arange(1, 33), n_prn, replace=False)]
        present = [p for p in prns if rng.random() < 0.6]
        sats = [dict(prn=p, doppler=float(rng.uniform(-drange, drange)), code_phase=float(rng.uniform(0, 1023)),
                     phase=float(rng.random()), amp=float(rng.uniform(2, 10))) for p in present] or \
               [dict(prn=prns[0], doppler=0.0, code_phase=1.0, phase=0.0, amp=0.0)]
        start = int(rng.integers(0, 200))
        if_hz = float(rng.choice([0.0, 0.0, 1250.0, -2000.0]))
        cap = (n * noncoh + start + 7) // 8 * 8
        eng.iq_alloc(cap, FMT_CI8)
        eng.code_slots(n_prn)
        for s, p in enumerate(prns):
            eng.load_gps_code(s, p)
        eng.iq_synth(sats, fs, float(rng.choice([4.0, 12.0, 30.0])), int(rng.integers(1, 1 << 30)), 0, cap)
        res = {}
        for fused in (1, 0):
            eng.set_option("pcps_fused", fused)
            try:
                res[fused] = eng.pcps(np.arange(n_prn), start, fs, if_hz, drange, dstep, 1, noncoh)
            finally:
                eng.set_option("pcps_fused", 1)
        assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1]), (r, fs, n_prn, nbins, noncoh)
        rel = np.abs(res[1][2] - res[0][2]) / np.abs(res[0][2])
        worst_vs_general = max(worst_vs_general, float(rel.max()))
        assert rel.max() < 1e-12, (r, float(rel.max()))
        searched += n_prn
        if r % oracle_every == 0:
            rf = orc.iq_to_complex(eng.iq_download(cap, 0))
            x = rf[start:start + n * noncoh].reshape(1, -1)
            for s in sorted({0, n_prn // 2, n_prn - 1}):
                m = orc.pcps_map(x, if_hz, fs, orc.code_spectrum(orc.gold_code(prns[s]), fs), drange, dstep, n, 1, noncoh)
                peak, ratio = orc.two_peak_compare(m, n, round(fs / orc.CODE_RATE))
                assert peak == [int(res[1][0][s]), int(res[1][1][s])], (r, fs, prns[s], peak, int(res[1][0][s]), int(res[1][1][s]))
                err = abs(res[1][2][s] - ratio) / abs(ratio)
                worst_ratio = max(worst_ratio, float(err))
                assert err < 1e-9, (r, prns[s], err)
                checked += 1
    return searched, checked, worst_vs_general, worst_ratio, time.time() - t0


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    searched, checked, wg, wr, dt = run(rounds, seed, every)
    print(f"{rounds} random searches ({searched} PRN searches) equal to the general kernels' (ratio within {wg:.2e}); "
          f"{checked} PRN searches equal to the oracle's map (ratio within {wr:.2e}); {dt:.1f} s")
