"""Randomised GPU-vs-oracle stress of the E/P/L kernel (both correlator variants, all ring formats, 1-8 taps,
ring wrap, chip switches on and near samples).  Usage: python tests/stress_epl.py [n_rounds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sydr_oracle as orc
from sydr_amd.engine import Engine, make_items, FMT_CI8, FMT_CI16, FMT_CF32, FMT_CF64



def run(rounds, seed, eng=None):
  """Returns (channel-epochs checked, worst relative error); raises AssertionError on the first mismatch."""
  rng = np.random.default_rng(seed)
  eng = eng or Engine(0)
  worst, checked = 0.0, 0
  for r in range(rounds):
      fmt = [FMT_CI8, FMT_CI8, FMT_CI16, FMT_CF32, FMT_CF64][r % 5]
      cap = int(rng.integers(40, 400)) * 1000 // 8 * 8
      if fmt == FMT_CI8:
          raw = rng.integers(-128, 128, 2 * cap).astype(np.int8)
      elif fmt == FMT_CI16:
          raw = rng.integers(-2000, 2000, 2 * cap).astype(np.int16)
      elif fmt == FMT_CF32:
          raw = rng.normal(0, 30, 2 * cap).astype(np.float32)
      else:
          raw = rng.normal(0, 30, 2 * cap).astype(np.float64)
      eng.iq_alloc(cap, fmt)
      eng.iq_upload(raw, 0)
      rf = raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)
      eng.code_slots(4)
      prns = rng.integers(1, 33, 4)
      for s, p in enumerate(prns):
          eng.load_gps_code(s, int(p))
      codes = [orc.pad_code(orc.gold_code(int(p))) for p in prns]
      n_taps = int(rng.integers(1, 9))
      spacing = tuple(np.sort(rng.uniform(-1.0, 1.0, n_taps)).round(int(rng.integers(1, 6))))
      # one launch = one correlator variant: draw the step regime per round
      regime = r % 4
      n_items = 24
      steps, ns = [], []
      for _ in range(n_items):
          if regime == 0:      # boundary variant, generic
              st = float(rng.uniform(0.004, 0.0599))
          elif regime == 1:    # boundary variant, steps that are exact binary fractions (switches exactly on samples)
              st = float(rng.choice([1 / 32, 1 / 64, 3 / 64, 1 / 128, 5 / 128, 7 / 128, 1 / 16, 3 / 32, 1 / 8, 7 / 64]))
          elif regime == 2:    # per-sample variant
              st = float(rng.uniform(0.0601, 0.6))
          else:                # mixed launch: one slow item forces the per-sample variant for all
              st = float(rng.uniform(0.004, 0.3))
          nmax = min(cap - 64, int(1020.0 / st))
          steps.append(st)
          ns.append(int(rng.integers(2, max(3, nmax))))
      starts = rng.integers(0, 4 * cap, n_items)          # absolute positions: the ring wraps
      f = rng.uniform(-8e3, 8e3, n_items)
      rc = rng.uniform(0, 2 * np.pi, n_items)
      rk = np.where(rng.random(n_items) < 0.3, rng.choice([0.0, 0.5, 0.25, 1e-12, 0.999999999999], n_items), rng.uniform(0, 1, n_items))
      slots = rng.integers(0, 4, n_items)
      fs = 1.023e6 / np.array(steps)
      # fs is a launch-wide parameter: use the first item's fs for all (the code step is what decides the variant)
      fs0 = float(fs[0])
      items = make_items(slots, ns, starts, f, rc, rk, steps)
      got = eng.epl_batch(items, spacing, fs0)
      for k in range(n_items):
          x = orc.ring_slice(rf, int(starts[k]) % cap, int(ns[k]))
          ref = np.asarray(orc.epl(x, codes[slots[k]], fs0, float(f[k]), float(rc[k]), float(rk[k]), float(steps[k]), spacing))
          g = got[k].reshape(-1, 2); q = ref.reshape(-1, 2)
          scale = np.maximum(np.hypot(q[:, 0], q[:, 1]), 1.0)
          err = float(np.max(np.hypot(g[:, 0] - q[:, 0], g[:, 1] - q[:, 1]) / scale))
          worst = max(worst, err)
          checked += 1
          if err > 1e-9:
              raise AssertionError(str(dict(round=r, item=k, fmt=fmt, n=ns[k], start=int(starts[k]), cap=cap, step=steps[k],
                                            f=f[k], rc=rc[k], rk=rk[k], spacing=spacing, fs=fs0, err=err)))
  return checked, worst


if __name__ == "__main__":
    t0 = time.time()
    checked, worst = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{checked} random channel-epochs checked in {time.time() - t0:.1f} s, worst relative error {worst:.2e}")
