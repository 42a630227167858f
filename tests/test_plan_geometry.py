"""What sdr_epl_plan_create works out per item for the straight-line correlators (one thread per item on the device; here a
HOST build of the same __host__ __device__ functions) -- the epoch geometry and the Q32.32 line
the block boundaries are predicted from (correlator_chip.h: chip_geometry, correlator_chip2.h: chipn_setup) -- held against
the reference's own chip-index expression ceil(linspace(...)) (tracking.py:111-112, through the oracle) on the CPU: first
and last partial chips, every block start, every tap switch.  The kernels trust a prediction unless it lies within 2^-16
of a sample (where they evaluate the reference expression exactly): so must this test."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO
from oracle import sydr_oracle as orc

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
NEAR = 1 << 16


def _dump(tmp_path, fs, n_items, seed):
    exe = tmp_path / "chip_geometry_dump"
    if not exe.exists():
        subprocess.check_call([HIPCC, "-O1", "-std=c++17", "--cuda-host-only", "-ffp-contract=off", "-o", str(exe),
                               os.path.join(REPO, "tests", "csrc", "chip_geometry_dump.hip")])
    out = subprocess.check_output([str(exe), repr(fs), str(n_items), str(seed)], text=True)
    items = []
    for line in out.splitlines():
        items.append({k: (float(v) if "." in v or "e" in v else int(v)) for k, v in re.findall(r"(\w+)=(\S+)", line)})
    return items


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("fs", [25e6, 10e6])
def test_plan_geometry_matches_the_reference_chip_indices(tmp_path, fs):
    items = _dump(tmp_path, fs, 24, 2026)
    assert len(items) == 24
    checked = 0
    for it in items:
        n, rem, step = it["n"], it["rem_code"], it["code_step"]
        idx = {t: orc.epl_indices(n, rem, step, sp) for t, sp in ((0, -0.5), (1, 0.0), (2, 0.5))}
        a = idx[1]
        q0, F = int(a[0]), int(a[-1]) - int(a[0]) - 1
        assert (it["q0"], it["F"]) == (q0, F)
        assert it["head_end"] == int(np.searchsorted(a, q0, side="right"))           # samples on the first (partial) chip
        assert it["tail_start"] == int(np.searchsorted(a, int(a[-1]), side="left"))  # first sample of the last chip
        assert it["bad"] == 0 and (it["JE"], it["JL"]) == (-1, 0)
        T, U = it["Tfx"], it["Ufx"]
        first = {t: {int(q): int(np.searchsorted(idx[t], q, side="left")) for q in range(q0 - 1, int(a[-1]) + 3)} for t in idx}
        for k in range(F):                                    # whole chip q = q0 + 1 + k of the prompt tap
            q = q0 + 1 + k
            uS = U + (q - 1) * T + (1 << 32)
            uE = uS + T
            for u, want in ((uS, first[1][q]), (uE, first[1][q + 1])):
                if (u + NEAR) % (1 << 32) >= 2 * NEAR:        # not within 2^-16 of a sample: the prediction must hold
                    assert u >> 32 == want, (q, u >> 32, want)
                    checked += 1
            S = first[1][q]
            for t, key, j in ((0, "dE", -1), (2, "dL", 0)):
                uT = uS + it[key]
                if all((u + NEAR) % (1 << 32) >= 2 * NEAR for u in (uS, uE, uT)):
                    E = first[1][q + 1]
                    want = min(max(first[t][q + j + 1], S), E) - S        # samples of the block before the tap's switch
                    assert min((uT >> 32) - S, E - S) == want, (q, t, (uT >> 32) - S, want)
                    checked += 1
        # the carrier rotations the plan carries (chip_rotations): exp(-1j*k*dphi), the offset's shares 4224*(1+1j)*sum_k r_k
        d, D = it["dphi"], it["Dmin"]
        assert D == (64 * T) >> 32
        for k, c, s_ in ((5, "urc5", "urs5"), (13, "urc13", "urs13"), (D + 1, "rd1c", "rd1s")):
            assert abs(it[c] - np.cos(-k * d)) < 4e-16 and abs(it[s_] - np.sin(-k * d)) < 4e-16, (k, it[c], np.cos(-k * d))
        r = np.exp(-1j * d * np.arange(13))
        for n_sum, c in ((13, "biasc2"), (11, "biasc0")):
            want = 4224.0 * (1 + 1j) * r[:n_sum].sum()
            assert abs(it[c] - want.real) < 1e-10 * 4224 * n_sum
        assert abs(it["biass2"] - (4224.0 * (1 + 1j) * r.sum()).imag) < 1e-10 * 4224 * 13
        if fs == 10e6:                                        # two chips per lane: pairs of whole chips, an odd one left to the edge
            assert it["c2"] == 1 and it["F2"] == F // 2
            want_tail = first[1][q0 + F] if F % 2 else it["tail_start"]
            assert it["c2_tail"] == want_tail
            assert it["c2_Dmin"] == (128 * T) >> 32
    assert checked > 20000
