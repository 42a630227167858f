"""Cross-check of the NumPy oracle against the reference's OWN compiled legacy C correlator
(oracle/_ref/tracking.so, built by oracle/Makefile straight from
/root/reference/sydr/c_functions/tracking.c).  Skipped where the build product is absent."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import REPO, load_golden
from oracle import sydr_oracle as orc

REF_SO = os.path.join(REPO, "oracle", "_ref", "tracking.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/tracking.so not built")


def ref_epl(rf, prn, fs, f, rem_carrier, rem_code, step, spacing):
    """generateReplica -> generateCarrier -> getCorrelator x taps (sydr/old/tracking/tracking_epl_c.py:105-137)."""
    lib = C.CDLL(REF_SO)
    n = len(rf)
    dp, zp = C.POINTER(C.c_double), C.c_void_p
    t = np.arange(0, n + 1) / fs
    replica = np.zeros(n, dtype=np.complex128)
    rem = np.zeros(1)
    lib.generateReplica.argtypes = [dp, C.c_size_t, C.c_double, C.c_double, dp, zp]
    lib.generateReplica(t.ctypes.data_as(dp), n, f, rem_carrier, rem.ctypes.data_as(dp), replica.ctypes.data_as(zp))
    i_sig, q_sig = np.zeros(n), np.zeros(n)
    rf = np.ascontiguousarray(rf, dtype=np.complex128)
    lib.generateCarrier.argtypes = [zp, zp, C.c_size_t, dp, dp]
    lib.generateCarrier(rf.ctypes.data_as(zp), replica.ctypes.data_as(zp), n, i_sig.ctypes.data_as(dp),
                        q_sig.ctypes.data_as(dp))
    code = orc.pad_code(orc.gold_code(prn)).astype(np.int32)
    lib.getCorrelator.argtypes = [dp, dp, C.POINTER(C.c_int), C.c_size_t, C.c_double, C.c_double, C.c_double, dp, dp]
    out = []
    for sp in spacing:
        ic, qc = np.zeros(1), np.zeros(1)
        lib.getCorrelator(i_sig.ctypes.data_as(dp), q_sig.ctypes.data_as(dp), code.ctypes.data_as(C.POINTER(C.c_int)),
                          n, step, rem_code, sp, ic.ctypes.data_as(dp), qc.ctypes.data_as(dp))
        out += [ic[0], qc[0]]
    return np.array(out)


def test_oracle_matches_reference_c_on_its_own_fixture():
    g = load_golden("g5_epl.npz")
    prn, fs, f, rc, rk, step = g["fixture_params"]
    rf = orc.iq_to_complex(g["fixture_iq"])
    ours = np.array(orc.epl(rf, orc.pad_code(orc.gold_code(int(prn))), fs, f, rc, rk, step, (-0.5, 0.0, 0.5)))
    theirs = ref_epl(rf, int(prn), fs, f, rc, rk, step, (-0.5, 0.0, 0.5))
    # the C code uses the GPS-ICD pi (tracking.c:10) and serial sums: 1e-11 is the reference's own bar
    np.testing.assert_allclose(ours, theirs, rtol=1e-11)


def test_oracle_matches_reference_c_random_cases():
    g = load_golden("g5_epl.npz")
    for tag in ("z0", "r00", "r01", "z1", "r10", "int16"):
        prn, fs, f, rc, rk, step, n = g[f"{tag}_params"]
        rf = orc.iq_to_complex(g[f"{tag}_iq"])
        sp = tuple(g[f"{tag}_spacing"])
        theirs = ref_epl(rf, int(prn), fs, f, rc, rk, step, sp)
        ours = g[f"{tag}_out"]
        scale = np.repeat(np.hypot(ours[0::2], ours[1::2]), 2)
        assert np.max(np.abs(ours - theirs) / scale) < 1e-9, tag
