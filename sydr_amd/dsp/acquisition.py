"""Function-level drop-ins for sydr/dsp/acquisition.py: same names, arguments and return values,
computed by the HIP library (no host arithmetic)."""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64
from ..runtime import get_engine


def PCPS(rfData, interFrequency, samplingFrequency, codeFFT, dopplerRange, dopplerStep, samplesPerCode,
         coherentIntegration=1, nonCoherentIntegration=1):
    """Correlation map [bins][samplesPerCode] of acquisition.py:9-74, on the GPU."""
    rf = np.squeeze(np.asarray(rfData, dtype=np.complex128))
    need = int(samplesPerCode) * int(coherentIntegration) * int(nonCoherentIntegration)
    if rf.size < need:
        raise ValueError(f"PCPS needs {need} samples, got {rf.size}")
    codeFFT = np.asarray(codeFFT, dtype=np.complex128).reshape(1, -1)
    if codeFFT.shape[1] != samplesPerCode:
        raise ValueError("codeFFT length must equal samplesPerCode")
    eng = get_engine()
    cap = (need + 7) // 8 * 8
    if eng.iq_fmt != FMT_CF64 or eng.iq_capacity < cap:
        eng.iq_alloc(cap, FMT_CF64)
    eng.iq_upload(rf[:need], 0)
    _, _, _, cmap = eng.pcps_spectra(codeFFT, 0, samplingFrequency, interFrequency, dopplerRange, dopplerStep,
                                     coherentIntegration, nonCoherentIntegration, want_map=True)
    return np.squeeze(np.squeeze(cmap[0]))


def TwoCorrelationPeakComparison(correlationMap, samplesPerCode, samplesPerCodeChip):
    """([bin, code], peak ratio) of acquisition.py:78-115 (same exclusion-window behaviour), on the GPU."""
    cmap = np.atleast_2d(np.asarray(correlationMap, dtype=np.float64))
    if cmap.shape[1] != samplesPerCode:
        raise ValueError("correlationMap row length must equal samplesPerCode")
    return get_engine().two_peak_compare(cmap, int(samplesPerCodeChip))


def SerialSearch(rfdata, code, dopplerRange, dopplerStep, samplingFrequency, samplesPerCode):
    """Brute-force map [bins][len(code)] of acquisition.py:119-155 (one code period of `rfdata`), on the GPU."""
    rf = np.squeeze(np.asarray(rfdata, dtype=np.complex128))
    if rf.size < samplesPerCode:
        raise ValueError(f"SerialSearch needs {samplesPerCode} samples, got {rf.size}")
    chips = np.asarray(code)
    eng = get_engine()
    if getattr(eng, "n_slots", 0) < 4:
        eng.code_slots(4, 4092)
    eng.set_code(3, chips.astype(np.int8))
    cap = (int(samplesPerCode) + 7) // 8 * 8
    if eng.iq_fmt != FMT_CF64 or eng.iq_capacity < cap:
        eng.iq_alloc(cap, FMT_CF64)
    eng.iq_upload(rf[:samplesPerCode], 0)
    _, _, _, cmap = eng.serial_search([3], 0, samplingFrequency, dopplerRange, dopplerStep, want_map=True,
                                      n_chips=len(chips))
    return np.squeeze(np.squeeze(cmap[0]))


def TwoCorrelationPeakComparison_SS(correlationMap):
    """([bin, chip], ratio) of acquisition.py:159-193 (3x3 exclusion block, Python slice semantics), on the GPU."""
    return get_engine().two_peak_compare_ss(np.atleast_2d(np.asarray(correlationMap, dtype=np.float64)))
