"""Function-level drop-ins for sydr/dsp/tracking.py.

EPL runs on the GPU.  The scalar discriminators / loop filters below are host arithmetic exactly
as in the reference (they stay on the host in the per-epoch plugin path; the closed-loop kernel
sydr_amd/csrc/track.hip carries its own device copy).  tracking.py:39-61,120-186,246-279."""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, make_items
from ..runtime import get_engine
from ..utils.constants import HALF_PI, PI, TWO_PI

_code_cache: dict[bytes, int] = {}


def EPL(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
        correlatorsSpacing):
    """[I_tap0, Q_tap0, I_tap1, ...] of tracking.py:92-116 for complex128 rfData; `code` is the padded
    replica [c[-1], c[0..L-1], c[0]] the reference plugins keep (channel_l1ca_kaplan.py:104-107)."""
    rf = np.squeeze(np.asarray(rfData, dtype=np.complex128))
    n = rf.size
    chips = np.asarray(code)[1:-1]
    eng = get_engine(0)
    if getattr(eng, "n_slots", 0) < 4:
        eng.code_slots(4, 4092)
    eng.set_code(2, chips.astype(np.int8))
    cap = (n + 15) // 8 * 8
    if eng.iq_fmt != FMT_CF64 or eng.iq_capacity < cap:
        eng.iq_alloc(cap, FMT_CF64)
    eng.iq_upload(rf, 0)
    items = make_items(2, n, 0, float(carrierFrequency), float(remainingCarrier), float(remainingCode),
                       float(codeStep))
    out = eng.epl_batch(items, tuple(float(s) for s in correlatorsSpacing), samplingFrequency)[0]
    return [float(v) for v in out]


def LoopFiltersCoefficients(loopNoiseBandwidth, dampingRatio, loopGain):
    Wn = loopNoiseBandwidth * 8.0 * dampingRatio / (4.0 * dampingRatio**2 + 1)
    tau1 = loopGain / Wn**2
    tau2 = 2.0 * dampingRatio / Wn
    return tau1, tau2


def DLL_NNEML(iEarly, qEarly, iLate, qLate):
    early = np.sqrt(iEarly**2 + qEarly**2)
    late = np.sqrt(iLate**2 + qLate**2)
    return (early - late) / (np.sqrt(iEarly**2 + qEarly**2) + np.sqrt(iLate**2 + qLate**2))


def PLL_costa(iPrompt, qPrompt):
    phaseError = np.arctan(qPrompt / iPrompt)
    phaseError /= TWO_PI
    return phaseError


def phase_unwrap(phase):
    if phase >= HALF_PI:
        return phase - PI
    if phase <= -HALF_PI:
        return phase + PI
    return phase


def FLL_ATAN(iPrompt, qPrompt, iPromptPrev, qPromptPrev, deltaT):
    frequencyError = np.arctan(qPrompt / iPrompt) - np.arctan(qPromptPrev / iPromptPrev)
    if np.isnan(frequencyError):
        frequencyError = 0.0
    frequencyError = phase_unwrap(frequencyError) / deltaT
    frequencyError /= TWO_PI
    return frequencyError


def FLL_ATAN2(iPrompt, qPrompt, iPromptPrev, qPromptPrev, deltaT):
    frequencyError = np.arctan2(iPromptPrev * iPrompt + qPromptPrev * qPrompt,
                                iPromptPrev * qPrompt - qPromptPrev * iPrompt) / deltaT
    frequencyError /= TWO_PI
    return frequencyError


def BorreLoopFilter(input, memory, tau1, tau2, pdi):
    output = tau2 / tau1 * (input - memory)
    output += pdi / tau1 * input
    return output


def FLLassistedPLL_2ndOrder(phaseInput, freqInput, w0f, w0p, a2, integrationTime, velMemory):
    update = (phaseInput * w0p**2 + freqInput * w0f) * integrationTime
    output = update + velMemory
    velMemory = update
    output += phaseInput * a2 * w0p
    return output, velMemory


def FLLassistedPLL_3rdOrder(phaseInput, freqInput, w0f, w0p, a2, a3, b3, integrationTime, velMemory, accMemory):
    """3rd-order PLL assisted by a 2nd-order FLL (tracking.py:283-327, [Kaplan 2006] p.180-182).
    Returns (output, velMemory, accMemory)."""
    acc_update = (phaseInput * w0p**3 + freqInput * w0f**2) * integrationTime
    output = acc_update + accMemory
    accMemory = acc_update
    vel_update = (output + (phaseInput * a3 * w0p**2 + freqInput * a2 * w0f)) * integrationTime
    output = vel_update + velMemory
    velMemory = vel_update
    output += phaseInput * b3 * w0p
    return output, velMemory, accMemory


def EPL_nonvector(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
                  correlatorsSpacing):
    """The reference's sample-by-sample formulation of EPL (tracking.py:65-88).  Its chip index is
    ceil(remCode + spacing + idx*codeStep) -- the same integers as EPL's linspace except where rounding differs in the
    last ulp -- so it is served by the same kernel."""
    return EPL(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
               correlatorsSpacing)


def generateReplica(time, nbSamples, carrierFrequency, remCarrier):
    """Carrier replica exp(1j*(-(f*2*pi*t) + rem)) over nbSamples and the phase left over for the next block
    (tracking.py:8-17).  Host arithmetic: the kernels generate their replica in registers and never store it."""
    t = np.asarray(time)[0:nbSamples + 1]
    phase = -(carrierFrequency * 2.0 * np.pi * t) + remCarrier
    return np.exp(1j * phase[:nbSamples]), phase[nbSamples] % (2 * np.pi)


def getCorrelator(iSignal, qSignal, correlatorSpacing, code, remainingCode, codeStep, nbSamples):
    """(I, Q) correlation of an already carrier-wiped signal with one tap of the code (tracking.py:21-35): the
    correlator kernel with a zero-frequency carrier and a single tap."""
    x = np.asarray(iSignal, dtype=np.float64)[:nbSamples] + 1j * np.asarray(qSignal, dtype=np.float64)[:nbSamples]
    i_corr, q_corr = EPL(x, code, 1.0, 0.0, 0.0, remainingCode, codeStep, (correlatorSpacing,))
    return i_corr, q_corr
