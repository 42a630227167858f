"""Function-level drop-ins for sydr/dsp/tracking.py.

`EPL` (tracking.py:92-116) -- the sample-touching function -- runs on the GPU.  The scalar discriminators and
loop filters (tracking.py:39-61,120-186,246-325) run on the DEVICE in closed-loop tracking
(sydr_amd/csrc/track.hip); the host functions below carry the reference's names and signatures for callers
that use them one at a time (a reference plugin behind the GpuCorrelatorSeams mixin may import either), and
are pinned bit for bit against values captured from the reference (tests/golden/g7_loopmath.npz).  They keep
the reference's operation order and its GPS-ICD pi: their results feed thresholds and NCOs.
`generateReplica` / `getCorrelator` / `EPL_nonvector` are the legacy per-piece forms (tracking.py:8-35,65-88)."""
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
    eng = get_engine()
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


def EPL_nonvector(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
                  correlatorsSpacing):
    """The reference's per-sample Python loop (tracking.py:65-88) computes what `EPL` computes; served by the same
    kernel (results agree with the loop to summation-order rounding, ~1e-15 relative)."""
    return EPL(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
               correlatorsSpacing)


# ---------------------------------------------------------------------------------------------- legacy pieces (host)
def generateReplica(time, nbSamples: int, carrierFrequency: float, remCarrier: float):
    """Carrier replica exp(j(-2 pi f t + rem)) over time[0:nbSamples] and the phase left at time[nbSamples], modulo
    NumPy's 2 pi (tracking.py:8-17)."""
    phase = -(carrierFrequency * 2.0 * np.pi * np.asarray(time)[0:nbSamples + 1]) + remCarrier
    return np.exp(1j * phase[:nbSamples]), phase[nbSamples] % (2 * np.pi)


def getCorrelator(iSignal, qSignal, correlatorSpacing: float, code, remainingCode: float, codeStep: float,
                  nbSamples: int):
    """One tap: the padded code sampled at ceil(linspace(...)) against the mixed I and Q (tracking.py:21-35)."""
    first = remainingCode + correlatorSpacing
    chips = np.asarray(code)[np.ceil(np.linspace(first, nbSamples * codeStep + first, nbSamples,
                                                 endpoint=False)).astype(int)]
    return np.sum(chips * iSignal), np.sum(chips * qSignal)


# ---------------------------------------------------------------------------------------------- scalar loop math (host)
def LoopFiltersCoefficients(loopNoiseBandwidth: float, dampingRatio: float, loopGain: float):
    """(tau1, tau2) of a second-order loop from noise bandwidth, damping and gain (tracking.py:39-61)."""
    wn = loopNoiseBandwidth * 8.0 * dampingRatio / (4.0 * dampingRatio**2 + 1)
    return loopGain / wn**2, 2.0 * dampingRatio / wn


def DLL_NNEML(iEarly: float, qEarly: float, iLate: float, qLate: float):
    """Normalised non-coherent early-minus-late envelope discriminator (tracking.py:120-129)."""
    early, late = np.sqrt(iEarly**2 + qEarly**2), np.sqrt(iLate**2 + qLate**2)
    return (early - late) / (early + late)


def PLL_costa(iPrompt: float, qPrompt: float):
    """Costas discriminator atan(Q/I) in cycles of the GPS-ICD 2 pi (tracking.py:133-142)."""
    return np.arctan(qPrompt / iPrompt) / TWO_PI


def FLL_ATAN2(iPrompt: float, qPrompt: float, iPromptPrev: float, qPromptPrev: float, deltaT: float):
    """Four-quadrant frequency discriminator, argument order as the reference has it (tracking.py:146-152)."""
    angle = np.arctan2(iPromptPrev * iPrompt + qPromptPrev * qPrompt, iPromptPrev * qPrompt - qPromptPrev * iPrompt)
    return angle / deltaT / TWO_PI


def phase_unwrap(phase):
    """Fold a difference of two atan values into (-pi/2, pi/2) (tracking.py:169-176)."""
    if phase >= HALF_PI:
        return phase - PI
    if phase <= -HALF_PI:
        return phase + PI
    return phase


def FLL_ATAN(iPrompt: float, qPrompt: float, iPromptPrev: float, qPromptPrev: float, deltaT: float):
    """Difference of two-quadrant phases over deltaT, NaN (0/0) read as no error (tracking.py:156-165)."""
    step = np.arctan(qPrompt / iPrompt) - np.arctan(qPromptPrev / iPromptPrev)
    if np.isnan(step):
        step = 0.0
    return phase_unwrap(step) / deltaT / TWO_PI


def BorreLoopFilter(input: float, memory: float, tau1: float, tau2: float, pdi: float):
    """Proportional + integral increment of Borre's second-order filter (tracking.py:180-186)."""
    out = tau2 / tau1 * (input - memory)
    out += pdi / tau1 * input
    return out


def FLLassistedPLL_2ndOrder(phaseInput: float, freqInput: float, w0f: float, w0p: float, a2: float,
                            integrationTime: float, velMemory: float):
    """Second-order PLL assisted by a first-order FLL, one velocity accumulator (tracking.py:246-279).
    Returns (output, new velocity memory)."""
    vel = (phaseInput * w0p**2 + freqInput * w0f) * integrationTime
    out = vel + velMemory
    out += phaseInput * a2 * w0p
    return out, vel


def FLLassistedPLL_3rdOrder(phaseInput: float, freqInput: float, w0f: float, w0p: float, a2: float, a3: float,
                            b3: float, integrationTime: float, velMemory: float, accMemory: float):
    """Third-order PLL assisted by a second-order FLL: acceleration then velocity accumulator
    (tracking.py:283-325).  Returns (output, new velocity memory, new acceleration memory)."""
    acc = (phaseInput * w0p**3 + freqInput * w0f**2) * integrationTime
    stage = acc + accMemory
    vel = (stage + (phaseInput * a3 * w0p**2 + freqInput * a2 * w0f)) * integrationTime
    out = vel + velMemory
    out += phaseInput * b3 * w0p
    return out, vel, acc


__all__ = ["EPL", "EPL_nonvector", "generateReplica", "getCorrelator", "LoopFiltersCoefficients", "DLL_NNEML",
           "PLL_costa", "FLL_ATAN2", "FLL_ATAN", "phase_unwrap", "BorreLoopFilter", "FLLassistedPLL_2ndOrder",
           "FLLassistedPLL_3rdOrder"]
