"""Host-side scalar lock indicators and C/N0 estimators under the names and signatures of
sydr/dsp/lockindicator.py:6-122.

On the product path these run on the device every epoch (sydr_amd/csrc/track.hip); the host functions exist
for callers that use them one at a time (a reference plugin behind the seams mixin, notebooks, the database
report) and are pinned bit for bit against values captured from the reference (tests/golden/g7_loopmath.npz).
Every expression keeps the reference's operation order: the results feed thresholds."""
from __future__ import annotations

import numpy as np


def lowPassFilter(new: float, old: float, alpha: float):
    """First-order recursive average, weight `alpha` on the new value (lockindicator.py:104-122)."""
    return (1 - alpha) * old + alpha * new


def FLL_Lock_Borre(iprompt, iprompt_prev, qprompt, qprompt_prev, fll_lock_prev, alpha=0.01):
    """Frequency lock detector: |cross-epoch dot, signed by the in-phase dot| / prompt power, smoothed
    (lockindicator.py:6-17)."""
    signed = iprompt * iprompt_prev - qprompt * qprompt_prev
    signed *= np.sign(iprompt * iprompt_prev + qprompt * qprompt_prev)
    signed /= (iprompt**2 + qprompt**2)
    return lowPassFilter(abs(signed), fll_lock_prev, alpha)


def PLL_Lock_Borre(iprompt, qprompt, pll_lock_prev, alpha=0.01):
    """Phase lock detector cos(2 phi) = (I^2 - Q^2) / (I^2 + Q^2), smoothed (lockindicator.py:22-35)."""
    narrow_diff, narrow_power = iprompt**2 - qprompt**2, iprompt**2 + qprompt**2
    return lowPassFilter(narrow_diff / narrow_power, pll_lock_prev, alpha)


def CN0_NWPR(iPromptSum: float, qPromptSum: float, iPromptSum2: float, qPromptSum2: float, nbAccum=20,
             integrationPeriod=1e-3):
    """Narrow-band / wide-band power ratio estimate in dB-Hz (lockindicator.py:40-71)."""
    ratio = (iPromptSum**2 + qPromptSum**2) / (iPromptSum2 + qPromptSum2)
    return 10 * np.log10(1 / integrationPeriod * (ratio - 1) / (nbAccum - ratio))


def CN0_Beaulieu(ratio: float, N: int, T: float, old: float):
    """Beaulieu's estimate from the accumulated Pn/Pd ratio over N epochs of T seconds, smoothed with
    alpha = 0.1 (lockindicator.py:75-99).  Linear units, as the reference leaves it."""
    lambda_c = 1 / (ratio / N)
    return lowPassFilter(lambda_c * (1 / T), old, alpha=0.1)


__all__ = ["FLL_Lock_Borre", "PLL_Lock_Borre", "CN0_NWPR", "CN0_Beaulieu", "lowPassFilter"]
