"""Scalar lock indicators / C-N0 estimator used by the host plugin (sydr/dsp/lockindicator.py:6-122)."""
import numpy as np


def FLL_Lock_Borre(iprompt, iprompt_prev, qprompt, qprompt_prev, fll_lock_prev, alpha=0.01):
    fll_lock = iprompt * iprompt_prev - qprompt * qprompt_prev
    fll_lock *= np.sign(iprompt * iprompt_prev + qprompt * qprompt_prev)
    fll_lock /= (iprompt**2 + qprompt**2)
    fll_lock = abs(fll_lock)
    return (1 - alpha) * fll_lock_prev + alpha * fll_lock


def PLL_Lock_Borre(iprompt, qprompt, pll_lock_prev, alpha=0.01):
    nbd = iprompt**2 - qprompt**2
    nbp = iprompt**2 + qprompt**2
    return (1 - alpha) * pll_lock_prev + alpha * (nbd / nbp)


def lowPassFilter(new, old, alpha):
    return (1 - alpha) * old + alpha * new


def CN0_Beaulieu(ratio, N, T, old):
    lambda_c = 1 / (ratio / N)
    cn0 = lambda_c * (1 / T)
    return lowPassFilter(cn0, old, alpha=0.1)


def CN0_NWPR(iPromptSum, qPromptSum, iPromptSum2, qPromptSum2, nbAccum=20, integrationPeriod=1e-3):
    """Narrow-band / wide-band power ratio C/N0 estimate in dB-Hz (lockindicator.py:40-71; imported by both
    reference plugins, used only in their commented-out alternatives)."""
    narrow = iPromptSum**2 + qPromptSum**2
    wide = iPromptSum2 + qPromptSum2
    ratio = narrow / wide
    return 10 * np.log10(1 / integrationPeriod * (ratio - 1) / (nbAccum - ratio))
