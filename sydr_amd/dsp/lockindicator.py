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
