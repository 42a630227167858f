"""Kaplan plugin with SerialSearch acquisition on the MI355X: the same three overrides the reference's
ChannelL1CA_Kaplan_SS makes (sydr/channel/channel_l1ca_kaplan_ss.py:10-54)."""
from __future__ import annotations

import numpy as np

from ..utils.constants import GPS_L1CA_CODE_FREQ
from .l1ca_kaplan import ChannelL1CA_Kaplan


class ChannelL1CA_Kaplan_SS(ChannelL1CA_Kaplan):
    def runSignalSearch(self):
        eng = self._ensure_code()
        n_code = round(self.rfSignal.samplingFrequency * 1023 / GPS_L1CA_CODE_FREQ)
        start = self._stage_slice(eng, self.currentSample, n_code * self.acq_nonCoherentIntegration)
        pb, pc, pr, cmap = eng.serial_search([self.codeSlot], start, self.rfSignal.samplingFrequency,
                                             self.acq_dopplerRange, self.acq_dopplerSteps,
                                             noncoh=self.acq_nonCoherentIntegration, want_map=True)
        self._acqMap, self._acqPeak, self._acqRatio = cmap[0], [int(pb[0]), int(pc[0])], float(pr[0])
        return self._acqMap

    def runPeakFinder(self, correlationMap):
        if correlationMap is getattr(self, "_acqMap", None):
            return self._acqPeak, self._acqRatio
        return self._engine().two_peak_compare_ss(np.asarray(correlationMap))

    def postAcquisitionUpdate(self, acqIndices):
        """SerialSearch peak [bin, code CHIP] -> NCO start values: the search runs in chips (scaled to samples here) and
        mixes with the opposite sign (channel_l1ca_kaplan_ss.py:38-52)."""
        samples_per_chip = self.rfSignal.samplingFrequency / GPS_L1CA_CODE_FREQ
        self.enterTracking(-(self.rfSignal.interFrequency + self.searchedFrequency(acqIndices[0])),
                           acqIndices[1] * samples_per_chip)
