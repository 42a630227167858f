"""GPS L1 C/A channel plugin, Borre-style loops (the reference's default plugin,
sydr/channel/channel_l1ca_borre.py:263-451 selected at receiver_gps_l1ca.py:17), hot path on the MI355X.

Same configuration keys ([ACQUISITION]/[TRACKING] of config/channels/channel_GPS_L1CA_borre.ini),
attribute names and packet keys.  PCPS + peak search and the E/P/L correlators run on the device
(`GpuCorrelatorSeams`); DLL(NNEML)/PLL(Costas) with Borre filters and the NCO bookkeeping are the
reference's scalar host arithmetic (NumPy pi in the NCO: SURVEY.md T3)."""
from __future__ import annotations

import numpy as np

from ..dsp.tracking import BorreLoopFilter, DLL_NNEML, LoopFiltersCoefficients, PLL_costa
from ..utils.constants import GPS_L1CA_CODE_FREQ, GPS_L1CA_CODE_MS, GPS_L1CA_CODE_SIZE_BITS, LNAV_MS_PER_BIT
from ..utils.enumerations import ChannelMessage, ChannelState, GNSSSignalType, GNSSSystems, TrackingFlags
from .base import Channel
from .seams import GpuCorrelatorSeams


class ChannelL1CA(GpuCorrelatorSeams, Channel):
    MIN_CONVERGENCE_TIME = 100

    def __init__(self, cid, sharedBuffer, resultQueue, rfSignal, configuration):
        super().__init__(cid, sharedBuffer, resultQueue, rfSignal, configuration)
        self.NCO_remainingCode = 0.0
        self.NCO_remainingCarrier = 0.0
        self.NCO_code = 0.0
        self.NCO_codeError = 0.0
        self.NCO_carrier = 0.0
        self.NCO_carrierError = 0.0
        self.codeFrequency = GPS_L1CA_CODE_FREQ
        self.carrierFrequency = 0.0
        self.initialFrequency = 0.0
        self.codeCounter = 0
        self.iPrompt = 0.0
        self.qPrompt = 0.0
        self.fll = 0.0
        self.setAcquisition(configuration['ACQUISITION'])
        self.setTracking(configuration['TRACKING'])
        self.navPromptSum, self.navPromptSumCounter, self.navBits = 0.0, 0, []

    def _lastPromptI(self):
        return self.iPrompt

    # NCO state lives under the Borre plugin's names
    def _nco_rem_carrier(self):
        return self.NCO_remainingCarrier

    def _nco_rem_code(self):
        return self.NCO_remainingCode

    def _processHandler(self):
        out = []
        if self.channelState == ChannelState.IDLE:
            raise Warning(f"Tracking channel {self.channelID} is in IDLE.")
        elif self.channelState == ChannelState.ACQUIRING:
            out.append(self.runAcquisition())
        elif self.channelState == ChannelState.TRACKING:
            out.append(self.runTracking())
            out.append(self.runDecoding())
        else:
            raise ValueError(f"Channel state {self.channelState} is not valid.")
        return [r for r in out if r is not None]

    def setSatellite(self, satelliteID):
        super().setSatellite(satelliteID)
        self.systemID = GNSSSystems.GPS
        self.signalID = GNSSSignalType.GPS_L1_CA
        eng = self._ensure_code()
        code = eng.read_code(self.codeSlot).astype(np.float64)
        self.code = np.r_[code[-1], code, code[0]]

    def getTimeSinceTOW(self):
        t = self.codeSinceTOW * GPS_L1CA_CODE_MS
        t += self.rfBuffer.getNbUnreadSamples(self.currentSample) / (self.rfSignal.samplingFrequency / 1e3)
        return t

    # ----------------------------------------------------------------- configuration (borre:193-259)
    def setAcquisition(self, configuration):
        self.acq_dopplerRange = float(configuration['doppler_range'])
        self.acq_dopplerSteps = float(configuration['doppler_steps'])
        self.acq_coherentIntegration = int(configuration['coherent_integration'])
        self.acq_nonCoherentIntegration = int(configuration['non_coherent_integration'])
        self.acq_threshold = float(configuration['threshold'])
        self.acq_requiredSamples = int(self.rfSignal.samplingFrequency * 1e-3 *
                                       self.acq_nonCoherentIntegration * self.acq_coherentIntegration)

    def setTracking(self, configuration):
        self.track_correlatorsSpacing = [float(configuration['correlator_early']),
                                         float(configuration['correlator_prompt']),
                                         float(configuration['correlator_late'])]
        self.track_dll_tau1, self.track_dll_tau2 = LoopFiltersCoefficients(
            loopNoiseBandwidth=float(configuration['dll_noise_bandwidth']),
            dampingRatio=float(configuration['dll_damping_ratio']), loopGain=float(configuration['dll_loop_gain']))
        self.track_pll_tau1, self.track_pll_tau2 = LoopFiltersCoefficients(
            loopNoiseBandwidth=float(configuration['pll_noise_bandwidth']),
            dampingRatio=float(configuration['pll_damping_ratio']), loopGain=float(configuration['pll_loop_gain']))
        self.track_dll_pdi = float(configuration['dll_pdi'])
        self.track_pll_pdi = float(configuration['pll_pdi'])
        self.codeStep = GPS_L1CA_CODE_FREQ / self.rfSignal.samplingFrequency
        self.track_requiredSamples = int(np.ceil((GPS_L1CA_CODE_SIZE_BITS - self.NCO_remainingCode) / self.codeStep))
        self.trackFlags = TrackingFlags.UNKNOWN

    # ----------------------------------------------------------------- acquisition (borre:263-329)
    def runAcquisition(self):
        if self.rfBuffer.getNbUnreadSamples(self.currentSample) < self.acq_requiredSamples:
            return None
        correlationMap = self.runSignalSearch()                    # <- GPU
        indices, peakRatio = self.runPeakFinder(correlationMap)    # <- GPU (same pass)
        dopplerShift = -((-self.acq_dopplerRange) + self.acq_dopplerSteps * indices[0])
        self.codeOffset = int(np.round(indices[1]))
        self.carrierFrequency = self.rfSignal.interFrequency + dopplerShift
        self.initialFrequency = self.rfSignal.interFrequency + dopplerShift
        self.currentSample = self.currentSample + self.acq_requiredSamples
        self.currentSample -= self.track_requiredSamples
        self.currentSample += self.codeOffset + 1
        self.channelState = ChannelState.TRACKING
        results = self.prepareResults()
        results["type"] = ChannelMessage.ACQUISITION_UPDATE
        results["carrierFrequency"] = self.carrierFrequency
        results["codeOffset"] = self.codeOffset
        results["frequency_idx"] = indices[0]
        results["code_idx"] = indices[1]
        results["correlation_map"] = correlationMap
        results["peak_ratio"] = peakRatio
        return results

    # ----------------------------------------------------------------- tracking (borre:333-451)
    def runTracking(self):
        if self.rfBuffer.getNbUnreadSamples(self.currentSample) < self.track_requiredSamples:
            return None
        correlatorResults = [float(v) for v in self._correlate()]  # <- GPU
        n = self.track_requiredSamples
        self.NCO_remainingCarrier -= self.carrierFrequency * 2.0 * np.pi * n / self.rfSignal.samplingFrequency
        self.NCO_remainingCarrier %= (2 * np.pi)
        iEarly, qEarly, iPrompt, qPrompt, iLate, qLate = correlatorResults
        codeError = DLL_NNEML(iEarly=iEarly, qEarly=qEarly, iLate=iLate, qLate=qLate)
        self.NCO_code = BorreLoopFilter(codeError, self.NCO_codeError, self.track_dll_tau1, self.track_dll_tau2,
                                        self.track_dll_pdi)
        self.NCO_codeError = codeError
        phaseError = PLL_costa(iPrompt=iPrompt, qPrompt=qPrompt)
        self.NCO_carrier = BorreLoopFilter(phaseError, self.NCO_carrierError, self.track_pll_tau1,
                                           self.track_pll_tau2, self.track_pll_pdi)
        self.NCO_carrierError = phaseError
        if not (self.trackFlags & TrackingFlags.BIT_SYNC):
            if (self.trackFlags & TrackingFlags.CODE_LOCK) and self.codeCounter > self.MIN_CONVERGENCE_TIME \
                    and np.sign(self.iPrompt) != np.sign(iPrompt):
                self.trackFlags |= TrackingFlags.BIT_SYNC
        self.trackFlags |= TrackingFlags.CODE_LOCK
        self.iPrompt = iPrompt
        self.qPrompt = qPrompt
        self.codeCounter += 1
        self.codeSinceTOW += 1
        self.codeFrequency -= self.NCO_code
        self.carrierFrequency += self.NCO_carrier
        self.NCO_remainingCode += n * self.codeStep - GPS_L1CA_CODE_SIZE_BITS
        self.codeStep = self.codeFrequency / self.rfSignal.samplingFrequency
        self.currentSample = (self.currentSample + n) % self.rfBuffer.maxSize
        self.track_requiredSamples = int(np.ceil((GPS_L1CA_CODE_SIZE_BITS - self.NCO_remainingCode) / self.codeStep))

        results = self.prepareResults()
        results['type'] = ChannelMessage.TRACKING_UPDATE
        for key, val in zip(("i_early", "q_early", "i_prompt", "q_prompt", "i_late", "q_late"), correlatorResults):
            results[key] = val
        results["dll"] = self.NCO_code
        results["pll"] = self.NCO_carrier
        results["fll"] = self.fll
        results["carrier_frequency"] = self.carrierFrequency
        results["code_frequency"] = self.codeFrequency
        results["cn0"] = np.nan
        results["pll_lock"] = 0.0
        results["fll_lock"] = 0.0
        results["lock_state"] = 0
        results["carrier_frequency_error"] = self.NCO_carrierError
        results["code_frequency_error"] = self.NCO_codeError
        return results

    def runDecoding(self):
        """Bit accumulation only (decodeBit: 20 prompts after bit sync -> Prompt2Bit); LNAV word / subframe
        decoding stays with the reference's sydr/dsp/decoding.py, which is outside the accelerated path."""
        self.decodeBit()
        return None

    def decodeBit(self):
        if not (self.trackFlags & TrackingFlags.BIT_SYNC):
            self.navPromptSum, self.navPromptSumCounter = 0.0, 0
            return False
        self.navPromptSum += self._lastPromptI()
        self.navPromptSumCounter += 1
        if self.navPromptSumCounter != LNAV_MS_PER_BIT:
            return False
        self.navBits.append(1 if self.navPromptSum > 0 else 0)
        self.navPromptSum, self.navPromptSumCounter = 0.0, 0
        return True
