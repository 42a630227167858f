"""The three hot-path seams of a SyDR channel plugin, served by the MI355X engine.

`ChannelL1CA_Kaplan` exposes `runSignalSearch` (channel_l1ca_kaplan.py:179), `runPeakFinder` (:203)
and `runCorrelators` (:378) as override points -- channel_l1ca_kaplan_ss.py:10-54 shows the
reference itself swapping them.  This mixin overrides exactly those, so it can sit in front of
either this package's host plugin or the reference's own class:

    class ChannelL1CA_Kaplan_MI355X(GpuCorrelatorSeams, sydr.channel.channel_l1ca_kaplan.ChannelL1CA_Kaplan):
        pass

With the reference's host `CircularBuffer` the needed slice is uploaded per call (compatibility
mode); with this package's device ring the kernels read the ring in place.
"""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, make_items
from ..utils.devicering import CircularBuffer as DeviceRing
from ..utils.constants import GPS_L1CA_CODE_FREQ, GPS_L1CA_CODE_SIZE_BITS, LNAV_MS_PER_BIT


class GpuCorrelatorSeams:
    codeSlot = None          # staged PRN replica of this channel (set by the manager or lazily)
    _injectedCorrelators = None   # set by a batching manager: this epoch's [IE,QE,IP,QP,IL,QL]
    _injectedAcquisition = None   # set by a batching manager: (map or None, [bin, code], ratio)

    # ------------------------------------------------------------------ plumbing
    def _engine(self):
        if isinstance(self.rfBuffer, DeviceRing):
            return self.rfBuffer.engine
        from ..runtime import get_engine
        return get_engine()

    def _ensure_code(self):
        eng = self._engine()
        if self.codeSlot is None:
            self.codeSlot = int(self.channelID)
        if getattr(eng, "n_slots", 0) <= self.codeSlot:
            eng.code_slots(max(32, self.codeSlot + 1))  # stand-alone use; a ChannelManager sizes this up front
        key = (self.satelliteID, getattr(eng, "code_generation", 0))
        if getattr(self, "_stagedPrn", None) != key:
            eng.load_gps_code(self.codeSlot, int(self.satelliteID))
            self._stagedPrn = key
        return eng

    def _stage_slice(self, eng, start, n):
        """Ring index the kernels should read: in place for the device ring, else upload the slice."""
        if isinstance(self.rfBuffer, DeviceRing):
            return int(start)
        data = np.squeeze(self.rfBuffer.getSlice(start, n)).astype(np.complex128)
        cap = (n + 7) // 8 * 8
        if eng.iq_fmt != FMT_CF64 or eng.iq_capacity < cap:
            eng.iq_alloc(cap, FMT_CF64)
        eng.iq_upload(data, 0)
        return 0

    # ------------------------------------------------------------------ acquisition seams
    def acquisitionRequest(self):
        """What a batching manager needs to search this channel together with others."""
        return dict(slot=self.codeSlot, start=int(self.currentSample), fs=self.rfSignal.samplingFrequency,
                    if_hz=self.rfSignal.interFrequency, doppler_range=self.acq_dopplerRange,
                    doppler_step=self.acq_dopplerSteps, coh=self.acq_coherentIntegration,
                    noncoh=self.acq_nonCoherentIntegration)

    def runSignalSearch(self):
        if self._injectedAcquisition is not None:
            cmap, self._acqPeak, self._acqRatio = self._injectedAcquisition
            self._injectedAcquisition = None
            self._acqMap = cmap
            return cmap
        eng = self._ensure_code()
        start = self._stage_slice(eng, self.currentSample, self.acq_requiredSamples)
        pb, pc, pr, cmap = eng.pcps([self.codeSlot], start, self.rfSignal.samplingFrequency,
                                    self.rfSignal.interFrequency, self.acq_dopplerRange, self.acq_dopplerSteps,
                                    self.acq_coherentIntegration, self.acq_nonCoherentIntegration, want_map=True)
        self._acqMap = cmap[0]
        self._acqPeak = [int(pb[0]), int(pc[0])]
        self._acqRatio = float(pr[0])
        return self._acqMap

    def runPeakFinder(self, correlationMap):
        if correlationMap is getattr(self, "_acqMap", None) and getattr(self, "_acqPeak", None) is not None:
            return self._acqPeak, self._acqRatio   # found on the device in the same pass as the map
        samplesPerCodeChip = round(self.rfSignal.samplingFrequency / GPS_L1CA_CODE_FREQ)
        return self._engine().two_peak_compare(np.asarray(correlationMap), samplesPerCodeChip)

    # ------------------------------------------------------------------ tracking seam
    def correlatorRequest(self):
        """sdr_epl_item fields of the next epoch, or None when the ring does not hold it yet."""
        if self.rfBuffer.getNbUnreadSamples(self.currentSample) < self.track_requiredSamples:
            return None
        return dict(code_slot=self.codeSlot, n_samples=int(self.track_requiredSamples),
                    start_sample=int(self.currentSample), carrier_hz=float(self.carrierFrequency),
                    rem_carrier=float(self._nco_rem_carrier()), rem_code=float(self._nco_rem_code()),
                    code_step=float(self.codeStep), spacing=tuple(float(s) for s in self.track_correlatorsSpacing))

    def _nco_rem_carrier(self):
        return self.remainingCarrier

    def _nco_rem_code(self):
        return self.remainingCode

    def _correlate(self):
        if self._injectedCorrelators is not None:
            out, self._injectedCorrelators = self._injectedCorrelators, None
            return out
        eng = self._ensure_code()
        req = self.correlatorRequest()
        start = self._stage_slice(eng, req["start_sample"], req["n_samples"])
        items = make_items(req["code_slot"], req["n_samples"], start, req["carrier_hz"], req["rem_carrier"],
                           req["rem_code"], req["code_step"])
        return eng.epl_batch(items, req["spacing"], self.rfSignal.samplingFrequency)[0]

    def runCorrelators(self):
        self.correlatorsResults[:] = self._correlate()
        if self.correlatorsAccumCounter == LNAV_MS_PER_BIT:
            self.correlatorsAccumCounter = 0
            self.correlatorsAccum[:] = 0.0
        self.correlatorsAccum += self.correlatorsResults[:]
        self.correlatorsAccumCounter += 1
