"""A channel plugin whose tracking state lives on the GPU.

The reference's plugins (sydr/channel/channel_l1ca_kaplan.py, channel_l1ca_borre.py) carry the loop
state as Python attributes and update it statement by statement every millisecond.  Here a channel is
a VIEW: its state is one row of the device-resident bank that belongs to the device ring
(`ChannelBank`, one per GPU), every tracking epoch -- correlators, discriminators, loop filters, NCO,
lock-state machine, bit decisions -- is one device step for all channels, and the reference's
attribute names are properties over the mirrored row.  What stays on the host is what the reference's
plugin surface demands: the constructor signature, the INI keys, the acquisition seams
(`runSignalSearch` / `runPeakFinder` / `postAcquisitionUpdate`), `_processHandler`, the packets.

Subclasses describe a plugin declaratively: which loop (`LOOP_KIND`), which INI key feeds which
`sdr_loop_cfg` field (`CFG_KEYS`, `FILTERS`), which reference attribute name maps to which state /
record field (`STATE_VIEW`, `RECORD_VIEW`).
"""
from __future__ import annotations

import numpy as np

from ..utils.constants import GPS_L1CA_CODE_FREQ, GPS_L1CA_CODE_MS, GPS_L1CA_CODE_SIZE_BITS
from ..utils.devicering import CircularBuffer as DeviceRing
from ..utils.enumerations import ChannelMessage, ChannelState, GNSSSignalType, GNSSSystems, TrackingFlags
from .bank import tracking_packet
from .navdecoder import HOST_FLAGS, default_decoder
from .base import Channel
from .seams import GpuCorrelatorSeams


def loop_filter_taus(noise_bandwidth: float, damping: float, gain: float):
    """tau1, tau2 of a second-order loop filter from its noise bandwidth (sydr/dsp/tracking.py:39-61)."""
    wn = noise_bandwidth * 8.0 * damping / (4.0 * damping**2 + 1)
    return gain / wn**2, 2.0 * damping / wn


def _state_property(field, cast):
    def getter(self):
        return cast(self._bank.state[field][self._row])

    def setter(self, value):
        self._bank.state[field][self._row] = value
        self._bank.touch(self._row)
    return property(getter, setter)


def _record_property(field):
    return property(lambda self: float(self._bank.last[field][self._row]))


class _ViewMeta(type(Channel)):
    """Turns the STATE_VIEW / RECORD_VIEW tables of a plugin class into properties."""

    def __new__(mcls, name, bases, ns):
        for attr, (field, cast) in ns.get("STATE_VIEW", {}).items():
            ns.setdefault(attr, _state_property(field, cast))
        for attr, field in ns.get("RECORD_VIEW", {}).items():
            ns.setdefault(attr, _record_property(field))
        return super().__new__(mcls, name, bases, ns)


class DeviceTrackedChannel(GpuCorrelatorSeams, Channel, metaclass=_ViewMeta):
    LOOP_KIND = None          # bank.KIND_*
    N_TAPS = 3
    CFG_KEYS = {}             # [TRACKING] ini key -> sdr_loop_cfg field
    FILTERS = ()              # (cfg prefix, ini prefix): <ini>_noise_bandwidth/_damping_ratio/_loop_gain -> <cfg>_tau1/_tau2
    STATE_VIEW = {}           # reference attribute name -> (sdr_track_state field, cast)
    RECORD_VIEW = {}          # reference attribute name -> sdr_track_epoch field of the latest epoch

    def __init__(self, cid, sharedBuffer, resultQueue, rfSignal, configuration):
        if not isinstance(sharedBuffer, DeviceRing):
            raise TypeError("a device-tracked channel needs the device ring (sydr_amd.utils.devicering.CircularBuffer); "
                            "to accelerate the reference's own plugin over its host ring, mix GpuCorrelatorSeams into it")
        sharedBuffer.bankFor(cid)                       # (grows the ring's bank to hold this channel)
        self._ring, self._row = sharedBuffer, int(cid)
        self._bank.state[self._row] = np.zeros((), dtype=self._bank.state.dtype)
        self._bank.tracking[self._row] = self._bank.lost[self._row] = False
        self._bank.code_since_tow[self._row] = 0
        self._bank.tow[self._row], self._bank.tow_decoded[self._row], self._bank.host_flags[self._row] = 0.0, False, 0
        del self._bank.nav_bits[self._row][:]
        super().__init__(cid, sharedBuffer, resultQueue, rfSignal, configuration)
        self.setDecoding()
        self.codeOffset = 0
        self.setAcquisition(configuration['ACQUISITION'])
        self.setTracking(configuration['TRACKING'])

    @property
    def _bank(self):
        """The ring's channel bank AS IT IS NOW: the bank is re-created when it has to grow (a 33rd channel), and a
        reference kept from construction would leave this channel on the old, closed one."""
        return self._ring.channelBank

    # ------------------------------------------------------------------ configuration ([ACQUISITION] / [TRACKING])
    def setAcquisition(self, configuration):
        self.acq_dopplerRange = float(configuration['doppler_range'])
        self.acq_dopplerSteps = float(configuration['doppler_steps'])
        self.acq_coherentIntegration = int(configuration['coherent_integration'])
        self.acq_nonCoherentIntegration = int(configuration['non_coherent_integration'])
        self.acq_threshold = float(configuration['threshold'])
        ms = self.acq_nonCoherentIntegration * self.acq_coherentIntegration
        self.acq_requiredSamples = int(self.rfSignal.samplingFrequency * 1e-3 * ms)

    def setTracking(self, configuration):
        """Fill this channel's sdr_loop_cfg row and the state tracking starts from (kaplan:256-338, borre:206-259)."""
        cfg, st = self._bank.cfg[self._row], self._bank.state[self._row]
        cfg["loop_kind"], cfg["n_taps"], cfg["fs"] = self.LOOP_KIND, self.N_TAPS, self.rfSignal.samplingFrequency
        for key, field in self.CFG_KEYS.items():
            cfg[field] = float(configuration[key])
        for cfg_prefix, ini_prefix in self.FILTERS:
            cfg[cfg_prefix + "_tau1"], cfg[cfg_prefix + "_tau2"] = loop_filter_taus(
                float(configuration[ini_prefix + "_noise_bandwidth"]), float(configuration[ini_prefix + "_damping_ratio"]),
                float(configuration[ini_prefix + "_loop_gain"]))
        self._configure_taps(configuration, cfg)
        st["code_hz"] = GPS_L1CA_CODE_FREQ
        st["code_step"] = GPS_L1CA_CODE_FREQ / self.rfSignal.samplingFrequency
        st["n_samples"] = int(np.ceil((GPS_L1CA_CODE_SIZE_BITS - 0.0) / st["code_step"]))   # 4001 at 4 MHz
        self._initial_loop_state(st, cfg)
        self._bank.touch(self._row)

    def _configure_taps(self, configuration, cfg):
        raise NotImplementedError

    def _initial_loop_state(self, st, cfg):
        pass

    # ------------------------------------------------------------------ views that need more than a cast
    @property
    def channelState(self):
        return self._channel_state

    @channelState.setter
    def channelState(self, value):
        self._channel_state = value
        self._bank.tracking[self._row] = value is ChannelState.TRACKING
        ring = self._bank.ring                                 # (the manager caches its channel lists against this)
        ring.stateVersion = getattr(ring, "stateVersion", 0) + 1

    @property
    def currentSample(self):
        return int(self._bank.state["current_sample"][self._row]) % self.rfBuffer.maxSize

    @currentSample.setter
    def currentSample(self, value):
        self._bank.state["current_sample"][self._row] = int(value)
        self._bank.touch(self._row)

    @property
    def trackFlags(self):
        v = int(self._bank.state["track_flags"][self._row]) | int(self._bank.host_flags[self._row])
        return TrackingFlags(v) if v in TrackingFlags._value2member_map_ else v

    @trackFlags.setter
    def trackFlags(self, value):
        """The tracking loop's bits live on the device, the decoder's (HOST_FLAGS) on the host."""
        device_bits = int(value) & ~HOST_FLAGS
        self._bank.host_flags[self._row] = int(value) & HOST_FLAGS
        if device_bits != int(self._bank.state["track_flags"][self._row]):
            self._bank.state["track_flags"][self._row] = device_bits
            self._bank.touch(self._row)

    @property
    def tow(self):
        return self._bank.channel_tow(self._row)

    @tow.setter
    def tow(self, value):
        self._bank.tow[self._row], self._bank.tow_decoded[self._row] = float(value), bool(value)

    @property
    def codeSinceTOW(self):
        return int(self._bank.code_since_tow[self._row])

    @codeSinceTOW.setter
    def codeSinceTOW(self, value):
        self._bank.code_since_tow[self._row] = int(value)

    @property
    def correlatorsResults(self):
        return self._bank.last["corr"][self._row][:2 * self.N_TAPS]

    @property
    def navBits(self):
        return self._bank.nav_bits[self._row]

    @property
    def lostLock(self):
        """True once the device stopped this channel because its NCO left the staged replica / the ring."""
        return bool(self._bank.lost[self._row])

    # ------------------------------------------------------------------ satellite
    def setSatellite(self, satelliteID):
        super().setSatellite(satelliteID)
        self.systemID = GNSSSystems.GPS
        self.signalID = GNSSSignalType.GPS_L1_CA
        eng = self._ensure_code()                       # Gold code generated by the device LFSR kernel
        self._bank.state["code_slot"][self._row] = self.codeSlot
        chips = eng.read_code(self.codeSlot).astype(np.float64)
        self.code = np.r_[chips[-1], chips, chips[0]]   # padded table of kaplan:104-107, kept for API compatibility

    # ------------------------------------------------------------------ decoding seam (navdecoder.py)
    DECODER_PLUGIN = "kaplan"      # whose subframe logic the default (reference-backed) decoder drives

    def setDecoding(self, decoder="default"):
        """Choose what turns this channel's navigation bits into DECODING_UPDATE packets / `tow` / the TOW and EPH
        flags (kaplan:680-698 sets up the reference's own buffers at this point).  "default": the reference's own
        subframe logic when `sydr` is importable on this host, else None (bits only); None; or any NavDecoder."""
        if isinstance(decoder, str):
            decoder = default_decoder(self.channelID, self.DECODER_PLUGIN)
        self._bank.decoders[self._row] = decoder
        if decoder is not None:
            decoder.reset()

    @property
    def navDecoder(self):
        return self._bank.decoders[self._row]

    def runDecoding(self):
        """The seam of kaplan:702-726 / borre:455: the DECODING_UPDATE packet completed by the epoch `runTracking` just
        ran, or None.  (The bits are decided on the device and pushed through the decoder as they arrive -- bank.py
        `_new_bit`; a ChannelManager collects the packets of all channels itself.)"""
        mine = [item for item in self._bank.decoded if item[0] == self._row]
        if not mine:
            return None
        self._bank.decoded.remove(mine[0])
        return mine[0][2]

    def getTimeSinceTOW(self):
        ms = self.codeSinceTOW * GPS_L1CA_CODE_MS
        return ms + self.rfBuffer.getNbUnreadSamples(self.currentSample) / (self.rfSignal.samplingFrequency / 1e3)

    # ------------------------------------------------------------------ per-tick entry of ONE channel
    def _processHandler(self):
        """What `Channel.run()` calls when a channel is driven on its own; a ChannelManager batches instead."""
        if self.channelState == ChannelState.IDLE:
            raise Warning(f"Tracking channel {self.channelID} is in IDLE.")
        if self.channelState == ChannelState.ACQUIRING:
            packet = self.runAcquisition()
        elif self.channelState == ChannelState.TRACKING:
            return [p for p in (self.runTracking(), self.runDecoding()) if p is not None]
        else:
            raise ValueError(f"Channel state {self.channelState} is not valid.")
        return [] if packet is None else [packet]

    # ------------------------------------------------------------------ acquisition (host seams around sdr_pcps)
    def runAcquisition(self):
        if self.rfBuffer.getNbUnreadSamples(self.currentSample) < self.acq_requiredSamples:
            return None
        correlationMap = self.runSignalSearch()
        indices, ratio = self.runPeakFinder(correlationMap)
        self.postAcquisitionUpdate(indices)
        return self.prepareResultsAcquisition(correlationMap, indices, ratio)

    def searchedFrequency(self, bin_idx):
        """Frequency of Doppler bin `bin_idx` of this channel's search grid (np.arange(-range, range + 1, step))."""
        return -self.acq_dopplerRange + self.acq_dopplerSteps * bin_idx

    def enterTracking(self, carrier_hz: float, code_offset_samples: float):
        """From an acquisition result to the first tracking epoch: NCO carrier, and the sample at which the code period
        found at `code_offset_samples` into the searched slab begins (the searched samples are skipped,
        the first epoch is taken back, + offset + 1)."""
        self.codeOffset = int(np.round(code_offset_samples))
        self.carrierFrequency = carrier_hz
        first_epoch = int(self._bank.state["n_samples"][self._row])
        self.currentSample = self.currentSample + self.acq_requiredSamples - first_epoch + self.codeOffset + 1
        self.channelState = ChannelState.TRACKING

    def postAcquisitionUpdate(self, acqIndices):
        """PCPS peak [bin, code sample] -> NCO start values (kaplan:217-235 / borre:301-316)."""
        self.enterTracking(self.rfSignal.interFrequency - self.searchedFrequency(acqIndices[0]), acqIndices[1])

    def prepareResultsAcquisition(self, correlationMap, acqIndices, acqPeakRatio):
        packet = self.prepareResults()
        packet.update(type=ChannelMessage.ACQUISITION_UPDATE, carrierFrequency=self.carrierFrequency,
                      codeOffset=self.codeOffset, frequency_idx=acqIndices[0], code_idx=acqIndices[1],
                      correlation_map=correlationMap, peak_ratio=acqPeakRatio)
        return packet

    # ------------------------------------------------------------------ tracking = one device step
    def runCorrelators(self):
        """The correlator seam (kaplan:378-401), kept for function-level use: the taps of the NEXT epoch for the
        current NCO state, open loop -- the state does not advance (runTracking / the manager's tick do that)."""
        return self._correlate()

    def runTracking(self):
        if self.lostLock or self.rfBuffer.getNbUnreadSamples(self.currentSample) < self.track_requiredSamples:
            return None
        rec, done = self._bank.step([self._row], 1)
        return tracking_packet(self.channelID, self.LOOP_KIND, rec[0, 0]) if done[0] else None
