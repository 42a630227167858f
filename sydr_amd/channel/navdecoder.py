"""The decoder seam: from the device's navigation bits to DECODING_UPDATE packets, `tow` and the TOW / EPH flags.

The reference's plugins follow every tracking epoch with `runDecoding` (channel_l1ca_kaplan.py:702-726,
channel_l1ca_borre.py:455-579): twenty prompt values make a bit (`decodeBit`), the bits are searched for the
TLM preamble with good parity on the first two words, two preambles 300 bits apart make SUBFRAME_SYNC, and
every complete subframe leaves as a DECODING_UPDATE packet (`cid, type, subframe_id, tow, bits`) which the
reference's receiver feeds to its ephemeris decoder (receiver_gps_l1ca.py:110-126); the CHANNEL_UPDATE of the
same tick carries the new `tow`, `code_since_tow = 0` and the TOW_DECODED / TOW_KNOWN / EPH_DECODED / EPH_KNOWN
flags from which the receiver forms pseudoranges.

Here `decodeBit` runs on the GPU (track.hip: the bit leaves the device in the epoch record that completes it).
Everything after the bit is 50 bit/s host work that BASELINE.json keeps untouched (`sydr/dsp/decoding.py` is NOT
re-implemented in this package): a device-tracked channel hands each bit to a `NavDecoder`:

    class NavDecoder(Protocol):
        def push(self, bit: int, track_flags: int) -> tuple[int, DecodedSubframe | None]: ...
        def reset(self) -> None: ...

`push` receives the bit (0/1) decided in an epoch and the channel's TrackingFlags as of that epoch, and returns
the flags as the decoder leaves them (it may set / clear SUBFRAME_SYNC, TOW_DECODED, TOW_KNOWN, EPH_DECODED,
EPH_KNOWN -- `HOST_FLAGS`; the device's CODE_LOCK / BIT_SYNC bits are passed through) plus, when the bit
completed a subframe, what the packet and the channel need.  `ReferencePluginDecoder` is the adapter that drives
the reference's OWN `decodeSubframe` / `postDecodingUpdate` / `prepareResultsDecoding` (Kaplan) or `runDecoding`
(Borre) on a state-only instance of the reference's plugin class -- imported lazily, on the host, only when a
channel asks for it; any other decoder (a C library, a vendor's frame synchroniser) implements the two methods.
"""
from __future__ import annotations

import importlib
from dataclasses import dataclass

from ..utils.enumerations import TrackingFlags

# flags owned by the decoder (kept on the host); the rest belong to the tracking loop on the device
HOST_FLAGS = int(TrackingFlags.SUBFRAME_SYNC | TrackingFlags.TOW_DECODED | TrackingFlags.EPH_DECODED
                 | TrackingFlags.TOW_KNOWN | TrackingFlags.EPH_KNOWN)


@dataclass
class DecodedSubframe:
    subframe_id: int
    tow: object          # the packet's `tow` (kaplan:865: int(self.tow) AFTER the alignment; borre:562: the HOW's value)
    bits: str            # the packet's `bits`: 300 characters, polarity corrected
    channel_tow: float   # what `Channel.tow` becomes: the HOW's TOW + the bits already received of the next subframe


def reference_available(plugin: str = "kaplan") -> bool:
    """True when the reference's plugin module (and with it sydr.dsp.decoding) imports on this host."""
    try:
        importlib.import_module(ReferencePluginDecoder._PLUGINS[plugin][0])
        return True
    except Exception:
        return False


class ReferencePluginDecoder:
    """Drives the reference's own subframe logic.  `plugin` = "kaplan" (decodeSubframe / postDecodingUpdate /
    prepareResultsDecoding, kaplan:756-868) or "borre" (the monolithic runDecoding, borre:455-579).

    The reference keeps this logic in methods of its channel class.  An instance made with `object.__new__` (no
    `__init__`: no process, no buffers, no configuration) carries exactly the attributes those methods read and
    write; nothing of the reference is copied or re-stated here."""

    _PLUGINS = {"kaplan": ("sydr.channel.channel_l1ca_kaplan", "ChannelL1CA_Kaplan"),
                "borre": ("sydr.channel.channel_l1ca_borre", "ChannelL1CA")}

    def __init__(self, cid: int, plugin: str = "kaplan"):
        module, name = self._PLUGINS[plugin]
        self._cls = getattr(importlib.import_module(module), name)
        self._const = importlib.import_module("sydr.utils.constants")
        self._flags_type = importlib.import_module("sydr.utils.enumerations").TrackingFlags
        self.plugin, self.cid = plugin, int(cid)
        self.reset()

    def reset(self):
        import numpy as np
        ref = object.__new__(self._cls)
        ref.channelID = self.cid
        ref.trackFlags = self._flags_type.UNKNOWN
        ref.codeSinceTOW = 0
        if self.plugin == "kaplan":
            self._cls.setDecoding(ref)
        else:                                     # the attributes borre:133-139 sets up inside setTracking
            c = self._const
            ref.navBitBufferSize = c.LNAV_SUBFRAME_SIZE + 2 * c.LNAV_WORD_SIZE + 2
            ref.navBitsBuffer = np.squeeze(np.empty((1, ref.navBitBufferSize), dtype=int))
            ref.navBitsCounter, ref.preambuleFound, ref.tow = 0, False, 0
            ref.subframeFlags = [False] * 5
            ref.IDX_I_PROMPT, ref.nbPrompt = 2, 1
            ref.correlatorsBuffer = np.zeros((1, 6))
        self._ref = ref

    def push(self, bit: int, track_flags: int):
        ref = self._ref
        ref.trackFlags = int(track_flags)
        if self.plugin == "kaplan":
            ref.navBitsBuffer[ref.navBitsCounter] = bit
            ref.navBitsCounter += 1
            if not (self._cls.decodeSubframe(ref) and self._cls.postDecodingUpdate(ref)):
                return int(ref.trackFlags), None
            pkt = self._cls.prepareResultsDecoding(ref)
        else:
            # borre:470-491 makes the bit itself from 20 prompts: hand it ONE prompt of the decided sign as the 20th
            ref.navPromptSum, ref.navPromptSumCounter = 0.0, self._const.LNAV_MS_PER_BIT - 1
            ref.correlatorsBuffer[0, ref.IDX_I_PROMPT] = 1.0 if bit else -1.0
            pkt = self._cls.runDecoding(ref)
            if pkt is None:
                return int(ref.trackFlags), None
        return int(ref.trackFlags), DecodedSubframe(int(pkt["subframe_id"]), pkt["tow"], pkt["bits"], ref.tow)


def default_decoder(cid: int, plugin: str):
    """What a device-tracked channel uses unless told otherwise: the reference's own logic when the reference is
    installed next to this package, else nothing (the channel then delivers `navBits` only and says so once)."""
    return ReferencePluginDecoder(cid, plugin) if reference_available(plugin) else None


__all__ = ["HOST_FLAGS", "DecodedSubframe", "ReferencePluginDecoder", "default_decoder", "reference_available"]
