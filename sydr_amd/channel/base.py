"""Channel plugin base class -- the drop-in boundary on the host side.

Mirrors `Channel` of the reference (sydr/channel/channel.py:21-228): same constructor signature,
attributes, `setSatellite`, abstract `_processHandler`, `prepareResults`, `prepareChannelUpdate`
and packet keys.  What differs, on purpose: a channel is a plain object, not a
`multiprocessing.Process` -- a HIP context does not survive `fork`, and one process per GPU drives
all of that GPU's channels.  `run()` therefore processes ONE manager tick instead of looping on
events; the per-millisecond barrier of channelManager.py:164-171 becomes a plain loop.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

from ..utils.enumerations import ChannelMessage, ChannelState, TrackingFlags


class Channel(ABC):
    TIMEOUT = 100000

    @abstractmethod
    def __init__(self, cid, sharedBuffer, resultQueue, rfSignal, configuration):
        self.configuration = configuration
        self.channelID = cid
        self.channelState = ChannelState.IDLE
        self.satelliteID = 0
        self.rfBuffer = sharedBuffer          # device ring shared by every channel of this GPU
        self.resultQueue = resultQueue        # optional: anything with .put(); the manager may pass None
        self.currentSample = 0
        self.rfSignal = rfSignal
        self.trackFlags = TrackingFlags.UNKNOWN
        self.tow = 0
        self.week = 0
        self.codeSinceTOW = 0
        self.name = f'CID{cid}'

    def setSatellite(self, satelliteID: int):
        self.satelliteID = satelliteID
        self.channelState = ChannelState.ACQUIRING

    def start(self):
        """Kept so `ChannelManager.requestTracking` reads like the reference's; nothing to fork."""
        return None

    def run(self):
        """One manager tick: process, append the channel update, hand the packets over."""
        results = self._processHandler()
        results.append(self.prepareChannelUpdate())
        if self.resultQueue is not None:
            self.resultQueue.put(results)
        return results

    @abstractmethod
    def _processHandler(self):
        return

    def getTimeSinceTOW(self):
        return 0

    def prepareResults(self):
        return {"cid": self.channelID}

    def prepareChannelUpdate(self):
        packet = self.prepareResults()
        packet['type'] = ChannelMessage.CHANNEL_UPDATE
        packet['state'] = self.channelState
        packet['tracking_flags'] = self.trackFlags
        packet['tow'] = self.tow
        packet['time_since_tow'] = self.getTimeSinceTOW()
        packet['unprocessed_samples'] = self.rfBuffer.getNbUnreadSamples(self.currentSample)
        packet['code_since_tow'] = self.codeSinceTOW
        return packet


class ChannelStatus(ABC):
    """Receiver-side mirror of a channel's status (sydr/channel/channel.py:232-263)."""

    def __init__(self, channelID: int, satelliteID: int):
        self.channelID = channelID
        self.satelliteID = satelliteID
        self.channelState = ChannelState.IDLE
        self.trackFlags = TrackingFlags.UNKNOWN
        self.week = 0
        self.tow = 0
        self.timeSinceTOW = 0
        self.subframeFlags = []
        self.unprocessedSamples = 0
        self.isTOWDecoded = False
