"""ChannelManager with the reference's surface (sydr/channel/channelManager.py:34-231) -- `addChannel`,
`requestTracking`, `addNewRFData`, `run`, `getChannel`, `close` -- and none of its plumbing:
no shared memory, no child processes, no Events, no pickled Queue.  One manager
drives the channels of ONE GPU in-process; the RF ring lives in HBM.

`run()` is the per-millisecond tick of the reference, but batched: every channel that is ready for
an epoch is correlated in ONE `sdr_epl_batch` launch, every channel ready for acquisition in ONE
`sdr_pcps` call; then each channel finishes its scalar bookkeeping and emits the reference's packets.

`runBlock(n)` is the fast path the reference has no counterpart for: n epochs of closed-loop
tracking for all tracking channels inside one persistent kernel (loop closure on the device),
returning the same per-epoch packets afterwards.
"""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, FMT_CI16, FMT_CI8, Engine, make_items
from ..utils.devicering import CircularBuffer
from ..utils.enumerations import ChannelState
from . import loopstate


def shard_channels(n_items: int, rank: int, world_size: int) -> list[int]:
    """Indices of the channels (PRNs) rank `rank` owns: contiguous blocks, sizes differing by at most 1.
    Channels are independent, so this is the whole multi-GPU story: no collective on the data path."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside world")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


class ChannelManager:
    TIMEOUT = 1

    def __init__(self, rfSignal, engine: Engine | None = None, device_id: int = 0, keepCorrelationMap: bool = True):
        self.rfSignal = rfSignal
        self.channels = {}
        self.nbChannels = 0
        if engine is None:
            from ..runtime import get_engine
            engine = get_engine(device_id)
        self.engine = engine
        buffersize = int(self.rfSignal.samplingFrequency * 1e-3 * 100)   # 100 ms, as channelManager.py:57
        fmt = {np.int8: FMT_CI8, np.int16: FMT_CI16}.get(getattr(rfSignal, "fileDataType", None), FMT_CF64)
        self.sharedBuffer = CircularBuffer(buffersize, rfSignal.dtype, engine=engine, fmt=fmt)
        self.resultQueue = None
        self.keepCorrelationMap = keepCorrelationMap
        self._slots = 0

    # ------------------------------------------------------------------ reference surface
    def addChannel(self, ChannelObject, configuration, nbChannels=1):
        total = self.nbChannels + nbChannels
        if total > self._slots:
            self._slots = max(32, total)
            self.engine.code_slots(self._slots)
        for _ in range(nbChannels):
            cid = self.nbChannels
            ch = ChannelObject(cid, self.sharedBuffer, self.resultQueue, self.rfSignal, configuration)
            ch.codeSlot = cid
            self.channels[cid] = ch
            self.nbChannels += 1

    def requestTracking(self, satelliteID: int):
        for channel in self.channels.values():
            if channel.channelState is ChannelState.IDLE:
                channel.setSatellite(satelliteID)
                channel.start()
                return channel
        raise Warning(f"Could not find an IDLE channel for tracking satellite [G{satelliteID}].")

    def addNewRFData(self, data):
        self.sharedBuffer.shift(data)

    def getChannel(self, channelID):
        if channelID not in self.channels:
            raise ValueError("Channel ID does not exist.")
        return self.channels[channelID]

    def close(self):
        self.channels.clear()

    # ------------------------------------------------------------------ the tick
    def run(self):
        """Flat list of result packets for this tick (channelManager.py:149-188)."""
        active = [ch for ch in self.channels.values() if ch.channelState is not ChannelState.IDLE]
        self._batchAcquisition(active)
        self._batchCorrelators(active)
        results = []
        for ch in active:
            packets = ch._processHandler()
            packets.append(ch.prepareChannelUpdate())
            results.extend(packets)
        return results

    def _batchAcquisition(self, active):
        groups = {}
        for ch in active:
            if ch.channelState is ChannelState.ACQUIRING and hasattr(ch, "acquisitionRequest") \
                    and self.sharedBuffer.getNbUnreadSamples(ch.currentSample) >= ch.acq_requiredSamples:
                ch._ensure_code()
                r = ch.acquisitionRequest()
                key = (r["start"], r["fs"], r["if_hz"], r["doppler_range"], r["doppler_step"], r["coh"], r["noncoh"])
                groups.setdefault(key, []).append(ch)
        for key, chans in groups.items():
            start, fs, if_hz, rng, step, coh, noncoh = key
            pb, pc, pr, cmap = self.engine.pcps([c.codeSlot for c in chans], start, fs, if_hz, rng, step, coh, noncoh,
                                                want_map=self.keepCorrelationMap)
            for k, ch in enumerate(chans):
                ch._injectedAcquisition = (cmap[k] if cmap is not None else None, [int(pb[k]), int(pc[k])], float(pr[k]))

    def _batchCorrelators(self, active):
        groups = {}
        for ch in active:
            if ch.channelState is ChannelState.TRACKING and hasattr(ch, "correlatorRequest"):
                req = ch.correlatorRequest()
                if req is not None:
                    groups.setdefault(req["spacing"], []).append((ch, req))
        for spacing, entries in groups.items():
            reqs = [r for _, r in entries]
            items = make_items([r["code_slot"] for r in reqs], [r["n_samples"] for r in reqs],
                               [r["start_sample"] for r in reqs], [r["carrier_hz"] for r in reqs],
                               [r["rem_carrier"] for r in reqs], [r["rem_code"] for r in reqs],
                               [r["code_step"] for r in reqs])
            out = self.engine.epl_batch(items, spacing, self.rfSignal.samplingFrequency)
            for (ch, _), row in zip(entries, out):
                ch._injectedCorrelators = row

    # ------------------------------------------------------------------ closed loop on the device
    def runBlock(self, nbEpochs: int):
        """Track every TRACKING channel for `nbEpochs` code periods inside one persistent kernel.

        The ring must already hold the samples those epochs will consume (a long resident ring, or
        a block of milliseconds added beforehand).  Returns the per-epoch TRACKING_UPDATE packets,
        channel by channel, epoch by epoch, followed by one CHANNEL_UPDATE per channel."""
        chans = [ch for ch in self.channels.values() if ch.channelState is ChannelState.TRACKING]
        if not chans:
            return []
        for ch in chans:
            need = nbEpochs * (int(ch.track_requiredSamples) + 1)
            if self.sharedBuffer.getNbUnreadSamples(ch.currentSample) < need:
                raise ValueError(f"runBlock({nbEpochs}): channel {ch.channelID} needs ~{need} unread samples in the "
                                 f"ring, has {self.sharedBuffer.getNbUnreadSamples(ch.currentSample)}")
        kinds ={loopstate.loop_kind(ch) for ch in chans}
        results = []
        for kind in sorted(kinds):
            group = [ch for ch in chans if loopstate.loop_kind(ch) == kind]
            cfg = loopstate.export_cfg(group[0])
            states = [loopstate.export_state(ch) for ch in group]
            states, traj, bits = self.engine.track_closed_loop(states, cfg, nbEpochs, want_traj=True, want_bits=True)
            for ch, st, tr, nav in zip(group, states, traj, bits):
                results.extend(loopstate.tracking_packet(ch, rec) for rec in tr)
                loopstate.import_state(ch, st, nbEpochs, last=tr[-1])
                if hasattr(ch, "navBits"):
                    ch.navBits.extend(int(b) for b in nav)   # bits decided on the device: 1 byte per 20 ms per channel
        results.extend(ch.prepareChannelUpdate() for ch in chans)
        return results
