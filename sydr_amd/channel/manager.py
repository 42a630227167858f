"""ChannelManager with the reference's surface (sydr/channel/channelManager.py:34-231) -- `addChannel`,
`requestTracking`, `addNewRFData`, `run`, `getChannel`, `close` -- and none of its plumbing: no shared
memory, no child processes, no Events, no pickled Queue.  One manager drives the channels
of ONE GPU in-process; the RF ring and the channels' tracking state both live in HBM.

A tick (`addNewRFData(slab)` + `run()`, the body of the receiver's outer loop, receiver.py:120-131) is
ONE device call: the slab goes into the ring and every tracking channel whose next epoch is complete
advances by one epoch on the device (`sdr_bank_tick`).  Channels still acquiring are searched together in
one `sdr_pcps` call.  `runBlock(n)` advances by up to n epochs per channel in one launch (loops closed
on the device for the whole block).

Multi-GPU (north_star: channels shard across the GPUs of a node, IQ replicated, no collective): give each
process / device its own manager with `channels=shard_channels(total, rank, world)`.
"""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, FMT_CI16, FMT_CI8, Engine
from ..utils.devicering import CircularBuffer
from ..utils.enumerations import ChannelState
from .bank import TickPackets, channel_update_builder, tracking_packet, tracking_packets_builder
from .tracked import DeviceTrackedChannel


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
    DEFER_BYTES = 1 << 20         # slabs up to this size ride in the next run()'s device call; longer ones upload at once

    def __init__(self, rfSignal, engine: Engine | None = None, device_id: int = 0, keepCorrelationMap: bool = True,
                 ring_ms: int = 100):
        self.rfSignal = rfSignal
        self.channels = {}
        self.nbChannels = 0
        if engine is None:
            from ..runtime import get_engine
            engine = get_engine(device_id)
        self.engine = engine
        buffersize = int(self.rfSignal.samplingFrequency * 1e-3 * ring_ms)   # 100 ms, as channelManager.py:57
        fmt = {np.int8: FMT_CI8, np.int16: FMT_CI16}.get(getattr(rfSignal, "fileDataType", None), FMT_CF64)
        self.sharedBuffer = CircularBuffer(buffersize, rfSignal.dtype, engine=engine, fmt=fmt)
        self.resultQueue = None
        self.keepCorrelationMap = keepCorrelationMap
        self._slots = 0
        self._pending = None          # (owned copy of the slab handed to addNewRFData, ring offset): uploaded by the next run()
        self._stage_buf = None        # the host buffer those copies live in (re-used from tick to tick)
        self._lists = None            # (state version, active, acquiring, host-side plugins, cids, states) of the last tick

    @property
    def bank(self):
        return self.sharedBuffer.channelBank

    # ------------------------------------------------------------------ reference surface
    def addChannel(self, ChannelObject, configuration, nbChannels=1):
        total = self.nbChannels + nbChannels
        if total > self._slots:
            self._flush_pending()
            self._slots = max(32, total)
            self.engine.code_slots(self._slots)
            for ch in self.channels.values():        # the tables were re-allocated: stage the codes again
                if ch.channelState is not ChannelState.IDLE:
                    ch._stagedPrn = None
                    ch._ensure_code()
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
        """Queue one slab for the ring.  The copy itself rides in the next run()'s device call (one call per tick);
        anything that looks at the ring before that (another addNewRFData, getSlice, runBlock) flushes it first."""
        self._flush_pending()
        staged, offset, count = self.sharedBuffer.stage(data)
        self._guard_unread(count)
        if staged.nbytes > self.DEFER_BYTES:
            self.engine.iq_upload(staged, offset)         # a long slab gains nothing from riding in the tick's call
        else:
            # the reference copies at this point (circularbuffer.py:54-82): keep an OWNED copy, so that a caller who
            # reuses its buffer before run() cannot change what enters the ring
            if self._stage_buf is None or self._stage_buf.dtype != staged.dtype or self._stage_buf.size < staged.size:
                self._stage_buf = np.empty(max(staged.size, 1), dtype=staged.dtype)
            own = self._stage_buf[:staged.size]
            np.copyto(own, staged.reshape(-1))
            self._pending = (own, offset)
        self.sharedBuffer.shiftIdxWrite(count)

    def _guard_unread(self, count: int):
        """Refuse to overwrite samples a tracking channel has not consumed yet (the reference would silently wrap:
        circularbuffer.py:54-82 -- and then track garbage)."""
        bank = self.bank
        if bank is None or not self.sharedBuffer.full:
            return
        idx = np.flatnonzero(bank.tracking & ~bank.lost)
        if idx.size and int(bank.unread(idx).max()) + count > self.sharedBuffer.maxSize:
            ch = int(idx[int(np.argmax(bank.unread(idx)))])
            raise ValueError(f"addNewRFData: {count} more samples would overwrite what channel {ch} has not read yet "
                             f"({int(bank.unread(idx).max())} unread of a {self.sharedBuffer.maxSize}-sample ring); "
                             "run the channels first")

    def _flush_pending(self):
        if self._pending is not None:
            staged, offset = self._pending
            self._pending = None
            self.engine.iq_upload(staged, offset)

    def getChannel(self, channelID):
        if channelID not in self.channels:
            raise ValueError("Channel ID does not exist.")
        return self.channels[channelID]

    def close(self):
        self._flush_pending()
        self.channels.clear()
        if self.bank is not None:
            self.bank.close()
            self.sharedBuffer.channelBank = None

    # ------------------------------------------------------------------ the tick
    def run(self):
        """Flat sequence of result packets for this tick (channelManager.py:149-188)."""
        out = TickPackets()
        # who is active / acquiring only changes when a channel changes state: device-tracked channels bump the ring's
        # stateVersion when they do, so a tracking receiver does not walk its channel objects every millisecond
        version = getattr(self.sharedBuffer, "stateVersion", None)
        if self._lists is not None and self._lists[0] == (version, self.nbChannels):
            _, active, acquiring, host_plugins, cids_active, states_active = self._lists
        else:
            active = [ch for ch in self.channels.values() if ch.channelState is not ChannelState.IDLE]
            acquiring = [ch for ch in active if ch.channelState is ChannelState.ACQUIRING]
            host_plugins = [ch for ch in active if not isinstance(ch, DeviceTrackedChannel)]
            cids_active = np.array([ch.channelID for ch in active], dtype=np.int64)
            states_active = [ch.channelState for ch in active]
            cacheable = version is not None and all(isinstance(ch, DeviceTrackedChannel) for ch in self.channels.values())
            self._lists = ((version, self.nbChannels), active, acquiring, host_plugins, cids_active, states_active) if cacheable else None
        if not active:
            self._flush_pending()
            return out
        bank = self.bank
        ready = bank.ready() if bank is not None else np.zeros(0, dtype=np.int32)
        # one device call: ring ingest + one epoch for every ready channel
        staged, offset = self._pending if self._pending is not None else (None, 0)
        try:
            if bank is not None:
                rec, done = bank.tick(staged, offset, ready)
            elif staged is not None:
                self.engine.iq_upload(staged, offset)
            self._pending = None
        except Exception:
            # the write index already counts this slab: whatever stopped the tick, its samples must still reach the ring
            # (uploading them twice is harmless) before anybody reads it again
            try:
                self._flush_pending()
            finally:
                self._pending = None
            raise
        if acquiring:
            out.add_ready(self._acquire(acquiring))
        for ch in host_plugins:   # plugins that keep their loops on the host (e.g. the reference's class behind the seams mixin)
            if ch.channelState is ChannelState.TRACKING:
                out.add_ready(ch._processHandler())
        if len(ready):
            ran = np.flatnonzero(done > 0)
            cids, kinds, rec = ready[ran], bank.cfg["loop_kind"][ready[ran]], rec[ran]
            out.add(len(ran), tracking_packets_builder(cids, kinds, rec))
            if bank.decoded:                                  # subframes completed by this tick's bits (kaplan:71-73)
                out.add_ready(pkt for _, _, pkt in bank.take_decoded())
        # channel updates: everything they report is captured now, the dicts are made when read (acquisition may have
        # moved channels to TRACKING during this tick: take the lists again if it did)
        if getattr(self.sharedBuffer, "stateVersion", None) != version or self._lists is None:
            cids = np.array([ch.channelID for ch in active], dtype=np.int64)
            states = [ch.channelState for ch in active]
        else:
            cids, states = cids_active, states_active
        if bank is not None:
            unread = bank.unread(cids)
            since = bank.code_since_tow[cids] * 1 + unread / (self.rfSignal.samplingFrequency / 1e3)
            out.add(len(active), channel_update_builder(cids, states, bank.flags(cids), bank.tow[cids].copy(),
                                                        bank.tow_decoded[cids].copy(), since, unread,
                                                        bank.code_since_tow[cids].copy()))
        else:
            out.add_ready(ch.prepareChannelUpdate() for ch in active)
        return out

    def _acquire(self, acquiring):
        """Channels with enough samples for their search, grouped by search geometry: ONE sdr_pcps call per group
        (the forward transforms of the Doppler-mixed slab are shared by all PRNs).  Plugins that replace the
        search seam (e.g. the SerialSearch plugin) run their own."""
        packets, groups = [], {}
        for ch in acquiring:
            if self.sharedBuffer.getNbUnreadSamples(ch.currentSample) < ch.acq_requiredSamples:
                continue
            if getattr(type(ch), "runSignalSearch", None) is not _default_search():
                packets.extend(ch._processHandler())
                continue
            ch._ensure_code()
            r = ch.acquisitionRequest()
            key = (r["start"], r["fs"], r["if_hz"], r["doppler_range"], r["doppler_step"], r["coh"], r["noncoh"])
            groups.setdefault(key, []).append(ch)
        for (start, fs, if_hz, rng, step, coh, noncoh), chans in groups.items():
            pb, pc, pr, cmap = self.engine.pcps([c.codeSlot for c in chans], start, fs, if_hz, rng, step, coh, noncoh,
                                                want_map=self.keepCorrelationMap)
            for k, ch in enumerate(chans):
                ch._injectedAcquisition = (cmap[k] if cmap is not None else None, [int(pb[k]), int(pc[k])], float(pr[k]))
                packets.append(ch.runAcquisition())
        return [p for p in packets if p is not None]

    # ------------------------------------------------------------------ many epochs per call
    def runBlock(self, nbEpochs: int):
        """Up to `nbEpochs` code periods for every TRACKING channel inside one persistent launch.

        Each channel runs as many whole epochs as the ring already holds for it (at most nbEpochs): nothing is
        read that has not been written.  Returns the per-epoch TRACKING_UPDATE packets, channel by channel, epoch
        by epoch, followed by one CHANNEL_UPDATE per tracking channel."""
        self._flush_pending()
        out = TickPackets()
        bank = self.bank
        chans = [ch for ch in self.channels.values() if ch.channelState is ChannelState.TRACKING and not ch.lostLock]
        if not chans or bank is None:
            return out
        cids = np.array([ch.channelID for ch in chans], dtype=np.int32)
        # epochs the ring holds: every epoch is at most n_samples + 1 long (the NCO moves it by a sample at most)
        budget = bank.unread(cids) // (bank.state["n_samples"][cids] + 1)
        groups = {}
        for cid, n in zip(cids, np.minimum(budget, nbEpochs)):
            if n > 0:
                groups.setdefault((int(n), int(bank.cfg["n_taps"][cid])), []).append(int(cid))
        for (n, _), members in sorted(groups.items()):
            members = np.array(members, dtype=np.int32)
            rec, done = bank.step(members, n)
            kinds = bank.cfg["loop_kind"][members]
            decoded = {(c, e): pkt for c, e, pkt in bank.take_decoded()}
            flat = []
            for r, (c, k) in enumerate(zip(members, kinds)):
                for e in range(done[r]):
                    flat.append((int(c), int(k), rec[r, e]))
                    if decoded and (int(c), e) in decoded:    # a subframe completed by this epoch's bit follows it
                        flat.append(decoded[(int(c), e)])
            out.add(len(flat), lambda i, flat=flat: flat[i] if isinstance(flat[i], dict) else tracking_packet(*flat[i]))
        out.add_ready(ch.prepareChannelUpdate() for ch in chans)
        return out


def _default_search():
    from .seams import GpuCorrelatorSeams
    return GpuCorrelatorSeams.runSignalSearch


__all__ = ["ChannelManager", "shard_channels", "FMT_CF64"]
