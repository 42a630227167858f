"""ChannelManager with the reference's surface (sydr/channel/channelManager.py:34-231) -- `addChannel`,
`requestTracking`, `addNewRFData`, `run`, `getChannel`, `close` -- and none of its plumbing: no shared
memory, no child processes, no Events, no pickled Queue.  One manager drives the channels
of ONE GPU in-process; the RF ring and the channels' tracking state both live in HBM.

A tick (`addNewRFData(slab)` + `run()`, the body of the receiver's outer loop, receiver.py:120-131) is
ONE device call: the slab goes into the ring and every tracking channel whose next epoch is complete
advances by one epoch on the device (`sdr_bank_tick`).  Channels still acquiring are searched together in
one `sdr_pcps` call.  `runBlock(n)` advances by up to n epochs per channel in one launch (loops closed
on the device for the whole block).

Multi-GPU (north_star: channels shard across the GPUs of a node, IQ replicated, no collective):
`ChannelManager(rfSignal, devices=[0, 1, ..., 7])` is ONE manager over the GPUs of a node in one process (multidevice.py:
a manager like this one per device behind the same surface, every device's tick begun before any is waited for); a
job of one process per GPU (bench.py under torch.distributed) gives each rank its own manager and
`shard_channels(total, rank, world)` of the channels.
"""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, FMT_CI16, FMT_CI8, Engine
from ..utils.devicering import CircularBuffer
from ..utils.enumerations import ChannelState
from ..utils.enumerations import ChannelMessage
from .bank import TickPackets, TrackingRows, UpdateRows, packet_templates, tracking_packet
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
    DEFER_BYTES = 1 << 20         # slabs up to this size are queued for the ring without waiting; longer ones upload at once
    STEADY_TICK = True            # all active channels tracking on the device: the tick is one sdr_bank_tick_mirrored call
    PREFETCH = True               # read-ahead: the next block is queued on the device while the current one is handed out

    def __new__(cls, rfSignal=None, *args, devices=None, engines=None, **kwargs):
        # ChannelManager(rfSignal, devices=[0, 1, ..., 7]) -- ONE manager, as the reference's receiver builds it
        # (receiver.py:86), over the GPUs of a node: multidevice.py
        if cls is ChannelManager and (engines is not None or (devices is not None and len(devices) != 1)):
            from .multidevice import MultiDeviceChannelManager
            return MultiDeviceChannelManager(rfSignal, *args, devices=devices, engines=engines, **kwargs)
        return super().__new__(cls)

    def __init__(self, rfSignal, engine: Engine | None = None, device_id: int | None = None, keepCorrelationMap: bool = True,
                 ring_ms: int = 100, devices=None, engines=None):
        if devices is not None:
            device_id = int(devices[0])
        self.rfSignal = rfSignal
        self.channels = {}
        self.nbChannels = 0
        if engine is None:
            from ..runtime import get_engine
            engine = get_engine(device_id)
        self.engine = engine
        self._engine_queues_slabs = hasattr(engine, "iq_upload_begin")      # (asked once, not every millisecond)
        buffersize = int(self.rfSignal.samplingFrequency * 1e-3 * ring_ms)   # 100 ms, as channelManager.py:57
        fmt = {np.int8: FMT_CI8, np.int16: FMT_CI16}.get(getattr(rfSignal, "fileDataType", None), FMT_CF64)
        self.sharedBuffer = CircularBuffer(buffersize, rfSignal.dtype, engine=engine, fmt=fmt)
        self.resultQueue = None
        self.keepCorrelationMap = keepCorrelationMap
        self._slots = 0
        self._readahead = None        # EpochSchedule once enableReadAhead() was called (readahead.py)
        self._ra_ms = 0
        self._pending = False         # a slab's transfer into the ring is queued on the engine's stream, not waited for
        self._lists = None            # (state version, active, acquiring, host-side plugins, cids, states, ...) of the last tick
        self._unread_max = None       # most unread samples of any running channel after the last tick, when known
        self._ahead = None            # read-ahead: the block queued on the device after the one being handed out (_track_ahead)
        self._samples_per_ms = self.rfSignal.samplingFrequency / 1e3

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

    def _addChannelsAt(self, cids, ChannelObject, configuration):
        """addChannel for a manager that is one DEVICE of a multi-device manager (multidevice.py): the channels get the
        numbers the whole receiver knows them by; rows of this device's bank and code table that belong to channels of other
        devices stay empty."""
        cids = [int(c) for c in cids]
        if not cids:
            return
        need = max(cids) + 1
        if need > self._slots:
            self._flush_pending()
            self._slots = max(32, need)
            self.engine.code_slots(self._slots)
            for ch in self.channels.values():
                if ch.channelState is not ChannelState.IDLE:
                    ch._stagedPrn = None
                    ch._ensure_code()
        for cid in cids:
            if cid in self.channels:
                raise ValueError(f"channel {cid} exists already")
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

    def enableReadAhead(self, nbMilliseconds: int = 50):
        """Serve the per-millisecond loop from blocks computed ahead (readahead.py): whenever every active channel is
        tracking and `rfSignal` is this package's file reader, the next `nbMilliseconds` of the recording are uploaded
        and tracked in one launch, and the following ticks hand out what they would have computed.  The reference's
        calls (`addNewRFData(rfSignal.getMilliseconds(1)); run()`) stay as they are; a slab that is not the
        recording's next millisecond is refused while a block is being replayed.  0 switches it off (once the
        epochs already computed have been handed out)."""
        from .readahead import EpochSchedule
        self._ra_ms = max(0, int(nbMilliseconds))
        self._EpochSchedule = EpochSchedule

    def _recording_position(self, data):
        """Sample index of `data` inside the recording when it is the slab `rfSignal` handed out last (or equals it
        byte for byte), else None."""
        rec_of = getattr(self.rfSignal, "_recording", None)
        if rec_of is None or not getattr(self.rfSignal, "filepath", None) or not isinstance(data, np.ndarray):
            return None
        try:
            rec = rec_of()
        except OSError:
            return None
        n = data.size // 2
        first = int(self.rfSignal.position) - n
        if first < 0 or data.dtype != rec.dtype or data.ndim != 1:
            return None
        there = rec[2 * first:2 * (first + n)]
        same = data.__array_interface__["data"][0] == there.__array_interface__["data"][0] or np.array_equal(data, there)
        return first if same else None

    def _open_window(self, data) -> bool:
        """Track the next block ahead if everything allows it (or take over the one that was started while the last block
        was being handed out); True when `data` was consumed that way."""
        ra, bank, ring = self._readahead, self.bank, self.sharedBuffer
        spt = ra.spt
        ahead = self._ahead
        if ahead is not None and ahead["bank"] is not bank:
            ahead = self._ahead = None                        # (the bank was re-created from the mirror: it starts the block again)
        if data.size != 2 * spt or self.nbChannels == 0:
            if ahead is not None:
                raise ValueError("addNewRFData: a block of the recording is being tracked ahead; the slab must be its next millisecond")
            return False
        first = self._recording_position(data)
        if ahead is not None:
            if first != ahead["first"]:
                raise ValueError("addNewRFData: a block of the recording is being tracked ahead; the slab must be the recording's "
                                 f"next millisecond (sample {ahead['first']}); call enableReadAhead(0) and finish the block to feed other data")
            cids, k, block = ahead["cids"], ahead["k"], ahead["block"]
        else:
            chans = [ch for ch in self.channels.values() if ch.channelState is not ChannelState.IDLE]
            if not chans or any(not isinstance(ch, DeviceTrackedChannel) or ch.channelState is not ChannelState.TRACKING
                                or ch.lostLock for ch in chans):
                return False
            if first is None:
                return False
            rec = self.rfSignal._recording()
            cids = np.array([ch.channelID for ch in chans], dtype=np.int32)
        unread = bank.unread(cids)
        if ahead is None:
            room = (ring.maxSize - int(unread.max())) // spt - 1      # milliseconds the ring can take without overwriting
            k = min(self._ra_ms, (rec.size // 2 - first) // spt, room)
            if any(ch.channelState is ChannelState.IDLE for ch in self.channels.values()):
                # a channel started later begins reading at ring position 0 (channel.py:93) with everything up to the write
                # index unread: samples written there ahead of the write index's own wrap would be searched in place of the
                # stale ones the plain loop still holds -- while a channel is IDLE no block crosses the ring's end
                k = min(k, (ring.maxSize - ring.idxWrite) // spt)
            if k < 4:
                return False
            self._flush_pending()
            block = rec[2 * first:2 * (first + k * spt)]
            self.engine.iq_upload(block, ring.idxWrite)
            bank.flush()
        # every epoch that is complete inside the block: ONE persistent launch of as many epochs as every channel has room
        # for, then a launch per group of channels with equal numbers of epochs left (one or two more: an epoch that just fits)
        avail = unread + k * spt
        width = k + 2
        records = np.zeros((len(cids), width), dtype=bank.last.dtype)
        done = np.zeros(len(cids), dtype=np.int64)
        used = np.zeros(len(cids), dtype=np.int64)               # samples of the epochs run so far
        states = bank.state[cids].copy()
        pending = np.ones(len(cids), dtype=bool)
        first_pass = True
        while pending.any():
            rows = np.flatnonzero(pending)
            left = avail[rows] - used[rows]
            n_next = states["n_samples"][rows].astype(np.int64)
            budget = np.minimum(left // (n_next + 1), width - done[rows])
            budget = np.where((budget == 0) & (left >= n_next) & (done[rows] < width), 1, budget)   # the last epoch just fits
            pending[rows[budget <= 0]] = False
            if first_pass and ahead is None and (budget > 0).any():
                budget = np.where(budget > 0, budget[budget > 0].min(), budget)   # (everybody's common part first: one launch)
            groups = [(int(n_ep), rows[budget == n_ep]) for n_ep in np.unique(budget[budget > 0])]
            if first_pass and ahead is not None:
                # the launch that was queued while the previous block was handed out: its channels, its epoch count
                groups = [(ahead["n_ep"], np.arange(len(cids)))]
                pending[:] = True
            for n_ep, grp in groups:
                if first_pass and ahead is not None:
                    rec_g, st_g, done_g = bank.device.step_end()
                    self._ahead = ahead = None
                else:
                    rec_g, st_g, done_g, _ = bank.device.step(cids[grp], n_ep, want_records=True, want_bits=False)
                if (done_g == n_ep).all() and (done[grp] == done[grp[0]]).all():
                    records[grp, done[grp[0]]:done[grp[0]] + n_ep] = rec_g      # (the usual case: one slice assignment)
                    done[grp] += n_ep
                else:
                    for i, r in enumerate(grp):
                        records[r, done[r]:done[r] + done_g[i]] = rec_g[i, :done_g[i]]
                        done[r] += done_g[i]
                        if done_g[i] < n_ep:                  # the device parked the channel (NCO ran away)
                            bank.lost[cids[r]] = True
                            pending[r] = False
                used[grp] += np.where(np.arange(n_ep)[None, :] < done_g[:, None], rec_g["n_samples"], 0).sum(axis=1)
                states[grp] = st_g
            first_pass = False
        ra.load(cids, records, done, states, unread)
        ra.raw, ra.first, ra.slabs_left, ra.slab_no = block, first, k, 0
        ra.raw_address = block.__array_interface__["data"][0]
        ra.version = getattr(ring, "stateVersion", None)       # (while it stands, the scheduled channels are all there is)
        lists = self._lists
        ra.covers_active = lists is not None and np.array_equal(lists[4], ra.cids64)
        # (a block whose epochs spill past its own ticks -- a late joiner working off its backlog, one epoch per tick -- is
        # followed by plain ticks until the next one can start: nothing may be queued on the device behind it)
        if ra.n_ticks <= k:
            self._track_ahead(cids, first + k * spt, k, avail - used, states, pending_lost=bank.lost[cids])
        self._accept_prefetched(data)
        return True

    def _track_ahead(self, cids, first, k_now, unread_then, states, pending_lost):
        """Queue the block AFTER the one just loaded (sdr_bank_step_begin: its samples into the ring, one launch of the epochs
        every channel has room for) so that the device works on it while this one's packets are handed out; `_open_window`
        takes it over when its first millisecond arrives.  Nothing is queued when the recording ends, the ring cannot hold
        both blocks, a channel was parked, or read-ahead was switched off."""
        ring, bank, spt = self.sharedBuffer, self.bank, self._readahead.spt
        if not self._ra_ms or not self.PREFETCH or not hasattr(bank.device, "step_begin") or pending_lost.any():
            return
        rec = self.rfSignal._recording()
        # the ring has to hold both blocks beside whatever any channel has not read yet -- the block's own channels (what
        # they will have left at its end) and everybody else who is active (a late joiner lags by its acquisition time)
        mine = set(int(c) for c in cids)
        behind = [int(unread_then.max())] + [ring.getNbUnreadSamples(ch.currentSample) for ch in self.channels.values()
                                             if ch.channelState is not ChannelState.IDLE and ch.channelID not in mine]
        room = (ring.maxSize - max(behind)) // spt - k_now - 1
        k = min(self._ra_ms, (rec.size // 2 - first) // spt, room)
        if any(ch.channelState is ChannelState.IDLE for ch in self.channels.values()):
            # a channel started later begins reading at ring position 0 (channel.py:93: currentSample = 0) -- samples the
            # plain loop would still hold there must not be replaced ahead of their time: the block stops at the ring's end
            start = (ring.idxWrite + k_now * spt) % ring.maxSize
            k = min(k, (ring.maxSize - start) // spt) if start else 0
        if k < 4:
            return
        avail = unread_then + k * spt
        n_next = states["n_samples"].astype(np.int64)
        budget = np.minimum(avail // (n_next + 1), k + 2)
        if (budget <= 0).any():
            return
        n_ep = int(budget.min())
        block = rec[2 * first:2 * (first + k * spt)]
        self.engine.iq_upload(block, (ring.idxWrite + k_now * spt) % ring.maxSize)
        bank.flush()
        bank.device.step_begin(cids, n_ep)
        self._ahead = dict(bank=bank, first=first, k=k, cids=cids, n_ep=n_ep, block=block)

    def _accept_prefetched(self, data):
        ra = self._readahead
        spt = ra.spt
        # (the usual case costs one address comparison: the slab IS the recording's next millisecond)
        ok = isinstance(data, np.ndarray) and data.size == 2 * spt and (
            data.__array_interface__["data"][0] == ra.raw_address + 2 * ra.slab_no * spt * ra.raw.itemsize
            or np.array_equal(data, ra.raw[2 * ra.slab_no * spt:2 * (ra.slab_no + 1) * spt]))
        if not ok:
            raise ValueError("addNewRFData: while a read-ahead block is replayed the slab must be the recording's next "
                             f"millisecond (sample {ra.first + ra.slab_no * spt}); call enableReadAhead(0) to feed other data")
        ra.slab_no += 1
        ra.slabs_left -= 1
        self.sharedBuffer.shiftIdxWrite(spt)

    def addNewRFData(self, data):
        """Queue one slab for the ring.  The copy itself rides in the next run()'s device call (one call per tick);
        anything that looks at the ring before that (another addNewRFData, getSlice, runBlock) flushes it first."""
        ra = self._readahead
        if self._ra_ms and self.bank is not None and (ra is None or (ra.bank is not self.bank and ra.empty)):
            ra = self._readahead = self._EpochSchedule(self.bank, int(self.rfSignal.samplingFrequency * 1e-3))
        if ra is not None:
            if ra.slabs_left:
                return self._accept_prefetched(data)
            if (self._ra_ms or self._ahead is not None) and ra.empty and self._open_window(data):
                return None
        if self._pending:
            self._flush_pending()
        ring = self.sharedBuffer
        if (type(data) is np.ndarray and data.ndim == 1 and data.dtype == ring.rawDtype and data.flags.c_contiguous
                and not data.size & 1 and data.size):
            # the usual slab -- interleaved I,Q in the ring's own element type: nothing to convert (stage() does the rest)
            staged, offset, count = data, ring.idxWrite, data.size >> 1
            if ring.maxSize % count:
                raise ValueError("Data shift need to be a multiple from the max buffer size.")
        else:
            staged, offset, count = ring.stage(data)
        if ring.full and (self._unread_max is None or self._unread_max + count > ring.maxSize):
            self._guard_unread(count)
        self._unread_max = None
        if staged.nbytes > self.DEFER_BYTES or not self._engine_queues_slabs:
            self.engine.iq_upload(staged, offset)
        else:
            # the reference copies at this point (circularbuffer.py:54-82); so does this: the samples are copied out of
            # the caller's buffer before the call returns, and their transfer into the ring runs while the caller is on
            # its way to run() -- ordered before the tick's launch, waited for by the tick's one synchronisation
            self.engine.iq_upload_begin(staged, offset)
            self._pending = True
        ring.shiftIdxWrite(count)

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
        """Wait for a slab whose transfer was only queued (anything that may read the ring from another stream, or
        hand the ring's memory to somebody else, comes through here first)."""
        if self._pending:
            self._pending = False
            self.engine.sync()

    def getChannel(self, channelID):
        if channelID not in self.channels:
            raise ValueError("Channel ID does not exist.")
        return self.channels[channelID]

    def close(self):
        self._flush_pending()
        if self._ahead is not None and self._ahead["bank"] is self.bank:
            self.bank.device.step_end()                       # (a block queued ahead: let it finish before the bank goes)
        self._ahead = None
        self.channels.clear()
        if self.bank is not None:
            self.bank.close()
            self.sharedBuffer.channelBank = None

    # ------------------------------------------------------------------ the tick
    def run(self):
        """Flat sequence of result packets for this tick (channelManager.py:149-188)."""
        return self._run_end(self._run_begin(queue=False))

    def _run_begin(self, queue=True):
        """First half of run(): when the tick is the steady one (every active channel tracking on the device), the ready
        channels' epoch is QUEUED on this manager's device and a token returned; anything else returns None and the whole
        tick happens in `_run_end`.  A manager of several devices begins every device's tick before it ends any
        (multidevice.py) -- the reference starts every channel process, then waits for each (channelManager.py:164-171)."""
        version = getattr(self.sharedBuffer, "stateVersion", None)
        lists = self._lists
        if lists is None or lists[0] != (version, self.nbChannels) or not (lists[6] and self.STEADY_TICK):
            return None
        bank, ra = self.bank, self._readahead
        if bank is None or (ra is not None and (ra.bank is not bank or not ra.empty)):
            return None
        if queue:       # (a manager of one device makes the whole tick ONE library call in _tick_steady_end instead)
            bank.tick_ready_begin(None, 0)
        return (TickPackets(), bank, lists[4], lists[5], lists[7], queue)

    def _run_end(self, token):
        if token is not None:
            return self._tick_steady_end(*token)
        return self._run_general()

    def _run_general(self):
        out = TickPackets()
        # who is active / acquiring only changes when a channel changes state: device-tracked channels bump the ring's
        # stateVersion when they do, so a tracking receiver does not walk its channel objects every millisecond
        version = getattr(self.sharedBuffer, "stateVersion", None)
        if self._lists is not None and self._lists[0] == (version, self.nbChannels):
            _, active, acquiring, host_plugins, cids_active, states_active, steady, upd_templates = self._lists
        else:
            active = [ch for ch in self.channels.values() if ch.channelState is not ChannelState.IDLE]
            acquiring = [ch for ch in active if ch.channelState is ChannelState.ACQUIRING]
            host_plugins = [ch for ch in active if not isinstance(ch, DeviceTrackedChannel)]
            cids_active = np.array([ch.channelID for ch in active], dtype=np.int64)
            states_active = [ch.channelState for ch in active]
            cacheable = version is not None and all(isinstance(ch, DeviceTrackedChannel) for ch in self.channels.values())
            # steady: every active channel is tracking on the device -- the tick is one library call (_tick_steady)
            steady = bool(cacheable and active and not acquiring and not host_plugins
                          and all(st is ChannelState.TRACKING for st in states_active))
            upd_templates = packet_templates(ChannelMessage.CHANNEL_UPDATE, cids_active)
            self._lists = (((version, self.nbChannels), active, acquiring, host_plugins, cids_active, states_active, steady,
                            upd_templates) if cacheable else None)
            self._unread_max = None
        if not active:
            self._flush_pending()
            return out
        bank = self.bank
        ra = self._readahead
        if ra is not None and ra.bank is not bank:
            ra.follow(bank)
        if ra is None or ra.empty:
            if steady and self.STEADY_TICK and self._lists is not None and bank is not None:
                return self._tick_steady(out, bank, cids_active, states_active, upd_templates)
        elif (not self._pending and self._lists is not None and not acquiring and not host_plugins
                and ra.version == version and ra.covers_active):
            # replaying a read-ahead block and nothing else is going on: everything this tick reports was worked out
            # when the block was computed (readahead.py)
            k = ra.tick
            entry, decoded = ra.release()
            if entry is not None:
                out.add_lazy(TrackingRows(entry[0], bank.kinds, entry[1]))
            if decoded:
                out.add_ready(decoded)
            unread, flags, code, tow, tow_dec = ra.updates(k)
            out.add_lazy(UpdateRows(cids_active, states_active, flags, tow, tow_dec, unread, code, self._samples_per_ms,
                                    upd_templates))
            return out
        self._unread_max = None
        ready = bank.ready() if bank is not None else np.zeros(0, dtype=np.int32)
        released, ra_tick = None, None
        if ra is not None and not ra.empty:
            if len(ready):
                ready = ready[~ra.busy[ready]]               # (their next epochs are computed already: not to be run again)
            ra_tick = ra.tick
            released = ra.release()                          # epochs (and subframes) a block run computed for this tick
        # one device call: one epoch for every ready channel, behind the slab addNewRFData queued
        if bank is not None and len(ready):
            rec, done = bank.tick(None, 0, ready)
            self._pending = False                            # (the call ended with a synchronisation of the stream)
        else:
            rec, done = None, None                           # (nothing for the device to do in this tick)
            self._flush_pending()
        if acquiring:
            out.add_ready(self._acquire(acquiring))
        for ch in host_plugins:   # plugins that keep their loops on the host (e.g. the reference's class behind the seams mixin)
            if ch.channelState is ChannelState.TRACKING:
                out.add_ready(ch._processHandler())
        if released is not None:
            entry, decoded = released
            if entry is not None:
                out.add_lazy(TrackingRows(entry[0], bank.kinds, entry[1]))
            if decoded:
                out.add_ready(decoded)
        if len(ready):
            ran = np.flatnonzero(done > 0)
            out.add_lazy(TrackingRows(ready[ran], bank.kinds, rec[ran]))
            if bank.decoded:                                  # subframes completed by this tick's bits (kaplan:71-73)
                out.add_ready(pkt for _, _, pkt in bank.take_decoded())
        # channel updates: everything they report is captured now, the dicts are made when read (acquisition may have
        # moved channels to TRACKING during this tick: take the lists again if it did)
        if getattr(self.sharedBuffer, "stateVersion", None) != version or self._lists is None:
            cids = np.array([ch.channelID for ch in active], dtype=np.int64)
            states = [ch.channelState for ch in active]
            upd_templates = None
        else:
            cids, states = cids_active, states_active
        if bank is not None:
            unread, flags, code = bank.unread(cids), bank.flags(cids), bank.code_since_tow[cids].copy()
            tow, tow_dec = bank.tow[cids].copy(), bank.tow_decoded[cids].copy()
            if ra_tick is not None:                          # channels of a replayed block report the TICK's state, not the mirror's
                rows = ra.row_of[cids]
                # (... up to the tick of their last computed epoch: from then on the mirror is their state again, and the
                # device may already have run them further)
                sel = (rows >= 0) & (ra.last_tick[cids] >= ra_tick)
                if sel.any():
                    for dst, src in zip((unread, flags, code, tow, tow_dec), ra.updates(ra_tick)):
                        dst[sel] = src[rows[sel]]
            out.add_lazy(UpdateRows(cids, states, flags, tow, tow_dec, unread, code, self._samples_per_ms, upd_templates))
        else:
            out.add_ready(ch.prepareChannelUpdate() for ch in active)
        return out

    def _tick_steady(self, out, bank, cids_active, states_active, upd_templates):
        """The tick of a receiver whose active channels are all tracking on the device: ONE library call
        (sdr_bank_tick_mirrored, behind the slab addNewRFData queued) decides who is ready, runs their epoch, brings
        the bank's mirror up to date and leaves what the packets report; the packets themselves are made when read."""
        return self._tick_steady_end(out, bank, cids_active, states_active, upd_templates, queued=False)

    def _tick_steady_end(self, out, bank, cids_active, states_active, upd_templates, queued=True):
        ran, rec, upd, self._unread_max = bank.tick_ready_end() if queued else bank.tick_ready(None, 0)
        if len(ran):
            self._pending = False                            # (an epoch ran: the call ended with a synchronisation of the stream)
        # (no channel ready: nothing was launched and nothing waited for -- the slab stays "queued, not waited for", and
        # whoever looks at the ring next flushes it; the library guards its own staging halves with events either way)
        decoded = ()
        if len(ran):
            rows = TrackingRows(ran, bank.kinds, rec)
            bank.hold(rows)                                   # (`ran` / `rec` / `upd` are views of the device's arrays: bank.hold)
            out.add_lazy(rows)
            if bank.decoded:                                  # subframes completed by this tick's bits (kaplan:71-73)
                decoded = [pkt for _, _, pkt in bank.take_decoded()]
                out.add_ready(decoded)
        self._steady_rows = (ran, rec, upd, decoded)         # (what a manager of several devices merges into one packet list)
        if len(upd) != len(cids_active):                     # (cannot happen while the lists stand; never guess)
            raise RuntimeError("channel bank and channel manager disagree about the tracking channels")
        rows = UpdateRows(cids_active, states_active, upd, bank.tow.copy(), bank.tow_decoded.copy(), None, None,
                          self._samples_per_ms, upd_templates)            # (flags / unread / code count: the rows of `upd`)
        bank.hold(rows)
        out.add_lazy(rows)
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
        if (self._readahead is not None and (self._readahead.slabs_left or not self._readahead.empty)) or self._ahead is not None:
            raise RuntimeError("runBlock while a read-ahead block is being replayed or tracked ahead: finish its ticks first")
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
