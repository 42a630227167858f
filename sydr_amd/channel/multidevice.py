"""ONE ChannelManager over the GPUs of a node, in one process.

The reference's receiver builds one manager (sydr/receiver/receiver.py:86) and allocates its channels from one pool
(sydr/channel/channelManager.py:70-127); its only parallelism is one OS process per channel, all started every
millisecond before any is waited for (channelManager.py:164-171).  north_star shards that pool over the GPUs of a
node: the IQ stream is replicated on every device, channels are independent, there is no collective.

`ChannelManager(rfSignal, devices=[0, 1, ..., 7])` keeps that shape: behind the reference's surface (`addChannel`,
`requestTracking`, `addNewRFData`, `run`, `getChannel`, `close`) there is one single-device manager per GPU -- its
own engine, ring, code table and channel bank -- and

* `addChannel(ChannelObject, configuration, n)` deals the n new channels out in `shard_channels(n, d, N)` order
  (contiguous blocks, sizes differing by at most one); a channel keeps the number the whole receiver knows it by;
* `requestTracking(prn)` takes the first IDLE channel of the device with the fewest busy ones (devices fill
  round-robin: eight satellites on eight GPUs are one per GPU, not eight on the first);
* `addNewRFData(slab)` queues the slab on EVERY device (`sdr_iq_upload_begin`: copied out of the caller's buffer,
  the transfer rides on each engine's stream) -- the replicated input of SURVEY 8(e);
* `run()` BEGINS every device's tick (`sdr_bank_tick_mirrored_begin`: readiness + the queued launch, nothing waited
  for) before it ENDS any (`..._end`: wait + mirrors), so the devices track at the same time under one host thread,
  and returns the packets of all devices as one list in channel order per packet kind -- for the steady tick exactly
  the list a single-device manager of the same channels returns.

Ticks in which some device still acquires (or replays a read-ahead block) take that device's general path one after
the other; their packets are merged in the same order.
"""
from __future__ import annotations

import numpy as np

from ..engine import Engine
from ..utils.enumerations import ChannelMessage, ChannelState
from .bank import TickPackets, TrackingRows, UpdateRows
from .manager import ChannelManager, shard_channels

_KIND_RANK = {ChannelMessage.ACQUISITION_UPDATE: 0, ChannelMessage.TRACKING_UPDATE: 1, ChannelMessage.DECODING_UPDATE: 2,
              ChannelMessage.CHANNEL_UPDATE: 3}


class MultiDeviceChannelManager:
    TIMEOUT = 1

    def __init__(self, rfSignal, *, devices=None, engines=None, keepCorrelationMap: bool = True, ring_ms: int = 100):
        """devices: HIP device numbers, one part each (a number may repeat: a second engine on the same card -- how the
        one-GPU test box rehearses two devices); engines: ready-made engines instead (not closed by close())."""
        self.rfSignal = rfSignal
        self._owned = []
        if engines is None:
            if not devices:
                raise ValueError("devices=[...] or engines=[...] must name at least one device")
            from ..runtime import get_engine
            engines, seen = [], set()
            for d in devices:
                d = int(d)
                if d in seen:
                    eng = Engine(d)               # (a second engine on a device already in the list: this manager's own)
                    self._owned.append(eng)
                else:
                    eng = get_engine(d)
                    seen.add(d)
                engines.append(eng)
        self.engines = list(engines)
        self.parts = [ChannelManager(rfSignal, engine=eng, keepCorrelationMap=keepCorrelationMap, ring_ms=ring_ms)
                      for eng in self.engines]
        self.channels = {}
        self.nbChannels = 0
        self.resultQueue = None
        self._part_of = {}            # channel number -> index of the part (device) that owns it
        self._merge_cache = None      # per-tick constants of the merged steady tick, valid while no channel changes state
        self._spms = rfSignal.samplingFrequency / 1e3

    # ------------------------------------------------------------------ reference surface
    @property
    def sharedBuffer(self):
        """The first device's ring (every device's ring holds the same samples at the same positions)."""
        return self.parts[0].sharedBuffer

    def addChannel(self, ChannelObject, configuration, nbChannels=1):
        first, n_dev = self.nbChannels, len(self.parts)
        for d, part in enumerate(self.parts):
            cids = [first + k for k in shard_channels(nbChannels, d, n_dev)]
            part._addChannelsAt(cids, ChannelObject, configuration)
            for c in cids:
                self._part_of[c] = d
        for c in range(first, first + nbChannels):          # (the receiver's view: one pool, numbered in order)
            self.channels[c] = self.parts[self._part_of[c]].channels[c]
        self.nbChannels += nbChannels
        self._merge_cache = None

    def requestTracking(self, satelliteID: int):
        busy = [sum(ch.channelState is not ChannelState.IDLE for ch in part.channels.values()) for part in self.parts]
        for d in sorted(range(len(self.parts)), key=lambda d: (busy[d], d)):
            for cid in sorted(self.parts[d].channels):
                channel = self.parts[d].channels[cid]
                if channel.channelState is ChannelState.IDLE:
                    channel.setSatellite(satelliteID)
                    channel.start()
                    self._merge_cache = None
                    return channel
        raise Warning(f"Could not find an IDLE channel for tracking satellite [G{satelliteID}].")

    def addNewRFData(self, data):
        for part in self.parts:                              # the same slab into every device's ring; none is waited for
            part.addNewRFData(data)

    def enableReadAhead(self, nbMilliseconds: int = 50):
        for part in self.parts:
            part.enableReadAhead(nbMilliseconds)

    def getChannel(self, channelID):
        if channelID not in self.channels:
            raise ValueError("Channel ID does not exist.")
        return self.channels[channelID]

    def deviceOf(self, channelID) -> int:
        """Index (into `devices` / `engines`) of the device that tracks this channel."""
        return self._part_of[channelID]

    def close(self):
        for part in self.parts:
            part.close()
        self.channels.clear()
        for eng in self._owned:
            eng.close()
        self._owned = []

    # ------------------------------------------------------------------ the tick
    def run(self):
        """Every device's tick begun, then every device's tick ended; one packet list (channelManager.py:149-188)."""
        parts = [p for p in self.parts if p.nbChannels]
        tokens = [part._run_begin() for part in parts]
        outs, error = [], None
        for part, tok in zip(parts, tokens):                 # (a tick that was begun is ended, whatever happened elsewhere)
            try:
                outs.append(part._run_end(tok))
            except BaseException as exc:                     # noqa: BLE001 -- re-raised below, after every device was ended
                outs.append(None)
                error = error or exc
        if error is not None:
            raise error
        steady = [p for p, tok in zip(parts, tokens) if tok is not None]
        if steady and not any(len(o) for o, tok in zip(outs, tokens) if tok is None):
            return self._merge_steady(steady)                # (devices without an active channel report nothing)
        return self._merge_general(outs)

    def _merge_steady(self, parts):
        """The steady ticks of all devices as ONE list: TRACKING_UPDATE packets in channel order, the subframes this
        tick's bits completed, CHANNEL_UPDATE packets in channel order -- built from the devices' rows, lazily as ever."""
        out = TickPackets()
        rows = [p._steady_rows for p in parts]
        ran = np.concatenate([r[0] for r in rows])
        if len(ran):
            rec = np.concatenate([r[1] for r in rows])
            kinds = np.concatenate([np.asarray(p.bank.kinds)[r[0]] for p, r in zip(parts, rows)])
            if len(ran) > 1 and (np.diff(ran) < 0).any():
                order = np.argsort(ran, kind="stable")
                ran, rec, kinds = ran[order], rec[order], kinds[order]
            out.add_lazy(TrackingRows(ran, kinds, rec))
            decoded = [pkt for r in rows for pkt in r[3]]
            if decoded:
                out.add_ready(sorted(decoded, key=lambda pkt: pkt["cid"]))
        upd = np.concatenate([r[2] for r in rows])
        cids = upd["channel"].astype(np.int64)
        tows = np.concatenate([p.bank.tow[r[2]["channel"]] for p, r in zip(parts, rows)])
        tow_dec = np.concatenate([p.bank.tow_decoded[r[2]["channel"]] for p, r in zip(parts, rows)])
        if len(cids) > 1 and (np.diff(cids) < 0).any():
            order = np.argsort(cids, kind="stable")
            upd, cids, tows, tow_dec = upd[order], cids[order], tows[order], tow_dec[order]
        cache = self._merge_cache
        key = tuple(p._lists[0] for p in parts)
        if cache is None or cache[0] != key or len(cache[1]) != len(cids):
            states = [self.channels[int(c)].channelState for c in cids]
            cache = self._merge_cache = (key, states)
        out.add_lazy(UpdateRows(cids, cache[1], upd, tows, tow_dec, None, None, self._spms))
        return out

    @staticmethod
    def _merge_general(outs):
        """Packets of ticks that took the devices' general paths (acquisition, read-ahead replay, host-side plugins):
        one list, ordered by packet kind as a single manager orders them and by channel number within a kind."""
        packets = [pkt for o in outs for pkt in o]
        packets.sort(key=lambda pkt: (_KIND_RANK.get(dict.__getitem__(pkt, "type"), 4), dict.__getitem__(pkt, "cid")))
        out = TickPackets()
        out.add_ready(packets)
        return out

    def runBlock(self, nbEpochs: int):
        """Up to `nbEpochs` epochs per tracking channel on every device (each device one persistent launch); packets device
        by device (within a device: channel by channel, epoch by epoch, then its CHANNEL_UPDATEs)."""
        out = TickPackets()
        for part in self.parts:
            if part.nbChannels:
                out.add_ready(list(part.runBlock(nbEpochs)))
        return out


__all__ = ["MultiDeviceChannelManager"]
