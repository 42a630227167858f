"""Host side of the device-resident channel bank.

The reference keeps one Python process per channel, each with its own copy of the loop state, and
synchronises them every millisecond through Events and a pickled Queue
(sydr/channel/channelManager.py:70-127,149-188; sydr/channel/channel.py:121-160).  Here the tracking
state of every channel of one GPU lives in HBM (`sdr_bank_*`, include/sydr_amd.h) and is advanced there
-- correlators, discriminators, loop filters, NCO update, lock-state machine, bit decisions in one
launch for all channels.  What the host holds is a MIRROR: two NumPy structured arrays (`state`,
`last`) refreshed by every step, from which the channel objects read their attributes and the
packets are built.  A write to a mirrored attribute marks the channel dirty; it is uploaded before
the next step.
"""
from __future__ import annotations

from collections.abc import Sequence

import weakref

import numpy as np

from .._lib import LOOP_CFG_DTYPE, TRACK_EPOCH_DTYPE, TRACK_STATE_DTYPE
from ..utils.enumerations import ChannelMessage, ChannelState, LoopLockState, TrackingFlags
from .navdecoder import HOST_FLAGS

KIND_BORRE, KIND_KAPLAN = 0, 1

_CORR_KEYS = ("i_early", "q_early", "i_prompt", "q_prompt", "i_late", "q_late")


def _flags(value: int):
    """TrackingFlags member when the bit set is a named one, else the plain integer (as the reference's
    `trackFlags |= ...` arithmetic produces for unnamed combinations)."""
    try:
        return TrackingFlags(value)
    except ValueError:
        return value


class ChannelBank:
    def __init__(self, engine, max_channels: int, ring):
        self.engine = engine
        self.ring = ring
        self.max_channels = int(max_channels)
        self.device = engine.bank(self.max_channels)
        self.state = np.zeros(self.max_channels, dtype=TRACK_STATE_DTYPE)
        self.cfg = np.zeros(self.max_channels, dtype=LOOP_CFG_DTYPE)
        self.last = np.zeros(self.max_channels, dtype=TRACK_EPOCH_DTYPE)   # most recent epoch record per channel
        self.last["nav_bit"] = -1
        self.code_since_tow = np.zeros(self.max_channels, dtype=np.int64)
        self.tow = np.zeros(self.max_channels, dtype=np.float64)           # Channel.tow once a subframe was decoded ...
        self.tow_decoded = np.zeros(self.max_channels, dtype=bool)         # ... (the plain int 0 of channel.py:93 before)
        self.host_flags = np.zeros(self.max_channels, dtype=np.int64)      # TrackingFlags owned by the decoder (HOST_FLAGS)
        self.decoders = [None] * self.max_channels                         # NavDecoder per channel (navdecoder.py)
        self.decoded = []                                                  # DECODING_UPDATE packets of the last step
        self.tracking = np.zeros(self.max_channels, dtype=bool)            # channels in ChannelState.TRACKING
        self.lost = np.zeros(self.max_channels, dtype=bool)                # NCO ran away on the device: channel parked
        self._dirty = np.zeros(self.max_channels, dtype=bool)
        self._any_dirty = False
        self._kinds = None                                                 # cfg["loop_kind"] as a list (see kinds)
        self._bound = None                                                 # the device's sdr_tick_mirror over these arrays
        self._held = ([], [])                                              # per output set of the device: weak references (hold())
        self.nav_bits = [[] for _ in range(self.max_channels)]

    def grown(self, max_channels: int) -> "ChannelBank":
        """A larger bank holding this one's channels (the mirror is complete, so the rows are simply re-uploaded)."""
        big = ChannelBank(self.engine, max_channels, self.ring)
        n = self.max_channels
        for name in ("state", "cfg", "last", "code_since_tow", "tow", "tow_decoded", "host_flags", "tracking", "lost"):
            getattr(big, name)[:n] = getattr(self, name)
        big.nav_bits[:n] = self.nav_bits
        big.decoders[:n] = self.decoders
        big._dirty[:n] = self.cfg["n_taps"] != 0
        big._any_dirty = True
        self.close()
        return big

    # ------------------------------------------------------------------ mirror <-> HBM
    def touch(self, ch: int):
        self._dirty[ch] = True
        self._any_dirty = True
        self._kinds = None

    def flush(self):
        """Upload every channel whose mirror was written by the host since the last step."""
        if not self._any_dirty:
            return
        for ch in np.flatnonzero(self._dirty):
            self.device.put(int(ch), self.state[ch], self.cfg[ch])
        self._dirty[:] = False
        self._any_dirty = False

    @property
    def kinds(self):
        """cfg["loop_kind"] of every channel as a plain list (re-made after a channel was configured)."""
        if self._kinds is None:
            self._kinds = self.cfg["loop_kind"].tolist()
        return self._kinds

    def refresh(self, ch: int):
        self.state[ch] = self.device.get(ch)

    def close(self):
        self.device.close()

    # ------------------------------------------------------------------ readiness (channel.py:137-146 / kaplan:347)
    def unread(self, channels) -> np.ndarray:
        """CircularBuffer.getNbUnreadSamples(currentSample) for many channels at once."""
        cur = self.state["current_sample"][channels] % self.ring.maxSize
        w = self.ring.idxWrite
        return np.where(cur <= w, w - cur, self.ring.maxSize - cur + w)

    def ready(self) -> np.ndarray:
        """Indices of the tracking channels whose next epoch is completely inside the ring."""
        idx = np.flatnonzero(self.tracking & ~self.lost)
        if idx.size == 0:
            return idx.astype(np.int32)
        ok = self.unread(idx) >= self.state["n_samples"][idx]
        return idx[ok].astype(np.int32)

    # ------------------------------------------------------------------ advancing
    def _absorb(self, channels, records, states, done):
        n_ep = records.shape[1]
        lo, n = int(channels[0]), len(channels)
        if int(channels[-1]) - lo + 1 == n and (n < 3 or bool((np.diff(channels) == 1).all())):
            sel = slice(lo, lo + n)               # the usual case: a run of channel numbers -- plain slices
        else:
            sel = channels
        self.state[sel] = states
        self.code_since_tow[sel] += done
        if n_ep == 1 and int(done.min()) == 1:    # a tick in which every listed channel ran its epoch
            self.last[sel] = records[:, 0]
        else:
            ran = done > 0
            if ran.any():
                rows = np.flatnonzero(ran)
                self.last[channels[rows]] = records[rows, done[rows] - 1]
            short = done < n_ep
            if short.any():
                self.lost[channels[short]] = True
        bits = records["nav_bit"]
        if (bits >= 0).any():
            for r, e in zip(*np.nonzero(bits >= 0)):      # (row-major: a channel's bits arrive in epoch order)
                if e < done[r]:
                    self._new_bit(int(channels[r]), int(bits[r, e]), int(records["track_flags"][r, e]),
                                  int(e), int(done[r]))

    def _new_bit(self, ch: int, bit: int, device_flags: int, epoch: int, epochs_run: int):
        """One navigation bit left the device: keep it, and let the channel's decoder look at it -- the `runDecoding`
        that follows `runTracking` in the reference's tick (kaplan:71-73).  A completed subframe becomes a
        DECODING_UPDATE packet (kaplan:858-868), `tow`, and restarts the code count (kaplan:833) from this epoch."""
        self.nav_bits[ch].append(bit)
        decoder = self.decoders[ch]
        if decoder is None:
            return
        flags, event = decoder.push(bit, device_flags | int(self.host_flags[ch]))
        self.host_flags[ch] = flags & HOST_FLAGS
        if event is not None:
            self.tow[ch], self.tow_decoded[ch] = event.channel_tow, True
            self.code_since_tow[ch] = epochs_run - 1 - epoch
            self.decoded.append((ch, epoch, {"cid": ch, "type": ChannelMessage.DECODING_UPDATE,
                                             "subframe_id": event.subframe_id, "tow": event.tow, "bits": event.bits}))

    def take_decoded(self):
        """[(channel, epoch within the step, DECODING_UPDATE packet)] produced by the last step, handed over once."""
        out, self.decoded = self.decoded, []
        return out

    def flags(self, channels):
        """TrackingFlags of `channels`: the device's bits (code lock, bit sync) and the decoder's."""
        return self.state["track_flags"][channels] | self.host_flags[channels]

    def channel_tow(self, ch: int):
        return float(self.tow[ch]) if self.tow_decoded[ch] else 0

    def tick(self, raw, ring_offset, channels):
        """Ring ingest + one epoch for `channels` in one device call; returns the records [n]."""
        self.flush()
        rec, states, done = self.device.tick(raw, ring_offset, channels)
        if len(channels):
            self._absorb(channels, rec.reshape(-1, 1), states, done)
        return rec, done

    def tick_ready(self, raw, ring_offset):
        """The whole tick in one library call (sdr_bank_tick_mirrored): readiness, one epoch for the ready channels,
        this mirror updated in place.  -> (ran, records, updates, max_unread): OWNED copies of the channels that
        completed an epoch, their records, and one sdr_tick_update row per tracking channel."""
        self.flush()
        dev = self.device
        if self._bound is None:
            self._bound = dev.bind_mirror(self.state, self.last, self.code_since_tow, self.tracking, self.lost, self.host_flags)
        self._release_next_set()
        return self.tick_ready_end(dev.tick_mirrored(raw, ring_offset, self.ring.idxWrite))

    # The device writes a tick's `ran` / `records` / `updates` into one of TWO sets of arrays, alternately: what tick_ready_end
    # hands out are VIEWS of them (three copies of together 8 KB cost 5 of a tick's 35 us from Python), good until the tick
    # after next.  Whoever builds something longer-lived on such views -- the lazy packet rows -- registers it with hold();
    # before the device comes round to a set again, what is still alive of it is told to take its own copy (detach()).
    def hold(self, *row_objects):
        if getattr(self.device, "double_buffered", False):
            self._held[self.device.out_set].extend(weakref.ref(o) for o in row_objects)

    def _release_next_set(self):
        dev = self.device
        if not getattr(dev, "double_buffered", False):
            return
        held = self._held[dev.out_set ^ 1]
        if held:
            for ref in held:
                rows = ref()
                if rows is not None:
                    rows.detach()
            held.clear()

    def tick_ready_begin(self, raw, ring_offset):
        """First half of `tick_ready` (sdr_bank_tick_mirrored_begin): the ready channels' epoch is queued on this bank's
        device, nothing is waited for -- a manager of several devices begins every bank's tick before it ends any."""
        self.flush()
        dev = self.device
        if self._bound is None:
            self._bound = dev.bind_mirror(self.state, self.last, self.code_since_tow, self.tracking, self.lost, self.host_flags)
        self._release_next_set()
        dev.tick_mirrored_begin(raw, ring_offset, self.ring.idxWrite)

    def tick_ready_end(self, m=None):
        dev = self.device
        if m is None:
            m = dev.tick_mirrored_end()
        n = m.n_ran
        if getattr(dev, "double_buffered", False):      # (views: see hold())
            ran, rec, upd = dev.ran[:n], dev.records[:n], dev.updates[:m.n_updates]
        else:
            ran, rec, upd = dev.ran[:n].copy(), dev.records[:n].copy(), dev.updates[:m.n_updates].copy()
        if m.n_nav_bits:
            bits, decoders, through_decoder = rec["nav_bit"], self.decoders, False
            for r in np.flatnonzero(bits >= 0).tolist():
                ch = int(ran[r])
                if decoders[ch] is None:
                    self.nav_bits[ch].append(int(bits[r]))
                else:
                    self._new_bit(ch, int(bits[r]), int(rec["track_flags"][r]), 0, 1)
                    through_decoder = True
            if through_decoder:     # the decoder owns flag bits and restarts the code count at a subframe (kaplan:833)
                chans = upd["channel"]
                upd["epochs_since_tow"] = self.code_since_tow[chans]
                upd["track_flags"] = self.state["track_flags"][chans] | self.host_flags[chans]
        return ran, rec, upd, m.max_unread

    def step(self, channels, n_epochs: int = 1, stream: int = 0):
        """`n_epochs` epochs for `channels`; returns (records [n][n_epochs], epochs_done [n])."""
        self.flush()
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        rec, states, done, _ = self.device.step(channels, n_epochs, want_records=True, want_bits=False, stream=stream)
        self._absorb(channels, rec, states, done)
        return rec, done


def tracking_packet(cid: int, kind: int, rec) -> dict:
    """TRACKING_UPDATE packet (keys of channel_l1ca_kaplan.py:653-676 = DB columns, io/database.py:76-93) from one
    device epoch record.  Borre has no lock indicators: channel_l1ca_borre.py:430-449 sends NaN / zeros there."""
    corr = rec["corr"]
    pkt = {"cid": cid, "type": ChannelMessage.TRACKING_UPDATE}
    pkt.update(zip(_CORR_KEYS, corr[:6].tolist()))
    pkt["carrier_frequency"] = float(rec["carrier_hz"])
    pkt["code_frequency"] = float(rec["code_hz"])
    pkt["carrier_frequency_error"] = float(rec["carrier_err"])
    pkt["code_frequency_error"] = float(rec["code_err"])
    pkt["dll"], pkt["pll"], pkt["fll"] = float(rec["dll"]), float(rec["pll"]), float(rec["fll"])
    if kind == KIND_KAPLAN:
        pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"] = float(rec["cn0"]), float(rec["pll_lock"]), float(rec["fll_lock"])
        pkt["lock_state"] = LoopLockState(int(rec["lock_state"]))
    else:
        pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"], pkt["lock_state"] = np.nan, 0.0, 0.0, 0
    return pkt


class LazyPacket(dict):
    """A result packet (a dict, as the reference's channels send them) whose values are filled in when somebody asks
    for one: it is born holding "cid" and "type" -- what `Receiver._updateDatabaseFromChannels` (receiver.py:291-299)
    looks at to route it -- and everything else (`_src.full(cid)`) arrives on the first access to another key or to
    the packet as a whole (len, iteration, items, ==, pickling ...).  Keys the consumer adds before that are kept.
    At 32 channels the 64 dicts of a tick cost more host time than the tick's device call; most consumers read a
    handful of keys of a handful of packets."""
    __slots__ = ("_src",)

    def _fill(self):
        try:
            src = self._src
        except AttributeError:
            return
        if src is None:
            return
        self._src = None
        mine = dict.copy(self)
        full = src.full(dict.__getitem__(self, "cid"))
        dict.clear(self)
        dict.update(self, full)
        dict.update(self, mine)        # (what the consumer wrote meanwhile wins; "cid" / "type" keep their places)

    def __missing__(self, key):
        try:
            pending = self._src is not None
        except AttributeError:
            pending = False
        if not pending:
            raise KeyError(key)
        self._fill()
        return dict.__getitem__(self, key)

    def _filled(name):
        plain = getattr(dict, name)

        def method(self, *args, **kwargs):
            self._fill()
            return plain(self, *args, **kwargs)
        method.__name__ = name
        return method

    for _name in ("__iter__", "__len__", "__contains__", "__reversed__", "__delitem__", "keys", "values", "items", "get",
                  "pop", "popitem", "setdefault", "copy", "__repr__", "__or__", "__ror__", "__ior__"):
        locals()[_name] = _filled(_name)
    del _name, _filled

    def __eq__(self, other):
        if not isinstance(other, dict):      # (`packet == None` of receiver.py:292 must not cost a fill)
            return NotImplemented
        self._fill()
        if isinstance(other, LazyPacket):
            other._fill()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        eq = self.__eq__(other)
        return eq if eq is NotImplemented else not eq

    __hash__ = None

    def clear(self):
        self._src = None
        dict.clear(self)

    def __reduce_ex__(self, protocol):
        self._fill()
        return (dict, (dict.copy(self),))    # travels (pickle, queues) as the plain dict it stands for


_TEMPLATES = {}       # (message type's value, cid) -> the two keys a packet is born with
_TEMPLATE_LISTS = {}  # (message type's value, the channel numbers' type and bytes) -> the list of those for a whole tick


def packet_templates(kind, cids):
    """The {"cid", "type"} dicts LazyPackets of `kind` for the channels `cids` (an integer array or a list of ints) are
    copied from.  A receiver's ticks report the same channels over and over: the list itself is remembered per channel
    set (a few entries: the sets differ only while channels come and go)."""
    code = kind.value                    # (an int: hashing the enum member itself is a Python-level call per key)
    key = None
    if isinstance(cids, np.ndarray):
        key = (code, cids.dtype.char, cids.tobytes())
        hit = _TEMPLATE_LISTS.get(key)
        if hit is not None:
            return hit
        cids = cids.tolist()
    out = []
    for c in cids:
        t = _TEMPLATES.get((code, c))
        if t is None:
            t = _TEMPLATES[(code, c)] = {"cid": c, "type": kind}
        out.append(t)
    if key is not None:
        if len(_TEMPLATE_LISTS) > 4096:
            _TEMPLATE_LISTS.clear()
        _TEMPLATE_LISTS[key] = out
    return out


class TrackingRows:
    """Source of one tick's TRACKING_UPDATE packets (keys of channel_l1ca_kaplan.py:653-676 = the DB columns,
    io/database.py:76-93; Borre has no lock indicators: channel_l1ca_borre.py:430-449 sends NaN / zeros there): the
    tick's epoch records, turned into plain Python values in ONE call the first time any packet is filled (field
    access on NumPy records costs more per packet than the dict itself)."""
    __slots__ = ("cids", "kinds", "records", "_rows", "_row_of", "_templates", "__weakref__")
    _AT = {name: k for k, name in enumerate(TRACK_EPOCH_DTYPE.names)}

    def detach(self):
        """The arrays this was built on are about to be overwritten (ChannelBank.hold): own copies, unless the rows have been
        turned into Python values already."""
        if self._rows is None:
            self.records = self.records.copy()
            if isinstance(self.cids, np.ndarray):
                self.cids = self.cids.copy()

    def __init__(self, cids, kinds, records, templates=None):
        """cids: channel per record (array or list); kinds: loop kind per CHANNEL NUMBER (a list indexed by cid) or per
        record (an array of len(cids))."""
        self.cids, self.kinds, self.records = cids, kinds, records
        self._rows = self._row_of = None
        self._templates = templates

    def __len__(self):
        return len(self.cids)

    def templates(self):
        if self._templates is None:
            self._templates = packet_templates(ChannelMessage.TRACKING_UPDATE, self.cids)
        return self._templates

    def full(self, cid):
        if self._rows is None:
            self._rows = self.records.tolist()
            cids = self.cids.tolist() if isinstance(self.cids, np.ndarray) else list(self.cids)
            self._row_of = {c: i for i, c in enumerate(cids)}
            if isinstance(self.kinds, np.ndarray):
                self.kinds = dict(zip(cids, self.kinds.tolist()))
        return self.row(self._row_of[cid], cid)

    def row(self, i, cid):
        at = self._AT
        r = self._rows[i]
        corr = r[at["corr"]]
        pkt = {"cid": cid, "type": ChannelMessage.TRACKING_UPDATE,
               "i_early": corr[0], "q_early": corr[1], "i_prompt": corr[2], "q_prompt": corr[3], "i_late": corr[4],
               "q_late": corr[5], "carrier_frequency": r[at["carrier_hz"]], "code_frequency": r[at["code_hz"]],
               "carrier_frequency_error": r[at["carrier_err"]], "code_frequency_error": r[at["code_err"]],
               "dll": r[at["dll"]], "pll": r[at["pll"]], "fll": r[at["fll"]]}
        if self.kinds[cid] == KIND_KAPLAN:
            pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"] = r[at["cn0"]], r[at["pll_lock"]], r[at["fll_lock"]]
            pkt["lock_state"] = LoopLockState(r[at["lock_state"]])
        else:
            pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"], pkt["lock_state"] = np.nan, 0.0, 0.0, 0
        return pkt


def tracking_packets_builder(cids, kinds, records):
    """build(i) -> the (eager) TRACKING_UPDATE packet of row i of a tick's records."""
    src = TrackingRows(cids, kinds, records)

    def build(i):
        return src.full(int(cids[i]))
    return build


class UpdateRows:
    """Source of one tick's CHANNEL_UPDATE packets (channel.py:205-228) from values captured at the end of the tick.
    `tow` is what the reference's `Channel.tow` holds: the int 0 until a subframe was decoded, then HOW TOW + 1.24 s
    (kaplan:810-822)."""
    __slots__ = ("cids", "states", "flags", "tows", "tow_decoded", "unread", "code", "samples_per_ms", "_rows", "_templates",
                 "__weakref__")

    def detach(self):
        """As TrackingRows.detach: `flags` may be a tick's sdr_tick_update rows (a view of the device's array)."""
        if self._rows is None and isinstance(self.flags, np.ndarray):
            self.flags = self.flags.copy()

    def __init__(self, cids, states, flags, tows, tow_decoded, unread, code_since_tow, samples_per_ms, templates=None):
        """cids / flags / unread / code_since_tow: one value per packet (or flags = the tick's sdr_tick_update rows and
        unread = code_since_tow = None); states: ChannelState per packet; tows / tow_decoded: per packet (arrays of
        len(cids)) or per CHANNEL NUMBER (longer arrays, indexed by cid)."""
        self.cids, self.states, self.flags, self.tows, self.tow_decoded = cids, states, flags, tows, tow_decoded
        self.unread, self.code, self.samples_per_ms = unread, code_since_tow, samples_per_ms
        self._rows = None
        self._templates = templates

    def __len__(self):
        return len(self.cids)

    def _cid_list(self):
        return self.cids.tolist() if isinstance(self.cids, np.ndarray) else list(self.cids)

    def templates(self):
        if self._templates is None:
            self._templates = packet_templates(ChannelMessage.CHANNEL_UPDATE, self.cids)
        return self._templates

    def full(self, cid):
        if self._rows is None:
            if self.unread is None:      # `flags` holds a tick's sdr_tick_update rows: the three columns come from it
                upd = self.flags
                self.flags, self.unread, self.code = upd["track_flags"], upd["unread"], upd["epochs_since_tow"]
            cids = self._cid_list()
            tows, dec = np.asarray(self.tows), np.asarray(self.tow_decoded)
            if len(tows) != len(cids):
                tows, dec = tows[cids], dec[cids]
            unread = np.asarray(self.unread)
            code = np.asarray(self.code)
            since = code + unread / self.samples_per_ms
            self._rows = {c: row for c, row in zip(cids, zip(self.states, np.asarray(self.flags).tolist(), tows.tolist(),
                                                             dec.tolist(), since.tolist(), unread.tolist(), code.tolist()))}
        state, flags, tow, dec, since, unread, code = self._rows[cid]
        return {"cid": cid, "type": ChannelMessage.CHANNEL_UPDATE, "state": state, "tracking_flags": _flags(flags),
                "tow": tow if dec else 0, "time_since_tow": since, "unprocessed_samples": unread, "code_since_tow": code}


class TickPackets(Sequence):
    """The flat packet list `ChannelManager.run()` returns (channelManager.py:149-188), built on demand.

    Everything a packet needs is captured when the tick ends (epoch records, flags, unread counts), so reading it
    later gives what an eager list would have held.  Nothing is made until somebody looks at the sequence; then the
    tick's TRACKING_UPDATE / CHANNEL_UPDATE packets come as LazyPackets (two keys each, the rest on demand)."""

    def __init__(self):
        self._parts = []      # (count, source with templates() / full(cid)  |  builder(i) -> dict)
        self._list = None
        self._n = 0

    def add(self, count: int, builder):
        if count:
            self._parts.append((count, builder))
            self._n += count
            self._list = None

    def add_lazy(self, source):
        """`source`: TrackingRows / UpdateRows -- len(), templates(), full(cid)."""
        self.add(len(source), source)

    def add_ready(self, packets):
        packets = list(packets)
        self.add(len(packets), packets.__getitem__)

    def _packets(self):
        out = self._list
        if out is None:
            out = []
            for count, src in self._parts:
                if callable(src):
                    out.extend(map(src, range(count)))
                else:
                    fresh = list(map(LazyPacket, src.templates()))
                    for p in fresh:
                        p._src = src
                    out += fresh
            self._list = out
        return out

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._packets()[i]

    def __iter__(self):
        return iter(self._packets())

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return f"TickPackets({list(self)!r})"


def channel_update_builder(cids, states, flags, tows, tow_decoded, since_tow_ms, unread, code_since_tow):
    """build(i) -> the (eager) CHANNEL_UPDATE packet of row i."""
    def build(i):
        return {"cid": int(cids[i]), "type": ChannelMessage.CHANNEL_UPDATE, "state": states[i],
                "tracking_flags": _flags(int(flags[i])), "tow": float(tows[i]) if tow_decoded[i] else 0,
                "time_since_tow": float(since_tow_ms[i]),
                "unprocessed_samples": int(unread[i]), "code_since_tow": int(code_since_tow[i])}
    return build


__all__ = ["ChannelBank", "TickPackets", "LazyPacket", "TrackingRows", "UpdateRows", "packet_templates", "tracking_packet",
           "tracking_packets_builder", "channel_update_builder", "KIND_BORRE", "KIND_KAPLAN", "ChannelState"]
