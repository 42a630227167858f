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
        self.close()
        return big

    # ------------------------------------------------------------------ mirror <-> HBM
    def touch(self, ch: int):
        self._dirty[ch] = True

    def flush(self):
        """Upload every channel whose mirror was written by the host since the last step."""
        for ch in np.flatnonzero(self._dirty):
            self.device.put(int(ch), self.state[ch], self.cfg[ch])
        self._dirty[:] = False

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


def tracking_packets_builder(cids, kinds, records):
    """build(i) -> the TRACKING_UPDATE packet of row i, for a whole tick's records at once: the structured array is
    turned into plain Python tuples in ONE call the first time any packet is read (field access on NumPy records costs
    more per packet than the dict itself)."""
    names = records.dtype.names
    at = {name: k for k, name in enumerate(names)}
    i_corr, i_chz, i_code, i_cerr, i_coderr = at["corr"], at["carrier_hz"], at["code_hz"], at["carrier_err"], at["code_err"]
    i_dll, i_pll, i_fll, i_cn0, i_plock, i_flock, i_lock = (at["dll"], at["pll"], at["fll"], at["cn0"], at["pll_lock"],
                                                            at["fll_lock"], at["lock_state"])
    rows = []

    def build(i):
        if not rows:
            rows.extend(records.tolist())
        r = rows[i]
        corr = r[i_corr]
        pkt = {"cid": int(cids[i]), "type": ChannelMessage.TRACKING_UPDATE,
               "i_early": corr[0], "q_early": corr[1], "i_prompt": corr[2], "q_prompt": corr[3], "i_late": corr[4],
               "q_late": corr[5], "carrier_frequency": r[i_chz], "code_frequency": r[i_code],
               "carrier_frequency_error": r[i_cerr], "code_frequency_error": r[i_coderr],
               "dll": r[i_dll], "pll": r[i_pll], "fll": r[i_fll]}
        if kinds[i] == KIND_KAPLAN:
            pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"] = r[i_cn0], r[i_plock], r[i_flock]
            pkt["lock_state"] = LoopLockState(r[i_lock])
        else:
            pkt["cn0"], pkt["pll_lock"], pkt["fll_lock"], pkt["lock_state"] = np.nan, 0.0, 0.0, 0
        return pkt
    return build


class TickPackets(Sequence):
    """The flat packet list `ChannelManager.run()` returns (channelManager.py:149-188), built on demand.

    Everything a packet needs is captured when the tick ends (epoch records, flags, unread counts), so reading it
    later gives what an eager list would have held; the dicts themselves are only made when somebody looks --
    at 32 channels they cost more host time than the whole device step."""

    def __init__(self):
        self._parts = []      # (count, builder(i) -> dict)
        self._cache = {}
        self._n = 0

    def add(self, count: int, builder):
        if count:
            self._parts.append((self._n, count, builder))
            self._n += count

    def add_ready(self, packets):
        packets = list(packets)
        self.add(len(packets), packets.__getitem__)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        hit = self._cache.get(i)
        if hit is None:
            for first, count, builder in self._parts:
                if first <= i < first + count:
                    hit = self._cache[i] = builder(i - first)
                    break
        return hit

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return f"TickPackets({list(self)!r})"


def channel_update_builder(cids, states, flags, tows, tow_decoded, since_tow_ms, unread, code_since_tow):
    """CHANNEL_UPDATE packets (channel.py:205-228) from values captured at the end of the tick.  `tow` is what the
    reference's `Channel.tow` holds: the int 0 until a subframe was decoded, then HOW TOW + 1.24 s (kaplan:810-822)."""
    def build(i):
        return {"cid": int(cids[i]), "type": ChannelMessage.CHANNEL_UPDATE, "state": states[i],
                "tracking_flags": _flags(int(flags[i])), "tow": float(tows[i]) if tow_decoded[i] else 0,
                "time_since_tow": float(since_tow_ms[i]),
                "unprocessed_samples": int(unread[i]), "code_since_tow": int(code_since_tow[i])}
    return build


__all__ = ["ChannelBank", "TickPackets", "tracking_packet", "tracking_packets_builder", "channel_update_builder", "KIND_BORRE", "KIND_KAPLAN",
           "ChannelState"]
