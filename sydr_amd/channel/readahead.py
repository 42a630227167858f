"""Read-ahead for the literal per-millisecond drop-in loop.

The reference's receiver feeds its ChannelManager one millisecond at a time (`addNewRFData(rfSignal.getMilliseconds(1));
run()`, receiver.py:120-131).  Served literally, every tick is one device call -- ~30 us of launch + synchronisation
around a 15 us kernel -- and the path runs at ~10x real time however fast the correlators are.  When the RF source is
this package's file reader the manager can look AHEAD in the recording instead: it uploads the next `K` milliseconds in
one copy, advances every tracking channel through the epochs they hold in ONE persistent launch (`sdr_bank_step`: loops
closed on the device, the arithmetic of K ticks), and then hands each tick exactly the packets that tick would have
produced: an epoch is released in the first tick whose slab completes it, at most one per channel and tick, as
`sdr_bank_tick` would have run it.  `addNewRFData(slab)` only checks that the slab IS the recording's next millisecond
and moves the write index; nothing is uploaded twice.

Everything a tick reports is worked out when the block is computed -- which epochs it releases, and for its
CHANNEL_UPDATE packets the unread samples, flags, code count and TOW of every channel as they stand after that tick
(the navigation bits of the block go through the channels' decoders in order at that point; a subframe they complete
is released with the tick of its bit) -- so a replayed tick costs a dictionary lookup.

The reference's calls do not change.  What differs while a block is replayed: attributes read from a channel OBJECT
(carrierFrequency, currentSample, navBits ...) show the state at the END of the block; the packets -- all the
reference's receiver ever sees of a channel, which lives in another process there -- are those of the tick.
"""
from __future__ import annotations

import numpy as np

from .bank import tracking_packets_builder
from .navdecoder import HOST_FLAGS
from ..utils.enumerations import ChannelMessage


def _take_records(records, rows, cols):
    """records[rows, cols] of a C-contiguous 2-D structured array whose item size is a multiple of 8, as a row gather
    over 64-bit words (NumPy's fancy indexing of structured items goes item by item through its generic copy: 200 us
    for the 1600 records of a block against ~10)."""
    dt = records.dtype
    if records.ndim != 2 or dt.itemsize % 8 or not records.flags.c_contiguous:
        return records[rows, cols]
    words = records.view(np.uint64).reshape(records.shape[0] * records.shape[1], dt.itemsize // 8)
    return words[np.asarray(rows) * records.shape[1] + np.asarray(cols)].view(dt).reshape(-1)


class EpochSchedule:
    """Epochs a block run computed ahead of the ticks that release them."""

    def __init__(self, bank, samples_per_tick: int):
        self.bank, self.spt = bank, int(samples_per_tick)
        self._starts = self._cids_sorted = self._records_sorted = None   # the block's epochs in tick order + the slice per tick
        self._finishing = {}         # tick number -> channels whose last computed epoch that tick releases
        self.decoded = {}            # tick number -> [DECODING_UPDATE packets]
        self.tick = 0                # ticks released so far
        self.n_ticks = 0             # ticks this block's epochs spread over
        self.busy = np.zeros(bank.max_channels, dtype=bool)   # channels with epochs still to be released
        self.last_tick = np.zeros(bank.max_channels, dtype=np.int64)
        self.raw = None              # the prefetched samples
        self.slabs_left = 0
        self.version = object()      # the ring's stateVersion when the block was computed (manager.run's fast path)
        self.cids64 = np.zeros(0, dtype=np.int64)
        self.row_of = np.full(bank.max_channels, -1, dtype=np.int64)
        self.upd = None              # per tick and scheduled channel: what a CHANNEL_UPDATE reports
        self.covers_active = False   # the scheduled channels are exactly the manager's active ones, in its order

    @property
    def empty(self) -> bool:
        return self.tick >= self.n_ticks

    def follow(self, bank):
        """The manager's bank was re-created to hold more channels (addChannel during a replay): its mirror rows were
        copied over, so the schedule simply moves to it."""
        if bank is not self.bank:
            grow = bank.max_channels - len(self.busy)
            if grow > 0:
                self.busy = np.concatenate([self.busy, np.zeros(grow, dtype=bool)])
                self.last_tick = np.concatenate([self.last_tick, np.full(grow, -1, dtype=np.int64)])
                self.row_of = np.concatenate([self.row_of, np.full(grow, -1, dtype=np.int64)])
            self.bank = bank

    def load(self, channels, records, done, states, unread_now):
        """Schedule `done[r]` epochs of `channels[r]`: epoch e is released by the first tick k whose slab completes it
        (unread_now + (k + 1) * spt >= samples up to its end), one epoch per channel and tick; move the bank's mirror
        to the end of the block and work out every tick's channel updates."""
        bank, spt = self.bank, self.spt
        n_ch = len(channels)
        self.tick, self.decoded, self.n_ticks = 0, {}, 0
        self._starts = self._cids_sorted = self._records_sorted = None
        self._finishing = {}
        self.cids64 = channels.astype(np.int64)
        self.row_of[:] = -1
        self.row_of[self.cids64] = np.arange(n_ch)
        n_max = int(done.max()) if n_ch else 0
        if n_max == 0:
            return
        lengths = records["n_samples"][:, :n_max].astype(np.int64)
        ends = np.cumsum(lengths, axis=1)
        first = np.maximum(0, -(-(ends - unread_now[:, None]) // spt) - 1)          # ceil(.) - 1
        valid = np.arange(n_max)[None, :] < done[:, None]
        # at most one epoch per channel and tick: first[e] >= first[e - 1] + 1, i.e. first[e] - e never decreases
        steps = np.arange(n_max)[None, :]
        first = np.maximum.accumulate(first - steps, axis=1) + steps
        first = np.where(valid, first, -1)
        n_ticks = self.n_ticks = int(first.max()) + 1
        rows, cols = np.nonzero(valid)
        ticks = first[rows, cols]
        # the epochs in the order of their ticks (channels ascending inside a tick): tick k releases one slice of these
        order = np.argsort(ticks, kind="stable")
        rows_s, cols_s = rows[order], cols[order]
        self._starts = np.searchsorted(ticks[order], np.arange(n_ticks + 1)).tolist()
        self._cids_sorted = self.cids64[rows_s]
        self._records_sorted = _take_records(records, rows_s, cols_s)
        self.busy[channels[done > 0]] = True
        last = first.max(axis=1)
        self.last_tick[channels] = last
        for t in np.unique(last[done > 0]).tolist():       # tick -> the channels whose last computed epoch it releases
            self._finishing[t] = channels[(last == t) & (done > 0)]

        # ---- per tick: samples consumed, device flags, code count (one epoch per channel and tick at most)
        epoch_at = np.full((n_ticks, n_ch), -1, dtype=np.int64)                     # epoch released by (tick, channel)
        epoch_at[ticks, rows] = cols
        ran = epoch_at >= 0
        safe = np.where(ran, epoch_at, 0)
        ch_rows = np.arange(n_ch)[None, :]
        consumed = np.cumsum(np.where(ran, lengths[ch_rows, safe], 0), axis=0)
        unread = unread_now[None, :] + (np.arange(n_ticks)[:, None] + 1) * spt - consumed
        latest = np.maximum.accumulate(np.where(ran, epoch_at, -1), axis=0)          # newest released epoch so far
        flags0 = bank.state["track_flags"][channels].astype(np.int64)
        rec_flags = records["track_flags"]
        dev_flags = np.where(latest >= 0, rec_flags[ch_rows, np.maximum(latest, 0)], flags0[None, :])
        count = np.cumsum(ran, axis=0)
        code_count = bank.code_since_tow[channels][None, :] + count
        host = np.repeat(bank.host_flags[channels][None, :], n_ticks, axis=0)
        tow = np.repeat(bank.tow[channels][None, :], n_ticks, axis=0)
        tow_dec = np.repeat(bank.tow_decoded[channels][None, :], n_ticks, axis=0)

        # ---- the block's navigation bits: kept per channel in epoch order, and through the channel's decoder when it
        # has one (navdecoder.py)
        nav = records["nav_bit"][:, :n_max]
        bit_rows, bit_cols = np.nonzero((nav >= 0) & valid)               # (row-major: a channel's bits in epoch order)
        if len(bit_rows):
            bit_values = nav[bit_rows, bit_cols].tolist()
            bounds = np.searchsorted(bit_rows, np.arange(n_ch + 1)).tolist()
            bit_epochs = bit_cols.tolist()
            for r in np.unique(bit_rows).tolist():
                ch = int(channels[r])
                lo, hi = bounds[r], bounds[r + 1]
                decoder = bank.decoders[ch]
                if decoder is None:
                    bank.nav_bits[ch].extend(bit_values[lo:hi])
                    continue
                for e, bit in zip(bit_epochs[lo:hi], bit_values[lo:hi]):
                    k = int(first[r, e])
                    bank.nav_bits[ch].append(bit)
                    flags, event = decoder.push(bit, int(rec_flags[r, e]) | int(host[k, r]))
                    host[k:, r] = flags & HOST_FLAGS
                    if event is not None:
                        tow[k:, r], tow_dec[k:, r] = event.channel_tow, True
                        code_count[k:, r] = count[k:, r] - count[k, r]              # (kaplan:833: the count restarts here)
                        self.decoded.setdefault(k, []).append({"cid": ch, "type": ChannelMessage.DECODING_UPDATE,
                                                               "subframe_id": event.subframe_id, "tow": event.tow,
                                                               "bits": event.bits})
        self.upd = dict(unread=unread, flags=dev_flags | host, code=code_count, tow=tow, tow_dec=tow_dec)

        # ---- the mirror moves to the end of the block
        has = done > 0
        lo, hi = int(channels[0]), int(channels[-1])
        if hi - lo + 1 == n_ch and bool((np.diff(channels) == 1).all()):   # (the usual case: a run of channel numbers)
            bank.state[lo:hi + 1] = states
        else:
            bank.state[channels] = states
        bank.last[channels[has]] = _take_records(records, np.flatnonzero(has), done[has] - 1)
        bank.code_since_tow[channels] = code_count[-1]
        bank.host_flags[channels], bank.tow[channels], bank.tow_decoded[channels] = host[-1], tow[-1], tow_dec[-1]

    def release(self):
        """((channel ids, records) | None, DECODING_UPDATE packets | None) of the tick that has just received its slab."""
        k = self.tick
        self.tick += 1
        entry = None
        if self._starts is not None and k < self.n_ticks:
            lo, hi = self._starts[k], self._starts[k + 1]
            if hi > lo:
                entry = (self._cids_sorted[lo:hi], self._records_sorted[lo:hi])
        if k + 1 >= self.n_ticks:
            self.busy[:] = False
        else:
            finishing = self._finishing.get(k)
            if finishing is not None:
                self.busy[finishing] = False
        return entry, self.decoded.pop(k, None)

    def updates(self, k):
        """What the CHANNEL_UPDATE packets of tick k report for the scheduled channels (rows in the order of cids64)."""
        u = self.upd
        k = min(k, self.n_ticks - 1)
        return u["unread"][k], u["flags"][k], u["code"][k], u["tow"][k], u["tow_dec"][k]


def packets_of(bank, cids, recs):
    return len(cids), tracking_packets_builder(cids, bank.cfg["loop_kind"][cids], recs)
