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

While a block is handed out the NEXT one is already on the device (`ChannelManager._track_ahead`: its samples in the ring,
its epochs queued with `sdr_bank_step_begin`); the tick that needs it only collects the results -- no tick waits for a
launch.  It is queued only where the plain loop could not tell the difference: the ring must hold both blocks beside
what every active channel has not read yet, and while a channel is still IDLE (it would start reading at ring position 0)
no block -- the one handed out or the one queued ahead -- crosses the ring's end, and the one queued ahead does not start
there either: samples written at ring position 0 before the write index's own wrap would be searched by a channel started
meanwhile in place of the stale ones the plain loop still holds.

The reference's calls do not change.  What differs while a block is replayed: attributes read from a channel OBJECT
(carrierFrequency, currentSample, navBits ...) show the state at the END of the block; the packets -- all the
reference's receiver ever sees of a channel, which lives in another process there -- are those of the tick.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .bank import tracking_packets_builder
from .navdecoder import HOST_FLAGS
from ..utils.enumerations import ChannelMessage


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
        # ---- which tick releases which epoch, and what every tick's channel updates report: one pass in the library
        # (sdr_block_schedule, csrc/schedule.hip -- host code; the same in NumPy array operations was 0.4 ms per block)
        lib = _lib.load()
        records = np.ascontiguousarray(records)
        n_cols = records.shape[1]
        done32 = np.ascontiguousarray(done, dtype=np.int32)
        unread64 = np.ascontiguousarray(unread_now, dtype=np.int64)
        flags0 = bank.state["track_flags"][channels].astype(np.int64)
        since0 = np.ascontiguousarray(bank.code_since_tow[channels], dtype=np.int64)
        max_ticks = n_cols + 8
        total = int(done32.sum())
        first = np.empty((n_ch, n_cols), dtype=np.int32)
        n_ticks_c = C.c_int32(0)
        rows_s, cols_s = np.empty(total, dtype=np.int32), np.empty(total, dtype=np.int32)
        starts = np.empty(max_ticks + 1, dtype=np.int32)
        records_sorted = np.empty(total, dtype=records.dtype)
        last_records = np.empty(n_ch, dtype=records.dtype)
        unread = np.empty((max_ticks, n_ch), dtype=np.int64)
        dev_flags, code_count = np.empty_like(unread), np.empty_like(unread)
        last = np.empty(n_ch, dtype=np.int32)
        bit_rows, bit_cols, bit_vals = (np.empty(total, dtype=np.int32) for _ in range(3))
        n_bits_c = C.c_int32(0)
        status = lib.sdr_block_schedule(records.ctypes.data, n_ch, n_cols, done32.ctypes.data, unread64.ctypes.data, self.spt,
                                        flags0.ctypes.data, since0.ctypes.data, max_ticks, first.ctypes.data, C.byref(n_ticks_c),
                                        rows_s.ctypes.data, cols_s.ctypes.data, starts.ctypes.data, records_sorted.ctypes.data,
                                        last_records.ctypes.data, unread.ctypes.data, dev_flags.ctypes.data, code_count.ctypes.data,
                                        last.ctypes.data, bit_rows.ctypes.data, bit_cols.ctypes.data, bit_vals.ctypes.data, C.byref(n_bits_c))
        if status:
            _lib.check(status)
        n_ticks = self.n_ticks = n_ticks_c.value
        unread, dev_flags, code_count = unread[:n_ticks], dev_flags[:n_ticks], code_count[:n_ticks]
        self._starts = starts[:n_ticks + 1].tolist()
        self._cids_sorted = self.cids64[rows_s]
        self._records_sorted = records_sorted
        self.busy[channels[done > 0]] = True
        self.last_tick[channels] = last
        by_tick = {}                                       # tick -> the channels whose last computed epoch it releases
        for ch, t in zip(channels.tolist(), last.tolist()):
            if t >= 0:
                by_tick.setdefault(t, []).append(ch)
        self._finishing = {t: np.array(chs, dtype=channels.dtype) for t, chs in by_tick.items()}
        count = code_count - since0[None, :]                # (epochs released so far: a subframe restarts the code count from it)
        rec_flags = records["track_flags"]
        decoders = [bank.decoders[int(c)] for c in channels]
        if any(d is not None for d in decoders):            # (the decoders own flag bits and the time of week: per tick from here on)
            host = np.repeat(bank.host_flags[channels][None, :], n_ticks, axis=0)
            tow = np.repeat(bank.tow[channels][None, :], n_ticks, axis=0)
            tow_dec = np.repeat(bank.tow_decoded[channels][None, :], n_ticks, axis=0)
        else:                                                # (nobody writes them: one row serves every tick)
            host = np.broadcast_to(bank.host_flags[channels], (n_ticks, n_ch))
            tow = np.broadcast_to(bank.tow[channels], (n_ticks, n_ch))
            tow_dec = np.broadcast_to(bank.tow_decoded[channels], (n_ticks, n_ch))

        # ---- the block's navigation bits: kept per channel in epoch order, and through the channel's decoder when it
        # has one (navdecoder.py)
        n_bits = n_bits_c.value
        if n_bits:
            bit_rows = bit_rows[:n_bits]
            bit_values = bit_vals[:n_bits].tolist()
            bit_epochs = bit_cols[:n_bits].tolist()
            bounds = np.searchsorted(bit_rows, np.arange(n_ch + 1)).tolist()
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
        if hi - lo + 1 == n_ch and bool(has.all()) and bool((np.diff(channels) == 1).all()):
            bank.last[lo:hi + 1] = last_records
        else:
            bank.last[channels[has]] = last_records[has]
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
