"""Engine: one MI355X's correlator state (IQ ring, staged PRN replicas, work buffers)
behind the C-ABI.  NumPy arrays in, NumPy arrays out; the arithmetic all runs in the HIP
library (there is deliberately no host implementation to fall back to)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (EPL_ITEM_DTYPE, FMT_CF32, FMT_CF64, FMT_CI16, FMT_CI8, LOOP_CFG_DTYPE, TRACK_EPOCH_DTYPE,
                   TRACK_STATE_DTYPE, LoopCfg, SynthSat, TrackState, check, ptr)

__all__ = ["Engine", "EplPlan", "Bank", "make_items", "FMT_CI8", "FMT_CI16", "FMT_CF32", "FMT_CF64"]


def make_items(code_slot, n_samples, start_sample, carrier_hz, rem_carrier, rem_code, code_step) -> np.ndarray:
    """Pack per-epoch NCO parameters (scalars or equal-length arrays) into sdr_epl_item records."""
    arrs = np.broadcast_arrays(code_slot, n_samples, start_sample, carrier_hz, rem_carrier, rem_code, code_step)
    items = np.zeros(arrs[0].shape, dtype=EPL_ITEM_DTYPE).reshape(-1)
    for name, a in zip(EPL_ITEM_DTYPE.names, arrs):
        items[name] = np.asarray(a).reshape(-1)
    return items


_addressof = C.addressof
_char_from_buffer = C.c_char.from_buffer


class EplPlan:
    """Items + outputs resident in HBM; run() only launches kernels (asynchronous)."""

    def __init__(self, engine: "Engine", items, spacing, fs: float, device_items: int = 0, n_items: int = 0):
        """items: the list (host array), or device_items: the address of n_items items in DEVICE memory
        (sdr_epl_plan_create_dev: copied there and checked there)."""
        self._e = engine
        self._lib = _lib.load()
        spacing = np.ascontiguousarray(spacing, dtype=np.float64)
        self.n_taps = len(spacing)
        self._h = C.c_void_p()
        if device_items:
            self.n_items = int(n_items)
            check(self._lib.sdr_epl_plan_create_dev(engine._h, C.c_void_p(int(device_items)), self.n_items, ptr(spacing),
                                                    self.n_taps, float(fs), C.byref(self._h)))
            return
        items = np.ascontiguousarray(items, dtype=EPL_ITEM_DTYPE)
        self.n_items = len(items)
        check(self._lib.sdr_epl_plan_create(engine._h, ptr(items), self.n_items, ptr(spacing), self.n_taps,
                                            float(fs), C.byref(self._h)))

    def run(self, first=None, count=None, stream=0):
        """Asynchronous launch of the whole plan or of items [first, first+count), on the engine's stream or on
        one made by Engine.stream_create() (one stream per channel batch)."""
        if first is None:
            first, count = 0, self.n_items
        check(self._lib.sdr_epl_plan_run_range_on(self._e._h, self._h, int(first), int(count), int(stream)))

    @property
    def variant(self) -> int:
        """Which correlator variant the items selected (sdr_epl_plan_variant: diagnostics)."""
        return int(self._lib.sdr_epl_plan_variant(self._h))

    def fetch(self) -> np.ndarray:
        out = np.empty((self.n_items, 2 * self.n_taps), dtype=np.float64)
        check(self._lib.sdr_epl_plan_fetch(self._e._h, self._h, ptr(out)))
        return out

    def close(self):
        if self._h:
            self._lib.sdr_epl_plan_destroy(self._e._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bank:
    """The tracking state of one GPU's channels, resident in HBM (sdr_bank_*): `put` a channel after acquisition,
    `step` / `tick` advance the listed channels on the device and hand back the epoch records and the new states."""

    def __init__(self, engine: "Engine", max_channels: int):
        self._e = engine
        self._lib = _lib.load()
        self.max_channels = int(max_channels)
        self._h = C.c_void_p()
        check(self._lib.sdr_bank_create(engine._h, self.max_channels, C.byref(self._h)))

    def put(self, ch: int, state: np.ndarray, cfg: np.ndarray):
        state = np.ascontiguousarray(state, dtype=TRACK_STATE_DTYPE).reshape(1)
        cfg = np.ascontiguousarray(cfg, dtype=LOOP_CFG_DTYPE).reshape(1)
        check(self._lib.sdr_bank_put(self._e._h, self._h, int(ch), ptr(state), ptr(cfg)))

    def get(self, ch: int) -> np.ndarray:
        out = np.zeros(1, dtype=TRACK_STATE_DTYPE)
        check(self._lib.sdr_bank_get(self._e._h, self._h, int(ch), ptr(out)))
        return out[0]

    def step(self, channels, n_epochs: int = 1, want_records=True, want_bits=False, stream=0, epochs_per_bit=20):
        """-> (records[n][n_epochs] or None, states[n], epochs_done[n], list of nav-bit arrays or None)"""
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        n = len(channels)
        rec = np.zeros((n, n_epochs), dtype=TRACK_EPOCH_DTYPE) if want_records else None
        states = np.zeros(n, dtype=TRACK_STATE_DTYPE)
        done = np.zeros(n, dtype=np.int32)
        max_bits = n_epochs // max(1, int(epochs_per_bit)) + 2
        bits = np.zeros((n, max_bits), dtype=np.int8) if want_bits else None
        n_bits = np.zeros(n, dtype=np.int32) if want_bits else None
        check(self._lib.sdr_bank_step(self._e._h, self._h, ptr(channels), n, int(n_epochs),
                                      ptr(rec) if want_records else None, ptr(states), ptr(done),
                                      ptr(bits) if want_bits else None, max_bits if want_bits else 0,
                                      ptr(n_bits) if want_bits else None, int(stream)))
        nav = [bits[c, :n_bits[c]].copy() for c in range(n)] if want_bits else None
        return rec, states, done, nav

    def step_begin(self, channels, n_epochs: int):
        """`step` without the wait (sdr_bank_step_begin): the launch and the copies of its results are queued on the
        engine's stream; `step_end()` hands them out.  One step in flight per bank."""
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        check(self._lib.sdr_bank_step_begin(self._e._h, self._h, ptr(channels), len(channels), int(n_epochs)))
        self._in_flight = (len(channels), int(n_epochs))

    def step_end(self):
        """-> (records[n][n_epochs], states[n], epochs_done[n]) of the step `step_begin` queued."""
        if getattr(self, "_in_flight", None) is None:
            # (never ask the library: a step begun through another wrapper would be collected into nothing and discarded)
            raise RuntimeError("step_end(): no step of this bank was begun through this object (step_begin first)")
        n, n_epochs = self._in_flight
        self._in_flight = None
        rec = np.empty((n, n_epochs), dtype=TRACK_EPOCH_DTYPE)
        states = np.empty(n, dtype=TRACK_STATE_DTYPE)
        done = np.zeros(n, dtype=np.int32)
        check(self._lib.sdr_bank_step_end(self._e._h, self._h, ptr(rec), ptr(states), ptr(done)))
        return rec, states, done

    def tick(self, raw, ring_offset: int, channels):
        """One receiver tick: `raw` (the ring's format, may be None) goes into the ring at ring_offset, then the
        listed channels run one epoch.  -> (records[n], states[n], epochs_done[n])"""
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        n = len(channels)
        rec = np.empty(n, dtype=TRACK_EPOCH_DTYPE)          # (all three are filled by the call)
        states = np.empty(n, dtype=TRACK_STATE_DTYPE)
        done = np.zeros(n, dtype=np.int32)
        n_samples = 0
        if raw is not None:
            raw = self._e._ring_samples(raw)
            n_samples = raw.size // 2
        check(self._lib.sdr_bank_tick(self._e._h, self._h, ptr(raw) if n_samples else None, n_samples, int(ring_offset),
                                      ptr(channels) if n else None, n, ptr(rec) if n else None,
                                      ptr(states) if n else None, ptr(done) if n else None))
        return rec, states, done

    def bind_mirror(self, states, last, epochs_since_tow, tracking, lost, host_flags):
        """The caller's mirrors of the bank (NumPy arrays of max_channels rows, kept alive by the caller) as the
        sdr_tick_mirror `tick_mirrored` updates in place.  Per-tick outputs: `ran`, `records`, `updates` (views of
        arrays this object owns: copy what has to outlive the next tick)."""
        n = self.max_channels
        for a, dt in ((states, TRACK_STATE_DTYPE), (last, TRACK_EPOCH_DTYPE), (epochs_since_tow, np.int64),
                      (tracking, np.bool_), (lost, np.bool_), (host_flags, np.int64)):
            if a.dtype != dt or a.shape != (n,) or not a.flags.c_contiguous:
                raise ValueError("tick mirror arrays must be contiguous, one row per bank channel, in the bank's dtypes")
        self._mirror_arrays = (states, last, epochs_since_tow, tracking, lost, host_flags)
        # TWO sets of per-tick outputs, used alternately (`out_set`): what a tick left in `ran` / `records` / `updates` stays
        # untouched through the NEXT tick, so a caller that hands out views of them (the channel bank's lazy packets) need not
        # copy 8 KB per millisecond -- only what is still looked at when the set comes round again (ChannelBank.hold)
        self._sets = []
        for _ in range(2):
            ran = np.zeros(n, dtype=np.int32)
            records = np.zeros(n, dtype=TRACK_EPOCH_DTYPE)
            updates = np.zeros(n, dtype=_lib.TICK_UPDATE_DTYPE)
            m = _lib.TickMirror()
            m.max_channels = n
            m.states, m.last, m.epochs_since_tow = states.ctypes.data, last.ctypes.data, epochs_since_tow.ctypes.data
            m.tracking, m.lost, m.host_flags = tracking.ctypes.data, lost.ctypes.data, host_flags.ctypes.data
            m.ran, m.records, m.updates = ran.ctypes.data, records.ctypes.data, updates.ctypes.data
            self._sets.append((ran, records, updates, m, C.byref(m)))
        self.out_set = 0
        self.ran, self.records, self.updates, self._mirror, self._mirror_ref = self._sets[0]
        self._tick_no_slab = (self._e._h, self._h, None, 0)        # leading arguments of a tick whose slab is queued already
        return self._mirror

    double_buffered = True      # (per-tick outputs alternate between two sets: see bind_mirror)

    def _next_set(self):
        k = self.out_set = self.out_set ^ 1
        self.ran, self.records, self.updates, self._mirror, self._mirror_ref = self._sets[k]

    def tick_mirrored(self, raw, ring_offset: int, write_index: int):
        """One receiver tick with the readiness test and the mirror updates in the library (sdr_bank_tick_mirrored);
        `raw` None when the slab went in with Engine.iq_upload_begin.  -> the bound sdr_tick_mirror (n_ran, n_updates,
        n_nav_bits, n_lost, max_unread; the rows in `ran` / `records` / `updates`)."""
        self._next_set()
        if raw is None:
            status = self._lib.sdr_bank_tick_mirrored(*self._tick_no_slab, ring_offset, write_index, self._mirror_ref)
        else:
            raw = self._e._ring_samples(raw)
            status = self._lib.sdr_bank_tick_mirrored(self._e._h, self._h, ptr(raw), raw.size // 2, ring_offset, write_index,
                                                      self._mirror_ref)
        if status:
            check(status)
        return self._mirror

    def tick_mirrored_begin(self, raw, ring_offset: int, write_index: int):
        """First half of `tick_mirrored` (sdr_bank_tick_mirrored_begin): who is ready is decided and their epoch queued on
        the engine's stream; nothing is waited for.  A manager of several devices begins every device's tick, then ends
        each."""
        self._next_set()
        if raw is None:
            status = self._lib.sdr_bank_tick_mirrored_begin(*self._tick_no_slab, ring_offset, write_index, self._mirror_ref)
        else:
            raw = self._e._ring_samples(raw)
            status = self._lib.sdr_bank_tick_mirrored_begin(self._e._h, self._h, ptr(raw), raw.size // 2, ring_offset, write_index,
                                                            self._mirror_ref)
        if status:
            check(status)

    def tick_mirrored_end(self):
        """Second half: wait, mirrors and per-tick rows updated in place.  -> the bound sdr_tick_mirror."""
        status = self._lib.sdr_bank_tick_mirrored_end(self._e._h, self._h, self._mirror_ref)
        if status:
            check(status)
        return self._mirror

    def close(self):
        if self._h and self._e._h:
            self._lib.sdr_bank_destroy(self._e._h, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    def __init__(self, device_id: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.sdr_engine_create(int(device_id), C.byref(self._h)))
        self.device_id = device_id
        self.iq_fmt = None
        self.iq_capacity = 0
        self._host_blocks = {}        # page-locked blocks handed out by host_alloc: address -> block

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if self._h:
            for block in list(getattr(self, "_host_blocks", {}).values()):
                self._lib.sdr_host_free(self._h, block)
            self._host_blocks = {}
            self._lib.sdr_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.sdr_engine_sync(self._h))

    def prof_enable(self, on=True, calls_only=False):
        """Per-stage HIP-event timing (`prof_read`); calls_only: one event pair around each whole call instead
        (scopes named "call_*": the in-stream time of everything the call launched)."""
        check(self._lib.sdr_prof_enable(self._h, (2 if calls_only else 1) if on else 0))

    def prof_reset(self):
        check(self._lib.sdr_prof_reset(self._h))

    def prof_read(self, prefix=""):
        tot, cnt = C.c_double(0), C.c_int64(0)
        check(self._lib.sdr_prof_read(self._h, prefix.encode(), C.byref(tot), C.byref(cnt)))
        return tot.value, cnt.value

    def set_option(self, name: str, value: int):
        check(self._lib.sdr_set_option(self._h, name.encode(), int(value)))

    def hbm_copy_rate(self, n_bytes: int = 1 << 30, reps: int = 10) -> float:
        """Measured GB/s (read + written) of a hand-written stream-copy kernel on this GPU."""
        g = C.c_double(0)
        check(self._lib.sdr_hbm_copy_rate(self._h, int(n_bytes), int(reps), C.byref(g)))
        return g.value

    # ------------------------------------------------------------------ IQ ring
    def iq_alloc(self, capacity_samples: int, fmt: int = FMT_CI8):
        check(self._lib.sdr_iq_alloc(self._h, int(capacity_samples), int(fmt)))
        self.iq_fmt, self.iq_capacity = fmt, int(capacity_samples)

    def _ring_samples(self, raw) -> np.ndarray:
        """Interleaved I,Q in the ring's element type (complex input is split; no copy when already right)."""
        dt = _lib.fmt_dtype(self.iq_fmt)
        if np.iscomplexobj(raw):
            raw = np.ascontiguousarray(raw, dtype=np.complex128).view(np.float64)
            if self.iq_fmt != FMT_CF64:
                raw = raw.astype(dt)
        raw = np.ascontiguousarray(raw, dtype=dt).reshape(-1)
        if raw.size % 2:
            raise ValueError("interleaved IQ needs an even number of elements")
        return raw

    def iq_upload(self, raw: np.ndarray, ring_offset: int = 0):
        """raw: interleaved I,Q in the ring's element type (complex128 accepted for FMT_CF64)."""
        raw = self._ring_samples(raw)
        check(self._lib.sdr_iq_upload(self._h, ptr(raw), raw.size // 2, int(ring_offset)))

    def iq_upload_begin(self, raw: np.ndarray, ring_offset: int = 0):
        """iq_upload without the wait (sdr_iq_upload_begin): `raw` is copied before the call returns, the transfer is
        ordered before everything queued on the engine's stream afterwards; `sync()` completes it."""
        if not (type(raw) is np.ndarray and raw.ndim == 1 and raw.flags.c_contiguous and raw.dtype == _lib.fmt_dtype(self.iq_fmt)):
            raw = self._ring_samples(raw)
        # (the array's address through the buffer protocol: `raw.ctypes.data` builds a helper object per call -- a microsecond of
        # a tick that is made of thirty; a read-only array has no writable buffer to offer and takes the long way)
        try:
            address = _addressof(_char_from_buffer(raw))
        except (TypeError, ValueError):
            address = raw.ctypes.data
        status = self._lib.sdr_iq_upload_begin(self._h, address, raw.size >> 1, ring_offset)
        if status:
            check(status)

    def iq_upload_queue(self, raw: np.ndarray, ring_offset: int = 0):
        """A chunk of a recording queued for the ring without a copy of its own (sdr_iq_upload_queue): `raw` -- contiguous,
        the ring's element type -- must stay alive and unchanged until `sync()`.  From `host_alloc` memory the call returns at
        once and the transfer overlaps what other streams compute."""
        if not (isinstance(raw, np.ndarray) and raw.ndim == 1 and raw.flags.c_contiguous and raw.dtype == _lib.fmt_dtype(self.iq_fmt)
                and not raw.size & 1):
            raise ValueError("iq_upload_queue takes a contiguous 1-D array of interleaved I,Q in the ring's element type")
        status = self._lib.sdr_iq_upload_queue(self._h, raw.ctypes.data, raw.size >> 1, int(ring_offset))
        if status:
            check(status)

    def host_alloc(self, n_elements: int, dtype=np.int8) -> np.ndarray:
        """Page-locked host memory as a NumPy array (sdr_host_alloc); `host_free(array)` gives it back -- the array must not
        be used afterwards."""
        dtype = np.dtype(dtype)
        block = C.c_void_p()
        check(self._lib.sdr_host_alloc(self._h, int(n_elements) * dtype.itemsize, C.byref(block)))
        buf = (C.c_char * (int(n_elements) * dtype.itemsize)).from_address(block.value)
        arr = np.frombuffer(buf, dtype=dtype)
        self._host_blocks[arr.ctypes.data] = block
        return arr

    def host_free(self, arr: np.ndarray):
        block = self._host_blocks.pop(arr.ctypes.data, None)
        if block is None:
            raise ValueError("not a block of host_alloc (or freed already)")
        check(self._lib.sdr_host_free(self._h, block))

    def iq_download(self, n_samples: int, ring_offset: int = 0) -> np.ndarray:
        out = np.empty(2 * int(n_samples), dtype=_lib.fmt_dtype(self.iq_fmt))
        check(self._lib.sdr_iq_download(self._h, ptr(out), int(n_samples), int(ring_offset)))
        return out

    def iq_synth(self, sats, fs, noise_sigma, seed, first_sample, n_samples):
        arr = (SynthSat * max(1, len(sats)))()
        for i, s in enumerate(sats):
            arr[i].prn = int(s["slot"]) if "slot" in s else int(s["prn"])
            arr[i].flags = (1 if "slot" in s else 0) | (2 if s.get("boc") else 0)
            arr[i].doppler_hz = float(s["doppler"])
            arr[i].code_phase = float(s["code_phase"])
            arr[i].carrier_phase = float(s.get("phase", 0.0))
            arr[i].amplitude = float(s["amp"])
        check(self._lib.sdr_iq_synth(self._h, arr, len(sats), float(fs), float(noise_sigma), int(seed),
                                     int(first_sample), int(n_samples)))

    # ------------------------------------------------------------------ PRN replicas
    def code_slots(self, n_slots: int, max_chips: int = 1023, max_periods: int = 1):
        check(self._lib.sdr_code_slots_ex(self._h, int(n_slots), int(max_chips), int(max_periods)))
        self.n_slots = int(n_slots)
        self.code_generation = getattr(self, "code_generation", 0) + 1  # staged codes are gone

    def load_gps_code(self, slot: int, prn: int):
        check(self._lib.sdr_code_gps_l1ca(self._h, int(slot), int(prn)))

    def set_code(self, slot: int, chips):
        chips = np.ascontiguousarray(chips, dtype=np.int8)
        check(self._lib.sdr_code_custom(self._h, int(slot), ptr(chips), chips.size))

    def read_code(self, slot: int, max_chips: int = 65536) -> np.ndarray:
        out = np.empty(max_chips, dtype=np.int8)
        n = C.c_int(0)
        check(self._lib.sdr_code_read(self._h, int(slot), ptr(out), max_chips, C.byref(n)))
        return out[:n.value].copy()

    def upsample(self, slot: int, fs: float, n_samples: int) -> np.ndarray:
        out = np.empty(int(n_samples), dtype=np.int8)
        check(self._lib.sdr_code_upsample(self._h, int(slot), float(fs), int(n_samples), ptr(out)))
        return out

    # ------------------------------------------------------------------ correlators
    def epl_batch(self, items: np.ndarray, spacing, fs: float) -> np.ndarray:
        items = np.ascontiguousarray(items, dtype=EPL_ITEM_DTYPE)
        spacing = np.ascontiguousarray(spacing, dtype=np.float64)
        out = np.empty((len(items), 2 * len(spacing)), dtype=np.float64)
        check(self._lib.sdr_epl_batch(self._h, ptr(items), len(items), ptr(spacing), len(spacing), float(fs),
                                      ptr(out)))
        return out

    def epl_plan_dev(self, device_items: int, n_items: int, spacing, fs) -> EplPlan:
        """A plan of n_items items that are already in device memory at address device_items."""
        return EplPlan(self, None, spacing, fs, device_items=device_items, n_items=n_items)

    def epl_plan(self, items, spacing, fs) -> EplPlan:
        return EplPlan(self, items, spacing, fs)

    # ------------------------------------------------------------------ acquisition
    def pcps(self, code_slots, start_sample, fs, if_hz, doppler_range, doppler_step, coh=1, noncoh=1,
             want_map=False):
        slots = np.ascontiguousarray(code_slots, dtype=np.int32)
        n = len(slots)
        nbins = self._lib.sdr_pcps_bins(float(doppler_range), float(doppler_step))
        n_code = int(round(fs * 1023 / 1.023e6))
        pb = np.empty(n, dtype=np.int64)
        pc = np.empty(n, dtype=np.int64)
        pr = np.empty(n, dtype=np.float64)
        cmap = np.empty((n, nbins, n_code), dtype=np.float64) if want_map else None
        nb = C.c_int(0)
        check(self._lib.sdr_pcps(self._h, ptr(slots), n, int(start_sample), float(fs), float(if_hz),
                                 float(doppler_range), float(doppler_step), int(coh), int(noncoh), ptr(pb),
                                 ptr(pc), ptr(pr), ptr(cmap) if want_map else None, C.byref(nb)))
        return pb, pc, pr, cmap

    def pcps_spectra(self, code_spectra, start_sample, fs, if_hz, doppler_range, doppler_step, coh=1, noncoh=1,
                     want_map=True):
        """PCPS with caller-supplied codeFFT rows (complex128 [n_prn][n_code]) -- the reference's PCPS() signature."""
        spec = np.ascontiguousarray(np.atleast_2d(code_spectra), dtype=np.complex128)
        n, n_code = spec.shape
        nbins = self._lib.sdr_pcps_bins(float(doppler_range), float(doppler_step))
        pb = np.empty(n, dtype=np.int64)
        pc = np.empty(n, dtype=np.int64)
        pr = np.empty(n, dtype=np.float64)
        cmap = np.empty((n, nbins, n_code), dtype=np.float64) if want_map else None
        nb = C.c_int(0)
        check(self._lib.sdr_pcps_spectra(self._h, ptr(spec), n, n_code, int(start_sample), float(fs), float(if_hz),
                                         float(doppler_range), float(doppler_step), int(coh), int(noncoh), ptr(pb),
                                         ptr(pc), ptr(pr), ptr(cmap) if want_map else None, C.byref(nb)))
        return pb, pc, pr, cmap

    def serial_search(self, code_slots, start_sample, fs, doppler_range, doppler_step, noncoh=1, want_map=False,
                      n_chips=1023):
        slots = np.ascontiguousarray(code_slots, dtype=np.int32)
        n = len(slots)
        nbins = self._lib.sdr_pcps_bins(float(doppler_range), float(doppler_step))
        pb, pc, pr = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64), np.empty(n, dtype=np.float64)
        cmap = np.empty((n, nbins, n_chips), dtype=np.float64) if want_map else None
        nb = C.c_int(0)
        check(self._lib.sdr_serial_search(self._h, ptr(slots), n, int(start_sample), float(fs), float(doppler_range),
                                          float(doppler_step), int(noncoh), ptr(pb), ptr(pc), ptr(pr),
                                          ptr(cmap) if want_map else None, C.byref(nb)))
        return pb, pc, pr, cmap

    def two_peak_compare_ss(self, cmap: np.ndarray):
        cmap = np.ascontiguousarray(cmap, dtype=np.float64)
        pb, pc, pr = C.c_int64(0), C.c_int64(0), C.c_double(0)
        check(self._lib.sdr_two_peak_compare_ss(self._h, ptr(cmap), cmap.shape[0], cmap.shape[1], C.byref(pb),
                                                C.byref(pc), C.byref(pr)))
        return [pb.value, pc.value], pr.value

    def two_peak_compare(self, cmap: np.ndarray, samples_per_chip: int):
        cmap = np.ascontiguousarray(cmap, dtype=np.float64)
        pb, pc, pr = C.c_int64(0), C.c_int64(0), C.c_double(0)
        check(self._lib.sdr_two_peak_compare(self._h, ptr(cmap), cmap.shape[0], cmap.shape[1],
                                             int(samples_per_chip), C.byref(pb), C.byref(pc), C.byref(pr)))
        return [pb.value, pc.value], pr.value

    # ------------------------------------------------------------------ closed loop
    def track_cluster(self, parts: int = 0):
        """Workgroups cooperating on one channel in `track_closed_loop` (0 = fill the GPU; 1, 2, 4, 8)."""
        check(self._lib.sdr_track_cluster(self._h, int(parts)))

    def tick_server_stats(self) -> dict:
        """The resident tick server of this engine (set_option("tick_server", 1)): is one resident now, requests answered,
        servers started, and whether the engine went back to plain ticks for good (a launch refused, a server that died)."""
        out = (C.c_int64 * 4)()
        check(self._lib.sdr_tick_server_stats(self._h, out))
        ph = (C.c_double * 4)()
        check(self._lib.sdr_tick_server_phases(self._h, ph))
        return {"running": bool(out[0]), "served": int(out[1]), "starts": int(out[2]), "disabled": bool(out[3]),
                "device_us_total": {"slab": ph[0], "release": ph[1], "channels": ph[2], "gather": ph[3]},
                "channel0_us_total": dict(zip(("release_seen", "samples_visible", "correlated", "exchanged", "updated", "answered"),
                                              self._tracker_phases()))}

    def _tracker_phases(self):
        t = (C.c_double * 6)()
        check(self._lib.sdr_tick_server_tracker_phases(self._h, t))
        return list(t)

    def stream_create(self) -> int:
        """A further HIP stream of this engine (north_star: one stream per channel batch); 0 is the default one."""
        sid = C.c_int(0)
        check(self._lib.sdr_stream_create(self._h, C.byref(sid)))
        return sid.value

    def stream_sync(self, stream: int = 0):
        check(self._lib.sdr_stream_sync(self._h, int(stream)))

    def bank(self, max_channels: int) -> Bank:
        return Bank(self, max_channels)

    def track_closed_loop_ex(self, states, cfgs, n_epochs: int, want_traj=True, want_bits=False, epochs_per_bit=20):
        """Per-channel outcome: -> (end states, trajectory or None, bits or None, epochs_done[n_ch]).  `cfgs`: one
        LoopCfg for all channels or a sequence with one per channel."""
        n_ch = len(states)
        arr = (TrackState * n_ch)(*states)
        per_channel = not isinstance(cfgs, LoopCfg)
        carr = (LoopCfg * n_ch)(*cfgs) if per_channel else (LoopCfg * 1)(cfgs)
        traj = np.zeros((n_ch, n_epochs), dtype=TRACK_EPOCH_DTYPE) if want_traj else None
        max_bits = n_epochs // max(1, int(epochs_per_bit)) + 2
        bits = np.zeros((n_ch, max_bits), dtype=np.int8) if want_bits else None
        n_bits = np.zeros(n_ch, dtype=np.int32) if want_bits else None
        done = np.zeros(n_ch, dtype=np.int32)
        check(self._lib.sdr_track_closed_loop_ex(self._h, n_ch, arr, carr, 1 if per_channel else 0, int(n_epochs),
                                                 ptr(traj) if want_traj else None, ptr(bits) if want_bits else None,
                                                 max_bits if want_bits else 0, ptr(n_bits) if want_bits else None,
                                                 ptr(done)))
        nav = [bits[c, :n_bits[c]].copy() for c in range(n_ch)] if want_bits else None
        return list(arr), traj, nav, done

    def track_closed_loop(self, states, cfg: LoopCfg, n_epochs: int, want_traj=True, want_bits=False):
        """Returns (end states, trajectory or None[, list of per-channel nav-bit arrays])."""
        n_ch = len(states)
        arr = (TrackState * n_ch)(*states)
        traj = np.zeros((n_ch, n_epochs), dtype=TRACK_EPOCH_DTYPE) if want_traj else None
        if not want_bits:
            check(self._lib.sdr_track_closed_loop(self._h, n_ch, arr, C.byref(cfg), int(n_epochs),
                                                  ptr(traj) if want_traj else None))
            return list(arr), traj
        max_bits = n_epochs // 20 + 2
        bits = np.zeros((n_ch, max_bits), dtype=np.int8)
        n_bits = np.zeros(n_ch, dtype=np.int32)
        check(self._lib.sdr_track_closed_loop_bits(self._h, n_ch, arr, C.byref(cfg), int(n_epochs),
                                                   ptr(traj) if want_traj else None, ptr(bits), max_bits, ptr(n_bits)))
        return list(arr), traj, [bits[c, :n_bits[c]].copy() for c in range(n_ch)]
