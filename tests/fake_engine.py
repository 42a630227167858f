"""A stand-in for sydr_amd.engine.Engine built on the CPU oracle -- TEST INFRASTRUCTURE ONLY.

It lets the host layer (ChannelManager, the Kaplan / Borre plugins, the ring bookkeeping) run on a
machine without a GPU so that its state machines can be checked, bit for bit, against the golden
trajectories captured from the reference.  The product never imports this."""
import numpy as np

from oracle import sydr_oracle as orc
from sydr_amd._lib import TRACK_EPOCH_DTYPE, TRACK_STATE_DTYPE
from sydr_amd.engine import FMT_CF64, FMT_CI16, FMT_CI8

_NP = {FMT_CI8: np.int8, FMT_CI16: np.int16, FMT_CF64: np.float64}


class OracleEngine:
    def __init__(self):
        self.iq_fmt, self.iq_capacity, self.ring = None, 0, None
        self.n_slots, self.codes, self.code_generation = 0, {}, 0
        self.calls = dict(pcps=0, epl_batch=0, epl_items=0)

    # ring
    def iq_alloc(self, capacity, fmt=FMT_CI8):
        self.iq_fmt, self.iq_capacity = fmt, int(capacity)
        self.ring = np.zeros(2 * capacity, dtype=_NP[fmt])

    def iq_upload(self, raw, ring_offset=0):
        raw = np.asarray(raw)
        if np.iscomplexobj(raw):
            raw = np.ascontiguousarray(raw, dtype=np.complex128).view(np.float64)
        raw = raw.astype(_NP[self.iq_fmt]).reshape(-1)
        n = raw.size // 2
        idx = (ring_offset + np.arange(n)) % self.iq_capacity
        self.ring[2 * idx] = raw[0::2]
        self.ring[2 * idx + 1] = raw[1::2]

    def iq_download(self, n, ring_offset=0):
        idx = (ring_offset + np.arange(n)) % self.iq_capacity
        out = np.empty(2 * n, dtype=self.ring.dtype)
        out[0::2], out[1::2] = self.ring[2 * idx], self.ring[2 * idx + 1]
        return out

    def _complex(self, start, n):
        raw = self.iq_download(n, start % self.iq_capacity).astype(np.float64)
        return raw[0::2] + 1j * raw[1::2]

    # codes
    def code_slots(self, n_slots, max_chips=1023):
        self.n_slots, self.codes = int(n_slots), {}
        self.code_generation += 1

    def load_gps_code(self, slot, prn):
        self.codes[slot] = orc.gold_code(int(prn))

    def set_code(self, slot, chips):
        self.codes[slot] = np.asarray(chips, dtype=np.float64)

    def read_code(self, slot, max_chips=65536):
        return self.codes[slot].astype(np.int8)

    # kernels
    def pcps(self, code_slots, start_sample, fs, if_hz, doppler_range, doppler_step, coh=1, noncoh=1, want_map=False):
        self.calls["pcps"] += 1
        n = orc.samples_per_code(fs)
        rf = self._complex(start_sample, n * coh * noncoh).reshape(1, -1)
        maps, pb, pc, pr = [], [], [], []
        for s in code_slots:
            m = orc.pcps_map(rf, if_hz, fs, orc.code_spectrum(self.codes[int(s)], fs), doppler_range, doppler_step, n,
                             coh, noncoh)
            peak, ratio = orc.two_peak_compare(m, n, round(fs / orc.CODE_RATE))
            maps.append(m)
            pb.append(peak[0])
            pc.append(peak[1])
            pr.append(ratio)
        return np.array(pb), np.array(pc), np.array(pr), (np.stack(maps) if want_map else None)

    def two_peak_compare(self, cmap, samples_per_chip):
        return orc.two_peak_compare(np.asarray(cmap), cmap.shape[1], samples_per_chip)

    def epl_batch(self, items, spacing, fs):
        self.calls["epl_batch"] += 1
        self.calls["epl_items"] += len(items)
        out = np.empty((len(items), 2 * len(spacing)))
        for k, it in enumerate(items):
            x = self._complex(int(it["start_sample"]), int(it["n_samples"]))
            out[k] = orc.epl(x, orc.pad_code(self.codes[int(it["code_slot"])]), fs, float(it["carrier_hz"]),
                             float(it["rem_carrier"]), float(it["rem_code"]), float(it["code_step"]), spacing)
        return out


class OracleBank:
    """sydr_amd.engine.Bank on the oracle's closed-loop channel models (oracle.KaplanLoop / BorreLoop): every step
    rebuilds the model from the sdr_track_state / sdr_loop_cfg rows, runs it, and writes the rows back."""

    def __init__(self, engine, max_channels):
        self.e, self.max_channels = engine, int(max_channels)
        self.states, self.cfgs = {}, {}
        self.calls = dict(step=0, tick=0, channels=0)

    def put(self, ch, state, cfg):
        self.states[int(ch)] = np.array(state, dtype=TRACK_STATE_DTYPE).reshape(1)[0].copy()
        self.cfgs[int(ch)] = np.array(cfg).reshape(1)[0].copy()

    def get(self, ch):
        return self.states[int(ch)].copy()

    def close(self):
        pass

    def _model(self, st, cfg):
        kaplan = int(cfg["loop_kind"]) == 1
        m = object.__new__(orc.KaplanLoop if kaplan else orc.BorreLoop)
        m.fs = float(cfg["fs"])
        m.code = orc.pad_code(self.e.codes[int(st["code_slot"])])
        m.dll_tau1, m.dll_tau2, m.dll_pdi = float(cfg["dll_tau1"]), float(cfg["dll_tau2"]), float(cfg["dll_pdi"])
        m.carrier_hz, m.code_hz = float(st["carrier_hz"]), float(st["code_hz"])
        m.rem_carrier, m.rem_code, m.code_step = float(st["rem_carrier"]), float(st["rem_code"]), float(st["code_step"])
        m.n, m.current_sample = int(st["n_samples"]), int(st["current_sample"])
        m.flags, m.code_counter = int(st["track_flags"]), int(st["code_counter"])
        m.nav_sum, m.nav_count, m.nav_bits = float(st["nav_prompt_sum"]), int(st["nav_sum_counter"]), []
        nt = int(cfg["n_taps"])
        m.prompt = nt // 2
        m.epoch_chips = float(cfg["epoch_chips"]) if float(cfg["epoch_chips"]) > 0 else orc.CODE_CHIPS   # 0: the reference's
        m.epochs_per_bit = int(cfg["epochs_per_bit"]) if int(cfg["epochs_per_bit"]) > 0 else orc.MS_PER_BIT
        if kaplan:
            m.dt = float(cfg["epoch_seconds"]) if float(cfg["epoch_seconds"]) > 0 else 1e-3
            m.sp_wide, m.sp_narrow = list(cfg["spacing_wide"][:nt]), list(cfg["spacing_narrow"][:nt])
            m.spacing = m.sp_narrow if int(st["spacing_sel"]) else m.sp_wide
            m.cfg = dict(fll_threshold_narrow=cfg["fll_thr_narrow"], pll_threshold_narrow=cfg["pll_thr_narrow"],
                         fll_threshold_wide=cfg["fll_thr_wide"], pll_threshold_wide=cfg["pll_thr_wide"],
                         fll_bandwidth_narrow=cfg["fll_bw_narrow"], pll_bandwidth_narrow=cfg["pll_bw_narrow"],
                         fll_bandwidth_wide=cfg["fll_bw_wide"], pll_bandwidth_wide=cfg["pll_bw_wide"],
                         fll_bandwidth_pullin=cfg["fll_bw_pullin"])
            m.cfg = {k: float(v) for k, v in m.cfg.items()}
            m.dll_thr = float(cfg["dll_threshold"])
            m.corr, m.accum_counter = [0.0] * (2 * nt), int(st["accum_counter"])
            m.ip_prev, m.qp_prev = float(st["i_prompt_prev"]), float(st["q_prompt_prev"])
            m.cn0_ratio, m.cn0 = float(st["cn0_ratio_acc"]), float(st["cn0"])
            m.dll, m.pll, m.fll = float(st["dll_mem"]), 0.0, 0.0
            m.fll_bw, m.pll_bw = float(st["fll_bw"]), float(st["pll_bw"])
            m.dll_lock, m.fll_lock, m.pll_lock = float(st["cn0"]), float(st["fll_lock"]), float(st["pll_lock"])
            m.vel_mem, m.time_in_state, m.lock_state = float(st["pll_mem"]), int(st["time_in_state"]), int(st["lock_state"])
        else:
            m.spacing = list(cfg["spacing_wide"][:nt])
            m.pll_tau1, m.pll_tau2, m.pll_pdi = float(cfg["pll_tau1"]), float(cfg["pll_tau2"]), float(cfg["pll_pdi"])
            m.code_err_mem, m.carrier_err_mem = float(st["dll_mem"]), float(st["pll_mem"])
            m.ip_prev = float(st["i_prompt_prev"])
        return m, kaplan

    @staticmethod
    def _store(m, kaplan, st, qp):
        st["carrier_hz"], st["code_hz"] = m.carrier_hz, m.code_hz
        st["rem_carrier"], st["rem_code"], st["code_step"] = m.rem_carrier, m.rem_code, m.code_step
        st["n_samples"], st["current_sample"] = m.n, m.current_sample
        st["track_flags"], st["code_counter"] = m.flags, m.code_counter
        st["nav_prompt_sum"], st["nav_sum_counter"] = m.nav_sum, m.nav_count
        st["nav_bits_emitted"] += len(m.nav_bits)
        if kaplan:
            st["spacing_sel"] = 1 if m.spacing is m.sp_narrow else 0
            st["accum_counter"], st["i_prompt_prev"], st["q_prompt_prev"] = m.accum_counter, m.ip_prev, m.qp_prev
            st["cn0_ratio_acc"], st["cn0"], st["dll_mem"] = m.cn0_ratio, m.cn0, m.dll
            st["fll_bw"], st["pll_bw"], st["fll_lock"], st["pll_lock"] = m.fll_bw, m.pll_bw, m.fll_lock, m.pll_lock
            st["pll_mem"], st["time_in_state"], st["lock_state"] = m.vel_mem, m.time_in_state, m.lock_state
        else:
            st["dll_mem"], st["pll_mem"], st["i_prompt_prev"], st["q_prompt_prev"] = m.code_err_mem, m.carrier_err_mem, m.ip_prev, qp

    def step(self, channels, n_epochs=1, want_records=True, want_bits=False, stream=0, epochs_per_bit=20):
        self.calls["step"] += 1
        self.calls["channels"] += len(channels)
        n = len(channels)
        rec = np.zeros((n, n_epochs), dtype=TRACK_EPOCH_DTYPE)
        rec["nav_bit"] = -1
        done = np.zeros(n, dtype=np.int32)
        bits = []
        for r, ch in enumerate(int(c) for c in channels):
            st, cfg = self.states[ch], self.cfgs[ch]
            m, kaplan = self._model(st, cfg)
            qp = float(st["q_prompt_prev"])
            for k in range(n_epochs):
                if not (0 < m.n <= self.e.iq_capacity and m.code_step > 0 and abs(m.carrier_hz) < 1e9):
                    break
                out = m.step(self.e._complex(m.current_sample, m.n))
                e = rec[r, k]
                e["start_sample"], e["n_samples"] = out["start"], out["n"]
                e["carrier_hz_in"], e["rem_carrier_in"] = out["carrier_hz_in"], out["rem_carrier_in"]
                e["rem_code_in"], e["code_step_in"] = out["rem_code_in"], out["code_step_in"]
                e["corr"][:6] = out["corr"]
                e["dll"], e["pll"], e["fll"] = out["dll"], out["pll"], out.get("fll", 0.0)
                e["carrier_err"], e["code_err"] = out["carrier_err"], out["code_err"]
                e["carrier_hz"], e["code_hz"] = out["carrier_hz"], out["code_hz"]
                e["cn0"], e["pll_lock"], e["fll_lock"] = out.get("cn0", 0.0), out.get("pll_lock", 0.0), out.get("fll_lock", 0.0)
                e["lock_state"], e["track_flags"], e["nav_bit"] = out.get("lock_state", 0), out["flags"], out["nav_bit"]
                qp = out["corr"][3]
                done[r] = k + 1
            self._store(m, kaplan, st, qp)
            bits.append(np.array(m.nav_bits, dtype=np.int8))
        states = np.array([self.states[int(c)] for c in channels], dtype=TRACK_STATE_DTYPE)
        return (rec if want_records else None), states, done, (bits if want_bits else None)

    def tick(self, raw, ring_offset, channels):
        self.calls["tick"] += 1
        if raw is not None:
            self.e.iq_upload(raw, ring_offset)
        if not len(channels):
            return np.zeros(0, dtype=TRACK_EPOCH_DTYPE), np.zeros(0, dtype=TRACK_STATE_DTYPE), np.zeros(0, dtype=np.int32)
        rec, states, done, _ = self.step(channels, 1)
        return rec[:, 0], states, done


def _bind_mirror(self, states, last, epochs_since_tow, tracking, lost, host_flags):
    from types import SimpleNamespace
    from sydr_amd._lib import TICK_UPDATE_DTYPE
    n = self.max_channels
    self.ran = np.zeros(n, dtype=np.int32)
    self.records = np.zeros(n, dtype=TRACK_EPOCH_DTYPE)
    self.updates = np.zeros(n, dtype=TICK_UPDATE_DTYPE)
    self._mirror_arrays = (states, last, epochs_since_tow, tracking, lost, host_flags)
    self._mirror = SimpleNamespace(n_ran=0, n_updates=0, n_nav_bits=0, n_lost=0, max_unread=0)
    return self._mirror


def _tick_mirrored(self, raw, ring_offset, write_index):
    """sdr_bank_tick_mirrored restated (sydr_amd/csrc/track.hip): readiness from the mirror, one epoch for the ready
    channels, the mirror and the per-tick rows updated in place."""
    states, last, since, tracking, lost, host_flags = self._mirror_arrays
    m, cap = self._mirror, self.e.iq_capacity
    if raw is not None:
        self.e.iq_upload(raw, ring_offset)

    def unread_of(ch):
        cur = int(states["current_sample"][ch]) % cap
        return write_index - cur if cur <= write_index else cap - cur + write_index
    ready = [ch for ch in range(self.max_channels) if tracking[ch] and not lost[ch] and ch in self.states
             and unread_of(ch) >= int(states["n_samples"][ch])]
    m.n_ran = m.n_nav_bits = m.n_lost = 0
    if ready:
        self.calls["tick"] += 1
        rec, st, done, _ = self.step(ready, 1)
        w = 0
        for i, ch in enumerate(ready):
            states[ch] = st[i]
            if done[i] < 1:
                lost[ch] = True
                m.n_lost += 1
                continue
            self.records[w] = rec[i, 0]
            last[ch] = rec[i, 0]
            since[ch] += 1
            m.n_nav_bits += int(rec[i, 0]["nav_bit"] >= 0)
            self.ran[w] = ch
            w += 1
        m.n_ran = w
    nu, m.max_unread = 0, 0
    for ch in range(self.max_channels):
        if not tracking[ch]:
            continue
        u = self.updates[nu]
        u["channel"], u["track_flags"] = ch, int(states["track_flags"][ch]) | int(host_flags[ch])
        u["unread"], u["epochs_since_tow"] = unread_of(ch), since[ch]
        if not lost[ch]:
            m.max_unread = max(m.max_unread, int(u["unread"]))
        nu += 1
    m.n_updates = nu
    return m


def _step_begin(self, channels, n_epochs):
    rec, states, done, _ = self.step(channels, n_epochs)      # (nothing runs beside the host here: the work is done at once)
    self._in_flight = (rec, states, done)


def _step_end(self):
    out, self._in_flight = self._in_flight, None
    return out


def _tick_mirrored_begin(self, raw, ring_offset, write_index):
    self._tick_args = (raw, ring_offset, write_index)         # (nothing runs beside the host here)


def _tick_mirrored_end(self):
    args, self._tick_args = self._tick_args, None
    return _tick_mirrored(self, *args)


OracleBank.tick_mirrored_begin = _tick_mirrored_begin
OracleBank.tick_mirrored_end = _tick_mirrored_end
OracleBank.step_begin = _step_begin
OracleBank.step_end = _step_end
OracleBank.bind_mirror = _bind_mirror
OracleBank.tick_mirrored = _tick_mirrored


def _make_bank(self, max_channels):
    bank = OracleBank(self, max_channels)
    self.bank_calls = bank.calls          # (what the host layer asked of the device: ticks / block steps)
    return bank


OracleEngine.bank = _make_bank
OracleEngine._ring_samples = lambda self, raw: np.asarray(raw)
OracleEngine.iq_upload_begin = OracleEngine.iq_upload         # (nothing to wait for on the host)
OracleEngine.sync = lambda self: None


def _serial_search(self, code_slots, start_sample, fs, doppler_range, doppler_step, noncoh=1, want_map=False,
                   n_chips=1023):
    n = orc.samples_per_code(fs)
    maps, pb, pc, pr = [], [], [], []
    for s in code_slots:
        m = None
        for k in range(noncoh):
            rf = self._complex(start_sample + k * n, n).reshape(1, -1)
            part = orc.serial_search(rf, self.codes[int(s)], doppler_range, doppler_step, fs, n)
            m = part if m is None else m + part
        peak, ratio = orc.two_peak_compare_ss(m)
        maps.append(m)
        pb.append(peak[0])
        pc.append(peak[1])
        pr.append(ratio)
    return np.array(pb), np.array(pc), np.array(pr), (np.stack(maps) if want_map else None)


OracleEngine.serial_search = _serial_search
OracleEngine.two_peak_compare_ss = lambda self, cmap: orc.two_peak_compare_ss(np.asarray(cmap))
