"""A stand-in for sydr_amd.engine.Engine built on the CPU oracle -- TEST INFRASTRUCTURE ONLY.

It lets the host layer (ChannelManager, the Kaplan / Borre plugins, the ring bookkeeping) run on a
machine without a GPU so that its state machines can be checked, bit for bit, against the golden
trajectories captured from the reference.  The product never imports this."""
import numpy as np

from oracle import sydr_oracle as orc
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
