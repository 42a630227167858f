"""CPU oracle for the GNSS correlator hot path -- TEST INFRASTRUCTURE ONLY.

This module is a clean-room NumPy restatement of the arithmetic of the reference
receiver aproposorg/sydr for the one path this repository accelerates.  It exists
to CHECK the HIP kernels; nothing under ``sydr_amd/`` may import it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

Parity status: PINNED.  Every function below is checked, bit for bit where the
result is an integer and to <= 1e-12 relative otherwise, against golden vectors
captured by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).

Third-party arithmetic under the reference path (not vendored in the reference):
numpy==1.24.2 (``requirements.txt:23``) -- ``np.fft`` (pocketfft, mixed radix),
``np.exp`` on complex128, ``np.linspace``/``np.ceil``, pairwise ``np.sum``.  The
restatement calls the same NumPy primitives in the same order, so it inherits
their semantics; golden vectors record the NumPy version that produced them.

Reference locations (relative to the reference checkout):
  gold_code            sydr/signal/ca.py:70-112, sydr/signal/gnsssignal.py:9-31
  upsample_code        sydr/signal/gnsssignal.py:35-58
  samples_per_code     sydr/signal/gnsssignal.py:62-70
  pad_code             sydr/channel/channel_l1ca_kaplan.py:104-107
  pcps_map             sydr/dsp/acquisition.py:9-74
  two_peak_compare     sydr/dsp/acquisition.py:78-115
  epl                  sydr/dsp/tracking.py:92-116
  loop math            sydr/dsp/tracking.py:39-61,120-186,246-279
  lock indicators      sydr/dsp/lockindicator.py:6-122
  BorreLoop            sydr/channel/channel_l1ca_borre.py:206-451
  KaplanLoop           sydr/channel/channel_l1ca_kaplan.py:138-619
"""
from __future__ import annotations

import math

import numpy as np

# --- constants (sydr/utils/constants.py:4-6,71-85) ---------------------------------------------
GPS_PI = 3.1415926535898          # GPS-ICD pi, used by the Kaplan NCO and the discriminators
GPS_TWO_PI = GPS_PI * 2.0
GPS_HALF_PI = GPS_PI / 2.0
CODE_CHIPS = 1023
CODE_RATE = 1.023e6
MS_PER_BIT = 20
W0_BW_1 = 0.25
W0_BW_2 = 0.53
W0_A2 = 1.414

# TrackingFlags / LoopLockState values (sydr/utils/enumerations.py:121-147)
FLAG_CODE_LOCK = 1
FLAG_BIT_SYNC = 2
LOCK_PULL_IN = 1
LOCK_WIDE = 2
LOCK_NARROW = 3

# G2 delays, PRN 1..37 shown in IS-GPS-200 Table 3-Ia; 38..210 from Table 3-Ib (= ca.py:13-68)
_G2_DELAY = [
    0, 5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
    473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862, 863, 950, 947, 948, 950, 67, 103,
    91, 19, 679, 225, 625, 946, 638, 161, 1001, 554, 280, 710, 709, 775, 864, 558, 220, 397, 55,
    898, 759, 367, 299, 1018, 729, 695, 780, 801, 788, 732, 34, 320, 327, 389, 407, 525, 405, 221,
    761, 260, 326, 955, 653, 699, 422, 188, 438, 959, 539, 879, 677, 586, 153, 792, 814, 446, 264,
    1015, 278, 536, 819, 156, 957, 159, 712, 885, 461, 248, 713, 126, 807, 279, 122, 197, 693, 632,
    771, 467, 647, 203, 145, 175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130,
    359, 595, 68, 386, 797, 456, 499, 883, 307, 127, 211, 121, 118, 163, 628, 853, 484, 289, 811,
    202, 1021, 463, 568, 904, 670, 230, 911, 684, 309, 644, 932, 12, 314, 891, 212, 185, 675, 503,
    150, 395, 345, 846, 798, 992, 357, 995, 877, 112, 144, 476, 193, 109, 445, 291, 87, 399, 292,
    901, 339, 208, 711, 189, 263, 537, 663, 942, 173, 900, 30, 500, 935, 556, 373, 85, 652, 310,
]


def _m_sequence(taps):
    """10-stage LFSR, all-ones start, output = stage 10, feedback = xor of `taps` (1-based)."""
    reg = [1] * 10
    out = np.empty(CODE_CHIPS, dtype=np.int8)
    for i in range(CODE_CHIPS):
        out[i] = reg[9]
        fb = 0
        for t in taps:
            fb ^= reg[t - 1]
        reg = [fb] + reg[:9]
    return out


_G1 = _m_sequence((10, 3))
_G2 = _m_sequence((10, 9, 8, 6, 3, 2))


def gold_code(prn: int) -> np.ndarray:
    """C/A code of `prn` as float64 +-1 (bit 1 -> +1.0, bit 0 -> -1.0: ca.py:112)."""
    if not 1 <= prn <= 210:
        raise KeyError(prn)
    bits = _G1 ^ np.roll(_G2, _G2_DELAY[prn])
    return 2.0 * bits.astype(np.float64) - 1.0


def first_10_chips_octal(prn: int) -> int:
    """IS-GPS-200 'first 10 chips' check value (ca.py:131-136)."""
    bits = ((gold_code(prn)[:10] + 1.0) / 2.0).astype(int)
    value = 0
    for b in bits:
        value = 2 * value + int(b)
    return value


def samples_per_code(fs: float) -> int:
    return round(fs / (CODE_RATE / CODE_CHIPS))


def upsample_index(fs: float, n: int | None = None) -> np.ndarray:
    ts = 1 / fs
    tc = 1 / CODE_RATE
    if n is None:
        n = samples_per_code(fs)
    return np.trunc(ts * np.array(range(n)) / tc).astype(int)


def upsample_code(code: np.ndarray, fs: float) -> np.ndarray:
    return code[upsample_index(fs)]


def pad_code(code: np.ndarray) -> np.ndarray:
    return np.r_[code[-1], code, code[0]]


def code_spectrum(code: np.ndarray, fs: float) -> np.ndarray:
    """conj(fft(upsampled code)) as built at channel_l1ca_kaplan.py:184-185."""
    return np.conj(np.fft.fft(upsample_code(code, fs)))


def doppler_bins(doppler_range: float, doppler_step: float) -> np.ndarray:
    return np.arange(-doppler_range, doppler_range + 1, doppler_step)


def pcps_map(rf, if_hz, fs, code_fft, doppler_range, doppler_step, n_code, coh=1, noncoh=1):
    """Correlation-magnitude map [bins][n_code] (acquisition.py:9-74)."""
    rf = np.squeeze(rf)
    phase_points = np.array(range(coh * n_code)) * 2 * np.pi / fs
    bins = doppler_bins(doppler_range, doppler_step)
    out = np.zeros((len(bins), n_code))
    noncoh_sum = np.zeros((1, n_code))
    for row, b in enumerate(bins):
        freq = if_hz - b
        carrier = np.exp(-1j * freq * phase_points)
        noncoh_sum = noncoh_sum * 0.0
        for k in range(noncoh):
            block = rf[k * coh * n_code:(k + 1) * coh * n_code]
            mixed = np.multiply(carrier, block)
            coh_sum = noncoh_sum * 0.0
            for c in range(coh):
                spec = np.fft.fft(mixed[c * n_code:(c + 1) * n_code])
                coh_sum = coh_sum + np.fft.ifft(np.multiply(spec, code_fft))
            noncoh_sum = noncoh_sum + abs(coh_sum)
        out[row, :] = abs(noncoh_sum)
    return np.squeeze(np.squeeze(out))


def two_peak_compare(cmap: np.ndarray, n_code: int, samples_per_chip: int):
    """([bin, code], ratio) with the reference's exclusion-window quirks (acquisition.py:78-115)."""
    top = np.unravel_index(cmap.argmax(), cmap.shape)
    top = [int(top[0]), int(top[1])]
    p1 = cmap[top[0], top[1]]
    lo = int(top[1] - samples_per_chip)
    hi = int(top[1] + samples_per_chip)
    if lo < 1:
        cols = list(range(hi, n_code - 1))
    elif hi >= n_code:
        cols = list(range(0, lo))
    else:
        cols = list(range(0, lo)) + list(range(hi, n_code - 1))
    p2 = np.amax(cmap[top[0], cols])
    return top, p1 / p2


def _shift(arr, num):
    """Circular right shift used by SerialSearch (acquisition.py:196-206)."""
    out = np.empty_like(arr)
    if num > 0:
        out[:num] = arr[-num:]
        out[num:] = arr[:-num]
    else:
        out[:] = arr
    return out


def serial_search(rf, code, doppler_range, doppler_step, fs, n_code):
    """Brute-force acquisition map [bins][len(code)] (acquisition.py:119-155)."""
    bins = doppler_bins(doppler_range, doppler_step)
    phase_points = np.array(range(n_code)) * 2 * np.pi / fs
    out = np.zeros((len(bins), len(code)))
    up = upsample_index(fs, n_code)
    for row, freq in enumerate(bins):
        carrier = np.exp(-1j * -freq * phase_points)
        signal = np.multiply(rf, carrier)
        for k in range(len(code)):
            shifted = _shift(code, k)[up]
            i_sig = np.multiply(np.real(signal), shifted)
            q_sig = np.multiply(np.imag(signal), shifted)
            out[row, k] += np.sum(i_sig) ** 2 + np.sum(q_sig) ** 2
    return np.squeeze(np.squeeze(out))


def two_peak_compare_ss(cmap):
    """([bin, chip], ratio) of acquisition.py:159-193: second peak outside the 3x3 block around the first
    (Python slices: a block starting at index -1 selects nothing)."""
    top = np.unravel_index(cmap.argmax(), cmap.shape)
    top = [int(top[0]), int(top[1])]
    p1 = cmap[top[0], top[1]]
    work = np.copy(cmap)
    work[top[0] - 1:top[0] + 2, top[1] - 1:top[1] + 2] = 0.0
    return top, p1 / np.amax(work)


def epl_indices(n, rem_code, code_step, spacing):
    """Padded-code index per sample for one tap (tracking.py:111-112)."""
    shift = rem_code + spacing
    return np.ceil(np.linspace(shift, code_step * n + shift, n, endpoint=False)).astype(int)


def epl(rf, code_padded, fs, carrier_hz, rem_carrier, rem_code, code_step, spacings):
    """[I_tap0, Q_tap0, I_tap1, ...] (tracking.py:92-116), any number of taps.

    `code_padded` is [c[L-1], c[0..L-1], c[0]]; indices outside it (taps beyond +-1 chip,
    multi-period epochs) wrap periodically: padded index p is chip (p-1) mod L.
    """
    rf = np.squeeze(rf)
    n = len(rf)
    t = np.arange(0.0, n) / fs
    replica = np.exp(1j * (-(carrier_hz * 2.0 * np.pi * t) + rem_carrier))
    mixed = replica * rf
    i_sig = np.real(mixed)
    q_sig = np.imag(mixed)
    n_chips = len(code_padded) - 2
    out = []
    for sp in spacings:
        idx = epl_indices(n, rem_code, code_step, sp)
        if idx.min() < 0 or idx.max() > n_chips + 1:
            chips = code_padded[1:-1][(idx - 1) % n_chips]
        else:
            chips = code_padded[idx]
        out.append(np.sum(chips * i_sig))
        out.append(np.sum(chips * q_sig))
    return out


# --- scalar loop math (tracking.py) -------------------------------------------------------------

def loop_coefficients(noise_bw, damping, gain):
    wn = noise_bw * 8.0 * damping / (4.0 * damping**2 + 1)
    return gain / wn**2, 2.0 * damping / wn


def dll_nneml(ie, qe, il, ql):
    return (np.sqrt(ie**2 + qe**2) - np.sqrt(il**2 + ql**2)) / \
           (np.sqrt(ie**2 + qe**2) + np.sqrt(il**2 + ql**2))


def pll_costas(ip, qp):
    err = np.arctan(qp / ip)
    err /= GPS_TWO_PI
    return err


def _unwrap(phase):
    if phase >= GPS_HALF_PI:
        return phase - GPS_PI
    if phase <= -GPS_HALF_PI:
        return phase + GPS_PI
    return phase


def fll_atan(ip, qp, ip_prev, qp_prev, dt):
    err = np.arctan(qp / ip) - np.arctan(qp_prev / ip_prev)
    if np.isnan(err):
        err = 0.0
    err = _unwrap(err) / dt
    err /= GPS_TWO_PI
    return err


def borre_filter(x, memory, tau1, tau2, pdi):
    out = tau2 / tau1 * (x - memory)
    out += pdi / tau1 * x
    return out


def fll_assisted_pll_2nd(phase_in, freq_in, w0f, w0p, a2, dt, vel_mem):
    upd = (phase_in * w0p**2 + freq_in * w0f) * dt
    out = upd + vel_mem
    vel_mem = upd
    out += phase_in * a2 * w0p
    return out, vel_mem


def fll_lock_borre(ip, ip_prev, qp, qp_prev, prev, alpha=0.01):
    v = ip * ip_prev - qp * qp_prev
    v *= np.sign(ip * ip_prev + qp * qp_prev)
    v /= (ip**2 + qp**2)
    v = abs(v)
    return (1 - alpha) * prev + alpha * v


def pll_lock_borre(ip, qp, prev, alpha=0.01):
    nbd = ip**2 - qp**2
    nbp = ip**2 + qp**2
    return (1 - alpha) * prev + alpha * (nbd / nbp)


def cn0_beaulieu(ratio, count, dt, old):
    lam = 1 / (ratio / count)
    cn0 = lam * (1 / dt)
    return (1 - 0.1) * old + 0.1 * cn0


# --- ring buffer slice (circularbuffer.py:114-137) ------------------------------------------------

def ring_slice(ring: np.ndarray, start: int, n: int) -> np.ndarray:
    size = len(ring)
    stop = (start + n) % size
    if stop < start:
        return np.concatenate((ring[start:], ring[:stop]))
    return ring[start:stop]


# --- closed-loop channel models ---------------------------------------------------------------------
# These follow the per-epoch bookkeeping of the two reference plugins on a LINEAR sample array
# (absolute sample indices; the reference's ring modulo at kaplan:531 / borre:428 is applied by the
# caller when a ring is used).  Navigation-bit decoding is not part of the path and is left out.
#
# Generalisation (BASELINE configs 4-5; the reference has no counterpart, SURVEY.md 8c -- PARITY UNPINNED BY THE
# REFERENCE for anything but the defaults): `taps` = (wide, narrow) lists of any odd length whose centre tap is
# the prompt and whose neighbours feed the discriminators (the outer taps are only correlated and recorded),
# `epoch_chips` = chips per correlator epoch (code length x periods, in the units of `code`: half chips for a
# BOC(1,1) code passed as its doubled half-chip sequence), `epochs_per_bit` = epochs per navigation symbol,
# `dt` = the epoch duration the discriminators and filters are scaled with, `code_rate` = nominal chip rate of
# `code`.  With the defaults every statement below is the reference's, operation for operation (pinned by
# g6 / g6b / g6c bit for bit); the generalised arguments only replace the constants 1023, 20, 1e-3 and 1.023e6.

class BorreLoop:
    """runTracking of channel_l1ca_borre.py:333-451 (DLL NNEML + Costas PLL, Borre filters)."""

    def __init__(self, fs, code, cfg, carrier_hz, current_sample, *, taps=None, epoch_chips=CODE_CHIPS,
                 epochs_per_bit=MS_PER_BIT, code_rate=CODE_RATE):
        self.fs = fs
        self.code = pad_code(code)
        self.spacing = [cfg["correlator_early"], cfg["correlator_prompt"], cfg["correlator_late"]] if taps is None \
            else list(taps[0])
        self.prompt = len(self.spacing) // 2
        self.epoch_chips, self.epochs_per_bit = epoch_chips, epochs_per_bit
        self.dll_tau1, self.dll_tau2 = loop_coefficients(cfg["dll_noise_bandwidth"], cfg["dll_damping_ratio"],
                                                         cfg["dll_loop_gain"])
        self.pll_tau1, self.pll_tau2 = loop_coefficients(cfg["pll_noise_bandwidth"], cfg["pll_damping_ratio"],
                                                         cfg["pll_loop_gain"])
        self.dll_pdi = cfg["dll_pdi"]
        self.pll_pdi = cfg["pll_pdi"]
        self.carrier_hz = carrier_hz
        self.code_hz = code_rate
        self.rem_carrier = 0.0
        self.rem_code = 0.0
        self.code_step = code_rate / fs
        self.n = int(np.ceil((self.epoch_chips - self.rem_code) / self.code_step))
        self.current_sample = current_sample
        self.code_err_mem = 0.0
        self.carrier_err_mem = 0.0
        self.flags = 0
        self.code_counter = 0
        self.ip_prev = 0.0
        self.nav_sum, self.nav_count, self.nav_bits = 0.0, 0, []

    def step(self, samples):
        """One epoch on `samples` (the n samples starting at current_sample).  Returns a record."""
        rec = dict(start=self.current_sample, n=self.n, carrier_hz_in=self.carrier_hz,
                   rem_carrier_in=self.rem_carrier, rem_code_in=self.rem_code, code_step_in=self.code_step)
        corr = epl(samples, self.code, self.fs, self.carrier_hz, self.rem_carrier, self.rem_code,
                   self.code_step, self.spacing)
        self.rem_carrier -= self.carrier_hz * 2.0 * np.pi * self.n / self.fs
        self.rem_carrier %= (2 * np.pi)
        p = self.prompt
        ie, qe, ip, qp, il, ql = corr[2 * p - 2:2 * p + 4]
        code_err = dll_nneml(ie, qe, il, ql)
        nco_code = borre_filter(code_err, self.code_err_mem, self.dll_tau1, self.dll_tau2, self.dll_pdi)
        self.code_err_mem = code_err
        phase_err = pll_costas(ip, qp)
        nco_carrier = borre_filter(phase_err, self.carrier_err_mem, self.pll_tau1, self.pll_tau2, self.pll_pdi)
        self.carrier_err_mem = phase_err
        # bit sync: first prompt sign flip after 100 epochs (channel_l1ca_borre.py:384-391,401)
        if not (self.flags & FLAG_BIT_SYNC) and (self.flags & FLAG_CODE_LOCK) and self.code_counter > 100 \
                and np.sign(self.ip_prev) != np.sign(ip):
            self.flags |= FLAG_BIT_SYNC
        self.flags |= FLAG_CODE_LOCK
        self.ip_prev = ip
        self.code_counter += 1
        self.code_hz -= nco_code
        self.carrier_hz += nco_carrier
        self.rem_code += self.n * self.code_step - self.epoch_chips
        self.code_step = self.code_hz / self.fs
        self.current_sample += self.n
        self.n = int(np.ceil((self.epoch_chips - self.rem_code) / self.code_step))
        rec.update(corr=list(corr), dll=nco_code, pll=nco_carrier, carrier_hz=self.carrier_hz,
                   code_hz=self.code_hz, code_err=code_err, carrier_err=phase_err, flags=self.flags,
                   nav_bit=_decode_bit(self, ip))
        return rec


class KaplanLoop:
    """runTracking of channel_l1ca_kaplan.py:342-619 (FLL-assisted PLL, lock-state machine)."""

    def __init__(self, fs, code, cfg, carrier_hz, current_sample, *, taps=None, epoch_chips=CODE_CHIPS,
                 epochs_per_bit=MS_PER_BIT, dt=1e-3, code_rate=CODE_RATE):
        self.fs = fs
        self.code = pad_code(code)
        wide, narrow = cfg["correlator_epl_wide"], cfg["correlator_epl_narrow"]
        self.sp_wide = [-wide, 0.0, wide] if taps is None else list(taps[0])
        self.sp_narrow = [-narrow, 0.0, narrow] if taps is None else list(taps[1])
        self.prompt = len(self.sp_wide) // 2
        self.epoch_chips, self.epochs_per_bit, self.dt = epoch_chips, epochs_per_bit, dt
        self.spacing = self.sp_wide
        self.dll_tau1, self.dll_tau2 = loop_coefficients(cfg["dll_noise_bandwidth"], cfg["dll_damping_ratio"],
                                                         cfg["dll_loop_gain"])
        self.dll_pdi = cfg["dll_pdi"]
        self.cfg = cfg
        self.dll_thr = cfg["dll_threshold"]
        self.corr = [0.0] * 6
        self.accum_counter = 0
        self.ip_prev = 0.0
        self.qp_prev = 0.0
        self.cn0_ratio = 0.0
        self.cn0 = 0.0
        self.dll = self.pll = self.fll = 0.0
        self.fll_bw = cfg["fll_bandwidth_pullin"]
        self.pll_bw = cfg["pll_bandwidth_wide"]
        self.dll_lock = 0.0
        self.fll_lock = 0.0
        self.pll_lock = 0.0
        self.vel_mem = 0.0
        self.time_in_state = 0
        self.lock_state = LOCK_PULL_IN
        self.flags = 0
        self.rem_code = 0.0
        self.rem_carrier = 0.0
        self.code_step = code_rate / fs
        self.n = int(np.ceil((self.epoch_chips - self.rem_code) / self.code_step))
        self.code_counter = 0
        self.carrier_hz = carrier_hz
        self.code_hz = code_rate
        self.current_sample = current_sample
        self.nav_sum, self.nav_count, self.nav_bits = 0.0, 0, []

    def step(self, samples):
        c = self.cfg
        rec = dict(start=self.current_sample, n=self.n, carrier_hz_in=self.carrier_hz,
                   rem_carrier_in=self.rem_carrier, rem_code_in=self.rem_code, code_step_in=self.code_step,
                   spacing=list(self.spacing))
        # runCorrelators (:378-401)
        self.corr = epl(samples, self.code, self.fs, self.carrier_hz, self.rem_carrier, self.rem_code,
                        self.code_step, self.spacing)
        if self.accum_counter == self.epochs_per_bit:
            self.accum_counter = 0
        self.accum_counter += 1
        p, dt = self.prompt, self.dt
        ie, qe, ip, qp, il, ql = self.corr[2 * p - 2:2 * p + 4]
        # runDiscriminators (:405-430)
        fll_d = pll_d = 0.0
        if self.lock_state == LOCK_PULL_IN:
            if self.code_counter > 1:
                fll_d = fll_atan(ip, qp, self.ip_prev, self.qp_prev, dt)
            dll_d = dll_nneml(ie, qe, il, ql)
        else:
            fll_d = fll_atan(ip, qp, self.ip_prev, self.qp_prev, dt)
            pll_d = pll_costas(ip, qp)
            dll_d = dll_nneml(ie, qe, il, ql)
        # loop filters (:434-461)
        carrier_err, self.vel_mem = fll_assisted_pll_2nd(pll_d, fll_d, self.fll_bw / W0_BW_1, self.pll_bw / W0_BW_2,
                                                         W0_A2, 1 * dt, self.vel_mem)
        code_err = borre_filter(dll_d, self.dll, self.dll_tau1, self.dll_tau2, self.dll_pdi * 1)
        # runLoopIndicators (:465-502)
        if self.code_counter != 0:
            self.fll_lock = fll_lock_borre(ip, self.ip_prev, qp, self.qp_prev, self.fll_lock, alpha=0.005)
            if self.lock_state > LOCK_PULL_IN:
                self.pll_lock = pll_lock_borre(ip, qp, self.pll_lock, alpha=0.005)
            self.cn0_ratio += (ip**2 + qp**2) / (abs(ip) - abs(qp)) ** 2
            if self.accum_counter == self.epochs_per_bit:
                self.cn0 = cn0_beaulieu(self.cn0_ratio, self.accum_counter, self.accum_counter * dt, self.cn0)
                self.cn0_ratio = 0.0
            self.dll_lock = self.cn0
        # postTrackingUpdate (:506-534)
        self.code_counter += 1
        self.dll, self.fll, self.pll = dll_d, fll_d, pll_d
        self.rem_carrier -= self.carrier_hz * GPS_TWO_PI * self.n / self.fs
        self.rem_carrier %= GPS_TWO_PI
        self.code_hz -= code_err
        self.carrier_hz += carrier_err
        self.rem_code += self.n * self.code_step - self.epoch_chips
        self.code_step = self.code_hz / self.fs
        self.current_sample += self.n
        self.n = int(np.ceil((self.epoch_chips - self.rem_code) / self.code_step))
        # trackingStateUpdate (:538-619)
        if self.lock_state != LOCK_PULL_IN and self.dll_lock > self.dll_thr and not (self.flags & FLAG_CODE_LOCK):
            self.flags |= FLAG_CODE_LOCK
        elif self.dll_lock < self.dll_thr and (self.flags & FLAG_CODE_LOCK):
            self.flags ^= FLAG_CODE_LOCK
        if (self.flags & FLAG_CODE_LOCK) and not (self.flags & FLAG_BIT_SYNC):
            if np.sign(self.ip_prev) != np.sign(ip):
                self.flags |= FLAG_BIT_SYNC
                self.accum_counter = 1
                self.cn0_ratio = 0.0
        self.ip_prev, self.qp_prev = ip, qp
        if self.lock_state != LOCK_NARROW and self.fll_lock >= c["fll_threshold_narrow"] \
                and self.pll_lock >= c["pll_threshold_narrow"]:
            self.lock_state = LOCK_NARROW
            self.fll_bw, self.pll_bw = c["fll_bandwidth_narrow"], c["pll_bandwidth_narrow"]
            self.spacing = self.sp_narrow
            self.time_in_state = 0
        elif self.lock_state != LOCK_WIDE and c["fll_threshold_wide"] <= self.fll_lock < c["fll_threshold_narrow"]:
            self.lock_state = LOCK_WIDE
            self.fll_bw, self.pll_bw = c["fll_bandwidth_wide"], c["pll_bandwidth_wide"]
            self.spacing = self.sp_wide
            self.time_in_state = 0
        elif self.lock_state != LOCK_PULL_IN and self.fll_lock <= c["fll_threshold_wide"]:
            self.lock_state = LOCK_PULL_IN
            self.fll_bw, self.pll_bw = c["fll_bandwidth_pullin"], 0.0
            self.spacing = self.sp_wide
            self.time_in_state = 0
        else:
            self.time_in_state += 1
        rec.update(corr=list(self.corr), dll=dll_d, pll=pll_d, fll=fll_d, carrier_err=carrier_err,
                   code_err=code_err, carrier_hz=self.carrier_hz, code_hz=self.code_hz, cn0=self.cn0,
                   pll_lock=self.pll_lock, fll_lock=self.fll_lock, lock_state=self.lock_state, flags=self.flags,
                   nav_bit=_decode_bit(self, ip))
        return rec


def _decode_bit(loop, i_prompt):
    """decodeBit (channel_l1ca_kaplan.py:728-754, channel_l1ca_borre.py:470-491) + Prompt2Bit
    (dsp/decoding.py:16-27): after bit sync, 20 prompt values decide one bit.  Returns 0/1 or -1."""
    if not (loop.flags & FLAG_BIT_SYNC):
        loop.nav_sum, loop.nav_count = 0.0, 0
        return -1
    loop.nav_sum += i_prompt
    loop.nav_count += 1
    if loop.nav_count != getattr(loop, "epochs_per_bit", MS_PER_BIT):
        return -1
    bit = 1 if loop.nav_sum > 0 else 0
    loop.nav_bits.append(bit)
    loop.nav_sum, loop.nav_count = 0.0, 0
    return bit


def post_acquisition(if_hz, doppler_range, doppler_step, peak, current_sample, acq_required, track_required):
    """postAcquisitionUpdate (channel_l1ca_kaplan.py:217-235): (carrier_hz, code_offset, current_sample)."""
    doppler = -((-doppler_range) + doppler_step * peak[0])
    code_offset = int(np.round(peak[1]))
    carrier_hz = if_hz + doppler
    current_sample = current_sample + acq_required
    current_sample -= track_required
    current_sample += code_offset + 1
    return carrier_hz, code_offset, current_sample


def required_samples(rem_code, code_step):
    return int(np.ceil((CODE_CHIPS - rem_code) / code_step))


# --- synthetic IQ for tests (host side; the HIP generator is checked through downloads) ---------

def synth_iq(fs, n_samples, sats, noise_sigma, seed, dtype=np.int8):
    """Seeded multi-satellite int8 IQ: list of dicts(prn, doppler, code_phase, phase, amp)."""
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    x = np.zeros(n_samples, dtype=np.complex128)
    for s in sats:
        code = gold_code(s["prn"])
        cstep = CODE_RATE * (1.0 + s["doppler"] / 1575.42e6) / fs
        chips = s["code_phase"] + n * cstep
        idx = np.floor(chips).astype(np.int64) % CODE_CHIPS
        period = np.floor(chips / CODE_CHIPS).astype(np.int64)
        if "data" in s:     # caller-supplied navigation symbols (+-1 per 20 code periods), e.g. encoded LNAV subframes
            bits = np.asarray(s["data"])
        else:
            bits = rng.integers(0, 2, size=int(period.max() // MS_PER_BIT) + 2) * 2 - 1
        data = bits[period // MS_PER_BIT]
        carrier = np.exp(2j * np.pi * (s["doppler"] / fs * n + s.get("phase", 0.0)))
        x += s["amp"] * code[idx] * data * carrier
    x += noise_sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    info = np.iinfo(dtype)
    lim = min(info.max, 127 if dtype == np.int8 else 32767)
    re = np.clip(np.rint(x.real), -lim, lim).astype(dtype)
    im = np.clip(np.rint(x.imag), -lim, lim).astype(dtype)
    out = np.empty(2 * n_samples, dtype=dtype)
    out[0::2] = re
    out[1::2] = im
    return out


def synth_iq_stream(fs, n_samples, sats, noise_sigma, seed, chunk=1 << 22, dtype=np.int8):
    """`synth_iq` for streams of many seconds, made in chunks of `chunk` samples (bounded memory): the same signal
    model on absolute sample numbers; every satellite needs its `data` symbols (+-1 per 20 code periods, counted
    from the code period in which sample 0 lies); the noise of chunk k is drawn from default_rng([seed, k])."""
    out = np.empty(2 * n_samples, dtype=dtype)
    info = np.iinfo(dtype)
    lim = min(info.max, 127 if dtype == np.int8 else 32767)
    codes = [gold_code(s["prn"]) for s in sats]
    for k, first in enumerate(range(0, n_samples, chunk)):
        m = min(chunk, n_samples - first)
        n = first + np.arange(m, dtype=np.float64)
        x = np.zeros(m, dtype=np.complex128)
        for s, code in zip(sats, codes):
            cstep = CODE_RATE * (1.0 + s["doppler"] / 1575.42e6) / fs
            chips = s["code_phase"] + n * cstep
            period = np.floor(chips / CODE_CHIPS).astype(np.int64)
            idx = np.floor(chips).astype(np.int64) - period * CODE_CHIPS
            data = np.asarray(s["data"])[period // MS_PER_BIT]
            x += s["amp"] * code[idx] * data * np.exp(2j * np.pi * (s["doppler"] / fs * n + s.get("phase", 0.0)))
        rng = np.random.default_rng([seed, k])
        x += noise_sigma * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
        out[2 * first:2 * (first + m):2] = np.clip(np.rint(x.real), -lim, lim).astype(dtype)
        out[2 * first + 1:2 * (first + m):2] = np.clip(np.rint(x.imag), -lim, lim).astype(dtype)
    return out


def iq_to_complex(raw: np.ndarray) -> np.ndarray:
    """Interleaved integer I,Q -> complex128 exactly as rfsignal.py:127-130 does."""
    return raw[0::2] + 1j * raw[1::2]
