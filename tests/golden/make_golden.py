#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; the GPU box has neither the
reference nor any need for this script):

    python tests/golden/make_golden.py

It imports aproposorg/sydr's own functions (sydr.dsp.acquisition.PCPS, sydr.dsp.tracking.EPL,
sydr.signal.gnsssignal, the two channel plugins ...) on seeded synthetic IQ and stores
inputs + outputs as small .npz fixtures.  Nothing from the reference's sources is copied:
only numbers it computed.  Every file records the NumPy version that produced it.
"""
import configparser
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SYDR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
os.environ.setdefault("MPLBACKEND", "Agg")

# third-party module the reference imports for calendar maths only (not on the path)
_gt = types.ModuleType("gps_time")


class _GPSTime:
    """Stand-in so that `from gps_time import GPSTime` (sydr/utils/time.py:4) succeeds.  The reference constructs one at IMPORT
    time (sydr/space/ephemeris.py:50 evaluates `Time()` in a class body -> GPSTime.from_datetime), so construction must work;
    but the object is poison: reading ANY attribute of it (week_number, time_of_week, ...) raises, and so does any other
    attribute of the module -- a fixture that was produced is thereby proven not to contain a value that came from here."""

    def __init__(self, *a, **k):
        pass

    @classmethod
    def from_datetime(cls, _dt):
        return cls()

    def __getattr__(self, name):
        raise RuntimeError(f"gps_time stand-in used: GPSTime().{name}")


def _gt_getattr(name):
    if name.startswith("__"):                   # (the import machinery's own probes: __path__, __spec__, ...)
        raise AttributeError(name)
    raise RuntimeError(f"gps_time stand-in used: gps_time.{name}")


_gt.__getattr__ = _gt_getattr
_gt.GPSTime = _GPSTime
sys.modules.setdefault("gps_time", _gt)

from sydr.dsp import acquisition as ref_acq  # noqa: E402
from sydr.dsp import tracking as ref_trk  # noqa: E402
from sydr.dsp import lockindicator as ref_lock  # noqa: E402
from sydr.signal import gnsssignal as ref_sig  # noqa: E402
from sydr.signal import ca as ref_ca  # noqa: E402
from sydr.signal.rfsignal import RFSignal  # noqa: E402
from sydr.utils.circularbuffer import CircularBuffer  # noqa: E402
from sydr.utils.enumerations import ChannelState  # noqa: E402

from oracle import sydr_oracle as orc  # noqa: E402  (only its seeded IQ synthesiser is used here)

META = dict(numpy_version=np.__version__, reference="aproposorg/sydr@/root/reference")


STAND_INS = np.array(["gps_time: third-party calendar module the reference imports (sydr/utils/time.py:4), absent from this "
                      "image; replaced by a stub so that the import succeeds -- reading any attribute of a stub object RAISES (only its import-time construction is allowed), so no value in these fixtures came from it"])


def save(name, **arrays):
    path = os.path.join(os.environ.get("SYDR_GOLDEN_OUT", HERE), name)     # (SYDR_GOLDEN_OUT: regenerate beside, to compare)
    arrays.setdefault("stand_ins", STAND_INS)     # what was NOT the reference's own code when this file was produced
    np.savez_compressed(path, numpy_version=np.array(np.__version__), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def iq_hash(raw):
    return np.frombuffer(hashlib.sha256(raw.tobytes()).digest(), dtype=np.uint8)


# ---------------------------------------------------------------------------------------------- G1 / G2
def make_codes():
    prns = list(range(1, 38)) + [64, 120, 138, 193, 210]
    chips = np.stack([ref_sig.GenerateGPSGoldCode(p).astype(np.int8) for p in prns])
    octal = np.array([ref_ca.first_10_chips(p) for p in prns], dtype=np.int64)
    ups = {}
    for fs in (4e6, 10e6, 25e6, 50e6, 12e6):
        n = ref_sig.getSamplesPerCode(fs)
        ramp = np.arange(1023, dtype=np.float64)  # upsampling a ramp returns the index map itself
        ups[f"upsample_idx_{int(fs)}"] = ref_sig.UpsampleCode(ramp, fs).astype(np.int16)
        assert len(ups[f"upsample_idx_{int(fs)}"]) == n
    save("g1_codes.npz", prns=np.array(prns), chips=chips, first10=octal, **ups)


# ---------------------------------------------------------------------------------------------- G3
def make_pcps():
    cases = []

    def run(tag, fs, sats, seed, doppler_range, doppler_step, coh, noncoh, search_prns, if_hz=0.0, sigma=20.0):
        n = ref_sig.getSamplesPerCode(fs)
        spc = round(fs / 1.023e6)
        total = n * coh * noncoh
        raw = orc.synth_iq(fs, total, sats, sigma, seed)
        rf = (raw[0::2] + 1j * raw[1::2]).reshape(1, -1)
        out = dict()
        out[f"{tag}_iq"] = raw
        out[f"{tag}_params"] = np.array([fs, if_hz, doppler_range, doppler_step, coh, noncoh, n, spc], dtype=np.float64)
        out[f"{tag}_prns"] = np.array(search_prns)
        peaks, ratios, rows, cols, sums = [], [], [], [], []
        for prn in search_prns:
            code = ref_sig.GenerateGPSGoldCode(prn)
            code_fft = np.conj(np.fft.fft(ref_sig.UpsampleCode(code, fs)))
            cmap = ref_acq.PCPS(rfData=rf, interFrequency=if_hz, samplingFrequency=fs, codeFFT=code_fft,
                                dopplerRange=doppler_range, dopplerStep=doppler_step, samplesPerCode=n,
                                coherentIntegration=coh, nonCoherentIntegration=noncoh)
            idx, ratio = ref_acq.TwoCorrelationPeakComparison(cmap, n, spc)
            peaks.append(idx)
            ratios.append(ratio)
            rows.append(cmap[idx[0], :].copy())
            cols.append(cmap[:, idx[1]].copy())
            sums.append(cmap.sum(axis=1))
        out[f"{tag}_peak"] = np.array(peaks, dtype=np.int64)
        out[f"{tag}_ratio"] = np.array(ratios)
        out[f"{tag}_row"] = np.stack(rows)
        out[f"{tag}_col"] = np.stack(cols)
        out[f"{tag}_binsum"] = np.stack(sums)
        cases.append(tag)
        return out

    data = {}
    sat7 = [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=8.0)]
    multi = [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=8.0),
             dict(prn=12, doppler=-3210.0, code_phase=17.6, phase=0.7, amp=6.0),
             dict(prn=30, doppler=4400.0, code_phase=1010.9, phase=0.3, amp=7.0)]
    # config 1 geometry: 4 MHz, +-5 kHz @ 250 Hz, 1 ms; PRN 7 present, PRN 9 absent
    data.update(run("a", 4e6, sat7, 20260001, 5000.0, 250.0, 1, 1, [7, 9]))
    # kaplan.ini integration (1 coh x 10 noncoh) with its 300 Hz step (34 bins, asymmetric grid)
    data.update(run("b", 4e6, multi, 20260002, 5000.0, 300.0, 1, 10, [12, 30, 5]))
    # borre.ini integration (5 coh x 10 noncoh), 100 Hz step
    data.update(run("c", 4e6, multi, 20260003, 5000.0, 100.0, 5, 10, [7]))
    # mixed small coh/noncoh + non-zero IF
    data.update(run("d", 4e6, multi, 20260004, 5000.0, 250.0, 2, 3, [7, 12], if_hz=1.25e5))
    # 10 MHz (the reference's own recording rate)
    data.update(run("e", 10e6, multi, 20260005, 5000.0, 250.0, 1, 1, [30, 1]))
    # 25 MHz (BASELINE config 2 geometry), one present + one absent PRN
    data.update(run("f", 25e6, multi, 20260006, 5000.0, 250.0, 1, 1, [12, 3]))
    save("g3_pcps.npz", cases=np.array(cases), **data)


# ---------------------------------------------------------------------------------------------- G4
def make_peaks():
    rng = np.random.default_rng(20260400)
    n, bins, spc = 200, 5, 4
    maps, outs_idx, outs_ratio = [], [], []
    positions = [0, 1, 2, 3, 4, 5, 100, n - 1, n - 2, n - 3, n - 4, n - 5, n - 6]
    for pos in positions:
        for row in (0, 2, 4):
            m = rng.random((bins, n))
            m[row, pos] = 5.0 + rng.random()
            # plant decoys: inside the exclusion window, at the never-read last sample, in another row
            m[row, min(n - 1, pos + 2)] = 4.5
            m[row, n - 1] = max(m[row, n - 1], 4.0) if pos != n - 1 else m[row, n - 1]
            m[(row + 1) % bins, (pos + 50) % n] = 4.9
            idx, ratio = ref_acq.TwoCorrelationPeakComparison(m.copy(), n, spc)
            maps.append(m)
            outs_idx.append(idx)
            outs_ratio.append(ratio)
    # exact ties: first occurrence in row-major order wins
    m = rng.random((bins, n))
    m[3, 77] = m[1, 150] = m[1, 20] = 9.0
    idx, ratio = ref_acq.TwoCorrelationPeakComparison(m.copy(), n, spc)
    maps.append(m)
    outs_idx.append(idx)
    outs_ratio.append(ratio)
    save("g4_peaks.npz", maps=np.stack(maps), idx=np.array(outs_idx, dtype=np.int64), ratio=np.array(outs_ratio),
         geometry=np.array([n, bins, spc]))


# ---------------------------------------------------------------------------------------------- G5 / G8
def make_epl():
    data = {}
    # the reference's own fixture: PRN 2, 3700 Hz, 10 MHz (sydr/unitTest/tracking_in_c.py:32-35)
    with open(os.path.join(REF, "sydr/unitTest/data/i_rfdata.txt")) as f:
        rf = np.loadtxt(f, dtype=complex)
    assert np.all(rf.real == np.rint(rf.real)) and np.all(rf.imag == np.rint(rf.imag))
    raw = np.empty(2 * len(rf), dtype=np.int8)
    raw[0::2] = rf.real.astype(np.int8)
    raw[1::2] = rf.imag.astype(np.int8)
    code = ref_sig.GenerateGPSGoldCode(2)
    padded = np.r_[code[-1], code, code[0]]
    fs = 10e6
    res = ref_trk.EPL(rf.reshape(1, -1), padded, fs, 3700.0, 0.0, 0.0, 1.023e6 / fs, (-0.5, 0.0, 0.5))
    data["fixture_iq"] = raw
    data["fixture_params"] = np.array([2, fs, 3700.0, 0.0, 0.0, 1.023e6 / fs])
    data["fixture_out"] = np.array(res)

    rng = np.random.default_rng(20260500)
    cases = []

    def case(tag, fs, prn, n, carrier, rem_carrier, rem_code, code_step, spacing, seed, fmt=np.int8):
        sats = [dict(prn=prn, doppler=carrier, code_phase=1023.0 - rem_code if rem_code else 0.0, phase=0.0, amp=9.0)]
        raw = orc.synth_iq(fs, n, sats, 20.0, seed, dtype=fmt)
        rf = (raw[0::2] + 1j * raw[1::2]).reshape(1, -1)
        code = ref_sig.GenerateGPSGoldCode(prn)
        padded = np.r_[code[-1], code, code[0]]
        res = ref_trk.EPL(rf, padded, fs, carrier, rem_carrier, rem_code, code_step, spacing)
        idx = [np.ceil(np.linspace(rem_code + s, code_step * n + rem_code + s, n, endpoint=False)).astype(np.int16)
               for s in spacing]
        data[f"{tag}_iq"] = raw
        data[f"{tag}_params"] = np.array([prn, fs, carrier, rem_carrier, rem_code, code_step, n], dtype=np.float64)
        data[f"{tag}_spacing"] = np.array(spacing)
        data[f"{tag}_out"] = np.array(res)
        data[f"{tag}_idx"] = np.stack(idx)
        cases.append(tag)

    for k, fs in enumerate((4e6, 25e6, 50e6)):
        step0 = 1.023e6 / fs
        n0 = int(np.ceil(1023 / step0))
        # exact-integer start phase (T9), nominal step
        case(f"z{k}", fs, 1 + k, n0, 1234.5 * (k + 1), 0.0, 0.0, step0, (-0.5, 0.0, 0.5), 20260510 + k)
        # random NCO state, n = N-1 / N+1, perturbed code step
        for v in range(2):
            step = (1.023e6 + rng.uniform(-3, 3)) / fs
            rem_code = rng.uniform(0, step)
            n = int(np.ceil((1023 - rem_code) / step)) + (-1 if v == 0 else 1) * (k % 2)
            case(f"r{k}{v}", fs, int(rng.integers(1, 33)), n, rng.uniform(-5000, 5000), rng.uniform(0, 2 * np.pi),
                 rem_code, step, (-0.5, 0.0, 0.5), 20260520 + 10 * k + v)
    # narrow correlator + large IF (phase of thousands of radians) + int16 samples
    step = (1.023e6 - 1.7) / 25e6
    case("narrow", 25e6, 17, 25000, 4.092e6 + 812.0, 1.0, 0.01, step, (-0.1, 0.0, 0.1), 20260590)
    case("int16", 4e6, 22, 4001, -2400.0, 5.5, 0.1, 1.023e6 / 4e6, (-0.5, 0.0, 0.5), 20260591, fmt=np.int16)
    data["cases"] = np.array(cases)

    # G8: replica known answers (sydr/c_functions/tracking.c:243-247), evaluated by the live NumPy formula
    t = np.arange(0, 6) / 1e7
    rep, rem = ref_trk.generateReplica(t, 5, -1500.0, 0.0)
    data["replica_known"] = rep
    data["replica_rem"] = np.array(rem)
    save("g5_epl.npz", **data)


# ---------------------------------------------------------------------------------------------- G6
def _channel_config(path, overrides):
    cfg = configparser.ConfigParser()
    cfg.read(path)
    for sec, kv in overrides.items():
        for k, v in kv.items():
            cfg[sec][k] = str(v)
    return cfg


def make_trajectories(fname="g6_trajectories.npz", ms=510, amp=8.0, sigma=20.0, seed=20260001,
                      plugins=("borre", "kaplan"), track_over=None, fs=4e6):
    from sydr.channel.channel_l1ca_borre import ChannelL1CA as RefBorre
    from sydr.channel.channel_l1ca_kaplan import ChannelL1CA_Kaplan as RefKaplan

    spms = int(fs * 1e-3)
    sats = [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=amp)]
    raw = orc.synth_iq(fs, ms * spms, sats, sigma, seed)
    rf = raw[0::2] + 1j * raw[1::2]
    acq_over = {"ACQUISITION": dict(doppler_steps=250, coherent_integration=1, non_coherent_integration=1)}
    if track_over:
        acq_over["TRACKING"] = track_over
    out = dict(iq_sha256=iq_hash(raw), synth=np.array([fs, ms * spms, 7, 1750.0, 300.25, 0.1, amp, sigma, seed]),
               track_override_keys=np.array(sorted(track_over or {})),
               track_override_vals=np.array([float((track_over or {})[k]) for k in sorted(track_over or {})]))

    for tag, cls, ini in (("borre", RefBorre, "channel_GPS_L1CA_borre.ini"),
                          ("kaplan", RefKaplan, "channel_GPS_L1CA_kaplan.ini")):
        if tag not in plugins:
            continue
        cfg = _channel_config(os.path.join(REF, "config/channels", ini), acq_over)
        rfs = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0,
                            data_size=8))
        buf = CircularBuffer(100 * spms, complex)
        ch = cls(0, buf, None, rfs, cfg)
        ch.setSatellite(7)
        acq, epochs = None, []
        for k in range(ms):
            buf.shift(rf[k * spms:(k + 1) * spms])
            pre = None
            if ch.channelState == ChannelState.TRACKING:
                rem_c = ch.NCO_remainingCarrier if tag == "borre" else ch.remainingCarrier
                rem_k = ch.NCO_remainingCode if tag == "borre" else ch.remainingCode
                pre = [ch.currentSample, ch.track_requiredSamples, ch.carrierFrequency, rem_c, rem_k, ch.codeStep]
            bits_before = getattr(ch, "navBitsCounter", 0)
            results = ch._processHandler()
            bits_after = getattr(ch, "navBitsCounter", 0)
            nav_bit = float(ch.navBitsBuffer[bits_after - 1]) if bits_after == bits_before + 1 else -1.0
            for r in results:
                if "correlation_map" in r:
                    acq = [r["frequency_idx"], r["code_idx"], r["peak_ratio"], r["carrierFrequency"], r["codeOffset"],
                           ch.currentSample, ch.track_requiredSamples]
                elif "i_prompt" in r:
                    epochs.append(pre + [r["i_early"], r["q_early"], r["i_prompt"], r["q_prompt"], r["i_late"],
                                         r["q_late"], r["dll"], r["pll"], r["fll"], r["carrier_frequency"],
                                         r["code_frequency"], r["carrier_frequency_error"],
                                         r["code_frequency_error"], r["cn0"], r["pll_lock"], r["fll_lock"],
                                         float(int(r["lock_state"])), float(int(ch.trackFlags)), nav_bit])
        out[f"{tag}_acq"] = np.array(acq, dtype=np.float64)
        out[f"{tag}_epochs"] = np.array(epochs, dtype=np.float64)
        print(tag, "acq", acq, "epochs", len(epochs))
    out["epoch_columns"] = np.array(["currentSample", "n", "carrier_in", "rem_carrier_in", "rem_code_in",
                                     "code_step_in", "ie", "qe", "ip", "qp", "il", "ql", "dll", "pll", "fll",
                                     "carrier_hz", "code_hz", "carrier_err", "code_err", "cn0", "pll_lock",
                                     "fll_lock", "lock_state", "flags", "nav_bit"])
    save(fname, **out)


# ---------------------------------------------------------------------------------------------- G10
def make_decoding(fname="g10_decoding.npz", seconds=20.6, lead_bits=30, fs=4e6):
    """The Kaplan plugin over a stream that carries valid LNAV subframes (oracle/lnav.py encodes them): the
    reference's own DECODING_UPDATE packets (kaplan:702-868), the `tow` / `time_since_tow` / `tracking_flags`
    / `code_since_tow` of every CHANNEL_UPDATE (channel.py:205-228) and the bit decided in every epoch.
    The stream starts `lead_bits` navigation bits before a subframe boundary; three whole subframes (IDs 1, 2, 3)
    follow, so the plugin passes SUBFRAME_SYNC, TOW_DECODED and EPH_DECODED.  IQ is not stored (seed + sha256)."""
    from sydr.channel.channel_l1ca_kaplan import ChannelL1CA_Kaplan as RefKaplan
    from sydr.utils.enumerations import ChannelMessage
    from oracle import lnav

    spms = int(fs * 1e-3)
    ms = int(round(seconds * 1000))
    first_tow, lnav_seed, n_sub = 57600, 20261000, 6
    frames = lnav.lnav_stream(first_tow, 5, n_sub, lnav_seed)          # subframe 5, then 1, 2, 3, 4, 5
    symbols = (2 * frames.astype(np.int64) - 1)[300 - lead_bits:]
    prn, doppler, code_phase, phase, amp, sigma, seed = 7, 1750.0, 300.25, 0.1, 30.0, 10.0, 20261001
    # code_phase chips into a period at sample 0; the bit edges fall on code-period boundaries
    sats = [dict(prn=prn, doppler=doppler, code_phase=code_phase, phase=phase, amp=amp, data=symbols)]
    raw = orc.synth_iq_stream(fs, ms * spms, sats, sigma, seed)
    track_over = dict(fll_threshold_wide=0.3, fll_threshold_narrow=0.7, pll_threshold_narrow=0.7, dll_threshold=3.0,
                      correlator_epl_narrow=0.25)
    cfg = _channel_config(os.path.join(REF, "config/channels/channel_GPS_L1CA_kaplan.ini"),
                          {"ACQUISITION": dict(doppler_steps=250, coherent_integration=1, non_coherent_integration=1),
                           "TRACKING": track_over})
    rfs = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0,
                        data_size=8))
    buf = CircularBuffer(100 * spms, complex)
    ch = RefKaplan(0, buf, None, rfs, cfg)
    ch.setSatellite(prn)
    acq, ints, floats, dec_tick, dec_epoch, dec_meta, dec_bits = None, [], [], [], [], [], []
    updates = []
    for k in range(ms):
        slab = raw[2 * k * spms:2 * (k + 1) * spms].astype(np.float64)
        buf.shift(slab[0::2] + 1j * slab[1::2])
        pre = [ch.currentSample, ch.track_requiredSamples] if ch.channelState == ChannelState.TRACKING else None
        bits_before = ch.navBitsCounter
        results = ch._processHandler()
        nav_bit = -1
        for r in results:
            if r["type"] == ChannelMessage.ACQUISITION_UPDATE:
                acq = [r["frequency_idx"], r["code_idx"], r["peak_ratio"], r["carrierFrequency"], r["codeOffset"],
                       ch.currentSample, ch.track_requiredSamples]
            elif r["type"] == ChannelMessage.TRACKING_UPDATE:
                ints.append(pre + [int(ch.trackFlags), int(r["lock_state"]), -1])
                floats.append([r["i_prompt"], r["q_prompt"], r["carrier_frequency"], r["code_frequency"], r["cn0"]])
            elif r["type"] == ChannelMessage.DECODING_UPDATE:
                assert sorted(r) == ["bits", "cid", "subframe_id", "tow", "type"]
                dec_tick.append(k)
                dec_epoch.append(len(ints) - 1)
                dec_meta.append([r["subframe_id"], r["tow"]])
                dec_bits.append(np.frombuffer(r["bits"].encode(), dtype=np.uint8) - ord("0"))
        if ints and ch.navPromptSumCounter == 0 and (int(ch.trackFlags) & 2) and any("i_prompt" in r for r in results) \
                and getattr(ch, "_g10_last_bit_epoch", None) != len(ints):
            # a bit was decided in this epoch: it is the newest entry of the plugin's buffer, unless the buffer was
            # just re-packed by decodeSubframe (then it sits at the end of what was kept)
            nav_bit = int(ch.navBitsBuffer[ch.navBitsCounter - 1])
            ints[-1][-1] = nav_bit
            ch._g10_last_bit_epoch = len(ints)
        u = ch.prepareChannelUpdate()
        updates.append([k, int(u["state"].value), int(u["tracking_flags"]), float(u["tow"]), float(u["time_since_tow"]),
                        int(u["unprocessed_samples"]), int(u["code_since_tow"])])
    ints = np.array(ints, dtype=np.int64)
    print("g10: acq", acq, "epochs", len(ints), "bits", int((ints[:, 4] >= 0).sum()), "decoding packets", dec_meta,
          "at ticks", dec_tick, "final flags", int(ch.trackFlags), "tow", ch.tow)
    save(fname, iq_sha256=iq_hash(raw),
         synth=np.array([fs, ms * spms, prn, doppler, code_phase, phase, amp, sigma, seed]),
         lnav=np.array([first_tow, 5, n_sub, lnav_seed, lead_bits]),
         track_override_keys=np.array(sorted(track_over)),
         track_override_vals=np.array([float(track_over[k]) for k in sorted(track_over)]),
         acq=np.array(acq, dtype=np.float64), epoch_ints=ints.astype(np.int32),
         epoch_int_columns=np.array(["currentSample", "n", "flags", "lock_state", "nav_bit"]),
         epoch_floats=np.array(floats)[::50], epoch_float_stride=np.array(50),
         epoch_float_columns=np.array(["i_prompt", "q_prompt", "carrier_hz", "code_hz", "cn0"]),
         updates=np.array(updates, dtype=np.float64),
         update_columns=np.array(["tick", "state", "tracking_flags", "tow", "time_since_tow", "unprocessed_samples",
                                  "code_since_tow"]),
         dec_tick=np.array(dec_tick), dec_epoch=np.array(dec_epoch), dec_meta=np.array(dec_meta, dtype=np.int64),
         dec_bits=np.array(dec_bits, dtype=np.uint8))


# ---------------------------------------------------------------------------------------------- G9
def make_serial():
    """SerialSearch + TwoCorrelationPeakComparison_SS (acquisition.py:119-193), reduced Doppler grid."""
    fs, n = 4e6, 4000
    sats = [dict(prn=11, doppler=500.0, code_phase=700.5, phase=0.3, amp=9.0)]
    raw = orc.synth_iq(fs, 2 * n, sats, 15.0, 20260900)
    rf = (raw[0::2] + 1j * raw[1::2])
    code = ref_sig.GenerateGPSGoldCode(11)
    maps = [ref_acq.SerialSearch(rf[k * n:(k + 1) * n].reshape(1, -1), code, 750.0, 250.0, fs, n) for k in range(2)]
    idx, ratio = ref_acq.TwoCorrelationPeakComparison_SS(maps[0].copy())
    idx2, ratio2 = ref_acq.TwoCorrelationPeakComparison_SS((maps[0] + maps[1]).copy())
    rng = np.random.default_rng(20260901)
    edge_maps, edge_idx, edge_ratio = [], [], []
    for (r, c) in ((0, 5), (3, 0), (0, 0), (6, 1022), (3, 500), (6, 0), (1, 1)):
        m = rng.random((7, 1023))
        m[r, c] = 9.0
        m[min(6, r + 1), min(1022, c + 1)] = 8.0     # inside the 3x3 block when the block is not empty
        m[(r + 3) % 7, (c + 300) % 1023] = 7.0
        i, q = ref_acq.TwoCorrelationPeakComparison_SS(m.copy())
        edge_maps.append(m)
        edge_idx.append(i)
        edge_ratio.append(q)
    # the SerialSearch plugin end to end: acquisition packet + the first tracking epochs after it
    from sydr.channel.channel_l1ca_kaplan_ss import ChannelL1CA_Kaplan_SS as RefSS
    sats7 = [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=8.0)]
    spms = int(fs * 1e-3)
    raw7 = orc.synth_iq(fs, 40 * spms, sats7, 20.0, 20260001)
    rf7 = raw7[0::2] + 1j * raw7[1::2]
    cfg = _channel_config(os.path.join(REF, "config/channels/channel_GPS_L1CA_kaplan.ini"),
                          {"ACQUISITION": dict(doppler_range=2000, doppler_steps=250, coherent_integration=1,
                                               non_coherent_integration=2)})
    rfs = RFSignal(dict(filepath="none", sampling_frequency=fs, is_complex="true", intermediate_frequency=0.0, data_size=8))
    buf = CircularBuffer(100 * spms, complex)
    ch = RefSS(0, buf, None, rfs, cfg)
    ch.setSatellite(7)
    ss_acq, ss_epochs = None, []
    for k in range(40):
        buf.shift(rf7[k * spms:(k + 1) * spms])
        for r in ch._processHandler():
            if "correlation_map" in r:
                ss_acq = [r["frequency_idx"], r["code_idx"], r["peak_ratio"], r["carrierFrequency"], r["codeOffset"],
                          ch.currentSample, ch.track_requiredSamples]
            elif "i_prompt" in r:
                ss_epochs.append([r["i_early"], r["q_early"], r["i_prompt"], r["q_prompt"], r["i_late"], r["q_late"],
                                  r["carrier_frequency"], r["code_frequency"]])
    print("SS plugin acq", ss_acq, "epochs", len(ss_epochs))
    save("g9_serial.npz", ss_iq_sha256=iq_hash(raw7), ss_acq=np.array(ss_acq, dtype=np.float64),
         ss_epochs=np.array(ss_epochs), iq=raw, params=np.array([fs, n, 11, 750.0, 250.0]), map0=maps[0], map1=maps[1],
         peak=np.array(idx), ratio=np.array(ratio), peak_sum=np.array(idx2), ratio_sum=np.array(ratio2),
         edge_maps=np.stack(edge_maps), edge_idx=np.array(edge_idx), edge_ratio=np.array(edge_ratio))


# ---------------------------------------------------------------------------------------------- G7
def make_loopmath():
    rng = np.random.default_rng(20260700)
    m = 64
    c = rng.normal(0, 3e4, size=(m, 8))
    c[0, 2] = 0.0   # iPrompt = 0 -> atan(+-inf)
    c[1, 2:4] = 0.0  # 0/0 -> nan branch of FLL_ATAN
    with np.errstate(all="ignore"):
        dll = np.array([ref_trk.DLL_NNEML(r[0], r[1], r[4], r[5]) for r in c])
        pll = np.array([ref_trk.PLL_costa(r[2], r[3]) for r in c])
        fll = np.array([ref_trk.FLL_ATAN(r[2], r[3], r[6], r[7], 1e-3) for r in c])
        flock = np.array([ref_lock.FLL_Lock_Borre(r[2], r[6], r[3], r[7], 0.3, alpha=0.005) for r in c])
        plock = np.array([ref_lock.PLL_Lock_Borre(r[2], r[3], 0.4, alpha=0.005) for r in c])
    coeff = np.array([ref_trk.LoopFiltersCoefficients(b, z, g) for b, z, g in
                      ((2.0, 0.7, 1.0), (1.0, 0.7, 1.0), (8.0, 0.7, 0.25), (15.0, 0.7, 1.5))])
    bf = np.array([ref_trk.BorreLoopFilter(r[0] * 1e-5, r[1] * 1e-5, coeff[0, 0], coeff[0, 1], 0.001) for r in c])
    fp = np.array([ref_trk.FLLassistedPLL_2ndOrder(r[0] * 1e-6, r[1] * 1e-3, 100.0 / 0.25, 25.0 / 0.53, 1.414, 1e-3,
                                                   r[2] * 1e-4) for r in c])
    cn0 = np.array([ref_lock.CN0_Beaulieu(abs(r[0]) * 1e-3 + 1.0, 20, 20e-3, abs(r[1]) * 1e-3) for r in c])
    save("g7_loopmath.npz", inputs=c, dll=dll, pll=pll, fll=fll, fll_lock=flock, pll_lock=plock, coeff=coeff,
         borre_filter=bf, fll_pll=fp, cn0=cn0)


if __name__ == "__main__":
    which = sys.argv[1:] or ["codes", "pcps", "peaks", "epl", "traj", "loop", "serial", "decoding"]
    if "serial" in which:
        make_serial()
    if "codes" in which:
        make_codes()
    if "pcps" in which:
        make_pcps()
    if "peaks" in which:
        make_peaks()
    if "epl" in which:
        make_epl()
    if "traj" in which:
        make_trajectories()
        # strong signal, 1.5 s: drives the Kaplan plugin through PULL_IN -> WIDE -> NARROW, code lock, bit sync
        make_trajectories("g6b_kaplan_strong.npz", ms=1200, amp=30.0, sigma=10.0, seed=20260611, plugins=("kaplan",),
                          track_over=dict(fll_threshold_wide=0.3, fll_threshold_narrow=0.7, pll_threshold_narrow=0.7,
                                          dll_threshold=3.0, correlator_epl_narrow=0.25))
        # the headline sampling rate (BASELINE configs 2-3): both plugins at 25 MHz, PCPS + ~300 epochs.  The IQ
        # (15 MB) is not stored: seed + sha256, regenerated by the tests
        make_trajectories("g6c_25mhz.npz", ms=310, amp=8.0, sigma=20.0, seed=20260625, fs=25e6)
    if "loop" in which:
        make_loopmath()
    if "decoding" in which:
        make_decoding()
