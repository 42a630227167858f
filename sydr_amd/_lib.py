"""ctypes binding of libsydr_amd.so (the C-ABI declared in include/sydr_amd.h).

This is the thin layer that supersedes the reference's ctypes wrappers around
sydr/c_functions (sydr/old/tracking/tracking_epl_c.py:31-96,
sydr/old/acquisition/acquisition_pcps_c.py:32-66).  There is no CPU fallback: if the
shared library is missing or no MI355X is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYDR_AMD_LIB") or os.path.join(_HERE, "libsydr_amd.so")  # (override: A/B builds of tools/ab_compare.sh)

SDR_MAX_TAPS = 8
FMT_CI8, FMT_CI16, FMT_CF32, FMT_CF64 = 0, 1, 2, 3
_FMT_NP = {FMT_CI8: np.int8, FMT_CI16: np.int16, FMT_CF32: np.float32, FMT_CF64: np.float64}


class SdrError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"sydr_amd error {status}: {text}")
        self.status = status


class EplItem(C.Structure):
    _fields_ = [("code_slot", C.c_int32), ("n_samples", C.c_int32), ("start_sample", C.c_int64),
                ("carrier_hz", C.c_double), ("rem_carrier", C.c_double), ("rem_code", C.c_double),
                ("code_step", C.c_double)]


EPL_ITEM_DTYPE = np.dtype([("code_slot", np.int32), ("n_samples", np.int32), ("start_sample", np.int64),
                           ("carrier_hz", np.float64), ("rem_carrier", np.float64), ("rem_code", np.float64),
                           ("code_step", np.float64)], align=True)


class SynthSat(C.Structure):
    _fields_ = [("prn", C.c_int32), ("flags", C.c_int32), ("doppler_hz", C.c_double),
                ("code_phase", C.c_double), ("carrier_phase", C.c_double), ("amplitude", C.c_double)]


class TrackState(C.Structure):
    _fields_ = [("code_slot", C.c_int32), ("n_samples", C.c_int32), ("current_sample", C.c_int64),
                ("carrier_hz", C.c_double), ("code_hz", C.c_double), ("rem_carrier", C.c_double),
                ("rem_code", C.c_double), ("code_step", C.c_double), ("dll_mem", C.c_double),
                ("pll_mem", C.c_double), ("i_prompt_prev", C.c_double), ("q_prompt_prev", C.c_double),
                ("fll_lock", C.c_double), ("pll_lock", C.c_double), ("cn0", C.c_double),
                ("cn0_ratio_acc", C.c_double), ("fll_bw", C.c_double), ("pll_bw", C.c_double),
                ("code_counter", C.c_int32), ("accum_counter", C.c_int32), ("lock_state", C.c_int32),
                ("track_flags", C.c_int32), ("time_in_state", C.c_int32), ("spacing_sel", C.c_int32),
                ("nav_prompt_sum", C.c_double), ("nav_sum_counter", C.c_int32), ("nav_bits_emitted", C.c_int32)]


class LoopCfg(C.Structure):
    _fields_ = [("loop_kind", C.c_int32), ("n_taps", C.c_int32), ("fs", C.c_double),
                ("spacing_wide", C.c_double * SDR_MAX_TAPS), ("spacing_narrow", C.c_double * SDR_MAX_TAPS),
                ("dll_tau1", C.c_double), ("dll_tau2", C.c_double), ("dll_pdi", C.c_double),
                ("pll_tau1", C.c_double), ("pll_tau2", C.c_double), ("pll_pdi", C.c_double),
                ("dll_threshold", C.c_double),
                ("fll_bw_pullin", C.c_double), ("fll_bw_wide", C.c_double), ("fll_bw_narrow", C.c_double),
                ("fll_thr_wide", C.c_double), ("fll_thr_narrow", C.c_double),
                ("pll_bw_wide", C.c_double), ("pll_bw_narrow", C.c_double),
                ("pll_thr_wide", C.c_double), ("pll_thr_narrow", C.c_double),
                ("epoch_chips", C.c_double), ("epochs_per_bit", C.c_int32), ("reserved", C.c_int32),
                ("epoch_seconds", C.c_double)]


# The same records as NumPy dtypes: the host keeps the channel bank's mirror as structured arrays.
TRACK_STATE_DTYPE = np.dtype(TrackState)
LOOP_CFG_DTYPE = np.dtype(LoopCfg)


class TrackEpoch(C.Structure):
    _fields_ = [("start_sample", C.c_int64), ("n_samples", C.c_int32), ("lock_state", C.c_int32),
                ("carrier_hz_in", C.c_double), ("rem_carrier_in", C.c_double), ("rem_code_in", C.c_double),
                ("code_step_in", C.c_double), ("corr", C.c_double * (2 * SDR_MAX_TAPS)),
                ("dll", C.c_double), ("pll", C.c_double), ("fll", C.c_double),
                ("carrier_err", C.c_double), ("code_err", C.c_double),
                ("carrier_hz", C.c_double), ("code_hz", C.c_double),
                ("cn0", C.c_double), ("pll_lock", C.c_double), ("fll_lock", C.c_double),
                ("track_flags", C.c_int32), ("nav_bit", C.c_int32)]


TRACK_EPOCH_DTYPE = np.dtype([("start_sample", np.int64), ("n_samples", np.int32), ("lock_state", np.int32),
                              ("carrier_hz_in", np.float64), ("rem_carrier_in", np.float64),
                              ("rem_code_in", np.float64), ("code_step_in", np.float64),
                              ("corr", np.float64, (2 * SDR_MAX_TAPS,)),
                              ("dll", np.float64), ("pll", np.float64), ("fll", np.float64),
                              ("carrier_err", np.float64), ("code_err", np.float64),
                              ("carrier_hz", np.float64), ("code_hz", np.float64),
                              ("cn0", np.float64), ("pll_lock", np.float64), ("fll_lock", np.float64),
                              ("track_flags", np.int32), ("nav_bit", np.int32)], align=True)



class TickUpdate(C.Structure):
    _fields_ = [("channel", C.c_int32), ("track_flags", C.c_int32), ("unread", C.c_int64), ("epochs_since_tow", C.c_int64)]


TICK_UPDATE_DTYPE = np.dtype(TickUpdate)


class TickMirror(C.Structure):
    """sdr_tick_mirror (include/sydr_amd.h): the caller's mirrors of the bank, updated in place by sdr_bank_tick_mirrored."""
    _fields_ = [("max_channels", C.c_int32), ("reserved", C.c_int32),
                ("states", C.c_void_p), ("last", C.c_void_p), ("epochs_since_tow", C.c_void_p),
                ("tracking", C.c_void_p), ("lost", C.c_void_p), ("host_flags", C.c_void_p),
                ("ran", C.c_void_p), ("records", C.c_void_p), ("updates", C.c_void_p),
                ("n_ran", C.c_int32), ("n_updates", C.c_int32), ("n_nav_bits", C.c_int32), ("n_lost", C.c_int32),
                ("max_unread", C.c_int64)]


_VP = C.c_void_p
_PROTOTYPES = {
    "sdr_last_error": (C.c_char_p, []),
    "sdr_abi_version": (C.c_int, []),
    "sdr_build_id": (C.c_char_p, []),
    "sdr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sdr_engine_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "sdr_engine_destroy": (None, [_VP]),
    "sdr_engine_sync": (C.c_int, [_VP]),
    "sdr_prof_enable": (C.c_int, [_VP, C.c_int]),
    "sdr_prof_read": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "sdr_prof_reset": (C.c_int, [_VP]),
    "sdr_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "sdr_hbm_copy_rate": (C.c_int, [_VP, C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    "sdr_iq_alloc": (C.c_int, [_VP, C.c_int64, C.c_int]),
    "sdr_iq_upload": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64]),
    "sdr_iq_download": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64]),
    "sdr_iq_synth": (C.c_int, [_VP, C.POINTER(SynthSat), C.c_int, C.c_double, C.c_double, C.c_uint64,
                               C.c_int64, C.c_int64]),
    "sdr_code_slots": (C.c_int, [_VP, C.c_int, C.c_int]),
    "sdr_code_slots_ex": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    "sdr_code_gps_l1ca": (C.c_int, [_VP, C.c_int, C.c_int]),
    "sdr_code_custom": (C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    "sdr_code_read": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.POINTER(C.c_int)]),
    "sdr_code_upsample": (C.c_int, [_VP, C.c_int, C.c_double, C.c_int64, _VP]),
    "sdr_epl_batch": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, C.c_double, _VP]),
    "sdr_epl_plan_create": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, C.c_double, C.POINTER(_VP)]),
    "sdr_epl_plan_create_dev": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, C.c_double, C.POINTER(_VP)]),
    "sdr_epl_plan_run": (C.c_int, [_VP, _VP]),
    "sdr_epl_plan_run_range": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64]),
    "sdr_epl_plan_fetch": (C.c_int, [_VP, _VP, _VP]),
    "sdr_epl_plan_destroy": (None, [_VP, _VP]),
    "sdr_pcps": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                           C.c_int, C.c_int, _VP, _VP, _VP, _VP, C.POINTER(C.c_int)]),
    "sdr_pcps_spectra": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_int, C.c_int, _VP, _VP, _VP, _VP, C.POINTER(C.c_int)]),
    "sdr_pcps_bins": (C.c_int, [C.c_double, C.c_double]),
    "sdr_two_peak_compare": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sdr_serial_search": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int, _VP, _VP,
                                    _VP, _VP, C.POINTER(C.c_int)]),
    "sdr_two_peak_compare_ss": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_double)]),
    "sdr_track_cluster": (C.c_int, [_VP, C.c_int]),
    "sdr_track_closed_loop": (C.c_int, [_VP, C.c_int, _VP, C.POINTER(LoopCfg), C.c_int, _VP]),
    "sdr_track_closed_loop_bits": (C.c_int, [_VP, C.c_int, _VP, C.POINTER(LoopCfg), C.c_int, _VP, _VP, C.c_int, _VP]),
    "sdr_track_closed_loop_ex": (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int, C.c_int, _VP, _VP, C.c_int, _VP, _VP]),
    "sdr_bank_create": (C.c_int, [_VP, C.c_int, C.POINTER(_VP)]),
    "sdr_bank_destroy": (None, [_VP, _VP]),
    "sdr_bank_put": (C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    "sdr_bank_get": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "sdr_bank_step": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, C.c_int, _VP, C.c_int]),
    "sdr_bank_step_begin": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int]),
    "sdr_bank_step_end": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "sdr_bank_tick": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int, _VP, _VP, _VP]),
    "sdr_bank_tick_mirrored": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.POINTER(TickMirror)]),
    "sdr_bank_tick_mirrored_begin": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.POINTER(TickMirror)]),
    "sdr_bank_tick_mirrored_end": (C.c_int, [_VP, _VP, C.POINTER(TickMirror)]),
    "sdr_iq_upload_begin": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64]),
    "sdr_iq_upload_queue": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64]),
    "sdr_tick_server_stats": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "sdr_tick_server_phases": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "sdr_tick_server_tracker_phases": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "sdr_host_alloc": (C.c_int, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "sdr_host_free": (C.c_int, [_VP, _VP]),
    "sdr_block_schedule": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, C.c_int64, _VP, _VP, C.c_int] + [_VP] * 15),
    "sdr_stream_create": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "sdr_stream_sync": (C.c_int, [_VP, C.c_int]),
    "sdr_epl_plan_run_range_on": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int]),
    "sdr_epl_plan_variant": (C.c_int, [_VP]),
}
ABI_VERSION = 5

_lib = None


def load():
    """Load the shared library once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C sydr_amd/csrc` (there is no CPU fallback for the correlator engine)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.sdr_abi_version() != ABI_VERSION:
        raise ImportError("libsydr_amd.so ABI version mismatch")
    _lib = lib
    return lib


def source_build_id() -> str:
    """sdr_build_id() of a library built from the sources in the tree NOW (the Makefile's recipe): differs from the loaded
    library's when that was built from other sources."""
    import hashlib
    src = os.path.join(_HERE, "csrc")
    names = sorted(n for n in os.listdir(src) if (n.endswith(".hip") or n.endswith(".h")) and n != "build_id.h")
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(src, n), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_HERE), "include", "sydr_amd.h"), "rb").read())
    return h.hexdigest()[:16]


def exported_symbols():
    return sorted(_PROTOTYPES)


def check(status):
    if status != 0:
        raise SdrError(status, load().sdr_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().sdr_device_count(C.byref(n)))
    return n.value


def ptr(a: np.ndarray):
    return a.ctypes.data_as(_VP)


def fmt_dtype(fmt):
    return _FMT_NP[fmt]
