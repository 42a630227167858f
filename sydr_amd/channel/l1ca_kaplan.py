"""GPS L1 C/A channel, Kaplan-style loops (FLL-assisted PLL, PULL_IN / WIDE / NARROW lock states, C/N0) --
the plugin surface of sydr/channel/channel_l1ca_kaplan.py with the tracking state on the GPU.

Everything per-epoch (kaplan:342-619: correlators, discriminators, loop filters, lock indicators, NCO update,
lock-state machine, bit sync) runs in `track_kernel` (sydr_amd/csrc/track.hip, loop_kind 1); this file only says
which INI key feeds which `sdr_loop_cfg` field and under which of the reference's attribute names each state
field is visible.  LNAV word / subframe decoding (kaplan:703-868, sydr/dsp/decoding.py) is outside the accelerated path
(BASELINE.json: navigation untouched) and is not re-implemented: the device delivers the bits (`navBits`), and the decoder
seam (`setDecoding` / `runDecoding`, sydr_amd/channel/navdecoder.py) hands them to the reference's own subframe methods, so
that DECODING_UPDATE packets, `tow` and the TOW / EPH flags leave the manager as the reference's would."""
from __future__ import annotations

from ..utils.enumerations import LoopLockState
from .bank import KIND_KAPLAN
from .tracked import DeviceTrackedChannel


class ChannelL1CA_Kaplan(DeviceTrackedChannel):
    LOOP_KIND = KIND_KAPLAN
    IDX_I_EARLY, IDX_Q_EARLY, IDX_I_PROMPT, IDX_Q_PROMPT, IDX_I_LATE, IDX_Q_LATE = range(6)

    # config/channels/channel_GPS_L1CA_kaplan.ini [TRACKING] -> sdr_loop_cfg
    CFG_KEYS = {
        "dll_pdi": "dll_pdi", "dll_threshold": "dll_threshold",
        "fll_bandwidth_pullin": "fll_bw_pullin", "fll_bandwidth_wide": "fll_bw_wide", "fll_bandwidth_narrow": "fll_bw_narrow",
        "fll_threshold_wide": "fll_thr_wide", "fll_threshold_narrow": "fll_thr_narrow",
        "pll_bandwidth_wide": "pll_bw_wide", "pll_bandwidth_narrow": "pll_bw_narrow",
        "pll_threshold_wide": "pll_thr_wide", "pll_threshold_narrow": "pll_thr_narrow",
    }
    FILTERS = (("dll", "dll"),)

    # the reference's attribute names over the device state (sdr_track_state) ...
    STATE_VIEW = {
        "carrierFrequency": ("carrier_hz", float), "codeFrequency": ("code_hz", float),
        "remainingCarrier": ("rem_carrier", float), "remainingCode": ("rem_code", float),
        "codeStep": ("code_step", float), "track_requiredSamples": ("n_samples", int),
        "codeCounter": ("code_counter", int), "correlatorsAccumCounter": ("accum_counter", int),
        "dllDiscrim": ("dll_mem", float), "fll_vel_memory": ("pll_mem", float),
        "iPromptPrev": ("i_prompt_prev", float), "qPromptPrev": ("q_prompt_prev", float),
        "fllLockIndicator": ("fll_lock", float), "pllLockIndicator": ("pll_lock", float),
        "cn0": ("cn0", float), "dllLockIndicator": ("cn0", float), "cn0_PdPnRatio": ("cn0_ratio_acc", float),
        "fllBandwidth": ("fll_bw", float), "pllBandwidth": ("pll_bw", float),
        "timeSinceLastState": ("time_in_state", int), "loopLockState": ("lock_state", LoopLockState),
        "navPromptSum": ("nav_prompt_sum", float), "navPromptSumCounter": ("nav_sum_counter", int),
    }
    # ... and over the latest epoch record (sdr_track_epoch)
    RECORD_VIEW = {"pllDiscrim": "pll", "fllDiscrim": "fll", "carrierFrequencyError": "carrier_err",
                   "codeFrequencyError": "code_err"}

    def _configure_taps(self, configuration, cfg):
        wide, narrow = float(configuration["correlator_epl_wide"]), float(configuration["correlator_epl_narrow"])
        self.dll_epl_wide, self.dll_epl_narrow = [-wide, 0.0, wide], [-narrow, 0.0, narrow]
        cfg["spacing_wide"][:3], cfg["spacing_narrow"][:3] = self.dll_epl_wide, self.dll_epl_narrow

    def _initial_loop_state(self, st, cfg):
        st["lock_state"] = int(LoopLockState.PULL_IN)
        st["fll_bw"], st["pll_bw"] = cfg["fll_bw_pullin"], cfg["pll_bw_wide"]

    @property
    def track_correlatorsSpacing(self):
        return self.dll_epl_narrow if self._bank.state["spacing_sel"][self._row] else self.dll_epl_wide

    # thresholds / bandwidths under the reference's names (read-only views of the configuration row)
    dllLockThreshold = property(lambda self: float(self._bank.cfg["dll_threshold"][self._row]))
    track_dll_tau1 = property(lambda self: float(self._bank.cfg["dll_tau1"][self._row]))
    track_dll_tau2 = property(lambda self: float(self._bank.cfg["dll_tau2"][self._row]))
    track_dll_pdi = property(lambda self: float(self._bank.cfg["dll_pdi"][self._row]))
