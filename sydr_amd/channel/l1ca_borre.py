"""GPS L1 C/A channel, Borre-style loops (DLL NNEML + Costas PLL through second-order Borre filters) -- the
reference's default plugin (sydr/channel/channel_l1ca_borre.py, selected at receiver_gps_l1ca.py:17) with the
tracking state on the GPU.

The per-epoch arithmetic of borre:333-451 runs in `track_kernel` (sydr_amd/csrc/track.hip, loop_kind 0; NumPy's pi
in the NCO); this file maps config/channels/channel_GPS_L1CA_borre.ini onto `sdr_loop_cfg` and the
plugin's attribute names onto the device state."""
from __future__ import annotations

from .bank import KIND_BORRE
from .tracked import DeviceTrackedChannel


class ChannelL1CA(DeviceTrackedChannel):
    LOOP_KIND = KIND_BORRE
    DECODER_PLUGIN = "borre"          # the default decoder drives borre:455-579 (packet `tow` = the HOW's value)
    MIN_CONVERGENCE_TIME = 100        # epochs before bit sync is looked for (borre:384-391); fixed in the kernel

    CFG_KEYS = {"dll_pdi": "dll_pdi", "pll_pdi": "pll_pdi"}
    FILTERS = (("dll", "dll"), ("pll", "pll"))

    STATE_VIEW = {
        "carrierFrequency": ("carrier_hz", float), "codeFrequency": ("code_hz", float),
        "NCO_remainingCarrier": ("rem_carrier", float), "NCO_remainingCode": ("rem_code", float),
        "codeStep": ("code_step", float), "track_requiredSamples": ("n_samples", int),
        "codeCounter": ("code_counter", int),
        "NCO_codeError": ("dll_mem", float), "NCO_carrierError": ("pll_mem", float),
        "iPrompt": ("i_prompt_prev", float), "qPrompt": ("q_prompt_prev", float),
        "navPromptSum": ("nav_prompt_sum", float), "navPromptSumCounter": ("nav_sum_counter", int),
    }
    RECORD_VIEW = {"NCO_code": "dll", "NCO_carrier": "pll"}
    fll = 0.0                         # the plugin never runs its FLL (borre:440)

    def _configure_taps(self, configuration, cfg):
        self.track_correlatorsSpacing = [float(configuration[k]) for k in
                                         ("correlator_early", "correlator_prompt", "correlator_late")]
        cfg["spacing_wide"][:3] = cfg["spacing_narrow"][:3] = self.track_correlatorsSpacing

    def postAcquisitionUpdate(self, acqIndices):
        super().postAcquisitionUpdate(acqIndices)
        self.initialFrequency = self.carrierFrequency

    # the NCO state under the names the seams mixin reads
    def _nco_rem_carrier(self):
        return self.NCO_remainingCarrier

    def _nco_rem_code(self):
        return self.NCO_remainingCode

    track_dll_tau1 = property(lambda self: float(self._bank.cfg["dll_tau1"][self._row]))
    track_dll_tau2 = property(lambda self: float(self._bank.cfg["dll_tau2"][self._row]))
    track_pll_tau1 = property(lambda self: float(self._bank.cfg["pll_tau1"][self._row]))
    track_pll_tau2 = property(lambda self: float(self._bank.cfg["pll_tau2"][self._row]))
    track_dll_pdi = property(lambda self: float(self._bank.cfg["dll_pdi"][self._row]))
    track_pll_pdi = property(lambda self: float(self._bank.cfg["pll_pdi"][self._row]))
