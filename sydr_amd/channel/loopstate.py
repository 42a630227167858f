"""Bridges between host channel objects and the closed-loop tracking kernel's C structs
(sdr_track_state / sdr_loop_cfg, include/sydr_amd.h): export a channel's NCO + loop state, run
many epochs on the device, import the state back and rebuild the reference's tracking packets."""
from __future__ import annotations

from .._lib import LoopCfg, TrackState
from ..utils.enumerations import ChannelMessage, LoopLockState, TrackingFlags

KIND_BORRE, KIND_KAPLAN = 0, 1


def loop_kind(channel) -> int:
    return KIND_KAPLAN if hasattr(channel, "fll_vel_memory") else KIND_BORRE


def export_cfg(channel) -> LoopCfg:
    cfg = LoopCfg()
    cfg.loop_kind = loop_kind(channel)
    cfg.n_taps = 3
    cfg.fs = channel.rfSignal.samplingFrequency
    cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi = channel.track_dll_tau1, channel.track_dll_tau2, channel.track_dll_pdi
    if cfg.loop_kind == KIND_KAPLAN:
        for t in range(3):
            cfg.spacing_wide[t] = channel.dll_epl_wide[t]
            cfg.spacing_narrow[t] = channel.dll_epl_narrow[t]
        cfg.dll_threshold = channel.dllLockThreshold
        cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow = (channel.fll_bandwidth_pullin, channel.fll_bandwidth_wide,
                                                                 channel.fll_bandwidth_narrow)
        cfg.fll_thr_wide, cfg.fll_thr_narrow = channel.fll_threshold_wide, channel.fll_threshold_narrow
        cfg.pll_bw_wide, cfg.pll_bw_narrow = channel.pll_bandwidth_wide, channel.pll_bandwidth_narrow
        cfg.pll_thr_wide, cfg.pll_thr_narrow = channel.pll_threshold_wide, channel.pll_threshold_narrow
    else:
        for t in range(3):
            cfg.spacing_wide[t] = cfg.spacing_narrow[t] = channel.track_correlatorsSpacing[t]
        cfg.pll_tau1, cfg.pll_tau2, cfg.pll_pdi = channel.track_pll_tau1, channel.track_pll_tau2, channel.track_pll_pdi
    return cfg


def export_state(channel) -> TrackState:
    st = TrackState()
    st.code_slot = int(channel.codeSlot)
    st.n_samples = int(channel.track_requiredSamples)
    st.current_sample = int(channel.currentSample)
    st.carrier_hz = float(channel.carrierFrequency)
    st.code_hz = float(channel.codeFrequency)
    st.code_step = float(channel.codeStep)
    st.code_counter = int(channel.codeCounter)
    st.track_flags = int(channel.trackFlags)
    st.nav_prompt_sum = float(getattr(channel, "navPromptSum", 0.0))
    st.nav_sum_counter = int(getattr(channel, "navPromptSumCounter", 0))
    st.nav_bits_emitted = len(getattr(channel, "navBits", []))
    if loop_kind(channel) == KIND_KAPLAN:
        st.rem_carrier, st.rem_code = float(channel.remainingCarrier), float(channel.remainingCode)
        st.dll_mem, st.pll_mem = float(channel.dllDiscrim), float(channel.fll_vel_memory)
        st.i_prompt_prev, st.q_prompt_prev = float(channel.iPromptPrev), float(channel.qPromptPrev)
        st.fll_lock, st.pll_lock = float(channel.fllLockIndicator), float(channel.pllLockIndicator)
        st.cn0, st.cn0_ratio_acc = float(channel.cn0), float(channel.cn0_PdPnRatio)
        st.fll_bw, st.pll_bw = float(channel.fllBandwidth), float(channel.pllBandwidth)
        st.accum_counter = int(channel.correlatorsAccumCounter)
        st.lock_state = int(channel.loopLockState)
        st.time_in_state = int(channel.timeSinceLastState)
        st.spacing_sel = 1 if channel.track_correlatorsSpacing is channel.dll_epl_narrow else 0
    else:
        st.rem_carrier, st.rem_code = float(channel.NCO_remainingCarrier), float(channel.NCO_remainingCode)
        st.dll_mem, st.pll_mem = float(channel.NCO_codeError), float(channel.NCO_carrierError)
        st.i_prompt_prev, st.q_prompt_prev = float(channel.iPrompt), float(channel.qPrompt)
    return st


def import_state(channel, st: TrackState, epochs: int, last=None):
    """Write the device's end state back into the channel object (ring index re-wrapped)."""
    channel.track_requiredSamples = int(st.n_samples)
    channel.currentSample = int(st.current_sample) % channel.rfBuffer.maxSize
    channel.carrierFrequency = st.carrier_hz
    channel.codeFrequency = st.code_hz
    channel.codeStep = st.code_step
    channel.codeCounter = int(st.code_counter)
    channel.codeSinceTOW += epochs
    channel.navPromptSum, channel.navPromptSumCounter = st.nav_prompt_sum, int(st.nav_sum_counter)
    channel.trackFlags = TrackingFlags(int(st.track_flags)) if int(st.track_flags) in TrackingFlags._value2member_map_ \
        else int(st.track_flags)
    if loop_kind(channel) == KIND_KAPLAN:
        channel.remainingCarrier, channel.remainingCode = st.rem_carrier, st.rem_code
        channel.dllDiscrim, channel.fll_vel_memory = st.dll_mem, st.pll_mem
        channel.iPromptPrev, channel.qPromptPrev = st.i_prompt_prev, st.q_prompt_prev
        channel.fllLockIndicator, channel.pllLockIndicator = st.fll_lock, st.pll_lock
        channel.cn0 = channel.dllLockIndicator = st.cn0
        channel.cn0_PdPnRatio = st.cn0_ratio_acc
        channel.fllBandwidth, channel.pllBandwidth = st.fll_bw, st.pll_bw
        channel.correlatorsAccumCounter = int(st.accum_counter)
        channel.loopLockState = LoopLockState(int(st.lock_state))
        channel.timeSinceLastState = int(st.time_in_state)
        channel.track_correlatorsSpacing = channel.dll_epl_narrow if st.spacing_sel else channel.dll_epl_wide
        if last is not None:
            channel.correlatorsResults[:] = last["corr"][:6]
            channel.fllDiscrim, channel.pllDiscrim = float(last["fll"]), float(last["pll"])
            channel.carrierFrequencyError, channel.codeFrequencyError = float(last["carrier_err"]), float(last["code_err"])
    else:
        channel.NCO_remainingCarrier, channel.NCO_remainingCode = st.rem_carrier, st.rem_code
        channel.NCO_codeError, channel.NCO_carrierError = st.dll_mem, st.pll_mem
        channel.iPrompt, channel.qPrompt = st.i_prompt_prev, st.q_prompt_prev
        if last is not None:
            channel.NCO_code, channel.NCO_carrier = float(last["dll"]), float(last["pll"])


def tracking_packet(channel, rec) -> dict:
    """One TRACKING_UPDATE packet (keys of channel_l1ca_kaplan.py:653-676) from a device epoch record."""
    kaplan = loop_kind(channel) == KIND_KAPLAN
    c = rec["corr"]
    return {"cid": channel.channelID, "type": ChannelMessage.TRACKING_UPDATE,
            "i_early": float(c[0]), "q_early": float(c[1]), "i_prompt": float(c[2]), "q_prompt": float(c[3]),
            "i_late": float(c[4]), "q_late": float(c[5]),
            "carrier_frequency": float(rec["carrier_hz"]), "code_frequency": float(rec["code_hz"]),
            "carrier_frequency_error": float(rec["carrier_err"]), "code_frequency_error": float(rec["code_err"]),
            "cn0": float(rec["cn0"]), "pll_lock": float(rec["pll_lock"]), "fll_lock": float(rec["fll_lock"]),
            "dll": float(rec["dll"]), "pll": float(rec["pll"]), "fll": float(rec["fll"]),
            "lock_state": LoopLockState(int(rec["lock_state"])) if kaplan else 0}
