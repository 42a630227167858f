"""GPU parity of the SerialSearch acquisition path (sydr/dsp/acquisition.py:119-193) against vectors
captured from the reference: chip-shift / Doppler-bin indices exact, map values to 1e-9 relative."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import sydr_oracle as orc
from sydr_amd.engine import FMT_CI8

pytestmark = pytest.mark.gpu


def test_serial_search_golden(engine):
    g = load_golden("g9_serial.npz")
    fs, n, prn, rng_hz, step = g["params"]
    n = int(n)
    engine.iq_alloc(2 * n, FMT_CI8)
    engine.iq_upload(g["iq"], 0)
    engine.code_slots(2)
    engine.load_gps_code(0, int(prn))
    engine.load_gps_code(1, 5)
    pb, pc, pr, cmap = engine.serial_search([0, 1], 0, fs, rng_hz, step, want_map=True)
    np.testing.assert_allclose(cmap[0], g["map0"], rtol=0, atol=1e-9 * g["map0"].max())
    assert [int(pb[0]), int(pc[0])] == list(g["peak"]) and pr[0] == pytest.approx(float(g["ratio"]), rel=1e-9)
    # second millisecond alone, then the two added (channel_l1ca_kaplan_ss.py:14-21)
    _, _, _, cmap1 = engine.serial_search([0], n, fs, rng_hz, step, want_map=True)
    np.testing.assert_allclose(cmap1[0], g["map1"], rtol=0, atol=1e-9 * g["map1"].max())
    pb2, pc2, pr2, both = engine.serial_search([0], 0, fs, rng_hz, step, noncoh=2, want_map=True)
    np.testing.assert_allclose(both[0], g["map0"] + g["map1"], rtol=0, atol=2e-9 * g["map0"].max())
    assert [int(pb2[0]), int(pc2[0])] == list(g["peak_sum"]) and pr2[0] == pytest.approx(float(g["ratio_sum"]), rel=1e-9)
    # the absent PRN's map matches the oracle too
    rf = orc.iq_to_complex(g["iq"])[:n].reshape(1, -1)
    ref = orc.serial_search(rf, orc.gold_code(5), rng_hz, step, fs, n)
    np.testing.assert_allclose(cmap[1], ref, rtol=0, atol=1e-9 * ref.max())


def test_two_peak_compare_ss_edge_cases(engine):
    g = load_golden("g9_serial.npz")
    for m, idx, ratio in zip(g["edge_maps"], g["edge_idx"], g["edge_ratio"]):
        got_idx, got_ratio = engine.two_peak_compare_ss(m)
        assert got_idx == list(idx) and got_ratio == ratio


def test_serial_search_dropins_and_plugin_seams(engine):
    from sydr_amd.dsp.acquisition import SerialSearch, TwoCorrelationPeakComparison_SS
    g = load_golden("g9_serial.npz")
    fs, n, prn, rng_hz, step = g["params"]
    rf = orc.iq_to_complex(g["iq"])[:int(n)].reshape(1, -1)
    cmap = SerialSearch(rfdata=rf, code=orc.gold_code(int(prn)), dopplerRange=rng_hz, dopplerStep=step,
                        samplingFrequency=fs, samplesPerCode=int(n))
    np.testing.assert_allclose(cmap, g["map0"], rtol=0, atol=1e-9 * g["map0"].max())
    idx, ratio = TwoCorrelationPeakComparison_SS(cmap)
    assert idx == list(g["peak"]) and ratio == pytest.approx(float(g["ratio"]), rel=1e-9)


def test_serial_search_plugin_on_gpu(engine):
    from test_host_layer import KAPLAN_INI, channel_config, drive, rf_signal
    from sydr_amd.channel.l1ca_kaplan_ss import ChannelL1CA_Kaplan_SS
    from sydr_amd.channel.manager import ChannelManager
    from sydr_amd.utils.enumerations import ChannelMessage
    g = load_golden("g9_serial.npz")
    fs, spms = 4e6, 4000
    raw = orc.synth_iq(fs, 40 * spms, [dict(prn=7, doppler=1750.0, code_phase=300.25, phase=0.1, amp=8.0)], 20.0, 20260001)
    cfg = channel_config(KAPLAN_INI)
    cfg["ACQUISITION"].update(doppler_range="2000", doppler_steps="250", non_coherent_integration="2")
    mgr = ChannelManager(rf_signal(fs), engine=engine)
    mgr.addChannel(ChannelL1CA_Kaplan_SS, cfg, 1)
    mgr.requestTracking(7)
    ticks = drive(mgr, raw, spms, 40)
    acq = [p for t in ticks for p in t if p["type"] is ChannelMessage.ACQUISITION_UPDATE][0]
    trk = [p for t in ticks for p in t if p["type"] is ChannelMessage.TRACKING_UPDATE]
    ref = g["ss_acq"]
    assert (acq["frequency_idx"], acq["code_idx"], acq["carrierFrequency"], acq["codeOffset"]) == \
        (int(ref[0]), int(ref[1]), ref[3], int(ref[4]))
    assert acq["peak_ratio"] == pytest.approx(ref[2], rel=1e-9) and len(trk) == len(g["ss_epochs"])
    got = np.array([[p["i_early"], p["q_early"], p["i_prompt"], p["q_prompt"], p["i_late"], p["q_late"],
                     p["carrier_frequency"], p["code_frequency"]] for p in trk])
    want = g["ss_epochs"]
    scale = np.maximum(np.abs(want), 1.0)
    scale[:, :6] = np.maximum(np.repeat(np.hypot(want[:, 0:6:2], want[:, 1:6:2]), 2, axis=1), 1.0)
    assert np.all(np.abs(got - want) <= 1e-9 * scale)
