#!/usr/bin/env python3
"""Track an IQ recording with the reference's own configuration files.

    python examples/run_file.py receiver.ini [--ms 2000] [--block 80 | --read-ahead 50] [--csv out.csv]

`receiver.ini` is the reference's receiver configuration (config/receiver.ini: [DEFAULT] nb_channels /
ms_to_process, [RFSIGNAL], [SATELLITES] include_prn, [CHANNELS] gps_l1ca = <channel ini>).  What the reference's
Receiver does around the hot path for the first stage of processing -- read the file millisecond by millisecond, give
each requested PRN a channel, acquire, track -- is done here with the drop-in ChannelManager; once every channel
tracks, blocks of `--block` ms go through the closed-loop kernel (`ChannelManager.runBlock`); with `--read-ahead N`
the loop stays the reference's own (one millisecond per iteration: `addNewRFData(getMilliseconds(1)); run()`) and the
manager tracks N ms ahead behind it (`enableReadAhead`).  Subframes the channels decode are printed as they complete
(DECODING_UPDATE; needs a decoder: by default the reference's own, where `sydr` is importable).  Navigation,
measurements, database and report stay the reference's business (feed them the packets this script prints / writes)."""
import argparse
import configparser
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sydr_amd.channel.l1ca_borre import ChannelL1CA                # noqa: E402
from sydr_amd.channel.l1ca_kaplan import ChannelL1CA_Kaplan        # noqa: E402
from sydr_amd.channel.manager import ChannelManager                # noqa: E402
from sydr_amd.signal.iqsource import RFSignal                      # noqa: E402
from sydr_amd.utils.enumerations import ChannelMessage, ChannelState  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("receiver_ini")
    ap.add_argument("--ms", type=int, default=None, help="milliseconds to process (default: ms_to_process)")
    ap.add_argument("--block", type=int, default=80, help="epochs per closed-loop block once all channels track (0: per-tick only)")
    ap.add_argument("--read-ahead", type=int, default=0, help="keep the per-millisecond loop and let the manager track this many ms ahead (overrides --block)")
    ap.add_argument("--csv", default=None, help="write one line per tracking epoch")
    args = ap.parse_args(argv)

    rcfg = configparser.ConfigParser()
    rcfg.read(args.receiver_ini)
    base = os.path.dirname(os.path.abspath(args.receiver_ini))
    chan_ini = rcfg["CHANNELS"]["gps_l1ca"]
    chan_ini = chan_ini if os.path.isabs(chan_ini) else os.path.normpath(os.path.join(base, chan_ini))
    ccfg = configparser.ConfigParser()
    ccfg.read(chan_ini)
    plugin = ChannelL1CA_Kaplan if "correlator_epl_wide" in ccfg["TRACKING"] else ChannelL1CA
    rf = RFSignal(rcfg["RFSIGNAL"])
    prns = [int(p) for p in rcfg["SATELLITES"]["include_prn"].split(",") if p.strip()]
    ms_total = args.ms or int(rcfg["DEFAULT"]["ms_to_process"])

    mgr = ChannelManager(rf, keepCorrelationMap=False)
    mgr.addChannel(plugin, ccfg, max(len(prns), int(rcfg["DEFAULT"].get("nb_channels", len(prns)))))
    for p in prns:
        mgr.requestTracking(p)
    if args.read_ahead:
        mgr.enableReadAhead(args.read_ahead)
        args.block = 0
    out = open(args.csv, "w") if args.csv else None
    if out:
        out.write("ms,cid,i_prompt,q_prompt,carrier_frequency,code_frequency,cn0,lock_state\n")
    ring_ms = mgr.sharedBuffer.maxSize // rf.samplesPerMs
    t0, ms, n_track = time.perf_counter(), 0, 0

    def emit(packets, ms_now):
        nonlocal n_track
        for p in packets:
            if p["type"] is ChannelMessage.ACQUISITION_UPDATE:
                print(f"[{ms_now:6d} ms] channel {p['cid']}: acquisition bin {p['frequency_idx']} code {p['code_idx']} "
                      f"ratio {p['peak_ratio']:.2f} carrier {p['carrierFrequency']:+.1f} Hz")
            elif p["type"] is ChannelMessage.DECODING_UPDATE:
                print(f"[{ms_now:6d} ms] channel {p['cid']}: subframe {p['subframe_id']} decoded, TOW {p['tow']}")
            elif p["type"] is ChannelMessage.TRACKING_UPDATE:
                n_track += 1
                if out:
                    out.write(f"{ms_now},{p['cid']},{p['i_prompt']:.3f},{p['q_prompt']:.3f},{p['carrier_frequency']:.4f},"
                              f"{p['code_frequency']:.4f},{p['cn0']:.3f},{int(p['lock_state'])}\n")

    while ms < ms_total:
        all_tracking = all(ch.channelState is ChannelState.TRACKING for ch in mgr.channels.values()
                           if ch.channelState is not ChannelState.IDLE)
        if args.block and all_tracking and ms + args.block <= ms_total and args.block < ring_ms - 2:
            for _ in range(args.block):                       # fill the ring ahead of the channels, then one launch
                mgr.addNewRFData(rf.getMilliseconds(1))
            emit(mgr.runBlock(args.block), ms + args.block)            # (as many whole epochs as the ring holds)
            ms += args.block
        else:
            mgr.addNewRFData(rf.getMilliseconds(1))
            emit(mgr.run(), ms + 1)
            ms += 1
    dt = time.perf_counter() - t0
    for ch in mgr.channels.values():
        if ch.channelState is not ChannelState.IDLE:
            bits = "".join(str(int(b)) for b in getattr(ch, "navBits", [])[:40])
            print(f"channel {ch.channelID} G{ch.satelliteID:02d}: state {ch.channelState.name}, carrier {ch.carrierFrequency:+.2f} Hz, bits {bits}")
    print(f"{ms} ms of signal, {n_track} tracking epochs in {dt:.2f} s = {ms * 1e-3 / dt:.2f}x real time")
    if out:
        out.close()


if __name__ == "__main__":
    main()
