import sys, os, argparse, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from sydr_amd._lib import LoopCfg, TrackState
from sydr_amd.engine import Engine
import sydr_amd.engine as E
# monkeypatch the leg to print per-channel outcome
orig = bench.closed_loop_multignss_leg
def leg(eng, gps_first, e1_first, fs, taps, n_epochs):
    real = eng.track_closed_loop_ex
    def wrapped(states, cfgs, n, want_traj=False, **kw):
        end, traj, bits, done = real(states, cfgs, n, want_traj=True, **kw)
        if n > 10:
            for c,(e,s) in enumerate(zip(end, states)):
                tr = traj[c]
                mag = np.hypot(tr["corr"][:,4], tr["corr"][:,5])
                print(c, "done", done[c], "dcar %.1f" % (e.carrier_hz - s.carrier_hz), "lock", tr["lock_state"][-1], "flags", tr["track_flags"][-1], "P first/last %.3g %.3g" % (mag[0], mag[-1]), "code_hz-nom %.3f" % (e.code_hz - s.code_hz))
        return end, traj, bits, done
    eng.track_closed_loop_ex = wrapped
    return orig(eng, gps_first, e1_first, fs, taps, n_epochs)
bench.closed_loop_multignss_leg = leg
import torch
args = argparse.Namespace(stream_seconds=4.0, steps=2, warmup=1, no_closed_loop=False)
r = bench.multignss_workload(args, 0, 0, 1, torch, None, emit=False)
print(json.dumps(r["closed_loop"]))
