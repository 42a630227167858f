"""rates leg of the bench at one or more front-end rates (MHz): which kernel serves it, what fraction of the roof, (tools/pmc_rate.sh takes
counters of exactly this; a launch holds 2 s x 32 channels = one wave per channel-epoch)."""
import sys, os, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sydr_amd.engine import Engine
eng = Engine(0)
src = open(bench.__file__).read()
for mhz in ([float(a) for a in sys.argv[1:]] or (4.0, 10.0)):
    fs = mhz * 1e6
    mod = types.ModuleType("b2"); mod.__file__ = bench.__file__
    exec(compile(src.replace("for fs in (4e6, 10e6, 12e6, 16.368e6, 18e6, 20e6, 22e6, 25e6, 32e6, 40e6, 50e6):", f"for fs in ({fs!r},):"),
                 bench.__file__, "exec"), mod.__dict__)
    r = mod.rates_leg(eng)["rates"][0]
    print(mhz, r["kernel_variant"], "|", r["correlator"], "| frac", round(r["roofline_frac"], 3), "x_realtime", round(r["x_realtime"], 1),
          "err", r["max_rel_err_gpu_vs_oracle"])
