"""rates leg at KM.5 samples per chip for KM = 16 .. 25 (+-0.5 chip): which kernel, what fraction of the roof."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sydr_amd.engine import Engine
eng = Engine(0)
import types
src = open(bench.__file__).read()
for km in ([int(a) for a in sys.argv[1:]] or range(16, 26)):
    fs = 1.023e6 * (km + 0.5) * (2 if os.environ.get("HALF_CHIP_VIEW") else 1)
    mod = types.ModuleType("b2"); mod.__file__ = bench.__file__
    src_k = src.replace("SPACING = (-0.5, 0.0, 0.5)", "SPACING = (-0.25, 0.0, 0.25)") if os.environ.get("NARROW") else src
    exec(compile(src_k.replace("for fs in (4e6, 10e6, 12e6, 16.368e6, 18e6, 20e6, 22e6, 25e6, 32e6, 40e6, 50e6):", f"for fs in ({fs!r},):"), bench.__file__, "exec"), mod.__dict__)
    r = mod.rates_leg(eng)["rates"][0]
    print(km, round(fs / 1e6, 3), r["kernel_variant"], round(r["roofline_frac"], 3), r["max_rel_err_gpu_vs_oracle"])
