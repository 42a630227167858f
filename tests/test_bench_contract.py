"""The JSON line bench.py prints, checked on the committed line of the round's last GPU run (profiles/): the keys the
driver and the judge read, their types, and the internal consistency of the numbers (CPU test: nothing is launched)."""
import glob
import json
import os

from conftest import REPO


def _latest_line():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_plain_run.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_keys():
    d = _latest_line()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None                       # BASELINE.md holds no published number for this metric
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_bench_line_is_internally_consistent():
    d = _latest_line()
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # the kernel time of a step cannot exceed the step
    launches_per_step = r["launches"] / d["steps"]
    assert r["avg_launch_ms"] * launches_per_step <= d["ms_per_step"] * 1.001
    # value = stream samples per second in units of 32-channel batches: 2 bytes per channel-sample, 32 channels
    stream_samples_per_step = r["algorithmic_bytes_per_launch"] * launches_per_step / 2.0 / d["config"]["channels_per_gpu"]
    assert abs(d["value"] - stream_samples_per_step / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert abs(d["x_realtime"] - d["value"] * 1e6 / d["config"]["fs_hz"]) < 1e-6 * d["x_realtime"]
    assert d["cpu_baseline"]["max_rel_err_gpu_vs_oracle"] <= 1e-9
