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


def test_gpus_must_equal_world_size_for_every_value():
    """`--gpus 8` inside a one-rank environment (WORLD_SIZE=1) is refused before anything is imported or measured: an
    `n_gpus: 1` line for a `--gpus 8` command cannot be printed."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                         timeout=120, cwd=REPO)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True,
                         timeout=120, cwd=REPO)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()


def test_without_a_launcher_the_process_becomes_one(monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE: the parent starts `torch.distributed.run --nproc-per-node 4 bench.py
    <same arguments>` as a child and returns its exit code, without importing torch itself."""
    import subprocess
    import sys
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    had_torch = "torch" in sys.modules
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7                                   # the launcher's (= the worst rank's) exit code
    else:
        raise AssertionError("main() returned")
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert had_torch or "torch" not in sys.modules           # the launcher itself never imports torch


def test_flat_scalars_reach_the_parts_of_the_line_the_driver_keeps():
    import bench
    d = _latest_line()
    flat = bench.flat_scalars(json.loads(json.dumps(d)))
    r, c = flat["roofline"], flat["cpu_baseline"]
    for key in ("acq_ms_per_prn", "acq_kernel_ms_32_prn", "acq_frac", "acq_ms_per_prn_cold", "fp64_frac", "copy_peak_GBps",
                "single_use_x_realtime", "closed_loop_us_per_epoch", "closed_loop_dense_us_per_epoch", "per_tick_x_realtime"):
        assert isinstance(r[key], float), key
    assert r["acq_ms_per_prn"] == d["acquisition"]["value"] and r["acq_frac"] == d["acquisition"]["roofline"]["frac"]
    assert r["closed_loop_us_per_epoch"] == d["closed_loop"]["us_per_epoch"]
    assert c["reference_c_value"] == d["cpu_baseline_reference_c"]["value"] and c["mp_cores"] == d["cpu_baseline_mp"]["cores"]
    assert isinstance(d["roofline"]["fp64_vector"], dict)    # the nested forms stay
