"""bench.py's N > 1 path rehearsed on one GPU: two ranks (gloo, both on device 0) generate the SAME 64-satellite
stream, each tracks its shard of 32 channels, and rank 0 reports the job.  The driver's own 8-GPU run uses RCCL and
one device per rank; what is checked here is everything else -- the sharding, the reductions and the JSON contract."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal_reports_one_job():
    env = dict(os.environ, SYDR_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--stream-seconds", "2.5"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)   # (the first `import torch` on a fresh box pages the image in: minutes)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1                                  # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["config"]["channels_total"] == 64 and r["config"]["channels_per_gpu"] == 32
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["roofline"]["launches"] == 3   # (one launch per pass over the stream)
    # the job's value counts both ranks' channel-samples: twice what one rank's stream rate alone would give
    assert r["value"] == pytest.approx(2.0 * r["x_realtime"] * 25.0, rel=2e-3)
