"""the host halves that parse untrusted input or are shared with the device run clean under the address and
undefined-behaviour sanitizers -- `make -C sydr_amd/csrc check-sanitize` (CPU only; the GPU pool refuses sanitizer runs):
the per-item check and per-item setups of sdr_epl_plan_create fed 600 000 hostile items (tests/csrc/fuzz_items.hip), the
chip-geometry dump of test_plan_geometry.py and the exact-division identity of test_div_by_constant.py."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_halves_are_clean_under_asan_and_ubsan(tmp_path):
    out = subprocess.run(["make", "-C", os.path.join(REPO, "sydr_amd", "csrc"), "check-sanitize", f"SAN_DIR={tmp_path}"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "check-sanitize: clean" in out.stdout
    assert "runtime error" not in out.stdout + out.stderr and "AddressSanitizer" not in out.stdout + out.stderr
