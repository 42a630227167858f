"""The closed-loop kernel divides by run constants (fs, GPS 2*pi, the epoch duration) with a precomputed
reciprocal and two FMA corrections (track.hip: div_by) and claims the SAME bits as an IEEE division.  This
checks the identity on the CPU -- same IEEE-754 double arithmetic, hardware or correctly rounded software fma --
for the denominators a receiver meets, on 10^6 random numerators each here (the committed C program runs 4*10^7
per denominator when called without an argument: 0 mismatches in 6*10^8)."""
import os
import subprocess

from conftest import REPO


def test_division_by_a_run_constant_is_bit_exact(tmp_path):
    exe = tmp_path / "div_by_constant"
    subprocess.check_call(["gcc", "-O2", "-o", str(exe), os.path.join(REPO, "tests", "csrc", "div_by_constant.c"), "-lm"])
    out = subprocess.check_output([str(exe), "1000000"], text=True)
    lines = [l for l in out.splitlines() if l.startswith("b=")]
    assert len(lines) >= 12
    for l in lines:
        assert "5-op mismatches 0 " in l, l
