"""csrc/libm_bits.h -- exp for non-positive arguments, atan2, division by a fixed divisor in fewer instructions --
returns the bits of the device library's functions and of the compiler's division: the checker built from
tools/exp_check.hip compares them on the GPU over 2^31 arguments each, every binade and the special values."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libm_bits_match_the_device_library_bit_for_bit():
    exe = os.path.join(ROOT, "slam_amd", "exp_check")
    assert os.path.exists(exe), "slam_amd/exp_check is missing: make -C slam_amd/csrc exp_check"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if "differing" in l]
    assert len(lines) == 2 and all("differing: 0" in l for l in lines), out.stdout
