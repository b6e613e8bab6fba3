"""pmath.h on the device against pmath.h on the host, function by function and bit by bit (tests/native/gpu_pmath_check.hip):
4 M operands each for the short division sequences (against the IEEE division, over the operand ranges their call sites
guarantee and a wide plain range), n / 1e6, exp (and its plain-range forms, wave by wave), log, the controller's coarse log and the reciprocal square root."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_device_math_equals_host_math_bitwise(tmp_path):
    exe = tmp_path / "gpu_pmath_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-I", str(ROOT / "picles_amd" / "csrc"), str(ROOT / "tests" / "native" / "gpu_pmath_check.hip"), "-o", str(exe)],
                   check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 mismatches") == 11, r.stdout


@pytest.mark.gpu
def test_structured_jacobian_device_equals_host_bitwise(tmp_path):
    """physics.h rhs3_jac_plain (the Jacobian of the Rosenbrock23 attempt for a plain particle) and rhs3 on the device against the
    same functions compiled for the host, entry by entry: ordinary seas and slow young seas just above the speed floor, with and
    without the per-node metric term and the wind's slope (tests/native/gpu_jac_check.hip)"""
    exe = tmp_path / "gpu_jac_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-I", str(ROOT / "picles_amd" / "csrc"), str(ROOT / "tests" / "native" / "gpu_jac_check.hip"), "-o", str(exe)],
                   check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 mismatches") == 4, r.stdout
