"""Unit checks of classpp_public_amd/host/cpt_numerics.hpp (hinted spline look-ups, row-major spline sweeps, the Dormand-Prince
integrator's stage hand-over): tests/numerics_check.cpp, compiled here with g++ and run on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_numerical_building_blocks(tmp_path):
    exe = str(tmp_path / "numerics_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "numerics_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "numerics ok" in r.stdout
