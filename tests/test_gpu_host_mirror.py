"""The C++ host-side mirror of the reference interface (include/mgym.hpp) run through the reference's
own unit tests (tests/native/host_mirror_test.cpp); needs a GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_passes_reference_unit_tests():
    out = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_mirror_test")
    lib = os.path.join(ROOT, "modurl_gym_amd")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "host_mirror_test.cpp"),
                    f"-L{lib}", "-lmgym", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all reference unit tests passed" in r.stdout
