"""The C++ host mirror used as a C++ library (not through the C facade): tests/cpp/zinc_prover_test.cpp restates
src/zinc/tests.rs with zinc::ZincProver / zinc::ZincVerifier and is built with g++ and run here."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_zinc_tests_rs_through_the_cpp_classes(tmp_path):
    from zinc_amd import build, cabi

    build.build_all()
    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    exe = tmp_path / "zinc_prover_test"
    lib = os.path.join(ROOT, "zinc_amd", "lib")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT}/include", f"-I{ROOT}/zinc_amd/host",
                    os.path.join(ROOT, "tests", "cpp", "zinc_prover_test.cpp"), "-o", str(exe), f"-L{lib}", "-lzinc_zip",
                    "-lzip_hip", f"-Wl,-rpath,{lib}"], check=True)
    # the program is its own process: it initialises HIP itself (nothing is exec'ed from this one)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
