"""The C++ host mirror used as a C++ library (not through the C facade): tests/cpp/zinc_prover_test.cpp restates
src/zinc/tests.rs with zinc::ZincProver / zinc::ZincVerifier / PreparedCcs.  It is a program of its own, built with g++
and run by tests/conftest.py before the first GPU test (see there); this test checks its verdict."""
import pytest

pytestmark = pytest.mark.gpu


def test_zinc_tests_rs_through_the_cpp_classes(request):
    res = getattr(request.config, "_zinc_cpp_mirror", None)
    assert res is not None, "conftest.py did not run the C++ program"
    assert res["returncode"] == 0 and res["stdout"].strip().endswith("OK"), res["stdout"] + res["stderr"]
