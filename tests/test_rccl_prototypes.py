"""The librccl entry points zip_mctx binds with dlsym (zinc_amd/csrc/rccl_dyn.h) against RCCL's own header: compile-only,
no GPU, no RCCL call.  Round-3 verdict: the first real 8-GPU run must not be the first time those signatures meet librccl."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM_INC = "/opt/rocm/include"


@pytest.mark.skipif(not os.path.exists(os.path.join(ROCM_INC, "rccl", "rccl.h")), reason="no rccl/rccl.h on this machine")
def test_hand_declared_rccl_prototypes_match_the_header(tmp_path):
    cxx = shutil.which("g++") or shutil.which("hipcc")
    assert cxx, "no C++ compiler"
    cmd = [cxx, "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", f"-I{ROCM_INC}",
           f"-I{os.path.join(ROOT, 'zinc_amd', 'csrc')}", os.path.join(ROOT, "tests", "native", "rccl_prototypes_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_fallback_declarations_compile_without_the_header():
    """rccl_dyn.h with ZIP_RCCL_HAND_DECLARED_ONLY (what a toolchain without rccl.h sees) is self-contained."""
    cxx = shutil.which("g++") or shutil.which("hipcc")
    src = '#include "rccl_dyn.h"\nstatic_assert(rccl::kUint8 == 1, "");\nint main() { rccl::all_gather_t f = nullptr; return f != nullptr; }\n'
    cmd = [cxx, "-std=c++17", "-fsyntax-only", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-DZIP_RCCL_HAND_DECLARED_ONLY", f"-I{ROCM_INC}",
           f"-I{os.path.join(ROOT, 'zinc_amd', 'csrc')}", "-"]
    r = subprocess.run(cmd, input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
