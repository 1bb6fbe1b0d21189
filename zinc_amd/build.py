"""Builds the native pieces in-tree (so the .so files travel to the GPU box):

  zinc_amd/lib/libzip_hip.so   HIP kernels + C ABI (include/zip_hip.h), gfx950 only
  zinc_amd/lib/libzinc_zip.so  C++ host mirror of zinc::zip (RaaCode, PcsTranscript, ...)

hipcc cross-compiles without a GPU.  Rebuilds only when a source is newer.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return sorted(out)


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def build_hip(force=False, verbose=False, extra=()):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libzip_hip.so")
    deps = _sources(CSRC, (".hip", ".cuh", ".h", ".inc")) + _sources(INCLUDE, (".h",))
    if force or _newer(target, deps):
        cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-shared", "-fPIC",
               "-Wall", "-Wno-unused-function", *extra, "-o", target, os.path.join(CSRC, "zip_hip.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return target


def build_host(force=False, verbose=False):
    srcs = _sources(HOST, (".cpp",))
    if not srcs:
        return None
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libzinc_zip.so")
    deps = srcs + _sources(HOST, (".hpp", ".h")) + _sources(INCLUDE, (".h",))
    if force or _newer(target, deps):
        build_hip(force=False, verbose=verbose)  # libzinc_zip.so links against the HIP library
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", f"-I{INCLUDE}", f"-I{HOST}",
               "-o", target, *srcs, f"-L{LIB}", "-lzip_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return target


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
