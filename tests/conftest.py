import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (the checker) is built HERE, before any test can have touched the GPU: _oracle.lib() never builds,
    # because a fork + exec of `make` from a process that has initialised HIP takes a GPU box down.
    import _oracle

    _oracle.build()


@pytest.fixture(scope="session")
def oracle():
    import _oracle

    _oracle.lib()
    return _oracle


def _run_gpu_dist_rehearsal(session):
    """tests/test_gpu_dist.py: RowShardedZip + HipBackend + real collectives, two ranks over gloo, both on GPU 0.  Run
    HERE, before this process has touched the GPU (a GPU-initialised process must not fork + exec on the GPU boxes)."""
    if not any("test_gpu_dist" in item.nodeid for item in session.items):
        return
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = {"returncode": None, "stdout": "", "stderr": ""}
    try:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), "12", "--hip"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
        out.update(returncode=res.returncode, stdout=res.stdout, stderr=res.stderr)
    except Exception as e:  # noqa: BLE001
        out.update(returncode=-1, stderr=repr(e))
    session.config._zinc_gpu_dist = out


def _gpu_visible():
    """Counting devices does not initialise the GPU on this image (torch.cuda.is_available() would)."""
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_finish(session):
    """tests/test_gpu_cpp_mirror.py runs a separate C++ program.  It is built and run HERE, before any test has touched
    the GPU: a process that has initialised HIP must not exec another program on the GPU boxes, and a forked child
    that execs counts.  The test itself only looks at the stored result."""
    if not _gpu_visible():  # GPU tests collected on a box without a device (no -m filter): nothing to rehearse
        return
    _run_gpu_dist_rehearsal(session)
    if not any("test_gpu_cpp_mirror" in item.nodeid for item in session.items):
        return
    import subprocess
    import tempfile

    lib = os.path.join(ROOT, "zinc_amd", "lib")
    out = {"returncode": None, "stdout": "", "stderr": ""}
    try:
        exe = os.path.join(tempfile.mkdtemp(prefix="zinc_cpp_"), "zinc_prover_test")
        cc = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT}/include", f"-I{ROOT}/zinc_amd/host",
                             os.path.join(ROOT, "tests", "cpp", "zinc_prover_test.cpp"), "-o", exe, f"-L{lib}", "-lzinc_zip",
                             "-lzip_hip", f"-Wl,-rpath,{lib}"], capture_output=True, text=True, timeout=300)
        if cc.returncode != 0:
            out.update(returncode=cc.returncode, stderr="g++ failed:\n" + cc.stderr)
        else:
            res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
            out.update(returncode=res.returncode, stdout=res.stdout, stderr=res.stderr)
    except Exception as e:  # noqa: BLE001
        out.update(returncode=-1, stderr=repr(e))
    session.config._zinc_cpp_mirror = out
