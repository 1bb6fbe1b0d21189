#!/usr/bin/env python3
"""zip_open into a pageable host buffer (library bounce buffers + host copy) against the same buffer pinned once with
zip_host_register (direct DMA).  GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import cabi  # noqa: E402
from zinc_amd.perm import shuffle_seeded_perm  # noqa: E402
import torch  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
row_len, num_rows, cw = cabi.geometry(nv)
ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
zf = cabi.make_field(bench.BENCH_MODULUS, 4)
coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
com, _ = ctx.commit(evals)
ctx.synchronize()
total = ctx.proof_len(cols.size, 4)
host = np.zeros(total, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    com.open(evals, coeffs, cols, q0, zf, out=host)
    t1 = time.perf_counter()
    print(f"pageable: {(t1 - t0) * 1e3:8.1f} ms  ({total / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
ref = host.copy()
host[:] = 0
t0 = time.perf_counter()
rc = cabi.lib().zip_host_register(host.ctypes.data, host.nbytes)
print(f"zip_host_register rc={rc}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    com.open(evals, coeffs, cols, q0, zf, out=host)
    t1 = time.perf_counter()
    print(f"pinned  : {(t1 - t0) * 1e3:8.1f} ms  ({total / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
assert np.array_equal(host, ref)
cabi.lib().zip_host_unregister(host.ctypes.data)
