#!/usr/bin/env python3
"""Host-delivered proofs at full size: zip_open into one host buffer vs zip_open_stream (GPU box)."""
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
    print(f"zip_open -> pageable host buffer      : {(t1 - t0) * 1e3:8.1f} ms  ({total / (t1 - t0) / 1e9:.1f} GB/s)")
ref = host.copy()
for chunk in (16 << 20, 64 << 20, 256 << 20):
    for rep in range(2):
        got = np.empty(total, dtype=np.uint8)
        pos = [0]

        def sink(mv):
            n = len(mv)
            got[pos[0]:pos[0] + n] = np.frombuffer(mv, dtype=np.uint8)
            pos[0] += n

        t0 = time.perf_counter()
        com.open_stream(evals, coeffs, cols, q0, zf, sink, chunk_bytes=chunk)
        t1 = time.perf_counter()
        n = [0]
        t2 = time.perf_counter()
        com.open_stream(evals, coeffs, cols, q0, zf, lambda mv: n.__setitem__(0, n[0] + len(mv)), chunk_bytes=chunk)
        t3 = time.perf_counter()
    assert np.array_equal(got, ref) and n[0] == total
    print(f"zip_open_stream chunk {chunk >> 20:4d} MiB: copy-out sink {(t1 - t0) * 1e3:8.1f} ms ({total / (t1 - t0) / 1e9:.1f} GB/s), "
          f"discarding sink {(t3 - t2) * 1e3:8.1f} ms ({total / (t3 - t2) / 1e9:.1f} GB/s)")
