#!/usr/bin/env python3
"""Wall time of the device sumcheck rounds (product of two MLEs, degree 2 = ZincProver::sumcheck_2)
with the tables resident in HBM, and of the whole prove_as_subprotocol through the host mirror."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import cabi, pcs  # noqa: E402
import torch  # noqa: E402

for nv in [int(a) for a in sys.argv[1:]] or [20, 24]:
    fl, K, degree = 4, 2, 2
    n = 1 << nv
    rng = np.random.default_rng(1)
    mles = rng.integers(0, 1 << 62, size=(K, n, fl), dtype=np.uint64)
    mles[..., fl - 1] >>= np.uint64(6)
    dev = [torch.from_numpy(mles[k].view(np.int64)).cuda() for k in range(K)]
    zf = cabi.make_field(bench.BENCH_MODULUS, fl)
    r = np.array([3, 1, 4, 1], dtype=np.uint64)
    for rep in range(3):
        sc = cabi.Sumcheck(dev, nv, degree, zf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per = []
        for i in range(nv):
            t1 = time.perf_counter()
            sc.round(None if i == 0 else r)
            per.append(time.perf_counter() - t1)
        dt = time.perf_counter() - t0
        sc.free()
    bytes_moved = K * n * fl * 8 * (1 + 1.5)  # round 1 reads n; round 2 reads n, writes n/2; then halves
    print(f"2^{nv}: {nv} rounds {dt * 1e3:.2f} ms (first three: {', '.join(f'{p * 1e3:.3f}' for p in per[:3])} ms; "
          f"last: {per[-1] * 1e6:.0f} us); >= {bytes_moved / 1e9:.2f} GB moved")
    if nv <= 22:
        field = pcs.FieldConfig(bench.BENCH_MODULUS, fl)
        for rep in range(2):
            t = pcs.KeccakTranscript()
            t0 = time.perf_counter()
            pcs.sumcheck_prove_product(t, mles, degree, field)
            dt = time.perf_counter() - t0
        print(f"2^{nv}: prove_as_subprotocol through the host mirror (tables uploaded from the host, Keccak on the host): {dt * 1e3:.2f} ms")
