#!/usr/bin/env python3
"""Occupancy experiment (GPU box) for sumcheck_round_kernel<4,4,3> -- the R1CS-shaped CCS combination (M0 * M1 - M2) * eq,
degree 3, 210 VGPRs = 2 waves per SIMD: would MORE waves help?  Asked the other way round: ZIP_HIP_SUMCHECK_LDS_PAD adds
dynamic LDS so that only ONE 256-thread workgroup fits a CU (1 wave per SIMD).  If halving the occupancy does not slow
the big rounds down, the kernel is bound by its dependent 64-bit multiply chains (v_mad_u64_u32, quarter rate), not by
latency, and doubling the occupancy (<= 128 VGPRs) would not speed it up.
  python3 tools/exp_sumcheck_occupancy.py [nv]     (run once per value of ZIP_HIP_SUMCHECK_LDS_PAD)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import cabi  # noqa: E402
import torch  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 22
fl, K, degree = 4, 4, 3
n = 1 << nv
q = bench.BENCH_MODULUS
rng = np.random.default_rng(1)
dev = []
for k in range(K):
    t = rng.integers(0, 1 << 62, size=(n, fl), dtype=np.uint64)
    t[..., fl - 1] >>= np.uint64(6)
    dev.append(torch.from_numpy(t.view(np.int64)).cuda())
zf = cabi.make_field(q, fl)
R = 1 << (64 * fl)
limbs = lambda x: [(x >> (64 * i)) & ((1 << 64) - 1) for i in range(fl)]
comb = cabi.make_comb([0b011, 0b100], [limbs(R % q), limbs((q - 1) * R % q)])
r = np.array([3, 1, 4, 1], dtype=np.uint64)
best = None
for rep in range(4):
    sc = cabi.Sumcheck(dev, nv, degree, zf, comb=comb)
    torch.cuda.synchronize()
    per = []
    for i in range(4):
        t1 = time.perf_counter()
        sc.round(None if i == 0 else r)
        per.append(time.perf_counter() - t1)
    sc.free()
    best = per if best is None else [min(a, b) for a, b in zip(best, per)]
print(f"pad={os.environ.get('ZIP_HIP_SUMCHECK_LDS_PAD', '0')}: 2^{nv}, K=4, degree 3 (CCS): rounds 1..4 "
      + ", ".join(f"{p * 1e3:.3f}" for p in best) + " ms")
