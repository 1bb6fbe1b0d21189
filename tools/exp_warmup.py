#!/usr/bin/env python3
"""How long does a fresh process take to reach the steady step time?  Per-5-step averages of the first 120 steps. (GPU box.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from zinc_amd import cabi
from zinc_amd.perm import shuffle_seeded_perm

nv = 24
row_len, num_rows, cw = cabi.geometry(nv)
ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
zf = cabi.make_field(bench.BENCH_MODULUS, 4)
coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 0x5A494E43)
evals = torch.from_numpy(bench.splitmix64(0x5A494E43, 1 << nv).copy()).cuda()
proof = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
out = []
for blk in range(24):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.commit_open(evals, coeffs, cols, q0, zf, out=proof, want_roots=False, keep=False)
    ctx.synchronize()
    out.append((time.perf_counter() - t0) / 5 * 1e3)
print(" ".join(f"{x:.3f}" for x in out))
