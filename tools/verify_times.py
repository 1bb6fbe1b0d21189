#!/usr/bin/env python3
"""Per-kernel HIP-event times of the device verifier (zip_verify) on a GPU-made proof (GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from zinc_amd import cabi, pcs  # noqa: E402
from zinc_amd.perm import shuffle_seeded_perm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-vars", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch

    nv = args.num_vars
    row_len, num_rows, cw = cabi.geometry(nv)
    ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
    zf = cabi.make_field(bench.BENCH_MODULUS, 4)
    field = pcs.FieldConfig(bench.BENCH_MODULUS, 4)
    coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
    lr = num_rows.bit_length() - 1
    point = field.map_to_field(np.ones(nv, dtype=np.int64))  # [1; num_vars] as in zip_benches.rs:143
    q0 = field.build_eq_x_r(point[nv - lr:])
    q1 = field.build_eq_x_r(point[: nv - lr])
    evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
    proof = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
    com, roots = ctx.commit(evals)
    com.open(evals, coeffs, cols, q0, zf, out=proof)
    ev = ctx.mle_eval(evals, q0, q1, zf)
    ctx.synchronize()
    for rep in range(args.reps + 1):
        if rep == 1:
            ctx.set_profiling(True)
            t0 = time.perf_counter()
        r = ctx.verify(roots, proof, coeffs, cols, q0, q1, ev, zf)
        assert r["verdict"] == 0, r
    dt = (time.perf_counter() - t0) / args.reps * 1e3
    t = ctx.profile_read()
    print(f"zip_verify 2^{nv}: {dt:.3f} ms per call (proof resident in HBM)")
    for k, (n, ms) in sorted(t.items()):
        print(f"{k:28s} launches {n:3d}  avg {ms / n:8.4f} ms")


if __name__ == "__main__":
    main()
