#!/usr/bin/env python3
"""Per-kernel HIP-event times through the C ABI's measurement hooks (GPU box).  Also the workload of the PMC passes
of tools/profile_round.sh: one process runs the hinted AND the plain commit (their kernels are different template
instances, so rocprofv3 lists them separately) followed by the open."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import cabi  # noqa: E402
from zinc_amd.perm import shuffle_seeded_perm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-vars", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--serial", action="store_true", help="wait for the commit before opening (kernels timed alone)")
    ap.add_argument("--hint", action="store_true", help="zip_commit_hinted with the columns of the open")
    ap.add_argument("--both", action="store_true", help="zip_commit_open, hinted and plain commits in turn (PMC passes)")
    args = ap.parse_args()
    import torch

    # what is being measured: tools/pmc_summary.py reads this line back from the log of a counter pass
    print("kernel_src_sha", bench.kernel_sources_sha(), flush=True)
    nv = args.num_vars
    row_len, num_rows, cw = cabi.geometry(nv)
    ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
    zf = cabi.make_field(bench.BENCH_MODULUS, 4)
    coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
    evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
    proof = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
    for rep in range(args.reps + 1):
        if rep == 1:
            ctx.set_profiling(True)
        for hinted in (("one_call", True, False) if args.both else (args.hint,)):
            if hinted == "one_call":
                _, _, com = ctx.commit_open(evals, coeffs, cols, q0, zf, out=proof, want_roots=False, keep=True)
            else:
                ctx.set_speculation(bool(hinted))  # (a "plain" commit must not hint itself with the last opening's columns)
                com, _ = ctx.commit(evals, want_roots=False, hint_cols=cols if hinted else None)
                if args.serial:
                    ctx.synchronize()
                com.open(evals, coeffs, cols, q0, zf, out=proof)
            com.free()
        c2, _ = ctx.commit(evals, with_merkle=False)
        c2.free()
    ctx.synchronize()
    t = ctx.profile_read()
    for k, (n, ms) in sorted(t.items()):
        print(f"{k:28s} launches {n:3d}  avg {ms / n:8.4f} ms")
    print(f"shader clock during the last commit kernel: {ctx.commit_clock_mhz():.0f} MHz")


if __name__ == "__main__":
    main()
