#!/usr/bin/env python3
"""Timing experiment (GPU box): the hinted commit kernel ALONE, library given by ZIP_HIP_LIB_PATH -- e.g. the
-DZIPK_EXP_SCANS_ONLY build (rows without their hash phase and chunk ends: what do the two scan passes with their four
barriers cost per row?).  Results are garbage in that build; only the kernel time is read."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import cabi  # noqa: E402
from zinc_amd.perm import shuffle_seeded_perm  # noqa: E402


def main():
    import torch

    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    row_len, num_rows, cw = cabi.geometry(nv)
    ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
    _, cols, _ = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
    evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
    for rep in range(8):
        if rep == 2:
            ctx.set_profiling(True)
        com, _ = ctx.commit(evals, want_roots=False, hint_cols=cols)
        ctx.synchronize()
        com.free()
    for k, (n, ms) in sorted(ctx.profile_read().items()):
        print(f"{k:28s} launches {n:3d}  avg {ms / n:8.4f} ms   ({ms / n / (num_rows / 256) * 1e3:6.2f} us per row of a workgroup)")


if __name__ == "__main__":
    main()
