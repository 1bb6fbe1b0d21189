#!/usr/bin/env python3
"""Diagnosis: gather of commitment N beside the commit of N+1 (no data dependency) vs each alone."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zinc_amd import cabi
from zinc_amd.perm import shuffle_seeded_perm
import torch

nv = 24
row_len, num_rows, cw = cabi.geometry(nv)
ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
depth = cw.bit_length() - 1
wire = torch.empty(cols.size * num_rows * (40 + 32 * depth), dtype=torch.uint8, device="cuda")

def timed(fn, reps=8):
    fn(); ctx.synchronize(); torch.cuda.synchronize()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    k = ctx.profile_read(); ctx.set_profiling(False)
    return dt, {n: round(v[1] / v[0], 3) for n, v in k.items() if "wait" not in n}

com0, _ = ctx.commit(evals, want_roots=False); ctx.synchronize()
def only_commit():
    c, _ = ctx.commit(evals, want_roots=False); ctx.synchronize(); c.free()
def only_gather():
    com0.open_columns(cols, out=wire)
def both():
    c, _ = ctx.commit(evals, want_roots=False)   # async on s_commit
    com0.open_columns(cols, out=wire)            # main stream, syncs it
    ctx.synchronize(); c.free()
print("commit alone ", timed(only_commit))
print("gather alone ", timed(only_gather))
print("both         ", timed(both))

if os.environ.get("EXP_STREAMS"):
    # commit beside a plain streaming copy of the same volume as the gather (1.87 GB read + 1.87 GB write)
    src = torch.empty(wire.numel(), dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    def only_copy():
        with torch.cuda.stream(side):
            wire.copy_(src)
        side.synchronize()
    def commit_and_copy():
        c, _ = ctx.commit(evals, want_roots=False)
        with torch.cuda.stream(side):
            wire.copy_(src)
        side.synchronize(); ctx.synchronize(); c.free()
    def timed_wall(fn, reps=8):
        fn(); torch.cuda.synchronize()
        ev = []
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    print("copy alone    %.3f ms" % timed_wall(only_copy))
    print("commit + copy ", timed(commit_and_copy))
    # write-only companion: memset of 3.7 GB
    big = torch.empty(2 * wire.numel(), dtype=torch.uint8, device="cuda")
    def only_fill():
        with torch.cuda.stream(side):
            big.zero_()
        side.synchronize()
    def commit_and_fill():
        c, _ = ctx.commit(evals, want_roots=False)
        with torch.cuda.stream(side):
            big.zero_()
        side.synchronize(); ctx.synchronize(); c.free()
    print("fill alone    %.3f ms" % timed_wall(only_fill))
    print("commit + fill ", timed(commit_and_fill))
    
