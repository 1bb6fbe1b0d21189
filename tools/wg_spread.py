#!/usr/bin/env python3
"""When do the workgroups of the persistent commit kernel start and end?  (GPU box)

Needs the debug build of the library (per-workgroup wall-clock stamps, -DZIPK_DEBUG_STAMPS):
  python3 tools/wg_spread.py --build          # here or on the box: zinc_amd/lib/libzip_hip_stamps.so
  ZIP_HIP_LIB_PATH=zinc_amd/lib/libzip_hip_stamps.so python3 tools/wg_spread.py [--alone] [--md profiles/x.md]
Runs bench-style commit+open steps at 2^num_vars and prints, for the LAST step's commit kernel, the distribution of
workgroup start / end times (100 MHz wall clock), per XCD, and where a workgroup's time goes.
--alone: zip_commit_hinted only (no openings beside the kernel)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(defs=(), suffix=""):
    from zinc_amd import build as b

    src = os.path.join(b.CSRC, "zip_hip.hip")
    out = os.path.join(b.LIB, f"libzip_hip_stamps{suffix}.so")
    cmd = [b.hipcc(), "-O3", "-std=c++17", f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-DZIPK_DEBUG_STAMPS",
           *[f"-D{d}" for d in defs], "-Wno-unused-function", "-o", out, src]
    print(" ".join(cmd), flush=True)
    import subprocess

    subprocess.run(cmd, check=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--defs", default="", help="--build: extra -D macros, comma separated (timing experiments)")
    ap.add_argument("--suffix", default="", help="--build: suffix of the library name")
    ap.add_argument("--num-vars", type=int, default=24)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--alone", action="store_true")
    ap.add_argument("--md", default=None, help="append the report to this markdown file")
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    if args.build:
        print(build([d for d in args.defs.split(",") if d], args.suffix))
        return
    import numpy as np
    import torch

    import bench
    from zinc_amd import cabi
    from zinc_amd.perm import shuffle_seeded_perm

    L = cabi.lib()
    if not hasattr(L, "zip_debug_stamps"):
        raise SystemExit("this library has no stamps: set ZIP_HIP_LIB_PATH to the -DZIPK_DEBUG_STAMPS build (--build)")
    L.zip_debug_stamps.restype = C.POINTER(C.c_uint64)
    L.zip_debug_stamps.argtypes = [C.POINTER(C.c_uint32)]
    nv = args.num_vars
    row_len, num_rows, cw = cabi.geometry(nv)
    ctx = cabi.ZipContext(nv, shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw))
    zf = cabi.make_field(bench.BENCH_MODULUS, 4)
    coeffs, cols, q0 = bench.host_inputs(nv, row_len, num_rows, cw, 4, 1)
    evals = torch.from_numpy(bench.splitmix64(7, 1 << nv).copy()).cuda()
    proof = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
    import time

    for i in range(args.steps):
        if i == args.steps - 1:
            ctx.synchronize()
            t0 = time.perf_counter()
        if args.alone:
            com, _ = ctx.commit(evals, want_roots=False, hint_cols=cols)
            ctx.synchronize()
            com.free()
        else:
            ctx.commit_open(evals, coeffs, cols, q0, zf, out=proof, want_roots=False, keep=False)
    ctx.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e3
    rec = C.c_uint32(0)
    p = L.zip_debug_stamps(C.byref(rec))
    rec = rec.value
    G = min(256, num_rows)  # one workgroup per CU at the big geometries
    a = np.ctypeslib.as_array(p, shape=(2048 * rec,)).reshape(2048, rec)[:G].astype(np.int64)
    start, end, hw = a[:, 0], a[:, 1], a[:, 2]
    t00 = start.min()
    s_us, e_us = (start - t00) / 100.0, (end - t00) / 100.0
    xcc = hw & 0xF
    nrows = a[:, 6]
    out = []
    w = out.append
    w(f"### {args.label or ('commit alone' if args.alone else 'commit + open step')}, 2^{nv}, {G} workgroups, last of {args.steps} steps "
      f"(host time of that step {step_ms:.3f} ms)\n")
    w(f"kernel span (first start -> last end): **{e_us.max():.1f} us**; mean workgroup life {np.mean(e_us - s_us):.1f} us "
      f"(min {np.min(e_us - s_us):.1f}, max {np.max(e_us - s_us):.1f}); rows per workgroup {nrows.min()}..{nrows.max()}\n")
    q = lambda v: ", ".join(f"{np.percentile(v, p):.1f}" for p in (0, 10, 50, 90, 100))
    w(f"* start after the first workgroup's start, us (min, p10, p50, p90, max): {q(s_us)}")
    w(f"* end before the last workgroup's end, us (min, p10, p50, p90, max): {q(e_us.max() - e_us)}")
    w(f"* per workgroup, us: scan passes {a[:, 3].mean() / 100:.1f}, hash phase {a[:, 4].mean() / 100:.1f}, chunk ends {a[:, 5].mean() / 100:.1f}\n")
    w("| XCD | workgroups | start p50 / max (us) | life mean / min / max (us) | end p50 / max (us) |")
    w("|---|---|---|---|---|")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        life = (e_us - s_us)[m]
        w(f"| {x} | {int(m.sum())} | {np.median(s_us[m]):.1f} / {s_us[m].max():.1f} | {life.mean():.1f} / {life.min():.1f} / {life.max():.1f} | "
          f"{np.median(e_us[m]):.1f} / {e_us[m].max():.1f} |")
    # per-row durations: is a workgroup uniformly slow, or are single rows slow?
    nr = int(nrows.min())
    if nr >= 2:
        ends = a[:, 8:8 + min(nr, rec - 8)]
        d = np.diff(np.concatenate([start[:, None], ends], axis=1), axis=1) / 100.0
        w(f"\nrow time (us) over all workgroups and rows: mean {d.mean():.2f}, p1 {np.percentile(d, 1):.2f}, p50 {np.median(d):.2f}, "
          f"p99 {np.percentile(d, 99):.2f}, max {d.max():.2f}; per workgroup mean row time: min {d.mean(axis=1).min():.2f}, "
          f"max {d.mean(axis=1).max():.2f}")
        w("mean row time by position in the workgroup's sequence (us): " + " ".join(f"{v:.1f}" for v in d.mean(axis=0)))
    text = "\n".join(out) + "\n"
    print(text)
    if args.md:
        with open(args.md, "a") as fh:
            fh.write(text + "\n")
    ctx.close() if hasattr(ctx, "close") else None


if __name__ == "__main__":
    main()
