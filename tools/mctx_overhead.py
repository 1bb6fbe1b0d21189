#!/usr/bin/env python3
"""What does row-sharding through zip_mctx cost?  (GPU box, ONE GPU: every shard on device 0, so the shards run one
after the other on the hardware and `ms_per_step` is the SUM of their work -- the sharding overhead is what that sum
exceeds the unsharded step by.)  bench.py --gpus N with BENCH_MCTX_DEVICES=0,..,0 for N = 1, 2, 4, 8 at 2^24 and 2^26.
Writes a markdown table.  This process never touches the GPU: it only starts bench.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(nv, shards, steps):
    env = dict(os.environ, BENCH_MCTX_DEVICES=",".join(["0"] * shards))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(shards), "--shard", "mctx", "--num-vars", str(nv),
           "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        raise RuntimeError(out.stderr[-2000:])
    return json.loads(lines[-1])


def main():
    md = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mctx_overhead.md")
    rows = ["| 2^n | shards (all on device 0) | ms per step (shards run back to back) | vs 1 shard | shard 0: commit kernel ms | shard 0: gathers ms | rows per shard | roots gather |",
            "|---|---|---|---|---|---|---|---|"]
    for nv, steps in ((24, 10), (26, 4)):
        base = None
        for shards in (1, 2, 4, 8):
            d = run(nv, shards, steps)
            base = base or d["ms_per_step"]
            k = d["kernels_ms_per_step_shard0"]
            rows.append(f"| {nv} | {shards} | {d['ms_per_step']:.3f} | {d['ms_per_step'] / base:.3f} | {k.get('raa_commit_kernel', 0):.3f} | "
                        f"{k.get('open_columns_kernel', 0):.3f} | {d['config']['num_rows'] // shards} | {d['roots_gather']} |")
            print(rows[-1], flush=True)
    with open(md, "w") as fh:
        fh.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
