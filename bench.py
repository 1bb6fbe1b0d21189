#!/usr/bin/env python3
"""Headline benchmark: Zip commit + open MCoeffs/s on a 2^24-coefficient witness
(BASELINE.json `metric`, `configs[2]`).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = MultilinearZip::commit followed by MultilinearZip::open of a synthetic witness that is
already resident in HBM when the timed region starts; every output (encoded rows, Merkle layers,
roots, the 1.74 GiB proof stream) is produced in HBM.  The Fiat-Shamir outputs the Rust shim would
squeeze before launching (proximity coefficients, 1000 column indices, the eq tensor q_0 --
SURVEY.md A.4) are small host arrays handed over inside the timed region.

N > 1, default `--shard polys` (weak scaling): one process per GPU, every rank commits + opens its
own 2^24 polynomial (batch_commit / batch_open sharded over ranks, commit.rs:134-142,
open_z.rs:43-58); the roots of all ranks are all-gathered over RCCL inside the timed region.
`--shard mctx` (strong scaling): ONE process row-shards ONE polynomial over the N GPUs through the C ABI's
multi-device context (zip_mctx_commit_open) -- what a Rust ZincProver, one process making one call, can use;
under torchrun rank 0 works and the other ranks wait at the barriers.
`--shard rows` (strong scaling, one Python process per GPU): zinc_amd.dist over torch.distributed,
all-gather of roots and of the partial row combinations + exact on-device sum.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383  # benches/zip_benches.rs:253
STEADY_AFTER = 12  # a fresh process reaches its steady step time after about this many steps (see main())
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FIELD_LIMBS = 4


def splitmix64(seed, n):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.view(np.int64)


def algorithmic_bytes(n, row_len, num_rows, cw, depth, n_cols, fl):
    """SURVEY.md §8d / BASELINE.md, bytes per whole polynomial and per stage (full materialisation,
    as the reference does; the two row combinations share ONE pass over the witness)."""
    commit = n * 8 + num_rows * cw * 32 * 3  # witness read + rows + leaf hashes + inner nodes (200 B/coeff)
    combine = n * 8 + row_len * (64 + 8 * fl)
    gather = 2 * n_cols * num_rows * (32 + 8 + 32 * depth)  # sibling/value reads + wire-format writes
    return {"commit": commit, "combine": combine, "gather": gather}


def host_inputs(num_vars, row_len, num_rows, cw, fl, seed):
    """SURVEY.md §8d streams: coeffs seed+1, cols seed+2; point = [1; num_vars]
    (benches/zip_benches.rs:143) -> q_0 = eq tensor of its last log2(num_rows) coordinates
    (pcs/utils.rs:279-292), in Montgomery form."""
    q = BENCH_MODULUS
    R = 1 << (64 * fl)
    coeffs = splitmix64(seed + 1, num_rows)
    cols = (splitmix64(seed + 2, 1000).view(np.uint64) % np.uint64(cw)).astype(np.uint32)
    lr = num_rows.bit_length() - 1
    point = [1] * num_vars
    q0 = [1]
    for t, rt in enumerate(point[num_vars - lr:]):  # variable t <-> bit t of the row index
        q0 = [q0[i & ((1 << t) - 1)] * (rt if (i >> t) & 1 else (1 - rt)) % q for i in range(1 << (t + 1))]
    q0 = [v * R % q for v in q0]
    q0_arr = np.zeros((num_rows, fl), dtype=np.uint64)
    for i, v in enumerate(q0):
        for k in range(fl):
            q0_arr[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return coeffs, cols, q0_arr


def cpu_baseline(num_vars, seed, budget_s=20.0):
    """The CPU oracle (C restatement of the reference, OpenMP over rows / columns like the
    reference's Rayon build) timed on this box's host cores on a bounded sample: commit + open of
    the largest 2^k witness (k <= num_vars) estimated to fit the time budget.  kind = "port": the
    Rust reference cannot be built here (no cargo)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc

    f = orc.make_field(BENCH_MODULUS, FIELD_LIMBS)

    def run(nv):
        z = orc.Zip(nv)
        evals = splitmix64(seed, 1 << nv)
        point = orc.point_to_field(f, [1] * nv)
        t0 = time.perf_counter()
        rows, layers, _ = z.commit(evals)
        z.open(f, evals, rows, layers, point, orc.new_transcript())
        return time.perf_counter() - t0

    probe = min(num_vars, 18)
    t_probe = run(probe)
    nv = probe
    while nv < min(num_vars, 24) and t_probe * (1 << (nv + 1 - probe)) <= budget_s:
        nv += 1
    dt = run(nv) if nv != probe else t_probe
    return {"value": round((1 << nv) / dt / 1e6, 4), "unit": "MCoeffs/s", "cores": orc.lib().orc_num_threads(),
            "kind": "port",
            "sample": f"oracle commit+open of one 2^{nv}-coefficient witness, {dt:.2f} s wall (OpenMP over rows/columns)"}


def pmc_entry(kernel, num_vars, mode):
    """Per-launch counters of `kernel` from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from the raw CSVs of tools/profile_round.sh), or None.  Evidence measured once per round on the
    same code, not in this run (counter collection serialises the streams); `source` names the file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d.get(f"{kernel}:{num_vars}:{mode}") or d.get(f"{kernel}:{num_vars}:any")
    except (OSError, ValueError):
        return None


def kernel_sources_sha():
    """Digest of the kernel sources (the same one tools/pmc_summary.py stores with every pmc_traffic.json entry)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kernels_commit.cuh", "kernels_open.cuh", "blake3.cuh", "blake3_sched.inc"):
        with open(os.path.join(ROOT, "zinc_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def commit_moved_bytes(per, row_len, cw, depth, cols):
    """Bytes the commit kernel really moves per launch (as opposed to SURVEY 8d's full-materialisation figure): the
    witness, 16-byte row entries (the 96 significant bits + sign; Int<4> only on demand), the tree nodes, the
    children of levels >= 4 read back by the in-kernel upper levels, the roots.  cols = the opening hint (None: plain
    commit): then only the entries / leaf hashes / level-1 and -2 nodes an opening of those columns reads are stored."""
    upper_nodes = max(cw // 4 - 1, 0)  # levels 3 .. depth
    reread = 2 * 32 * max(cw // 8 - 1, 0)  # children of levels 4 .. depth
    if cols is None:
        per_row = 16 * cw + 32 * (2 * cw - 1)
    else:
        c = np.unique(np.asarray(cols, dtype=np.int64))
        n0 = c.size
        n1 = np.unique((c >> 1) ^ 1).size
        n2 = np.unique((c >> 2) ^ 1).size
        per_row = 16 * c.size + 32 * (n0 + n1 + n2 + upper_nodes)
    return int(per * (row_len * 8 + per_row + reread + 32))


_ROOT_VIEWS = {}


def roots_view(torch, ptr, rows, dev):
    """Zero-copy torch view of a commitment's device-resident roots.  The library's pool hands the same few blocks out
    again and again: the view of an address is built once (torch.as_tensor over __cuda_array_interface__ costs ~25 us)."""
    key = (ptr, rows, str(dev))
    v = _ROOT_VIEWS.get(key)
    if v is None:
        holder = type("_H", (), {})()
        holder.__cuda_array_interface__ = {"shape": (rows, 32), "typestr": "|u1", "data": (ptr, False), "version": 2}
        v = _ROOT_VIEWS[key] = torch.as_tensor(holder, device=dev)
    return v


def run_mctx(args, torch, dist, rank, world, cabi, perm1, perm2, zf, coeffs, cols, q0, nv, row_len, num_rows, cw, depth, fl):
    """--shard mctx: ONE process drives the GPUs through the C ABI's multi-device context (zip_mctx_commit_open): the
    rows of one polynomial split over them (strong scaling), witness resident on the devices, every output left in
    HBM (each shard's openings on its own device, u' and the evaluation row on the lead device).  Under torchrun only
    rank 0 works; the other ranks wait at the barriers.  BENCH_MCTX_DEVICES=0,0,0,0 repeats a device (one-GPU box)."""
    n = 1 << nv
    ndev = args.devices or (args.gpus if world == 1 else world)
    devs = ([int(x) for x in os.environ["BENCH_MCTX_DEVICES"].split(",")] if "BENCH_MCTX_DEVICES" in os.environ
            else list(range(ndev)))
    if rank == 0 and max(devs) >= cabi.device_count():
        raise SystemExit(f"--gpus {ndev}: this box has {cabi.device_count()} device(s); to rehearse {ndev} row shards on "
                         f"one GPU set BENCH_MCTX_DEVICES=" + ",".join(["0"] * ndev))
    m = None
    if rank == 0:
        m = cabi.ZipMultiContext(nv, perm1, perm2, devs)
        m.set_witness(splitmix64(args.seed, n))

    def step():
        if m is not None:
            m.commit_open(None, coeffs, cols, q0, zf, want_proof=False, want_roots=False)  # returns when every shard has drained

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    if m is not None:
        for sh in range(len(devs)):
            m.shard_profile(sh, on=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        ktimes = m.shard_profile(0)
        step_s = dt / args.steps
        ab = algorithmic_bytes(n, row_len, num_rows, cw, depth, cols.size, fl)
        per = num_rows // len(devs)
        launches, tot_ms = ktimes.get("raa_commit_kernel", (1, 0.0))
        avg_ms = tot_ms / max(launches, 1)
        commit_bytes = per * row_len * 8 + per * cw * 32 * 3
        out = {
            "metric": "Zip commit+open MCoeffs/s at 2^%d witness" % nv, "value": round(n / step_s / 1e6, 2), "unit": "MCoeffs/s",
            "n_gpus": len(devs), "distinct_devices": len(set(devs)),
            "roots_gather": m.roots_path(),  # "rccl": in-process ncclAllGather over xGMI; "copies": repeated ordinals (rehearsal)
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i64",
            "data": "synthetic (SplitMix64 full-range i64 witness; coefficient / column / point streams per SURVEY.md 8d)",
            "config": {"workload": "Zip commit+open_z 2^%d coeffs, ONE polynomial row-sharded over %d device context(s) by one process (zip_mctx)" % (nv, len(devs)),
                       "row_len": row_len, "num_rows": num_rows, "codeword_len": cw, "column_openings": int(cols.size),
                       "field_limbs": fl, "parallelism": "mctx rows%d" % len(devs), "devices": devs},
            "roofline": {"bound": "valu", "kernel": "raa_commit_kernel (shard 0)",
                         "achieved": round(commit_bytes / (avg_ms * 1e-3) / 1e9, 1) if avg_ms else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(commit_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_ms else None,
                         "traffic": None, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(commit_bytes)},
            "whole_path": {"algorithmic_bytes": int(sum(ab.values())),
                           "hbm_frac_per_gpu": round(sum(ab.values()) / step_s / 1e9 / HBM_PEAK_GBS / len(devs), 4)},
            "kernels_ms_per_step_shard0": {k: round(v[1] / args.steps, 4) for k, v in sorted(ktimes.items())},
            # the dominant kernel of every shard (ms per launch): on distinct GPUs they run side by side
            "commit_kernel_ms_per_shard": [round(v[1] / max(v[0], 1), 4) for v in
                                           (m.shard_profile(sh).get("raa_commit_kernel", (1, 0.0)) for sh in range(1, len(devs)))],
        }
        out["commit_kernel_ms_per_shard"].insert(0, round(avg_ms, 4))
        print(json.dumps(out), flush=True)
        m.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-vars", type=int, default=24)
    ap.add_argument("--shard", choices=["polys", "rows", "mctx"], default="polys")
    ap.add_argument("--devices", type=int, default=0, help="--shard mctx: GPUs driven by the one working process (default: --gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-calls", action="store_true",
                    help="zip_commit_hinted + zip_open instead of zip_commit_open (values / low siblings via the trees)")
    ap.add_argument("--no-hint", action="store_true",
                    help="plain zip_commit (every row entry and tree node stored) instead of zip_commit_hinted")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra leg that runs the same steps as jobs, two in flight (zip_commit_open_begin / "
                         "zip_job_wait); it is reported beside `value`, never as it")
    ap.add_argument("--steady-only", action="store_true",
                    help="only the warm-up, cold and steady one-call steps: no event-bracketed second region, no extra legs "
                         "(the run tools/profile_round.sh traces, so that rocprofv3's per-kernel averages are those of `value`)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="extra leg (1 GPU, reported beside `value`, never as it): this many independent commit+open jobs "
                         "in flight at once, one zip_ctx and one host thread each")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5A494E43)
    args = ap.parse_args()

    if not args.no_cpu_baseline and int(os.environ.get("RANK", "0")) == 0:
        # the checker library of the cpu_baseline leg is built (if stale) BEFORE anything touches the GPU:
        # a GPU-initialised process must not fork + exec `make` on the GPU boxes
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle

        _oracle.build()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (not used by the driver): BENCH_BACKEND=gloo with BENCH_DEVICE=0 runs the multi-rank
    # control flow with every rank on one GPU, the exchanges staged through host tensors.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if "BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE"])
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where collectives operate
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    from zinc_amd import cabi
    from zinc_amd.perm import shuffle_seeded_perm

    nv, fl = args.num_vars, FIELD_LIMBS
    n = 1 << nv
    row_len, num_rows, cw = cabi.geometry(nv)
    depth = cw.bit_length() - 1
    # seeds (1, 2) = what MockTranscript yields (src/zip/pcs/tests.rs:24-37); the tables are INPUTS of
    # the device path (the Rust shim computes them with rand itself)
    perm1, perm2 = shuffle_seeded_perm(1, cw), shuffle_seeded_perm(2, cw)
    zf = cabi.make_field(BENCH_MODULUS, fl)
    coeffs, cols, q0 = host_inputs(nv, row_len, num_rows, cw, fl, args.seed)
    n_cols = cols.size

    if world == 1 and args.gpus > 1 and args.shard == "polys":
        # `python3 bench.py --gpus N` without a launcher: ONE process drives the N GPUs through the C ABI's multi-device
        # context (strong scaling of one polynomial, the roots gathered with in-process RCCL) -- what a Rust ZincProver,
        # one process making one call, would use.  Under torchrun (WORLD_SIZE = N) the default stays `polys`.
        args.shard = "mctx"
    if args.shard == "mctx":
        return run_mctx(args, torch, dist, rank, world, cabi, perm1, perm2, zf, coeffs, cols, q0, nv, row_len, num_rows, cw, depth, fl)
    rows_mode = args.shard == "rows" and world > 1
    if rows_mode:
        from zinc_amd.dist import RowShardedZip

        sharded = RowShardedZip(nv, perm1, perm2, device=local_rank)
        ctx = sharded.backend.ctx
        per = sharded.row_count
        witness = sharded.local_slice(splitmix64(args.seed, n))
    else:
        ctx = cabi.ZipContext(nv, perm1, perm2, device=local_rank)
        per = num_rows
        witness = splitmix64(args.seed + 1000 * rank, n)
    if args.no_hint:
        ctx.set_speculation(False)  # everything stored, also after the first opening has named the columns
    evals_d = torch.from_numpy(np.ascontiguousarray(witness)).to(dev)
    if not rows_mode:
        proof = torch.empty(ctx.proof_len(n_cols, fl), dtype=torch.uint8, device=dev)
        roots_all = torch.empty((world * per, 32), dtype=torch.uint8, device=xdev) if world > 1 else None

    # one proof after the other of the same shape into the same buffers: the call's arguments are marshalled once
    # (cabi.commit_open_prepared; BENCH_PREPARED=0: through commit_open every step)
    one_call = None
    if not rows_mode and world == 1 and not (args.two_calls or args.no_hint) and os.environ.get("BENCH_PREPARED", "1") != "0":
        one_call = ctx.commit_open_prepared(evals_d, coeffs, cols, q0, zf, proof)

    def step():
        if one_call is not None:
            one_call()
            return
        if rows_mode:
            # hinted commit enqueued, the shard's open pipelined behind it, the roots all-gathered at the end
            com, _ = sharded.commit(evals_d, cols, gather_roots=False)
            sharded.open(com, evals_d, coeffs, cols, q0, zf)
            sharded.gather_roots(com)
        else:
            if args.two_calls or args.no_hint:
                # asynchronous: the open below overlaps it.  --no-hint: plain zip_commit, everything stored
                com, _ = ctx.commit(evals_d, want_roots=False, hint_cols=None if args.no_hint else cols)
                com.open(evals_d, coeffs, cols, q0, zf, out=proof)  # returns when the whole stream is in HBM
            else:
                # commit + open as ONE call, as the prover makes them (commit_z_mle_and_prove_evaluation opens with
                # a fresh PcsTranscript, src/zinc/prover.rs:305-328: the columns are known before the commit)
                _, _, com = ctx.commit_open(evals_d, coeffs, cols, q0, zf, out=proof, want_roots=False, keep=world > 1)
            if world > 1:
                # the one exchange of the commit: every rank's Merkle roots
                roots_ptr = com.roots_ptr()  # rows=NULL: the 16-byte row entries are not expanded
                mine = roots_view(torch, roots_ptr, per, dev)
                dist.all_gather_into_tensor(roots_all, mine if backend == "nccl" else mine.cpu())
                torch.cuda.current_stream().synchronize()  # the roots buffer returns to the pool below
        if com is not None:
            com.free()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # `--warmup W` means what it says: W untimed steps, then K timed steps = the COLD figure (`cold` in the JSON line: a
    # fresh process needs ~15 steps to its steady state -- first use of the 2.5 GB of pooled buffers, table uploads, clocks:
    # tools/exp_warmup.py).  Those K steps (topped up to STEADY_AFTER if K is smaller) are the spin-up of the steady
    # measurement that follows; `value` is the steady figure, `untimed_steps_before_value` says how many steps preceded it.
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_cold = time.perf_counter() - t0
    extra_spinup = max(0, STEADY_AFTER - args.steps)
    for _ in range(extra_spinup):
        step()
    barrier()
    untimed_before = args.warmup + args.steps + extra_spinup
    # HIP events on the stream the dominant kernel is launched on (its own); the other kernels of a step are timed in a
    # short second region below: events between the kernels of one stream delay every dependent launch by ~12 us
    ctx.set_profiling(0 if os.environ.get("BENCH_NO_EVENTS") else 2)  # (BENCH_NO_EVENTS=1: what do the two events cost?)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ktimes = ctx.profile_read()
    if os.environ.get("BENCH_NO_EVENTS"):
        ktimes = {"raa_commit_kernel": (1, float("nan"))}
    clock_mhz = ctx.commit_clock_mhz()  # shader clock during the last timed commit kernel (in-kernel stamps)
    steps_all = min(args.steps, 10)
    ktimes_all = {}
    if not args.steady_only:
        ctx.set_profiling(True)
        for _ in range(steps_all):
            step()
        barrier()
        ktimes_all = ctx.profile_read()
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([dt_cold], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_cold = float(t.item())

    pipelined = None
    if args.steady_only:
        args.no_pipelined = True
    if not args.no_pipelined and world == 1 and not rows_mode and not (args.two_calls or args.no_hint):
        # The same K steps as jobs, two in flight: the next polynomial's commit kernel is queued behind the current
        # one's and runs beside the end of its openings (the last chunk's gather).  Throughput of a prover that has a
        # queue of polynomials; `value` above stays the one-proof-at-a-time figure.
        proofs2 = [proof, torch.empty_like(proof)]
        torch.cuda.synchronize()
        for k in (2, args.steps):  # warm-up, then the timed run
            barrier()
            t1 = time.perf_counter()
            pending = []
            for i in range(k):
                if len(pending) == 2:
                    pending.pop(0).wait()
                pending.append(ctx.commit_open_begin(evals_d, coeffs, cols, q0, zf, proofs2[i & 1]))
            for j in pending:
                j.wait()
            barrier()
            dtp = time.perf_counter() - t1
        pipelined = {"jobs_in_flight": 2, "steps": args.steps, "ms_per_step": round(dtp / args.steps * 1e3, 4),
                     "value": round(n * args.steps / dtp / 1e6, 2), "unit": "MCoeffs/s",
                     "proofs_identical": bool(torch.equal(proofs2[0], proofs2[1])),
                     "api": "zip_commit_open_begin / zip_job_wait",
                     "note": "the same steps queued as jobs, two in flight; `value` above is one proof at a time"}

    two_call = None
    full_mat = None
    if world == 1 and not rows_mode and not (args.two_calls or args.no_hint) and not args.steady_only:
        # What MultilinearZip::commit returns in the reference: the WHOLE MultilinearZipData (every encoded row, every tree
        # node: src/zip/pcs/structs.rs:33-38; benches/zip_benches.rs:118 black-boxes it) -- plain zip_commit with no hint of
        # any kind (speculation off), then zip_open.  This is what SURVEY 8d's 200 B/coeff prices; reported beside `value`.
        ctx.set_speculation(False)

        def step_full():
            com, _ = ctx.commit(evals_d, want_roots=False)
            com.open(evals_d, coeffs, cols, q0, zf, out=proof)
            com.free()

        proof_ref = proof.clone()
        for _ in range(max(args.warmup, 3)):
            step_full()
        barrier()
        ctx.set_profiling(2)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_full()
        barrier()
        dtf = time.perf_counter() - t1
        kf = ctx.profile_read()
        ctx.set_profiling(False)
        fl_l, fl_ms = kf.get("raa_commit_kernel", (1, 0.0))
        full_bytes = per * row_len * 8 + per * cw * 32 * 3
        full_mat = {"api": "zip_commit (no hint, no speculation: everything MultilinearZipData holds is stored) + zip_open",
                    "steps": args.steps, "ms_per_step": round(dtf / args.steps * 1e3, 4),
                    "value": round(n * args.steps / dtf / 1e6, 2), "unit": "MCoeffs/s",
                    "commit_kernel_avg_launch_ms": round(fl_ms / max(fl_l, 1), 4),
                    "commit_kernel_hbm_frac": round(full_bytes / (fl_ms / max(fl_l, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fl_ms else None,
                    "whole_path_hbm_frac": round(sum(algorithmic_bytes(n, row_len, num_rows, cw, depth, n_cols, fl).values())
                                                 / (dtf / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                    "proof_identical_to_one_call": bool(torch.equal(proof, proof_ref))}
        del proof_ref

    if world == 1 and not rows_mode and not (args.two_calls or args.no_hint) and not args.steady_only:
        # The two calls an UNCHANGED ZincProver makes (src/zinc/prover.rs:315-320): plain zip_commit -- no columns in its
        # signature -- then zip_open.  The first opening names the columns; from then on the ctx hints its plain commits
        # with that list on its own, so this leg should match `value`.  The bench's witness is DEVICE memory, for which
        # that is opt-in (zip_ctx_set_speculation(ctx, 1): the witness then outlives its handles unchanged); the Rust
        # binding's HOST witness gets it by default.
        ctx.set_speculation(True)

        def step2():
            com, _ = ctx.commit(evals_d, want_roots=False)
            com.open(evals_d, coeffs, cols, q0, zf, out=proof)
            com.free()

        proof_one_call = proof.clone()
        for _ in range(max(args.warmup, 2)):
            step2()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        barrier()
        dt2c = time.perf_counter() - t1
        two_call = {"api": "zip_commit + zip_open (no column argument to the commit; speculative hint from the ctx's last opening, "
                           "zip_ctx_set_speculation(ctx, 1) for a device witness)",
                    "steps": args.steps, "ms_per_step": round(dt2c / args.steps * 1e3, 4),
                    "value": round(n * args.steps / dt2c / 1e6, 2), "unit": "MCoeffs/s",
                    "proof_identical_to_one_call": bool(torch.equal(proof, proof_one_call))}

    in_flight = None
    if args.in_flight > 1 and world == 1 and not rows_mode:
        # Independent jobs (separate provers' polynomials) overlapped: the end of one job's opening -- the last chunk's
        # gather, the fold, the host turnaround -- runs beside the next job's commit kernel.  Throughput, not latency.
        import threading

        jobs = [(ctx, evals_d, proof)]
        for j in range(1, args.in_flight):
            c2 = cabi.ZipContext(nv, perm1, perm2, device=local_rank)
            jobs.append((c2, evals_d.clone(), torch.empty_like(proof)))
        torch.cuda.synchronize()
        per_thread = max(1, args.steps // args.in_flight)

        def worker(job, k):
            c, ev, pr = job
            for _ in range(k):
                c.commit_open(ev, coeffs, cols, q0, zf, out=pr, want_roots=False, keep=False)
            c.synchronize()

        for k in (2, per_thread):  # warm-up round, then the timed one
            th = [threading.Thread(target=worker, args=(job, k)) for job in jobs]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
        same = all(bool(torch.equal(jobs[0][2], job[2])) for job in jobs[1:])
        in_flight = {"jobs_in_flight": args.in_flight, "steps": per_thread * args.in_flight,
                     "ms_per_step": round(dt2 / (per_thread * args.in_flight) * 1e3, 4),
                     "value": round(n * per_thread * args.in_flight / dt2 / 1e6, 2), "unit": "MCoeffs/s",
                     "proofs_identical": same,
                     "note": "independent jobs overlapped (one zip_ctx + one host thread each); `value` above is one job at a time"}

    if rank == 0:
        step_s = dt / args.steps
        coeffs_per_step = n * (1 if rows_mode else world)
        ab = algorithmic_bytes(n, row_len, num_rows, cw, depth, n_cols, fl)
        # dominant kernel: the persistent fused commit kernel (encode + every Merkle level).  One launch
        # per step covers `per` rows: witness slice read + rows + leaf hashes + all inner nodes
        # = 200 B/coeff x per*row_len coefficients.
        dom = "raa_commit_kernel"
        launches, tot_ms = ktimes[dom]
        avg_ms = tot_ms / launches
        commit_bytes = per * row_len * 8 + per * cw * 32 * 3
        achieved = commit_bytes / (avg_ms * 1e-3) / 1e9
        # get_hint_plan (zip_hip.hip): every hinted commit of the 8- and 16-entries-per-thread kernels
        packed = not args.no_hint and cw >= 512 and os.environ.get("ZIP_HIP_PACKED") != "0"
        mode = "plain" if args.no_hint else "packed" if packed else "hinted"
        moved = commit_moved_bytes(per, row_len, cw, depth, None if args.no_hint else cols)
        pe = pmc_entry(dom, nv, mode) if not rows_mode else None
        traffic = int((2 * pe["fetch_kib"] + pe["write_kib"]) * 1024) if pe else None  # reads x2: profiles/*_fetch_calibration.md
        simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
        valu = None
        if pe and pe.get("valu_insts") and clock_mhz > 0:
            # gfx950 issues the two-operand VOP2 integer opcodes (v_xor_b32, v_add_u32) every 2.7 cycles per SIMD and the
            # VOP3 ones (v_add3_u32, v_alignbit_b32) every 4.1; BLAKE3 is half and half, and a stream that alternates the
            # two kinds issues at 3.4 cycles per wave64 instruction (profiles/round3_valu_issue.md: tools/ubench_valu_ops,
            # tools/ubench_gsched).  That mix rate is the ceiling this kernel is priced against (round 2 priced 4.0, the
            # rate of hipcc's instruction order).
            cyc = 3.4
            peak_ginst = simds * clock_mhz * 1e6 / cyc / 1e9
            ach_ginst = pe["valu_insts"] / (avg_ms * 1e-3) / 1e9
            valu = {"kernel": dom, "insts_per_launch": int(pe["valu_insts"]), "insts_source": pe["source"],
                    "issue_cycles_per_inst": cyc, "simds": simds, "clock_mhz": round(clock_mhz, 1),
                    "achieved": round(ach_ginst, 1), "peak": round(peak_ginst, 1), "unit": "G wave-inst/s",
                    "frac": round(ach_ginst / peak_ginst, 4)}
        gather_bytes = ab["gather"]
        g_l, g_ms = ktimes_all.get("open_columns_kernel", (0, 0.0))  # (empty under --steady-only)
        g_ms *= args.steps / steps_all  # (the expressions below divide by args.steps)
        configs_idx = {24: "configs[2]", 26: "configs[3] geometry, whole polynomial on one GPU", 20: "configs[1] + open"}.get(nv, "non-headline size")
        out = {
            "metric": "Zip commit+open MCoeffs/s at 2^%d witness" % nv,
            "value": round(coeffs_per_step / step_s / 1e6, 2),
            "unit": "MCoeffs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "untimed_steps_before_value": untimed_before,  # --warmup, then the K timed COLD steps (`cold`), topped up to 12
            "ms_per_step": round(step_s * 1e3, 4),
            # the first K steps after only --warmup untimed ones (a fresh process; `value` is the steady state after them)
            "cold": {"steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt_cold / args.steps * 1e3, 4),
                     "value": round(coeffs_per_step * args.steps / dt_cold / 1e6, 2), "unit": "MCoeffs/s"},
            "higher_is_better": True,
            "scaling": "strong" if rows_mode else "weak",
            "vs_baseline": None,
            "dtype": "i64",
            "data": "synthetic (SplitMix64 full-range i64 witness; coefficient / column / point streams per SURVEY.md 8d)",
            "config": {"workload": "Zip commit+open_z 2^%d coeffs (BASELINE %s)" % (nv, configs_idx), "row_len": row_len,
                       "num_rows": num_rows, "codeword_len": cw, "column_openings": n_cols, "field_limbs": fl,
                       "parallelism": ("rows%d" % world if rows_mode else "polys%d" % world),
                       "commit": ("zip_commit + zip_open (everything stored)" if args.no_hint else
                                  "zip_commit_hinted + zip_open (the 1000 columns are known before the commit, "
                                  "prover.rs:316; stores no opening reads are skipped%s)"
                                  % ("; what the openings read of the entries and of tree levels 0..2 is stored packed"
                                     if packed else "") if args.two_calls else
                                  "zip_commit_open (one call, as commit_z_mle_and_prove_evaluation: the 1000 columns are "
                                  "known before the commit, prover.rs:316; stores no opening reads are skipped%s)"
                                  % ("; what the openings read of the entries and of tree levels 0..2 is stored packed"
                                     if packed else ""))},
            # the contract's HBM figure for the dominant kernel: SURVEY 8d algorithmic bytes / its launch time / 8 TB/s ...
            "roofline": {"bound": "valu", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         # the committed PMC pass was taken from other kernel sources than the ones this run uses
                         "traffic_stale": bool(pe) and pe.get("kernel_src_sha") != kernel_sources_sha(),
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(commit_bytes),
                         "moved_bytes_per_launch": moved,
                         "moved_gbs": round(moved / (avg_ms * 1e-3) / 1e9, 1),
                         "note": "bound = valu: %.1fM BLAKE3 compressions per launch at 678 int32 VALU instructions each; HBM is "
                                 "not what binds this kernel (roofline_valu).  `achieved` prices SURVEY 8d's full "
                                 "materialisation (200 B/coeff); `moved_bytes_per_launch` is what this build stores "
                                 "(16-byte row entries%s); `traffic` = FETCH_SIZE x2 + WRITE_SIZE of the committed PMC pass"
                                 % (per * (2 * cw - 1) / 1e6, "" if args.no_hint else ", only what the hinted openings read"
                                    + (", levels 0..2 and the entries packed" if packed else "")
                                    )},
            # ... and the roofline that does bind it
            "roofline_valu": valu,
            "roofline_gather": {"bound": "hbm", "kernel": "open_columns_kernel",
                                "algorithmic_bytes_per_step": int(gather_bytes),
                                "sum_ms_per_step": round(g_ms / args.steps, 4),
                                "achieved": round(gather_bytes / (g_ms / args.steps * 1e-3) / 1e9, 1) if g_ms else None,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gather_bytes / (g_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if g_ms else None,
                                "note": "per-chunk launches summed; they run beside the commit kernel"},
            "whole_path": {"algorithmic_bytes": int(sum(ab.values())),
                           "hbm_frac": round(sum(ab.values()) / step_s / 1e9 / HBM_PEAK_GBS, 4)},
            # (every kernel bracketed by events: a short second region; the dominant kernel: the timed region)
            "kernels_ms_per_step": dict({k: round(v[1] / steps_all, 4) for k, v in sorted(ktimes_all.items())},
                                        **{dom: round(avg_ms * launches / args.steps, 4)}),
        }
        if full_mat:
            out["full_materialisation"] = full_mat
        if two_call:
            out["two_call_unchanged_api"] = two_call
        if pipelined:
            out["pipelined"] = pipelined
        if in_flight:
            out["in_flight"] = in_flight
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(nv, args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
