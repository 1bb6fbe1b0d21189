#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the tracked summaries under profiles/.

  tools/pmc_summary.py --tag round2 --num-vars 24 --trace DIR [--fetch DIR --write DIR --sq DIR] [--cal DIR]
                       [--name-suffix _2pow26] [--verify DIR --sumcheck DIR --prover DIR]

* <tag><suffix>_kernel_stats.csv : the --kernel-trace --stats table (per-kernel calls / total / average ns)
* <tag><suffix>_pmc.md           : FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU per kernel and launch
* pmc_traffic.json               : per (kernel, num_vars, mode) HBM bytes and VALU wave-instructions per launch, read by
                                   bench.py for `roofline.traffic` and `roofline_valu.insts` (merged, not overwritten)
* <tag>_fetch_calibration.md     : tools/ubench_fetchcal under --pmc FETCH_SIZE: what the counter reads for the
                                   access patterns of this library (--cal)
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha():
    """Digest of the kernel sources the counters were taken from: bench.py flags a pmc_traffic.json entry whose digest is
    not that of the sources it runs with (`traffic_stale`)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kernels_commit.cuh", "kernels_open.cuh", "blake3.cuh", "blake3_sched.inc"):
        with open(os.path.join(ROOT, "zinc_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_sha(d):
    """The digest tools/kernel_times.py printed into the log of the counter pass itself (`<dir>.log`, written by
    tools/profile_round.sh): what was MEASURED, not what the tree holds when the summary is written."""
    if not d:
        return None
    try:
        for line in open(d.rstrip("/") + ".log"):
            if line.startswith("kernel_src_sha "):
                return line.split()[1]
    except OSError:
        pass
    return None


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def pmc(d, counter):
    """{kernel name: [value per dispatch]}, summed over the counter's dimensions (XCDs ...) per dispatch."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    f = find(d, "*counter_collection.csv") if d else None
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def avg(x):
    return sum(x) / len(x) if x else 0.0


def classify(kernel):
    """(short name, mode) of the kernels bench.py prices."""
    k = kernel
    if "raa_commit16_kernel<" in k or "raa_commit_kernel<" in k:
        args = k[k.index("<") + 1:k.index(">")].replace(" ", "").split(",")
        if args[1] != "true":  # encode only
            return None
        # raa_commit_kernel<E, HASH, MODE> / raa_commit16_kernel<T, HASH, MODE>: MODE 0 everything stored, 1 hinted at
        # the natural places, 3 hinted and packed
        mode = {"0": "plain", "1": "hinted", "3": "packed"}.get(args[2] if len(args) > 2 else "0", "plain")
        return "raa_commit_kernel", mode
    if "open_columns_kernel" in k or "open_columns_stream_kernel" in k or "open_columns_ilv_kernel" in k:
        return "open_columns_kernel", "any"
    return None


def steady_stats(trace_dir, bench_log, out_csv):
    """Per-kernel stats of the STEADY steps only, from the kernel trace itself (rocprofv3's own *_kernel_stats.csv averages
    every launch of the process, cold warm-up launches included -- round-3 verdict: its average was 8 % off the bench's).
    The traced `bench.py --steady-only` runs U untimed steps (`untimed_steps_before_value`) and then K timed ones, one
    commit launch per step: every dispatch that begins before the (U + 1)-th commit kernel is dropped.  Same columns as
    rocprofv3's table.  False if the trace or the log is not there."""
    tr = find(trace_dir, "*kernel_trace.csv")
    if not tr or not bench_log:
        return False
    untimed = None
    try:
        for line in open(bench_log):
            if line.startswith("{") and "untimed_steps_before_value" in line:
                untimed = json.loads(line)["untimed_steps_before_value"]
    except (OSError, ValueError):
        return False
    if untimed is None:
        return False
    rows = list(csv.DictReader(open(tr)))
    commits = sorted(int(r["Start_Timestamp"]) for r in rows if "raa_commit" in r["Kernel_Name"] and ", true" in r["Kernel_Name"].replace(",true", ", true"))
    if len(commits) <= untimed:
        return False
    t0 = commits[untimed]
    per = collections.defaultdict(list)
    for r in rows:
        if int(r["Start_Timestamp"]) >= t0:
            per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in per.values()) or 1
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            m = sum(v) / len(v)
            sd = (sum((x - m) ** 2 for x in v) / len(v)) ** 0.5
            w.writerow([k, len(v), sum(v), f"{m:.6f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v), f"{sd:.6f}"])
    with open(out_csv + ".note", "w") as fh:
        fh.write(f"steady-state launches only: every dispatch before the {untimed + 1}-th commit kernel of the traced "
                 f"`bench.py --steady-only` run dropped (tools/pmc_summary.py steady_stats)\n")
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace"), ap.add_argument("--fetch"), ap.add_argument("--write")
    ap.add_argument("--sq"), ap.add_argument("--verify"), ap.add_argument("--sumcheck"), ap.add_argument("--prover")
    ap.add_argument("--cal")
    ap.add_argument("--bench-log", help="the traced bench.py's output: its JSON line says how many steps preceded the timed ones")
    ap.add_argument("--tag", default="round2")
    ap.add_argument("--name-suffix", default="")
    ap.add_argument("--num-vars", type=int, default=24)
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    tag = a.tag + a.name_suffix
    if a.trace:
        if not steady_stats(a.trace, a.bench_log, os.path.join(prof, f"{tag}_kernel_stats.csv")):
            st = find(a.trace, "*kernel_stats.csv")
            if st:
                shutil.copy(st, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    for d, name in ((a.verify, "verify"), (a.sumcheck, "sumcheck"), (a.prover, "prover")):
        st = find(d, "*kernel_stats.csv") if d else None
        if st:
            shutil.copy(st, os.path.join(prof, f"{a.tag}_{name}_kernel_stats.csv"))
    if a.cal:
        fe = pmc(a.cal, "FETCH_SIZE")
        useful = {"cal_stream16": 4 << 20, "cal_stream8": 4 << 20, "cal_gather32": None}
        lines = [f"# {a.tag}: what FETCH_SIZE counts (tools/ubench_fetchcal under rocprofv3 --pmc FETCH_SIZE)", "",
                 "Every byte is read exactly once from a 4 GiB buffer (16x the Infinity Cache).  `useful` = bytes the",
                 "kernel asked for; ratio = FETCH_SIZE / useful.", "",
                 "| kernel (dispatch) | pattern | useful KiB | FETCH_SIZE KiB | ratio |", "|---|---|---|---|---|"]
        pats = {"cal_stream16": ["16 B per lane, streaming"], "cal_stream8": ["8 B per lane, streaming"],
                "cal_gather32": ["2 lanes x 16 B per 32-byte node, nodes 128 B apart, random order",
                                 "2 lanes x 16 B per 32-byte node, nodes 64 B apart, random order"]}
        usef = {"cal_stream16": [4 << 20], "cal_stream8": [4 << 20], "cal_gather32": [1 << 20, 2 << 20]}
        for k, vals in sorted(fe.items()):
            short = next((s for s in pats if s in k), None)
            if not short:
                continue
            for i, v in enumerate(vals):
                u = usef[short][min(i, len(usef[short]) - 1)]
                lines.append(f"| `{short}` ({i}) | {pats[short][min(i, len(pats[short]) - 1)]} | {u:,} | {v:,.0f} | {v / u:.3f} |")
        with open(os.path.join(prof, f"{a.tag}_fetch_calibration.md"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        print("\n".join(lines))
    fetch, write = pmc(a.fetch, "FETCH_SIZE"), pmc(a.write, "WRITE_SIZE")
    insts = pmc(a.sq, "SQ_INSTS_VALU")
    active, wcyc = pmc(a.sq, "SQ_ACTIVE_INST_VALU"), pmc(a.sq, "SQ_WAVE_CYCLES")
    shas = {x for x in (measured_sha(a.fetch), measured_sha(a.write), measured_sha(a.sq)) if x}
    sha = shas.pop() if len(shas) == 1 else None  # (passes taken from different sources: no digest)
    if fetch or write or insts:
        lines = [f"# {tag}: per-launch counters from rocprofv3 --pmc (separate passes), num_vars = {a.num_vars}", "",
                 "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; how to read FETCH_SIZE for each access pattern",
                 f"is measured in `{a.tag}_fetch_calibration.md`.  SQ_INSTS_VALU = VALU wave-instructions issued;",
                 "SQ_ACTIVE_INST_VALU = quad-cycles a VALU instruction was executing.  (SQ_WAVE_CYCLES is collected but NOT",
                 "tabulated as residency: on this chip it does not measure it -- profiles/round3_wg_spread.md -- and the",
                 "ratio round 3 printed from it exceeded 1.)  Kernel sources measured: " + (f"`{sha}`" if sha else "(not recorded)") + ".", "",
                 "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU |",
                 "|---|---|---|---|---|---|"]
        path = os.path.join(prof, "pmc_traffic.json")
        try:
            traffic = json.load(open(path))
        except (OSError, ValueError):
            traffic = {}
        for k in sorted(set(fetch) | set(write) | set(insts)):
            fk, wk, ik = avg(fetch.get(k, [])), avg(write.get(k, [])), avg(insts.get(k, []))
            n = max(len(fetch.get(k, [])), len(write.get(k, [])), len(insts.get(k, [])))
            ak, ck = avg(active.get(k, [])), avg(wcyc.get(k, []))
            lines.append(f"| `{k[:80]}` | {n} | {fk:,.0f} | {wk:,.0f} | {ik:,.0f} | {ak:,.0f} |")
            cl = classify(k)
            if cl:
                traffic[f"{cl[0]}:{a.num_vars}:{cl[1]}"] = {
                    "kernel": k[:120], "num_vars": a.num_vars, "mode": cl[1], "fetch_kib": fk, "write_kib": wk,
                    "valu_insts": ik, "valu_active_quadcycles": ak, "wave_quadcycles": ck, "source": f"profiles/{tag}_pmc.md",
                    # the digest the counter pass itself printed (tools/kernel_times.py); None = not recorded at
                    # measurement time: bench.py then reports the entry as stale
                    "kernel_src_sha": sha}
        with open(os.path.join(prof, f"{tag}_pmc.md"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        with open(path, "w") as fh:
            json.dump(traffic, fh, indent=1, sort_keys=True)
        print("\n".join(lines))


if __name__ == "__main__":
    main()
