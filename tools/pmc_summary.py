#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the tracked summaries under profiles/.

  tools/pmc_summary.py --trace DIR --fetch DIR --write DIR --tag round1 --num-vars 24

* <tag>_kernel_stats.csv : the --kernel-trace --stats table (per-kernel calls / total / average ns)
* <tag>_pmc.md           : FETCH_SIZE / WRITE_SIZE per kernel, gfx950 read correction applied
* pmc_traffic.json       : HBM bytes per launch of the dominant kernel, read by bench.py (`roofline.traffic`)
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def pmc(d, counter):
    out = collections.defaultdict(list)
    f = find(d, "*counter_collection.csv")
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace"), ap.add_argument("--fetch"), ap.add_argument("--write")
    ap.add_argument("--sq"), ap.add_argument("--verify"), ap.add_argument("--sumcheck"), ap.add_argument("--prover")
    ap.add_argument("--tag", default="round1")
    ap.add_argument("--num-vars", type=int, default=24)
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    if a.trace:
        st = find(a.trace, "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(prof, f"{a.tag}_kernel_stats.csv"))
    for d, name in ((a.verify, "verify"), (a.sumcheck, "sumcheck"), (a.prover, "prover")):
        st = find(d, "*kernel_stats.csv") if d else None
        if st:
            shutil.copy(st, os.path.join(prof, f"{a.tag}_{name}_kernel_stats.csv"))
    fetch, write = pmc(a.fetch, "FETCH_SIZE") if a.fetch else {}, pmc(a.write, "WRITE_SIZE") if a.write else {}
    lines = [f"# {a.tag}: HBM traffic per launch from rocprofv3 --pmc (separate passes)", "",
             "FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide",
             "streaming read (MI355X_MICROARCH.md, HBM section): the `read x2` column applies that correction; for",
             "the 32-byte gathers of open_columns_kernel the uncorrected value already matches the distinct-line",
             "estimate of DESIGN.md, so both are shown.", "",
             "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes (read x2 + write) | bytes (read x1 + write) |",
             "|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        fk = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        wk = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        b2, b1 = int((2 * fk + wk) * 1024), int((fk + wk) * 1024)
        lines.append(f"| `{k[:70]}` | {len(fetch.get(k, write.get(k, [])))} | {fk:,.0f} | {wk:,.0f} | {b2:,} | {b1:,} |")
        short = "raa_commit_kernel" if "raa_commit_kernel<" in k and ", true>" in k else (
            "open_columns_kernel" if "open_columns_kernel" in k else None)
        if short:
            traffic[short] = {"num_vars": a.num_vars, "fetch_kib": fk, "write_kib": wk,
                              "traffic_bytes": b2 if short == "raa_commit_kernel" else b1,
                              "source": f"profiles/{a.tag}_pmc.md"}
    if a.sq:
        # VALU utilisation of each kernel: issued VALU wave-instructions (one per 4 SIMD cycles) against the
        # SIMD cycles of its launch, SQ_BUSY_CYCLES being per shader engine x 4 SIMDs ... the ratio that is
        # unit-free is SQ_ACTIVE_INST_VALU (quad-cycles a VALU instruction was executing) / SQ_WAVE_CYCLES
        # (quad-cycles waves were resident) x waves per SIMD; both are listed as collected.
        sq = {c: pmc(a.sq, c) for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU")}
        lines += ["", "## SQ counters per launch (one pass)", "",
                  "| kernel | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES |", "|---|---|---|---|---|"]
        for k in sorted(set().union(*[set(v) for v in sq.values()])):
            avg = lambda c: sum(sq[c].get(k, [0])) / max(len(sq[c].get(k, [])), 1)
            lines.append(f"| `{k[:70]}` | {avg('SQ_INSTS_VALU'):,.0f} | {avg('SQ_ACTIVE_INST_VALU'):,.0f} | "
                         f"{avg('SQ_WAVE_CYCLES'):,.0f} | {avg('SQ_BUSY_CYCLES'):,.0f} |")
    with open(os.path.join(prof, f"{a.tag}_pmc.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic:
        with open(os.path.join(prof, "pmc_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
