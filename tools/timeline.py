#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV directory (GPU box or here):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
   python3 tools/timeline.py gpurun_out/tl [step_index | median]
Prints, for one commit+open step, every kernel's start / end relative to the start of the step's commit kernel."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    arg = sys.argv[2] if len(sys.argv) > 2 else "-2"
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Stream_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "raa_commit" in r[2]]
    if not starts:
        print("no commit kernel in trace")
        return
    if arg == "median":  # the step whose commit kernel has the median duration among the last 20 (a typical steady step)
        tail = starts[-21:-1] if len(starts) > 21 else starts[:-1] or starts
        by_dur = sorted(tail, key=lambda i: rows[i][1] - rows[i][0])
        which = starts.index(by_dur[len(by_dur) // 2]) - len(starts)
    else:
        which = int(arg)
    i0 = starts[which]
    i1 = starts[which + 1] if which + 1 < 0 and which + 1 + len(starts) < len(starts) else len(rows)
    try:
        i1 = starts[starts.index(i0) + 1]
    except IndexError:
        i1 = len(rows)
    t0 = rows[i0][0]
    prev_end = max((r[1] for r in rows[:i0]), default=t0)
    print(f"gap since the previous step's last kernel end: {(t0 - prev_end) / 1e3:8.1f} us")
    last = t0
    for s, e, n, st in rows[i0:i1]:
        print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  {n}")
        last = max(last, e)
    print(f"step kernels span {(last - t0) / 1e3:.1f} us")
    # step-to-step period over the trace
    per = [(rows[starts[k + 1]][0] - rows[starts[k]][0]) / 1e3 for k in range(len(starts) - 1)]
    print("commit-start to commit-start periods (us):", [round(p, 1) for p in per])


if __name__ == "__main__":
    main()
