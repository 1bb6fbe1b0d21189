#!/bin/bash
# GPU box: counter passes on open_columns_kernel ALONE at 2^24 (commit waited for before the open: --serial), one
# rocprofv3 --pmc run per counter group.  Raw CSVs -> gpurun_out/gather_pmc/<group>/.
#   gpurun --timeout 900 -- 'bash tools/gather_pmc.sh > gpurun_out/gather_pmc.log 2>&1'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gather_pmc; rm -rf $OUT; mkdir -p $OUT
make -C oracle > /dev/null 2>&1
export ZIP_HIP_CHUNKS=1
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
run() {  # name, counters...
  local name=$1; shift
  # (a counter set the hardware cannot collect makes rocprofv3 abort and the process linger: bounded)
  timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/kernel_times.py --num-vars 24 --hint --serial --reps 2 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcc2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_STALL_sum
run tcc3 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run ta2 TA_BUSY_avr TA_FLAT_WRITE_WAVEFRONTS_sum
run fw FETCH_SIZE
run ww WRITE_SIZE
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/gather_pmc"
rows = collections.OrderedDict()
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "open_columns" not in k and "raa_commit" not in k:
                continue
            key = (k[:60], r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
        disp = collections.Counter()
        for (k, c), (v, n) in acc.items():
            rows[(k, c)] = (v, n)
# per-dispatch average: a counter row appears once per dispatch (per XCD dims are summed by rocprofv3 in 'sum' metrics)
print("| kernel | counter | per launch |\n|---|---|---|")
for (k, c), (v, n) in rows.items():
    print(f"| `{k}` | {c} | {v / max(n,1):,.0f} |")
PY
