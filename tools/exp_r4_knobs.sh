#!/bin/bash
# GPU box, round 4: the knobs that lost in round 3 only because of the gather's requests, re-measured with the
# row-interleaved layout.  One bench line per setting, default at both ends and in the middle.
NV=${NV:-24}
run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars $NV --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'combine',k.get('combine_rows_kernel'), 'wait', k.get('wait_counter_kernel'))"; }
run A=1
run ZIP_HIP_GATHER_PRIO=0
run ZIP_HIP_GATHER_STREAM=1
run ZIP_HIP_GATHER_STREAM=1 ZIP_HIP_GATHER_PRIO=0
run ZIP_HIP_GATHER_RPB=16
run ZIP_HIP_GATHER_RPB=64
run A=1
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,4
run ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,3,1
run ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,2,2
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,2,2
run ZIP_HIP_CHUNK_ROUNDS=2,2,2,2,2,2,2,2
run ZIP_HIP_CHUNK_ROUNDS=2,3,3,3,3,2
run A=1
run ZIP_HIP_WIDE=1
run ZIP_HIP_WIDE=1 ZIP_HIP_GATHER_STREAM=1
run ZIP_HIP_WIDE=1 ZIP_HIP_GATHER_STREAM=1 ZIP_HIP_GATHER_PRIO=0
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2
run A=1
echo "--- alone (serial): hinted"
python3 tools/kernel_times.py --hint --serial 2>/dev/null | grep -v "^kernel_src"
