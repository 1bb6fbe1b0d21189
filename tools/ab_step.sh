#!/bin/bash
# GPU box: the step time of two builds of libzip_hip.so, alternated on ONE box (box-to-box spread is 3-5 %).
#   tools/ab_step.sh [reps] [old-lib] -- extra env assignments apply to both
# prints per run: ms/step, commit kernel (in step), gather sum, combine, wait
NV=${NV:-24}
REPS=${1:-3}
OLD=${2:-zinc_amd/lib/libzip_hip_prev.so}
one() {  # label, env...
    local label=$1; shift
    env "$@" python3 bench.py --num-vars $NV --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step']
t=d.get('two_call_unchanged_api') or {}
print('$label', d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'combine',k.get('combine_rows_kernel'),'wait',k.get('wait_counter_kernel'),'two_call',t.get('ms_per_step'), t.get('proof_identical_to_one_call'))"
}
for rep in $(seq $REPS); do
    one "new" A=1
    [ -f "$OLD" ] && one "old" ZIP_HIP_LIB_PATH=$PWD/$OLD
done
