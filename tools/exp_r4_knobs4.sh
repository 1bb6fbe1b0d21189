run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars ${NV:-24} --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];t=d.get('two_call_unchanged_api') or {};print(d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'combine',k.get('combine_rows_kernel'), 'wait', k.get('wait_counter_kernel'),'two_call',t.get('ms_per_step'), t.get('proof_identical_to_one_call'))"; }
for rep in 1 2; do
run A=1
run ZIP_HIP_GATHER_SC1=0
run ZIP_HIP_PER_ROUND=0
run ZIP_HIP_PER_ROUND=0 ZIP_HIP_GATHER_SC1=0
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=2,3,3,3,2,1,1,1
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,3,1
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,2,2
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2,2,2,2,1,1
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=4,4,4,2,1,1
run ZIP_HIP_GATHER_SC1=0 ZIP_HIP_CHUNK_ROUNDS=1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1
run ZIP_HIP_GATE=1
run ZIP_HIP_GATE=1 ZIP_HIP_CHUNK_ROUNDS=1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1
done
