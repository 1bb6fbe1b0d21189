run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars 24 --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'combine',k.get('combine_rows_kernel'), 'wait', k.get('wait_counter_kernel'))"; }
tools/ubench_signal
run A=1
run ZIP_HIP_CHUNK_ROUNDS=1,3,3,3,3,3
run ZIP_HIP_CHUNK_ROUNDS=1,2,3,3,3,3,1
run ZIP_HIP_CHUNK_ROUNDS=2,3,3,3,3,2
run ZIP_HIP_CHUNK_ROUNDS=2,3,3,3,3,1,1
run ZIP_HIP_CHUNK_ROUNDS=2,2,3,3,3,2,1
run ZIP_HIP_CHUNK_ROUNDS=2,4,4,4,2
run A=1
run ZIP_HIP_GATHER_STREAMS=2
run ZIP_HIP_GATHER_STREAMS=2 ZIP_HIP_CHUNK_ROUNDS=2,3,3,3,3,2
run ZIP_HIP_GATHER_STREAMS=2 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2,2,2,2,2
run A=1
