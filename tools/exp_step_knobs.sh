# GPU box: one bench line per knob setting (step time, kernel times per step)
NV=${NV:-24}
run() { echo "== $*"; env "$@" python3 bench.py --num-vars $NV --no-cpu-baseline --no-pipelined --steps 10 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k['raa_commit_kernel'],'gather',k['open_columns_kernel'],'combine',k['combine_rows_kernel'], 'wait', k['wait_counter_kernel'])"; }
for rep in 1 2; do
run A=1
run ZIP_HIP_CHUNKS=16
run ZIP_HIP_CHUNKS=11
run ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,3,3,3,3,2,2,2,2
run ZIP_HIP_CHUNK_ROUNDS=2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,1,1
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,4,4,4,2,2,2,1,1
done
