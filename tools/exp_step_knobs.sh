# GPU box: one bench line per knob setting (step time, kernel times per step, in-kernel clock)
run() { echo "== $*"; env "$@" python3 bench.py --no-cpu-baseline --no-pipelined --steps 20 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k['raa_commit_kernel'],'gather',k['open_columns_kernel'],'combine',k['combine_rows_kernel'], 'clk', d['roofline_valu'] and d['roofline_valu']['clock_mhz'])"; }
for rep in 1 2; do
run A=1
run ZIP_HIP_CHUNK_ROUNDS=2,2,4,4,4
run ZIP_HIP_CHUNK_ROUNDS=2,3,3,4,4
run ZIP_HIP_CHUNK_ROUNDS=2,2,3,3,3,3
run ZIP_HIP_CHUNK_ROUNDS=1,2,3,3,3,4
run ZIP_HIP_CHUNK_ROUNDS=2,2,4,8
run ZIP_HIP_CHUNK_ROUNDS=2,2,3,4,5
run ZIP_HIP_CHUNK_ROUNDS=2,3,4,4,3
run ZIP_HIP_CHUNK_ROUNDS=1,1,2,4,4,4
run ZIP_HIP_CHUNK_ROUNDS=2,4,4,4,2
done
run A=1
