# GPU box: one bench line per knob setting (step time, kernel times per step, in-kernel clock)
run() { echo "== $*"; env "$@" python3 bench.py --no-cpu-baseline --no-pipelined --steps 20 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k['raa_commit_kernel'],'gather',k['open_columns_kernel'],'combine',k['combine_rows_kernel'], 'clk', d['roofline_valu'] and d['roofline_valu']['clock_mhz'])"; }
run A=1
run ZIP_HIP_WIDE=1
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=1,2,2,2,1
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,1,1
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=1,1,1,1,1,1,1,1
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2 ZIP_HIP_GATHER_STREAM=0
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2 ZIP_HIP_GATHER_LEAN=1
run ZIP_HIP_WIDE=1 ZIP_HIP_CHUNK_ROUNDS=2,2,2,2 ZIP_HIP_GATHER_RPB=64
run A=1
