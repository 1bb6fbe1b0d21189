# GPU box: one bench line per knob setting (step time, kernel times per step, in-kernel clock)
run() { echo "== $*"; env "$@" python3 bench.py --no-cpu-baseline --no-pipelined --steps 20 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k['raa_commit_kernel'],'gather',k['open_columns_kernel'],'combine',k['combine_rows_kernel'], 'clk', d['roofline_valu'] and d['roofline_valu']['clock_mhz'])"; }
run A=1
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=32
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=16
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=8
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=16 ZIP_HIP_GATHER_PRIO=0
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=32 ZIP_HIP_GATHER_PRIO=0
run A=1
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=16
run ZIP_HIP_GATHER_LEAN=1 ZIP_HIP_GATHER_RPB=24
run A=1
