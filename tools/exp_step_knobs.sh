# GPU box: one bench line per knob setting (step time, kernel times per step, in-kernel clock)
run() { echo "== $*"; env "$@" python3 bench.py --no-cpu-baseline --no-pipelined --steps 20 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step'];print(d['ms_per_step'], 'commit',k['raa_commit_kernel'],'gather',k['open_columns_kernel'],'combine',k['combine_rows_kernel'], 'fold', k['combine_finalize_kernel'], 'two', d['two_call_unchanged_api']['ms_per_step'])"; }
run A=1
for rep in 1 2 3; do
run ZIP_HIP_COMBINE=split
run ZIP_HIP_COMBINE=first
run ZIP_HIP_COMBINE=first ZIP_HIP_COMBINE_PRIO=0
run ZIP_HIP_COMBINE=aux
done
