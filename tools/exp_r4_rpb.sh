run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars 24 --no-cpu-baseline --steady-only --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], 'commit', d['roofline']['avg_launch_ms'], 'cold', d['cold']['ms_per_step'])"; }
for rep in 1 2 3; do
run A=1
run ZIP_HIP_GATHER_RPB=32
run ZIP_HIP_GATHER_RPB=24
run ZIP_HIP_GATHER_RPB=28
done
