# GPU box: is a three-round LAST chunk worth its earlier... (six alternations, 40 steps each)
run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars 24 --no-cpu-baseline --steady-only --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], 'commit', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3 4 5 6; do
run A=1
run ZIP_HIP_CHUNK_ROUNDS=3,3,4,3,3
run ZIP_HIP_CHUNK_ROUNDS=3,4,3,3,3
run ZIP_HIP_CHUNK_ROUNDS=4,3,3,3,3
run ZIP_HIP_CHUNK_ROUNDS=3,3,3,4,3
done
