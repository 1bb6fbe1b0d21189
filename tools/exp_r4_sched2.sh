# GPU box: chunk schedules again, with 32 rows per gather workgroup for the last chunk too (ZIP_HIP_GATHER_RPB=32)
export ZIP_HIP_GATHER_RPB=32
run() { echo -n "$* : "; env "$@" python3 bench.py --num-vars 24 --no-cpu-baseline --steady-only --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], 'commit', d['roofline']['avg_launch_ms'], 'cold', d['cold']['ms_per_step'])"; }
for rep in 1 2 3; do
run A=1
run ZIP_HIP_CHUNK_ROUNDS=3,3,4,3,3
run ZIP_HIP_CHUNK_ROUNDS=2,4,4,4,2
run ZIP_HIP_CHUNK_ROUNDS=3,3,4,4,2
run ZIP_HIP_CHUNK_ROUNDS=2,3,4,4,3
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,4
run ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,2,2
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,3,1
done
