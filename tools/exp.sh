run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
for i in 1 2 3; do
run ZIP_HIP_CHUNK_ROUNDS=4,4,4,4
run ZIP_HIP_X=default
run ZIP_HIP_CHUNK_ROUNDS=6,5,3,2
run ZIP_HIP_CHUNK_ROUNDS=5,5,4,2
run ZIP_HIP_CHUNK_ROUNDS=4,4,3,3,2
run ZIP_HIP_CHUNK_ROUNDS=6,6,3,1
done
