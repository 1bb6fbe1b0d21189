run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run ZIP_HIP_CHUNKS=1 ZIP_HIP_COMBINE_LAST=1
run ZIP_HIP_CHUNKS=2
run ZIP_HIP_CHUNKS=4
run ZIP_HIP_CHUNKS=8
run ZIP_HIP_CHUNKS=4 ZIP_HIP_NO_PRIORITY=1
