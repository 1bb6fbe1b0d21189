run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
ZIP_HIP_SPLIT=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
run ZIP_HIP_CHUNKS=4
run ZIP_HIP_SPLIT=1 ZIP_HIP_COMBINE_LAST=1
run ZIP_HIP_SPLIT=2
run ZIP_HIP_SPLIT=4
run ZIP_HIP_SPLIT=8
run ZIP_HIP_SPLIT=16
run ZIP_HIP_SPLIT=8 ZIP_HIP_GATHER_PRIO=0
