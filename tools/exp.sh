run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
OLD=$PWD/tools/bin/libzip_hip_old.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
run ZIP_HIP_CHUNKS=1 ZIP_HIP_COMBINE_LAST=1
run ZIP_HIP_CHUNKS=1 ZIP_HIP_COMBINE_LAST=1 ZIP_HIP_LIB_PATH=$OLD
run ZIP_HIP_CHUNKS=4
run ZIP_HIP_CHUNKS=4 ZIP_HIP_LIB_PATH=$OLD
done
