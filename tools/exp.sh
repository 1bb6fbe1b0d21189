run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
NS=$PWD/tools/bin/libzip_hip_nostore.so
for i in 1 2; do
run ZIP_HIP_CHUNKS=1 ZIP_HIP_COMBINE_LAST=1
run ZIP_HIP_CHUNKS=1 ZIP_HIP_COMBINE_LAST=1 ZIP_HIP_LIB_PATH=$NS
run ZIP_HIP_CHUNKS=4 ZIP_HIP_LIB_PATH=$NS
done
