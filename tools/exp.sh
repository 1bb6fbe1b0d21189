#!/bin/bash
# A/B driver for experiments on the GPU box:  gpurun -- 'bash tools/exp.sh > gpurun_out/exp.log 2>&1'
# run <ENV=..>...  : one bench line (value, ms/step, per-kernel ms) under the given environment
# Knobs: ZIP_HIP_CHUNKS=n (pipeline chunks), ZIP_HIP_COMBINE=aux|first|last (where the row combinations run),
#        ZIP_HIP_GATHER_PRIO=0|1, ZIP_HIP_NO_PRIORITY=1, ZIP_HIP_LIB_PATH=<other build of libzip_hip.so>
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
for i in 1 2 3; do
run ZIP_HIP_CHUNKS=4
run ZIP_HIP_CHUNKS=4 ZIP_HIP_COMBINE=aux
run ZIP_HIP_CHUNKS=1
done
