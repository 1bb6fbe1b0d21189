#!/bin/bash
# A/B of environment knobs on the default bench (GPU box): edit the `run` lines, then
#   gpurun -- 'bash tools/exp.sh > gpurun_out/exp.log 2>&1; cat gpurun_out/exp.log'
# Box-to-box and run-to-run noise is 3-5 %: alternate the variants and repeat.
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
for i in 1 2 3 4; do
run ZIP_HIP_NO_COMPACT_ROWS=1
run ZIP_HIP_X=compact
done
