#!/bin/bash
run() { local flags="$1"; shift; echo -n "[$flags] $* : "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $flags 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
for i in 1 2 3 4; do
run "" X=0
run "--two-calls" X=0
done
