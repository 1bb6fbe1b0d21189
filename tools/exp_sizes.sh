# GPU box: the driver-style bench line, then the other sizes and the un-hinted commit, one line each
show() { python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['config']['workload'][:44], '|', d['ms_per_step'], 'ms', d['value'], 'MCoeffs/s', d.get('whole_path'), d['kernels_ms_per_step'], 'two_call', (d.get('two_call_unchanged_api') or {}).get('ms_per_step'), 'jobs', (d.get('pipelined') or {}).get('ms_per_step'))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show
for nv in 26 22 20; do python3 bench.py --num-vars $nv --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined 2>/dev/null | show; done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-hint 2>/dev/null | show
