run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'wait' not in k})"; }
ZIP_HIP_GATHER_DIRECT=8 timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
ovl() { echo "== $*"; env "$@" ZIP_HIP_CHUNKS=1 timeout -k 10 300 python tools/exp_overlap.py 2>&1 | grep -v amdgpu.ids; }
ovl ZIP_HIP_GATHER_DIRECT=4
ovl ZIP_HIP_GATHER_DIRECT=8
ovl ZIP_HIP_GATHER_DIRECT=16
ovl ZIP_HIP_GATHER_DIRECT=8 ZIP_HIP_GATHER_RPB=128
ovl ZIP_HIP_GATHER_DIRECT=8 ZIP_HIP_GATHER_PRIO=0
run ZIP_HIP_CHUNKS=4
run ZIP_HIP_CHUNKS=4 ZIP_HIP_GATHER_DIRECT=8
run ZIP_HIP_CHUNKS=4 ZIP_HIP_GATHER_DIRECT=4
run ZIP_HIP_CHUNKS=8 ZIP_HIP_GATHER_DIRECT=8
