cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -C oracle > /dev/null 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t6.log
tail -3 gpurun_out/t6.log
grep -q "rc=0" gpurun_out/t6.log || exit 1
one() {  # label, env...
    local label="$*"
    env "$@" python3 bench.py --num-vars ${NV:-24} --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step']
print('$label', d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'wait',k.get('wait_counter_kernel'))"
}
( for rep in 1 2 3 4; do
one A=1
one ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,4
one ZIP_HIP_LIB_PATH=$PWD/zinc_amd/lib/libzip_hip_prev.so
done
for NV in 20 22; do export NV; for rep in 1 2; do one NV=$NV; one NV=$NV ZIP_HIP_LIB_PATH=$PWD/zinc_amd/lib/libzip_hip_prev.so; done; done ) > gpurun_out/ab6.log 2>&1
cat gpurun_out/ab6.log
