cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
one() {  # label, env...
    local label="$*"
    env "$@" python3 bench.py --num-vars 24 --no-cpu-baseline --no-pipelined --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read());k=d['kernels_ms_per_step']
print('$label', d['ms_per_step'], 'commit',k.get('raa_commit_kernel'),'gather',k.get('open_columns_kernel'),'wait',k.get('wait_counter_kernel'))"
}
for rep in 1 2 3 4; do
for s in "A=1" "ZIP_HIP_CHUNK_ROUNDS=4,4,4,3,1" "ZIP_HIP_CHUNK_ROUNDS=6,4,3,2,1" "ZIP_HIP_CHUNK_ROUNDS=5,4,4,2,1" "ZIP_HIP_CHUNK_ROUNDS=5,5,3,2,1" "ZIP_HIP_CHUNK_ROUNDS=5,4,3,2,2" "ZIP_HIP_CHUNK_ROUNDS=4,4,3,3,2" "ZIP_HIP_CHUNK_ROUNDS=5,5,4,2" "ZIP_HIP_CHUNK_ROUNDS=6,5,4,1" "ZIP_HIP_CHUNK_ROUNDS=3,3,3,3,3,1"; do
  one $s
done
done
