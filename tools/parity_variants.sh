#!/bin/bash
# GPU box: the full-size parity tests (2^22 / 2^24 / 2^26 against the oracle) and the one-call parity tests under the knobs
# that select OTHER code paths than the defaults -- chunk schedules (single-round chunks, consecutive chunk ends, one
# chunk: the deferred stages of ChunkFinisher in every combination) and gather forms (two passes per workgroup, no LDS
# image, the general path for blocks that are not whole passes).  The knobs are read once per process, so each
# variant is its own pytest run.
#   gpurun --timeout 1100 -- 'bash tools/parity_variants.sh > gpurun_out/parity_variants.log 2>&1; cat gpurun_out/parity_variants.log'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -C oracle > /dev/null 2>&1
K="2pow24 or 2pow22 or 2pow26 or one_call_is_byte_identical or priority_classes"
rc=0
for s in "ZIP_HIP_CHUNK_ROUNDS=1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1" "ZIP_HIP_CHUNK_ROUNDS=5,4,3,2,1,1" "ZIP_HIP_CHUNK_ROUNDS=15,1" \
         "ZIP_HIP_CHUNKS=1" "ZIP_HIP_CHUNKS=3" "ZIP_HIP_GATHER_RPB=64" "ZIP_HIP_GATHER_STREAM=1" \
         "ZIP_HIP_GATHER_STREAM=1 ZIP_HIP_GATHER_RPB=64" "ZIP_HIP_GATHER_RPB=96" "ZIP_HIP_GATHER_RPB=20"; do
  echo "== $s"
  env $s timeout -k 10 400 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -n 1
  [ ${PIPESTATUS[0]} -eq 0 ] || rc=1
done
exit $rc
