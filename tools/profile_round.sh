#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench, then the two
# PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass) and an SQ pass for the VALU
# utilisation of the commit kernel, then kernel traces of the verifier, the sumcheck prover and the full
# ZincProver at 2^20 (BASELINE configs[4]);
# condensed into profiles/<tag>_*.
set -e
TAG="${1:-round1}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
# (counter collection serialises kernel dispatch: no commit/open pipelining in these passes)
export ZIP_HIP_CHUNKS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/kernel_times.py --reps 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/kernel_times.py --reps 2 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 tools/kernel_times.py --reps 2 > $OUT/sq.log 2>&1
unset ZIP_HIP_CHUNKS
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/verify -- python3 tools/verify_times.py --reps 5 > $OUT/verify.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck -- python3 tools/sumcheck_times.py 24 > $OUT/sumcheck.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prover -- python3 tools/zinc_prover_times.py 20 --reps 3 > $OUT/prover.log 2>&1
python3 tools/pmc_summary.py --prover $OUT/prover --trace $OUT/trace --fetch $OUT/fetch --write $OUT/write --sq $OUT/sq --verify $OUT/verify --sumcheck $OUT/sumcheck --tag "$TAG" --num-vars 24
cp profiles/* $OUT/ 2>/dev/null || true
tail -1 $OUT/bench_trace.log | cut -c1-600
tail -7 $OUT/verify.log; tail -2 $OUT/sumcheck.log; tail -3 $OUT/prover.log
