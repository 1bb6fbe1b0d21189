#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces + stats of the bench (2^24 hinted = default, 2^24 plain,
# 2^26), the PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass; SQ_INSTS_VALU) of hinted and
# plain commit + open at 2^24 and 2^26, and the FETCH_SIZE calibration; condensed into profiles/<tag>_*.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh round2 > gpurun_out/prof.log 2>&1; tail -40 gpurun_out/prof.log'
# "full" as second argument also re-traces the verifier, the sumcheck prover and the whole ZincProver.
set -e
TAG="${1:-round2}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nohint -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hint > $OUT/bench_trace_nohint.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace26 -- python3 bench.py --num-vars 26 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_trace26.log 2>&1
python3 tools/pmc_summary.py --tag "$TAG" --trace $OUT/trace
python3 tools/pmc_summary.py --tag "$TAG" --name-suffix _nohint --trace $OUT/trace_nohint
python3 tools/pmc_summary.py --tag "$TAG" --name-suffix _2pow26 --trace $OUT/trace26
# counter collection serialises kernel dispatch (no commit/open pipelining in these passes): one chunk
export ZIP_HIP_CHUNKS=1
for NV in 24 26; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/fetch$NV.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/write$NV.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/sq$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/sq$NV.log 2>&1
  SUF=""; [ $NV = 26 ] && SUF="_2pow26"
  python3 tools/pmc_summary.py --tag "$TAG" --name-suffix "$SUF" --num-vars $NV --fetch $OUT/fetch$NV --write $OUT/write$NV --sq $OUT/sq$NV > /dev/null
done
unset ZIP_HIP_CHUNKS
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal -- tools/ubench_fetchcal > $OUT/cal.log 2>&1
python3 tools/pmc_summary.py --tag "$TAG" --cal $OUT/cal
if [ "$2" = "full" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/verify -- python3 tools/verify_times.py --reps 5 > $OUT/verify.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck -- python3 tools/sumcheck_times.py 24 > $OUT/sumcheck.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prover -- python3 tools/zinc_prover_times.py 20 --reps 3 > $OUT/prover.log 2>&1
  python3 tools/pmc_summary.py --tag "$TAG" --verify $OUT/verify --sumcheck $OUT/sumcheck --prover $OUT/prover
fi
mkdir -p $OUT/profiles && cp profiles/* $OUT/profiles/ 2>/dev/null || true
for f in $OUT/bench_trace.log $OUT/bench_trace_nohint.log $OUT/bench_trace26.log; do tail -1 $f | cut -c1-400; done
cat profiles/${TAG}_pmc.md profiles/${TAG}_2pow26_pmc.md profiles/${TAG}_fetch_calibration.md
