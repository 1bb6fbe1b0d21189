#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces + stats of the bench (2^24 hinted = default, 2^24 plain,
# 2^26, 2^20), the PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass; SQ_INSTS_VALU,
# SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES) of hinted and plain commit + open at 2^24 and 2^26, the FETCH_SIZE calibration,
# the per-workgroup stamps of the commit kernel, the issue-rate micro-benchmarks; condensed into profiles/<tag>_*.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh round4 > gpurun_out/prof.log 2>&1; tail -40 gpurun_out/prof.log'
# "full" as second argument also re-traces the verifier, the sumcheck prover and the whole ZincProver.
# (every step appends to gpurun_out/prof_progress.log: a run that writes nothing for 7 minutes is taken to be hung)
set -e
TAG="${1:-round4}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_prof; rm -rf $OUT; mkdir -p $OUT
P=gpurun_out/prof_progress.log; : > $P
say() { echo "$(date +%T) $*" >> $P; }
make -C oracle > /dev/null 2>&1 || true
# (--steady-only: warm-up, cold and steady one-call steps, nothing else; the summaries keep the steady launches only, so
# that AverageNs of the dominant kernel is the `avg_launch_ms` of the traced run's JSON line)
say trace24; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-only > $OUT/bench_trace.log 2>&1
say trace24-nohint; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nohint -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-only --no-hint > $OUT/bench_trace_nohint.log 2>&1
say trace26; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace26 -- python3 bench.py --num-vars 26 --steps 8 --warmup 2 --no-cpu-baseline --steady-only > $OUT/bench_trace26.log 2>&1
say trace20; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace20 -- python3 bench.py --num-vars 20 --steps 20 --warmup 5 --no-cpu-baseline --steady-only > $OUT/bench_trace20.log 2>&1
python3 tools/pmc_summary.py --tag "$TAG" --trace $OUT/trace --bench-log $OUT/bench_trace.log
python3 tools/pmc_summary.py --tag "$TAG" --name-suffix _nohint --trace $OUT/trace_nohint --bench-log $OUT/bench_trace_nohint.log
python3 tools/pmc_summary.py --tag "$TAG" --name-suffix _2pow26 --trace $OUT/trace26 --bench-log $OUT/bench_trace26.log
python3 tools/pmc_summary.py --tag "$TAG" --name-suffix _2pow20 --trace $OUT/trace20 --bench-log $OUT/bench_trace20.log
python3 tools/timeline.py $OUT/trace median > $OUT/timeline24.txt 2>&1 || true
python3 tools/timeline.py $OUT/trace20 median > $OUT/timeline20.txt 2>&1 || true
# counter collection serialises kernel dispatch (no commit/open pipelining in these passes): one chunk
export ZIP_HIP_CHUNKS=1
for NV in 24 26; do
  say pmc-fetch$NV; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/fetch$NV.log 2>&1
  say pmc-write$NV; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/write$NV.log 2>&1
  say pmc-sq$NV; rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/sq$NV -- python3 tools/kernel_times.py --num-vars $NV --both --reps 2 > $OUT/sq$NV.log 2>&1
  SUF=""; [ $NV = 26 ] && SUF="_2pow26"
  python3 tools/pmc_summary.py --tag "$TAG" --name-suffix "$SUF" --num-vars $NV --fetch $OUT/fetch$NV --write $OUT/write$NV --sq $OUT/sq$NV > /dev/null
done
unset ZIP_HIP_CHUNKS
say fetchcal; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal -- tools/ubench_fetchcal > $OUT/cal.log 2>&1
python3 tools/pmc_summary.py --tag "$TAG" --cal $OUT/cal
# per-workgroup stamps of the commit kernel (debug build of the library, built here if it did not travel)
say wg-spread
[ -f zinc_amd/lib/libzip_hip_stamps.so ] || python3 tools/wg_spread.py --build > /dev/null 2>&1
: > $OUT/wg_spread.md
ZIP_HIP_LIB_PATH=$PWD/zinc_amd/lib/libzip_hip_stamps.so python3 tools/wg_spread.py --alone --md $OUT/wg_spread.md > /dev/null 2>&1 || true
ZIP_HIP_LIB_PATH=$PWD/zinc_amd/lib/libzip_hip_stamps.so python3 tools/wg_spread.py --md $OUT/wg_spread.md > /dev/null 2>&1 || true
# issue rates of the integer VALU opcodes, alone and mixed (what the fixed BLAKE3 instruction order is built on)
say ubench
for b in ubench_valu_ops ubench_valu_mix ubench_gsched ubench_blake3; do [ -x tools/$b ] && tools/$b > $OUT/$b.md 2>&1 || true; done
say mctx; python3 tools/mctx_overhead.py $OUT/mctx_overhead.md > /dev/null 2>&1 || true
if [ "$2" = "full" ]; then
  say verify; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/verify -- python3 tools/verify_times.py --reps 5 > $OUT/verify.log 2>&1
  say sumcheck; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck -- python3 tools/sumcheck_times.py 24 > $OUT/sumcheck.log 2>&1
  say prover; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prover -- python3 tools/zinc_prover_times.py 20 --reps 3 > $OUT/prover.log 2>&1
  python3 tools/pmc_summary.py --tag "$TAG" --verify $OUT/verify --sumcheck $OUT/sumcheck --prover $OUT/prover
fi
mkdir -p $OUT/profiles && cp profiles/* $OUT/profiles/ 2>/dev/null || true
say done
for f in $OUT/bench_trace.log $OUT/bench_trace_nohint.log $OUT/bench_trace26.log $OUT/bench_trace20.log; do tail -1 $f | cut -c1-400; done
cat profiles/${TAG}_pmc.md profiles/${TAG}_2pow26_pmc.md profiles/${TAG}_fetch_calibration.md
