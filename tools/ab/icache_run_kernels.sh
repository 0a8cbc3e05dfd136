#!/bin/bash
# r05: instruction-cache counters of the ONE-LAUNCH kernels (their waves do not walk the code together as a per-step launch's do) beside the per-step kernel.
# usage: bash tools/ab/icache_run_kernels.sh   (through gpurun; separate --pmc passes, kernel trace only)
set -u
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05 --steps 100"
for cfg in "8192 k_run_philox1 1" "8192 k_step_philox1 0" "1024 k_run_philox 1" "4096 k_run_philox1m 1"; do
  set -- $cfg; N=$1; KN=$2; P=$3
  export CC4_PERSIST=$P CC4_MULTISTEP=$P CC4_RUN1=$P
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/icA -- $BENCH --total-envs $N > /dev/null 2> $OUT/icA.err
  rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES -d $OUT/icB -- $BENCH --total-envs $N > /dev/null 2> $OUT/icB.err
  python tools/rocpd_summary.py counters $KN $OUT/r05_pmc_icache_${KN}_${N}env.json $OUT/icA $OUT/icB
  rm -rf $OUT/icA $OUT/icB
done
