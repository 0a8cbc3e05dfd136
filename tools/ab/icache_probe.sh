#!/bin/bash
# Instruction-cache behaviour of the step kernels (each is ~340 KB of code; the L1I is 64 KB per two CUs):
# separate --pmc passes, kernel-trace only.  usage: bash tools/icache_probe.sh TAG   (through gpurun)
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05"
for N in 8192 1024; do
  KN=k_step_philox1; [ $N -lt 4096 ] && KN=k_step_philox
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/icA$N -- $BENCH --total-envs $N > /dev/null 2> $OUT/icA$N.err
  rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES -d $OUT/icB$N -- $BENCH --total-envs $N > /dev/null 2> $OUT/icB$N.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/icC$N -- $BENCH --total-envs $N > /dev/null 2> $OUT/icC$N.err
  python tools/rocpd_summary.py counters $KN $OUT/${TAG}_pmc_icache_${N}env.json $OUT/icA$N $OUT/icB$N $OUT/icC$N
  rm -rf $OUT/icA$N $OUT/icB$N $OUT/icC$N
done
