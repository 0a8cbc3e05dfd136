#!/bin/bash
# PC sampling of the step kernel (rocprofv3 beta feature): where the waves' program counters are, and why they stall.
# usage (through gpurun): bash tools/pc_sample.sh <total_envs> <method: stochastic|host_trap>
set -u
N=${1:-8192}; M=${2:-stochastic}
OUT=gpurun_out/pcs_$N
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
UNIT=cycles; INT=1048576
if [ "$M" = host_trap ]; then UNIT=time; INT=100; fi
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $M --pc-sampling-interval $INT \
  --kernel-trace --output-format csv -d $OUT -- python bench.py --no-alt --no-cpu-baseline --min-seconds 0.2 --total-envs $N > $OUT/bench.json 2> $OUT/err.txt
echo "rc=$?"
tail -5 $OUT/err.txt
find $OUT -type f | head; du -sh $OUT
