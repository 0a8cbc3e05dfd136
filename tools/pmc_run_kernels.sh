#!/bin/bash
# HBM traffic of the ONE-LAUNCH kernels the timed regions run: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only; a launch =
# K steps of the whole batch, bytes per step = bytes of the launch / K (rocpd_summary.py pmc_run).  Part of tools/profile_round.sh.
# usage: bash tools/pmc_run_kernels.sh [tag]   (through gpurun)
set -u
TAG=${1:-r05}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05"
rm -f $OUT/${TAG}_pmc.json
for NK in "8192 100" "1024 100" "32768 50" "4096 100"; do
  set -- $NK; N=$1; K=$2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch$N -- $BENCH --total-envs $N --steps $K --warmup 5 > /dev/null 2> $OUT/fetch$N.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write$N -- $BENCH --total-envs $N --steps $K --warmup 5 > $OUT/bench_pmc$N.json 2> $OUT/write$N.err
  KN=$(python -c "import json,sys; d=[json.loads(l) for l in open('$OUT/bench_pmc$N.json') if l.startswith('{')][0]; print(d['roofline']['kernel'])")
  python tools/rocpd_summary.py pmc_run $OUT/fetch$N $OUT/write$N $KN $OUT/${TAG}_pmc.json $N $K > /dev/null
  rm -rf $OUT/fetch$N $OUT/write$N
done
cat $OUT/${TAG}_pmc.json
