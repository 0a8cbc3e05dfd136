#!/bin/bash
# r06: runs of steps (RunArgs.SA ..: the agent part stays in LDS inside a run) on the per-CU partitions, 6 waves per SIMD; one box.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_runs_ab.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
export CC4_LIB=$PWD/build_var/runs6.so
export CC4_PERSIST_POOL=0
echo "## self-check (CC4_PERSIST_VERIFY), runs 4,2,2,2" >> $OUT
for n in 5632 8192 7001; do timeout 600 python tools/verify_probe.py $n 1 40 2>&1 | tail -1 >> $OUT; done
echo "## stress against the oracle" >> $OUT
timeout 900 python tools/persist_stress.py 1 1 2>&1 | tail -6 >> $OUT
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0"
for rep in 1 2; do
for runs in "1,1,0,0" "2,1,0,2" "4,2,2,2" "4,2,1,1" "3,1,0,2" "6,3,1,2" "8,2,2,2"; do
  for K in 20 500; do
    CC4_PERSIST_RUNS=$runs $B --steps $K 2>/dev/null | line "runs=$runs K=$K" >> $OUT
  done
done
done
for n in 4096 5120 16384; do $B --steps 20 --total-envs $n 2>/dev/null | line "default runs, envs=$n K=20" >> $OUT; done
CC4_PERSIST_RUNS=4,2,2,2 python tools/persist_timeline.py 2>&1 | grep "cc4 timeline" | grep -v XCD >> $OUT
cat $OUT
